// Device helpers shared by the two Winograd F(4x4, 3x3) kernels (emb_winograd4.hip: one wave per SIMD and unit;
// emb_winograd4p.hip: two waves per SIMD, a unit's 36 points split between a pair of waves): the packed transforms,
// the pinned-accumulator MFMA macros, the per-stage staging descriptors.
#pragma once

#include "common.h"
#include "emb_winograd_geom.h"
#include "emb_winograd4_geom.h"

namespace pa {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds4_ptr_t;

__device__ __forceinline__ void wino4_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---- B^T x for a 6-vector of channel pairs (12 packed operations)
//   y0 = 4 x0 - 5 x2 + x4          y1 = (x4 - 4 x2) + (x3 - 4 x1)      y2 = (x4 - 4 x2) - (x3 - 4 x1)
//   y5 = 4 x1 - 5 x3 + x5          y3 = (x4 - x2) + 2 (x3 - x1)        y4 = (x4 - x2) - 2 (x3 - x1)
struct W4Const {
  f32x2 p4, m4, m5, p2, m2, m1;
};
__device__ __forceinline__ f32x2 w4fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void wino4_bt(const f32x2 (&x)[6], f32x2 (&y)[6], const W4Const& k) {
  const f32x2 a = w4fma(x[2], k.m4, x[4]);
  const f32x2 b = w4fma(x[1], k.m4, x[3]);
  const f32x2 c = w4fma(x[2], k.m1, x[4]);
  const f32x2 d = w4fma(x[1], k.m1, x[3]);
  y[0] = w4fma(x[0], k.p4, w4fma(x[2], k.m5, x[4]));
  y[1] = a + b;
  y[2] = w4fma(b, k.m1, a);
  y[3] = w4fma(d, k.p2, c);
  y[4] = w4fma(d, k.m2, c);
  y[5] = w4fma(x[1], k.p4, w4fma(x[3], k.m5, x[5]));
}

// ---- A^T m for a 6-vector of float4 (four consecutive output channels): 4 outputs
//   y0 = m0 + (m1 + m2) + (m3 + m4)     y1 = (m1 - m2) + 2 (m3 - m4)
//   y2 = (m1 + m2) + 4 (m3 + m4)        y3 = (m1 - m2) + 8 (m3 - m4) + m5
struct W4Const4 {
  f32x4 m1, p2, p4, p8, m2, m8;
};
__device__ __forceinline__ f32x4 w4fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void wino4_at(const f32x4 m0, const f32x4 m1, const f32x4 m2, const f32x4 m3,
                                         const f32x4 m4, const f32x4 m5, f32x4 (&y)[4], const W4Const4& k) {
  const f32x4 s1 = m1 + m2, d1 = w4fma4(m2, k.m1, m1), s2 = m3 + m4, d2 = w4fma4(m4, k.m1, m3);
  y[0] = m0 + s1 + s2;
  y[1] = w4fma4(d2, k.p2, d1);
  y[2] = w4fma4(s2, k.p4, s1);
  y[3] = w4fma4(d2, k.p8, d1) + m5;
}

// v_mfma_f32_16x16x4_f32 with the accumulator pinned to a register class ("a": AccVGPRs, "v": architectural)
#define W4_MFMA_A(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define W4_MFMA_A_ZERO(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b))
#define W4_MFMA_V(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define W4_MFMA_V_ZERO(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))

// opaque constants in SCALAR registers (a literal would be folded into unpacked single-lane arithmetic, a vector
// register per constant is what made the epilogue spill: 28 registers of splats)
__device__ __forceinline__ float w4_opaque(float v) {
  asm volatile("" : "+s"(v));
  return v;
}

// One stage's staging = 20 LDS-DMA pieces of 1 KB per wave (11 of the patch, 9 of the U slab).  A wave's DMA
// instruction costs it 150-200 cycles of issue on its own (tools/probes/dma_probe.py) but 5-25 inside its own MFMA
// run (interleave_probe.py) -- and with one wave per SIMD nobody else fills those cycles (first build of this kernel:
// all 20 in front of the transform, 14.8 k cycles per stage instead of the 6.5 k its instructions add up to,
// profiles/r4_wino4_v1_dma_exposed.txt).  So the pieces of stage s + 1 are issued from INSIDE the MFMA run of stage
// s, two behind each of its first ten point pairs; the rest of the run hides their flight.
struct Wino4Stage {       // wave-uniform
  __amdgpu_buffer_rsrc_t xsrd, usrd;
  int keep, usoff;
  unsigned char* pbuf;    // this wave's patch block
  unsigned char* ubuf;    // the U buffer being filled
};
// per (unit, cout slice): everything of a stage's staging that does not depend on the stage (computed once per tile;
// the stage adds 32 bytes to the patch origin and one slab to the U offset)
struct Wino4Ctx {
  const float* xp;   // patch origin of stage 0
  int xnum;          // bytes from there to the end of the image
  int keep, usoff;
};
__device__ __forceinline__ Wino4Ctx wino4_ctx(const float* __restrict__ X, int H, int W, int CIN, const Wino4Unit& u,
                                              int n0, int x0_last) {
  using G = Wino4Geom;
  const long img = (long)H * W * CIN;
  const long org = ((long)(u.y0 - 1) * W + (u.x0 - 1)) * CIN;
  Wino4Ctx c;
  c.xp = X + (long)u.b * img + org;
  c.xnum = (int)((img - org) * 4);
  c.keep = wino4_patch_keep(u, x0_last);
  c.usoff = (n0 / W_BN) * (CIN / G::CB) * G::USLAB_BYTES;
#ifdef PA_W4_NOPATCH   // development A/B (never in the product build): every patch lane out of bounds -> no traffic
  c.keep = -1;
#endif
  return c;
}
__device__ __forceinline__ Wino4Stage wino4_stage(const Wino4Ctx& c, const float* __restrict__ U, int COUT, int CIN,
                                                  int s, unsigned char* pbuf, unsigned char* ubuf) {
  using G = Wino4Geom;
  Wino4Stage st;
  st.xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.xp + s * G::CB), 0, c.xnum - s * G::CB * 4,
                                              0x00020000);
  st.usrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, 36 * COUT * CIN * 4, 0x00020000);
  st.keep = c.keep;
  st.usoff = c.usoff + s * G::USLAB_BYTES;
#ifdef PA_W4_NOU       // ... every U piece from slab 0 (L2-resident)
  st.usoff = 0;
#endif
  st.pbuf = pbuf;
  st.ubuf = ubuf;
  return st;
}
}  // namespace pa
