// Winograd F(4x4, 3x3) for the stride-1 BasicBlock convolutions of the WeSpeaker ResNet on gfx950
// (reference: models/embedding/wespeaker/resnet.py:84-145).  36 multiplies per 4x4 output tile and (cin, cout)
// pair instead of 64 with F(2x2, 3x3) (emb_winograd.hip) and 144 in the direct form: 1.78x fewer MFMAs than the
// kernel it replaces, at true fp32.
//
//     Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A         per 4x4 output tile, 6x6 input patch d
//
// Numerics (tools/probes/winograd_f4_numerics.py, oracle modules on the CPU): the transforms multiply by up to 8 and
// 1/24, yet the embeddings of the whole ResNet34 move by 1.3e-7 (F(2x2): 1.2e-7; float32 direct vs float64: 2.2e-7)
// of a 1e-5 + 1e-4 |ref| bound -- BatchNorm re-scales every layer.  U = G g G^T is prepared in float64 on the host.
//
// Shape of the kernel -- different from emb_winograd.hip because 36 points x 32 output channels = 288 accumulator
// registers per lane:
//   * ONE workgroup of 4 waves per CU, one wave per SIMD, 512 registers per lane (__launch_bounds__(256, 1)): the
//     accumulators live in the AccVGPRs, the transformed patch (72) and the transform's temporaries in the
//     architectural ones.  On gfx950 the f32 MFMA executes on the vector ALUs, nothing co-issues with it and SIMD
//     time is the SUM of the issue cycles of everything (profiles/r2_mfma_probe.txt), so a second wave per SIMD only
//     ever hid WAITS -- here there are none left to hide: both LDS images are double-buffered and the DMA of stage
//     s + 1 (the next tile's first stage behind a tile's last one) is issued right after the barrier that opens
//     stage s; it has 6 500 cycles to land.
//   * workgroup tile = 8 x 128 output pixels x 32 output channels; wave w: tile row w >> 1, 16 tile columns;
//     lane (t = lane & 15, g = lane >> 4): tile t, input-channel pair g of the 8-channel stage.  The lane
//     transforms the 6x6 patch of its tile for its two channels with packed f32 arithmetic (144 v_pk_* per stage:
//     B^T x = 12 operations per 6-vector) -- V never touches LDS -- and feeds it to v_mfma_f32_16x16x4_f32 as the B
//     operand; U is the A operand, so a lane ends up with four consecutive output channels of one tile for all 36
//     points: inverse transform and epilogue are lane-local float4 arithmetic with 16-byte accesses.
//   * per stage and wave: 144 MFMAs (4 608 cycles) against ~580 cycles of input transform, 36 + 72 ds_read_b64 and
//     ~20 LDS-DMA instructions; per tile the inverse transform (400 packed operations) + 32 stores (+ 32 residual
//     loads, issued per channel group in front of that group's inverse transform).
//   * staging is LDS-DMA (buffer_load_dwordx4 ... lds) as in emb_winograd.hip: the patch de-interleaved by column
//     mod 4 (the 16 lanes of a tile row read consecutive 32-B rows: every ds_read_b64 covers 512 contiguous bytes),
//     halo and out-of-image columns zero-filled by the buffer bounds check through class bits; U as one contiguous
//     36-KB image per (32-cout slice, 8-cin stage) (weights.winograd4_pack).
//   * tiles are claimed at run time (tile_queue.h), in the XCD-aware order of wino_decode.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "emb_winograd_geom.h"
#include "emb_winograd4_geom.h"

namespace pa {

#ifndef PA_W4_STAMP
#define PA_W4_STAMP 0
#endif
#if PA_W4_STAMP
// development instrumentation (never in the product build): s_memtime at the phases of the first 64 stages of
// workgroups 0 .. 7, per wave; read back with pa_wino4_read_stamps
__device__ unsigned long long g_w4_stamps[8 * 4 * 64 * 6];
#define W4_STAMP(p) st_[p] = __builtin_amdgcn_s_memtime()
#define W4_STAMP_FLUSH()                                                                      \
  do {                                                                                        \
    if (blockIdx.x < 8 && st_iter < 64 && lane == 0) {                                         \
      _Pragma("unroll") for (int p_ = 0; p_ < 6; ++p_)                                         \
          g_w4_stamps[((blockIdx.x * 4 + slw) * 64 + st_iter) * 6 + p_] = st_[p_];             \
    }                                                                                         \
    ++st_iter;                                                                                \
  } while (0)
#else
#define W4_STAMP(p)
#define W4_STAMP_FLUSH()
#endif

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
constexpr int W4_AGPR_POINTS = 32;   // 256 AccVGPRs; 4 points (32 registers) stay architectural
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
  unsigned char* buf;
};
__device__ __forceinline__ Wino4Stage wino4_stage(const float* __restrict__ X, int H, int W, int CIN,
                                                  const float* __restrict__ U, int COUT, const WinoTile& q, int c0,
                                                  unsigned char* buf, int x0_last) {
  using G = Wino4Geom;
  const long img = (long)H * W * CIN;
  const long org = ((long)(q.y0 - 1) * W + (q.x0 - 1)) * CIN + c0;
  Wino4Stage st;
  st.xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + (long)q.b * img + org), 0,
                                              (int)((img - org) * 4), 0x00020000);
  st.usrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, 36 * COUT * CIN * 4, 0x00020000);
  st.keep = wino_patch_keep(q, x0_last);
  st.usoff = ((q.n0 / W_BN) * (CIN / G::CB) + c0 / G::CB) * G::USLAB_BYTES;
#ifdef PA_W4_NOPATCH   // development A/B (never in the product build): every patch lane out of bounds -> no traffic
  st.keep = -1;
#endif
#ifdef PA_W4_NOU       // ... every U piece from slab 0 (L2-resident)
  st.usoff = 0;
#endif
  st.buf = buf;
  return st;
}
constexpr int W4_PIECES = Wino4Geom::NPP + 9;
struct Wino4Lanes {       // per-lane patch offsets + class bits (wino4_patch_lanes), one per piece of this wave
  int a[Wino4Geom::NPP];
};
// (I is a compile-time constant at every call site after unrolling)
__device__ __forceinline__ void wino4_piece(const int I, const Wino4Stage& st, const Wino4Lanes& pl, int lane,
                                            int slw) {
  using G = Wino4Geom;
  if (I < G::NPP) {
    const int k = slw + 4 * I;
    const int* prel = pl.a;
    if (k < G::PINSTR)   // wave-uniform
#ifdef PA_W4_NOPATCH
      __builtin_amdgcn_raw_ptr_buffer_load_lds(st.xsrd, (lds4_ptr_t)(st.buf + 1024 * k), 16, prel[I] | WCLS_PAD, 0, 0,
                                               0);
#else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(st.xsrd, (lds4_ptr_t)(st.buf + 1024 * k), 16, prel[I] & st.keep, 0, 0,
                                               0);
#endif
  } else {
    const int k = slw + 4 * (I - G::NPP);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(st.usrd, (lds4_ptr_t)(st.buf + G::PATCH_BYTES + 1024 * k), 16, lane * 16,
                                             st.usoff + 1024 * k, 0, 0);
  }
}
__device__ __forceinline__ void wino4_issue_all(const Wino4Stage& st, const Wino4Lanes& pl, int lane, int slw) {
#pragma unroll
  for (int i = 0; i < W4_PIECES; ++i) wino4_piece(i, st, pl, lane, slw);
}

template <bool HAS_R>
__global__ __launch_bounds__(256, 1) void k_conv3x3_wino4(
    const float* __restrict__ X, int H, int W, int CIN, const float* __restrict__ U,
    const float* __restrict__ shift, const float* __restrict__ R, float* __restrict__ Y, int COUT, int relu,
    int tiles_w, int tiles_hw, int n_tiles, int total_tiles, int num_pb, int* __restrict__ counters) {
  using G = Wino4Geom;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int t = lane & 15, g = lane >> 4;
  const int slw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = slw >> 1, wc = slw & 1;
  int* mail = reinterpret_cast<int*>(smem4 + 2 * G::BUF_BYTES);

  const TileQueue tq{counters, (int)(blockIdx.x & 7), total_tiles >> 3};
  if (tid == 0) mail[0] = tq_resolve(tq, tq_claim_own(tq));
  __syncthreads();
  int q = mail[0];
  if (q < 0) {
    if (tid == 0) tq_done(tq, gridDim.x);
    return;
  }
  const int x0_last = (tiles_w - 1) * G::TW;
  Wino4Lanes pl;
  wino4_patch_lanes(pl.a, W, CIN, lane, slw, x0_last);
  const int pbase = wino4_patch_base(t, g, wr, wc);
  const int ubase = wino4_u_base(t, g);
  W4Const kc;
  {
    const float p4 = w4_opaque(4.f), m4 = w4_opaque(-4.f), m5 = w4_opaque(-5.f), p2 = w4_opaque(2.f),
                m2 = w4_opaque(-2.f), m1 = w4_opaque(-1.f);
    kc.p4 = f32x2{p4, p4}; kc.m4 = f32x2{m4, m4}; kc.m5 = f32x2{m5, m5};
    kc.p2 = f32x2{p2, p2}; kc.m2 = f32x2{m2, m2}; kc.m1 = f32x2{m1, m1};
  }
  const int nstages = CIN / G::CB;

  WinoTile cur = wino_decode(q, tiles_w, tiles_hw, n_tiles, G::TH, G::TW, num_pb), nxt = cur;
  int buf = 0;
  wino4_issue_all(wino4_stage(X, H, W, CIN, U, COUT, cur, 0, smem4, x0_last), pl, lane, slw);
  int claim = 0;
  if (tid == 0) claim = tq_claim_own(tq);
  int nq = -1;
  f32x4 acca[W4_AGPR_POINTS][2];        // points 0 .. 31: AccVGPRs
  f32x4 accv[36 - W4_AGPR_POINTS][2];   // points 32 .. 35: architectural registers

#if PA_W4_STAMP
  unsigned long long st_[6] = {0, 0, 0, 0, 0, 0};
  int st_iter = 0;
#endif
  while (true) {
    for (int s = 0; s < nstages; ++s) {
      W4_STAMP(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this stage's images have landed (issued a stage ago)
      W4_STAMP(1);
      wino4_barrier();                                    // ... everybody's; and everybody is done with the other buffer
      W4_STAMP(2);
      unsigned char* mine = smem4 + buf * G::BUF_BYTES;
      unsigned char* other = smem4 + (buf ^ 1) * G::BUF_BYTES;
      // what the MFMA run below stages: the next stage of this tile, or the first stage of the next one
      bool stage_next = true;
      Wino4Stage nst;
      if (s + 1 < nstages) {
        nst = wino4_stage(X, H, W, CIN, U, COUT, cur, (s + 1) * G::CB, other, x0_last);
      } else {
        nq = mail[0];   // (written by thread 0 in front of this stage's barrier)
        stage_next = nq >= 0;
        nxt = wino_decode(stage_next ? nq : 0, tiles_w, tiles_hw, n_tiles, G::TH, G::TW, num_pb);
        nst = wino4_stage(X, H, W, CIN, U, COUT, nxt, 0, other, x0_last);
      }
      // ---- input transform V = B^T d B of this lane's tile and channel pair, in registers
      f32x2 v[6][6];
      {
        const unsigned char* pb = mine + pbase;
        f32x2 tt[6][6];
        // columns of d: tt[.][j] = B^T d[.][j]; column j + 1 is read while column j is combined (left alone, the
        // scheduler issues all 36 reads first and the 72 extra registers spill)
        f32x2 x[2][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) x[0][i] = *reinterpret_cast<const f32x2*>(pb + wino4_patch_k(i, 0));
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if (j + 1 < 6) {
#pragma unroll
            for (int i = 0; i < 6; ++i) x[(j + 1) & 1][i] = *reinterpret_cast<const f32x2*>(pb + wino4_patch_k(i, j + 1));
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x2 y[6];
          wino4_bt(x[j & 1], y, kc);
#pragma unroll
          for (int i = 0; i < 6; ++i) tt[i][j] = y[i];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {   // rows: v[i][.] = B^T tt[i][.]
          wino4_bt(tt[i], v[i], kc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#if PA_W4_STAMP
      asm volatile("s_nop 0" ::"v"(v[5][5]), "v"(v[0][0]));   // the transform is complete here
#endif
      W4_STAMP(3);
      // ---- 36 points x 2 channel groups x 2 k-steps, two points at a time (a dependent MFMA is four MFMAs behind
      // its producer); U fragments of the next pair are read while this pair's MFMAs issue.  The MFMAs are inline
      // assembly because the accumulators must be PINNED: 32 points in the 256 AccVGPRs, 4 in architectural
      // registers (left to the register allocator, 288 accumulators + the transform spill ~200 registers).
      auto mfma_run = [&](auto first_stage) {
        constexpr bool FIRST = decltype(first_stage)::value;
        const unsigned char* ub = mine + G::PATCH_BYTES + ubase;
        f32x2 uf[2][2][2];   // [pair parity][point of the pair][channel group]
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int cg = 0; cg < 2; ++cg) uf[0][e][cg] = *reinterpret_cast<const f32x2*>(ub + wino4_u_k(e, cg));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xp = 0; xp < 36; xp += 2) {
          const int par = (xp >> 1) & 1;
          // eight MFMAs per point pair, everything else spread between them one instruction at a time (an LDS read
          // or an LDS-DMA issued right behind an MFMA costs the wave nothing while that MFMA executes; two DMA
          // pieces back to back do: first build of this loop, 6 240 cycles per run instead of 4 608 + ~600)
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const int ks = m >> 2, e = (m >> 1) & 1, cg = m & 1;
            const int xi = xp + e;
            const f32x2 bv = v[xi / 6][xi % 6];
            const float a = ks ? uf[par][e][cg].y : uf[par][e][cg].x, b = ks ? bv.y : bv.x;
            if (xi < W4_AGPR_POINTS) {
              if (ks == 0 && FIRST) W4_MFMA_A_ZERO(acca[xi < W4_AGPR_POINTS ? xi : 0][cg], a, b);
              else W4_MFMA_A(acca[xi < W4_AGPR_POINTS ? xi : 0][cg], a, b);
            } else {
              if (ks == 0 && FIRST) W4_MFMA_V_ZERO(accv[xi >= W4_AGPR_POINTS ? xi - W4_AGPR_POINTS : 0][cg], a, b);
              else W4_MFMA_V(accv[xi >= W4_AGPR_POINTS ? xi - W4_AGPR_POINTS : 0][cg], a, b);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifndef PA_W4_NOUREAD   // (development A/B: the same fragments for every point -> what the U reads cost)
            if ((m == 0 || m == 2) && xp + 2 < 36) {      // U fragments of the next pair, one point per slot
              const int en = m >> 1;
#pragma unroll
              for (int c2 = 0; c2 < 2; ++c2)
                uf[par ^ 1][en][c2] = *reinterpret_cast<const f32x2*>(ub + wino4_u_k(xp + 2 + en, c2));
              __builtin_amdgcn_sched_barrier(0);
            }
#else
            if (xp == 0 && m == 0) {
#pragma unroll
              for (int en = 0; en < 2; ++en)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) uf[1][en][c2] = uf[0][en][c2];
            }
#endif
#ifndef PA_W4_NODMA     // (development A/B: no staging at all from inside the run)
            if ((m == 4 || m == 6) && stage_next) {       // wave-uniform; pieces xp, xp + 1 of the next stage
              const int piece = xp + ((m - 4) >> 1);
              if (piece < W4_PIECES) wino4_piece(piece, nst, pl, lane, slw);
              __builtin_amdgcn_sched_barrier(0);
            }
#endif
          }
        }
      };
      if (s == 0) mfma_run(std::true_type{});
      else mfma_run(std::false_type{});
      W4_STAMP(4);
      if (s == nstages - 2 && tid == 0) mail[0] = tq_resolve(tq, claim);   // published by the next barrier
      buf ^= 1;
      if (s + 1 < nstages) {
        W4_STAMP(5);
        W4_STAMP_FLUSH();
      }
    }
    // ---- inverse transform A^T M A + BN shift (+ residual) (+ ReLU), 16-byte stores
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (the compiler cannot see them)
    {
      W4Const4 k4;
      {
        const float m1 = w4_opaque(-1.f), p2 = w4_opaque(2.f), p4 = w4_opaque(4.f), p8 = w4_opaque(8.f),
                    m2 = w4_opaque(-2.f), m8 = w4_opaque(-8.f);
        k4.m1 = f32x4{m1, m1, m1, m1}; k4.p2 = f32x4{p2, p2, p2, p2};
        k4.p4 = f32x4{p4, p4, p4, p4}; k4.p8 = f32x4{p8, p8, p8, p8};
        k4.m2 = f32x4{m2, m2, m2, m2}; k4.m8 = f32x4{m8, m8, m8, m8};
      }
      const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
          Y + (long)cur.b * H * W * COUT, 0, H * W * COUT * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(HAS_R ? R + (long)cur.b * H * W * COUT : Y), 0, H * W * COUT * 4, 0x00020000);
      constexpr int OOB = (int)0x80000000;
      const int xl = cur.x0 + 64 * wc + 4 * t;
      const int obase = (((cur.y0 + 4 * wr) * W + xl) * COUT + cur.n0 + 4 * g) * 4;
      const int srow = W * COUT * 4, spix = COUT * 4;
      int offq[4];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) offq[qq] = (cur.valid && xl + qq < W) ? obase + qq * spix : OOB;
      const float lo = relu ? 0.f : -__builtin_inff();
      const f32x4 lo4 = {lo, lo, lo, lo};
      // Every accumulator is read ONCE, column by column of the 6x6 point grid: y = A^T M[:, b] (10 operations),
      // then its 18 non-zero contributions AT[q][b] y[p] go straight into the 4x4 outputs.  (Two passes over the
      // accumulators -- rows 0-1, then rows 2-3 -- made the compiler keep every accumulator's VGPR copy alive
      // between them: 180 registers, 70-100 spills, 17 k cycles per tile.)
#define W4_ACC(xi) ((xi) < W4_AGPR_POINTS ? acca[(xi) < W4_AGPR_POINTS ? (xi) : 0][cg] \
                                          : accv[(xi) >= W4_AGPR_POINTS ? (xi)-W4_AGPR_POINTS : 0][cg])
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + cur.n0 + 16 * cg + 4 * g);
        f32x4 rv[2][4];
        if (HAS_R) {   // rows 0-1 now (their latency hides under the inverse transform), rows 2-3 behind them
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
              rv[p][qq] = __builtin_bit_cast(
                  f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                             rsrd, offq[qq] == OOB ? OOB : offq[qq] + p * srow + 64 * cg, 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 o[4][4];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          // (an empty volatile asm re-defines each accumulator of this column HERE: instruction selection otherwise
          //  emits the AccVGPR -> VGPR copies of all 72 accumulators at the top of the epilogue -- sched_barrier only
          //  binds the machine scheduler -- and 288 live registers spill)
#pragma unroll
          for (int a6 = 0; a6 < 6; ++a6) {
            constexpr int dummy = 0;
            (void)dummy;
            const int xi = 6 * a6 + b;
            if (xi < W4_AGPR_POINTS) asm volatile("" : "+a"(acca[xi < W4_AGPR_POINTS ? xi : 0][cg]));
          }
          f32x4 y[4];
          wino4_at(W4_ACC(0 + b), W4_ACC(6 + b), W4_ACC(12 + b), W4_ACC(18 + b), W4_ACC(24 + b), W4_ACC(30 + b), y,
                   k4);
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            if (b == 0) {
              o[p][0] = y[p];
            } else if (b == 1) {
              o[p][0] = o[p][0] + y[p];
              o[p][1] = y[p];
              o[p][2] = y[p];
              o[p][3] = y[p];
            } else if (b == 2) {
              o[p][0] = o[p][0] + y[p];
              o[p][1] = w4fma4(y[p], k4.m1, o[p][1]);
              o[p][2] = o[p][2] + y[p];
              o[p][3] = w4fma4(y[p], k4.m1, o[p][3]);
            } else if (b == 3) {
              o[p][0] = o[p][0] + y[p];
              o[p][1] = w4fma4(y[p], k4.p2, o[p][1]);
              o[p][2] = w4fma4(y[p], k4.p4, o[p][2]);
              o[p][3] = w4fma4(y[p], k4.p8, o[p][3]);
            } else if (b == 4) {
              o[p][0] = o[p][0] + y[p];
              o[p][1] = w4fma4(y[p], k4.m2, o[p][1]);
              o[p][2] = w4fma4(y[p], k4.p4, o[p][2]);
              o[p][3] = w4fma4(y[p], k4.m8, o[p][3]);
            } else {
              o[p][3] = o[p][3] + y[p];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              f32x4 vv = o[2 * h + p][qq] + sh;
              if (HAS_R) vv = vv + rv[p][qq];
              vv = __builtin_elementwise_max(vv, lo4);
              __builtin_amdgcn_raw_buffer_store_b128(
                  __builtin_bit_cast(u32x4, vv), ysrd,
                  offq[qq] == OOB ? OOB : offq[qq] + (2 * h + p) * srow + 64 * cg, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
          if (HAS_R && h == 0) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
              for (int qq = 0; qq < 4; ++qq)
                rv[p][qq] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                               rsrd, offq[qq] == OOB ? OOB : offq[qq] + (2 + p) * srow + 64 * cg, 0, 0));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#undef W4_ACC
    }
    W4_STAMP(5);
    W4_STAMP_FLUSH();
    if (nq < 0) break;
    cur = nxt;
    if (tid == 0) claim = tq_claim_own(tq);
  }
  if (tid == 0) tq_done(tq, gridDim.x);
}

template <bool HAS_R>
static int launch_wino4(const float* X, int B, int H, int W, int CIN, const float* U, const float* shift,
                        const float* R, float* Y, int COUT, int relu, hipStream_t st) {
  using G = Wino4Geom;
  const int tiles_w = cdiv(W, G::TW), tiles_h = cdiv(H, G::TH);
  const size_t lds = 2 * (size_t)G::BUF_BYTES + 16;
  auto kernel = k_conv3x3_wino4<HAS_R>;
  constexpr int MAXDEV = 16;
  static int cus_of[MAXDEV] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= MAXDEV) dev = 0;
  if (!cus_of[dev]) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus_of[dev] = cus;
  }
  const int tiles_hw = tiles_w * tiles_h, n_tiles = COUT / W_BN;
  const long num_pb = (long)tiles_hw * B;
  const long total = ((num_pb + 7) / 8) * 8 * n_tiles;
  const int resident = cus_of[dev] & ~7;             // one workgroup per CU
  const int grid = (int)(total < resident ? total : resident);
  int* counters = tile_counters();
  if (counters == nullptr) {
    set_error("pa_conv3x3_wino4: cannot allocate the tile counters");
    return 2;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, st, X, H, W, CIN, U, shift, R, Y, COUT, relu, tiles_w,
                     tiles_hw, n_tiles, (int)total, (int)num_pb, counters);
  return 0;
}

}  // namespace pa

extern "C" {

#if PA_W4_STAMP
int pa_wino4_read_stamps(unsigned long long* host) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pa::g_w4_stamps), sizeof(unsigned long long) * 8 * 4 * 64 * 6) ==
                 hipSuccess
             ? 0
             : 1;
}
#endif

// conv3x3, stride 1, pad 1, via Winograd F(4x4,3x3): Y = [relu](conv(X) + shift [+ R]).  X, R, Y: NHWC float32.
// U: G g G^T (BatchNorm scale folded) in the slab layout of weights.winograd4_pack / pa_winograd4_pack_host:
// [cout/32][cin/8][row = 32 xi + n][8], xi = 6a + b.
int pa_conv3x3_wino4(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                     const float* R, float* Y, int cout, int relu, void* stream) {
  if (B <= 0) return 0;
  PA_REQUIRE(cin % 8 == 0 && cin >= 32 && cout % pa::W_BN == 0,
             "pa_conv3x3_wino4: cin %% 8 == 0, cin >= 32 and cout %% 32 == 0 required");
  PA_REQUIRE((long)H * W * (cin > cout ? cin : cout) * 4 < (1L << 28),
             "pa_conv3x3_wino4: one image must be smaller than 256 MB");
  // `flops` = the direct convolution's (the reference's operation); the kernel executes 36/144 of them
  pa::ProfScope prof("k_conv3x3_wino4", stream, 2.0 * 9 * cin * cout * (double)B * H * W,
                     4.0 * ((double)B * H * W * cin + (double)B * H * W * cout * (R ? 2 : 1) + 9.0 * cin * cout));
  hipStream_t st = (hipStream_t)stream;
  const int rc = R != nullptr ? pa::launch_wino4<true>(X, B, H, W, cin, U, shift, R, Y, cout, relu, st)
                              : pa::launch_wino4<false>(X, B, H, W, cin, U, shift, R, Y, cout, relu, st);
  if (rc != 0) return rc;
  PA_CHECK_LAUNCH("pa_conv3x3_wino4");
  return 0;
}

}  // extern "C"
