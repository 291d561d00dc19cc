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
//     ever hid WAITS -- here there are none left to hide: the staging of stage s + 1 (the next tile's first stage
//     behind a tile's last one) is issued from inside the MFMA run of stage s and has the rest of it to land.  (A
//     two-waves-per-SIMD variant with the points of a unit split between a pair of waves was built and measured
//     twice in round 4: slower both times, tools/probes/emb_winograd4_paired.hip.txt.)
//   * a wave owns a UNIT = 16 consecutive tiles of one tile row of one image (4 x 64 output pixels), a workgroup 4
//     consecutive units x 32 output channels -- any rows of any images, so that maps with an odd number of tile rows
//     (20 and 10 pixel rows: 5 and 3) leave no wave idle (the first build tiled 8 x 128 pixels per workgroup: one
//     wave in six / four had nothing to do there).  The patch of a unit is PRIVATE to its wave: no barrier, no
//     second buffer -- the wave issues the next stage's patch right behind its own input transform; only the U slab
//     is shared and double-buffered (one barrier per stage).
//     lane (t = lane & 15, g = lane >> 4): tile t, input-channel pair g of the 8-channel stage.  The lane
//     transforms the 6x6 patch of its tile for its two channels with packed f32 arithmetic (144 v_pk_* per stage:
//     B^T x = 12 operations per 6-vector) -- V never touches LDS -- and feeds it to v_mfma_f32_16x16x4_f32 as the B
//     operand; U is the A operand, so a lane ends up with four consecutive output channels of one tile for all 36
//     points: inverse transform and epilogue are lane-local float4 arithmetic with 16-byte accesses.
//   * per stage and wave (measured, 128 channels: 7 500 cycles): 144 MFMAs (4 608 cycles), the input transform
//     (144 packed operations at 8 cycles + its 36 ds_read_b64: 1 390), 72 ds_read_b64 of U fragments (free between
//     MFMAs), 22 LDS-DMA pieces (~24 cycles each from inside the run), one barrier; per tile the inverse transform
//     (~230 packed operations per channel group) + 32 stores of 16 scattered 64-byte segments (~200 cycles each;
//     those of channel group 0 are issued from inside the arithmetic of group 1 when there is no residual) + 32
//     residual loads: 13 000 cycles.
//   * staging is LDS-DMA (buffer_load_dwordx4 ... lds) as in emb_winograd.hip: the patch de-interleaved by column
//     mod 4 (the 16 lanes of a tile row read consecutive 32-B rows), halo and out-of-image columns zero-filled by the
//     buffer bounds check through class bits; U as one contiguous 36-KB image per (32-cout slice, 8-cin stage)
//     (weights.winograd4_pack).  22 pieces of 1 KB per wave and stage, issued as inline assembly with scalar-only
//     set-up (wino4_piece_asm).  Both images carry a bank swizzle (emb_winograd4_geom.h) and are read with
//     ds_read_b64 that the compiler may not fuse (w4_lds_read64): 0 LDS bank conflict cycles.
//   * groups of 4 units are claimed at run time (tile_queue.h), in an XCD-aware order (wino4_decode).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "emb_winograd_geom.h"
#include "emb_winograd4_geom.h"

namespace pa {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds4_ptr_t;

// ds_read_b64 that stays one: volatile, so that the load/store optimizer does not fuse two of them into a
// ds_read2_b64 (served in 16-lane groups on 32 banks: 4-way conflicts with these layouts), through an explicit LDS
// pointer (a volatile access through a generic pointer is left a flat_load)
typedef const volatile f32x2 __attribute__((address_space(3))) * w4_lds_f32x2_ptr;
__device__ __forceinline__ f32x2 w4_lds_read64(const unsigned char* p) { return *(w4_lds_f32x2_ptr)p; }

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
  st.pbuf = pbuf;
  st.ubuf = ubuf;
  return st;
}

#ifndef PA_W4_STAMP
#define PA_W4_STAMP 0
#endif
#ifndef PA_W4_DEFER_STORES   // stores of channel group 0 issued from inside the column arithmetic of group 1:
#define PA_W4_DEFER_STORES 1 // 0 = never, 1 = instantiation without residual only, 2 = both (spills: slower)
#endif
// cache-policy bits of the output stores / the residual loads (gfx942+: 1 = sc0, 2 = nt, 16 = sc1).  The outputs are
// written once and read by the NEXT kernel, 9 GB later: non-temporal stores leave the L2 to the patches, whose lines are
// touched by four consecutive stages (8 of a pixel's 32 channels each): -2 % per launch on average, -4 % at 256 channels
// (profiles/r5_xcd_ranges.txt, section 4).  Non-temporal residual loads measured slower.
#ifndef PA_W4_STORE_AUX
#define PA_W4_STORE_AUX 2
#endif
#ifndef PA_W4_RES_AUX
#define PA_W4_RES_AUX 0
#endif
#ifndef PA_W4_STORES_IN_FLIGHT   // a tile's first stage does not wait for the previous tile's stores (0: vmcnt(0) as in round 4)
#define PA_W4_STORES_IN_FLIGHT 1
#endif
#ifndef PA_W4_LATE_BARRIER   // the stage barrier behind most of the input transform (0: in front of it, as in round 4)
#define PA_W4_LATE_BARRIER 1
#endif
#if PA_W4_STAMP
// development instrumentation (never in the product build): s_memtime at the phases of the first 64 stages of
// workgroups 0 .. 7, per wave; read back with pa_wino4_read_stamps
__device__ unsigned long long g_w4_stamps[8 * 4 * 64 * 10];
#define W4_STAMP(p) st_[p] = __builtin_amdgcn_s_memtime()
#define W4_STAMP_FLUSH()                                                                      \
  do {                                                                                        \
    if (blockIdx.x < 8 && st_iter < 64 && lane == 0) {                                         \
      _Pragma("unroll") for (int p_ = 0; p_ < 10; ++p_)                                        \
          g_w4_stamps[((blockIdx.x * 4 + slw) * 64 + st_iter) * 10 + p_] = st_[p_];            \
    }                                                                                         \
    ++st_iter;                                                                                \
  } while (0)
#else
#define W4_STAMP(p)
#define W4_STAMP_FLUSH()
#endif

constexpr int W4_AGPR_POINTS = 32;   // 256 AccVGPRs; 4 points (32 registers) stay architectural
// The partial waits of the stage loop count vector-memory operations (they complete in order): the numbers below are
// what the issue loops and the epilogue are BUILT from, so that a change of either cannot silently turn a wait into a
// read-before-DMA race.
constexpr int W4_U_PIECES = Wino4Geom::UINSTR / 4;   // U pieces of a stage per wave: issued BEHIND its patch pieces
static_assert(W4_U_PIECES == 9 && Wino4Geom::UINSTR % 4 == 0, "36 U pieces of 1 KB per stage, 4 waves");
constexpr int W4_TILE_OUT = 16;                      // 4 x 4 outputs per lane, tile and channel group: one 16-byte access each
// vector-memory operations of an epilogue, all issued BEHIND the next tile's first staging: 2 channel groups x 16 stores,
// + 2 x 16 residual loads with a residual.  A tile's first stage waits until at most W4_TAIL_WAIT operations are
// outstanding: its staging has landed, the newest stores may still fly.
constexpr int W4_TAIL_WAIT = 2 * W4_TILE_OUT;
#define W4_STR2(x) #x
#define W4_STR(x) W4_STR2(x)
#define W4_U_PIECES_LIT 9
#define W4_TAIL_WAIT_LIT 32
static_assert(W4_U_PIECES == W4_U_PIECES_LIT && W4_TAIL_WAIT == W4_TAIL_WAIT_LIT, "the literals of the s_waitcnt strings");

// pieces of a stage per wave: PIN of the wave's patch (13 row-shaped / 18 tile-linear units) + its W4_U_PIECES of the 36 U pieces
template <int PIN>
struct Wino4Lanes {       // per-lane patch offsets (+ class bits, wino4_patch_lanes), one per patch piece
  int a[PIN];
};
// (I is a compile-time constant at every call site after unrolling)
template <int PIN>
__device__ __forceinline__ void wino4_piece(const int I, const Wino4Stage& st, const Wino4Lanes<PIN>& pl, int lane,
                                            int slw) {
  if (I < PIN) {
    const int off = pl.a[I < PIN ? I : 0] & st.keep;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(st.xsrd, (lds4_ptr_t)(st.pbuf + 1024 * I), 16, off, 0, 0, 0);
  } else {
    const int k = slw + 4 * (I - PIN);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(st.usrd, (lds4_ptr_t)(st.ubuf + 1024 * k), 16, lane * 16,
                                             st.usoff + 1024 * k, 0, 0);
  }
}
// The same piece from INSIDE the MFMA run, as inline assembly with scalar-only set-up: M0 and the scalar offset are
// base + immediate (the compiler's version kept 22 precomputed LDS addresses in SGPRs, spilled them to VGPR lanes and
// reloaded each with a v_readlane, and masked the patch offset with a v_and per piece: two vector-ALU instructions per
// piece that queue behind the MFMA in flight -- the f32 MFMA runs on the vector ALUs -- and, issue being in order,
// hold back the next MFMA: ~58 cycles per piece, profiles/r4_wino4_anatomy.txt).  `plm` = the patch offsets already
// masked with the unit's class bits; `lane16` = lane * 16.
struct Wino4Dma {           // wave-uniform LDS byte addresses
  unsigned patch_lds;       // this wave's patch block
  unsigned u_lds;           // the U buffer being filled + 1024 * wave
  int usoff;                // slab offset of the stage + 1024 * wave
};
__device__ __forceinline__ unsigned w4_lds_addr(const void* p) {
  return (unsigned)(size_t)(lds4_ptr_t)const_cast<void*>(p);
}
template <int I, int PIN>
__device__ __forceinline__ void wino4_piece_asm(const Wino4Stage& st, const Wino4Dma& d, const int (&plm)[PIN],
                                                int lane16) {
  if constexpr (I < PIN) {
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds"
                 :: "s"(d.patch_lds), "i"(1024 * I), "v"(plm[I]), "s"(st.xsrd)
                 : "memory");
  } else {
    int tmp;
    asm volatile("s_add_u32 m0, %1, %2\n\ts_add_u32 %0, %3, %2\n\tbuffer_load_dwordx4 %4, %5, %0 offen lds"
                 : "=&s"(tmp)
                 : "s"(d.u_lds), "i"(4096 * (I - PIN)), "s"(d.usoff), "v"(lane16), "s"(st.usrd)
                 : "memory");
  }
}
// (the piece number is a constant after unrolling: the switch folds)
template <int PIN>
__device__ __forceinline__ void wino4_piece_asm_n(const int piece, const Wino4Stage& st, const Wino4Dma& d,
                                                  const int (&plm)[PIN], int lane16) {
  switch (piece) {
#define W4_CASE(I) case I: if constexpr (I < PIN + W4_U_PIECES) wino4_piece_asm<I, PIN>(st, d, plm, lane16); break;
    W4_CASE(0) W4_CASE(1) W4_CASE(2) W4_CASE(3) W4_CASE(4) W4_CASE(5) W4_CASE(6) W4_CASE(7) W4_CASE(8) W4_CASE(9)
    W4_CASE(10) W4_CASE(11) W4_CASE(12) W4_CASE(13) W4_CASE(14) W4_CASE(15) W4_CASE(16) W4_CASE(17) W4_CASE(18)
    W4_CASE(19) W4_CASE(20) W4_CASE(21) W4_CASE(22) W4_CASE(23) W4_CASE(24) W4_CASE(25) W4_CASE(26)
#undef W4_CASE
    default: break;
  }
}
template <int PIN>
__device__ __forceinline__ void wino4_issue_all(const Wino4Stage& st, const Wino4Lanes<PIN>& pl, int lane, int slw) {
#pragma unroll
  for (int i = 0; i < PIN + W4_U_PIECES; ++i) wino4_piece<PIN>(i, st, pl, lane, slw);   // patch first, then U
}

// ---- tile-linear units (emb_winograd4_geom.h): what a lane needs of its unit
struct Wino4LinUnit {
  int b0;               // wave-uniform: first image of the unit
  Wino4LinTile tc;      // the tile this lane transforms and stores (t = lane & 15)
};
// per-lane DMA offsets of unit u (18 pieces) + the lane's compute tile
__device__ __forceinline__ Wino4LinUnit wino4_lin_unit(int u, int tcols, int trows, int total_tiles, int H, int W,
                                                       int CIN, int lane, int (&off)[Wino4LinGeom::PINSTR]) {
  Wino4LinUnit o;
  o.b0 = __builtin_amdgcn_readfirstlane(wino4_lin_b0(u, tcols, trows, total_tiles));
  o.tc = wino4_lin_tile(16 * u + (lane & 15), tcols, trows, total_tiles, o.b0);
  const Wino4LinTile td = wino4_lin_tile(16 * u + ((lane >> 1) & 15), tcols, trows, total_tiles, o.b0);
#pragma unroll
  for (int i = 0; i < Wino4LinGeom::PINSTR; ++i) off[i] = wino4_lin_patch_lane(i, td, H, W, CIN, lane);
  return o;
}
// the same for a RUN-shaped unit (emb_winograd4_geom.h): 15 lane offsets, the lane's tile and its patch slot
__device__ __forceinline__ Wino4LinUnit wino4_run_unit(int u, int tcols, int trows, int total_tiles, int H, int W,
                                                       int CIN, int nimg, int lane, int (&off)[Wino4RunGeom::PINSTR],
                                                       int* slot) {
  Wino4LinUnit o;
  int b0;
  const Wino4Runs R = wino4_runs(u, tcols, trows, &b0);
  o.b0 = __builtin_amdgcn_readfirstlane(b0);
  const int t = lane & 15;
  o.tc = wino4_run_tile(R, t, u, total_tiles);
  *slot = t + wino4_run_of_tile(R, t);
  wino4_run_patch_lanes(off, R, H, W, CIN, nimg - o.b0, lane);
  return o;
}
// staging context of a tile-linear unit: the descriptor starts at the unit's first image and runs to the end of the
// tensor (clamped to 2^31 - 1); every halo is an out-of-bounds lane offset, `keep` passes everything through
__device__ __forceinline__ Wino4Ctx wino4_lin_ctx(const float* __restrict__ X, int H, int W, int CIN, int nimg, int b0,
                                                  int n0) {
  using G = Wino4LinGeom;
  const long img = (long)H * W * CIN;
  const long long rest = (long long)(nimg - b0) * img * 4;
  Wino4Ctx c;
  c.xp = X + (long)b0 * img;
  c.xnum = (int)(rest > 0x7fffffffLL ? 0x7fffffffLL : rest);
  c.keep = -1;
  c.usoff = (n0 / W_BN) * (CIN / G::CB) * G::USLAB_BYTES;
  return c;
}

// LIN: tile-linear units (16 consecutive tiles of the raster order; `cgroups` = tile columns, `num_units` = total
// tiles, `nimg` = images) instead of row-shaped ones -- see emb_winograd4_geom.h and launch_wino4.
// MODE: 0 row-shaped units, 1 tile-linear units with tile-private patches, 2 tile-linear units in RUN shape
template <bool HAS_R, int MODE>
__global__ __launch_bounds__(256, 1) void k_conv3x3_wino4(
    const float* __restrict__ X, int H, int W, int CIN, const float* __restrict__ U,
    const float* __restrict__ shift, const float* __restrict__ R, float* __restrict__ Y, int COUT, int relu,
    int cgroups, int trows, int num_units, int num_groups, int n_tiles, int total_work, int nimg, int xranges,
    int* __restrict__ counters) {
  // the late stage barrier pays for the row-shaped units only (-0.8 %; the linear forms: +0.4 .. +1.3 %), and with the
  // barrier in FRONT of the transform the linear forms learn their next unit early enough to compute its lane offsets
  // before the transform's 144 live registers exist
  constexpr bool LATE = PA_W4_LATE_BARRIER && MODE == 0;
  constexpr bool LIN = MODE != 0;    // a unit = 16 consecutive tiles of the raster order (per-lane tiles)
  constexpr bool RUN = MODE == 2;    // ... whose patch keeps the row-shaped layout, run by run
  using G = std::conditional_t<MODE == 0, Wino4Geom, std::conditional_t<MODE == 1, Wino4LinGeom, Wino4RunGeom>>;
  constexpr int PIN = G::PINSTR;
  constexpr int W4_PIECES = PIN + W4_U_PIECES;
  static_assert(W4_PIECES <= 27, "wino4_piece_asm_n spells out 27 cases");
  // (the epilogue below issues W4_TILE_OUT stores per channel group, and as many residual loads with a residual)
  static_assert(2 * W4_TILE_OUT * (HAS_R ? 2 : 1) >= W4_TAIL_WAIT, "a tile's first-stage wait counts epilogue operations");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int t = lane & 15, g = lane >> 4;
  const int slw = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* my_patch = smem4 + slw * G::PATCH_BYTES;
  unsigned char* ubufs = smem4 + 4 * G::PATCH_BYTES;
  int* mail = reinterpret_cast<int*>(smem4 + G::LDS_BYTES);

  const TileQueue tq{counters, (int)(blockIdx.x & 7), total_work >> 3};
  if (tid == 0) mail[0] = tq_resolve(tq, tq_claim_own(tq));
  __syncthreads();
  int q = mail[0];
  if (q < 0) {
    if (tid == 0) tq_done(tq, gridDim.x);
    return;
  }
  const int x0_last = LIN ? 0 : (cgroups - 1) * Wino4Geom::TW;
  Wino4Lanes<PIN> pl;      // row-shaped units: per kernel; tile-linear: the CURRENT unit's lane offsets
  Wino4Lanes<PIN> npl;     // tile-linear: the next unit's (computed in the current tile's last stage)
  if constexpr (!LIN) wino4_patch_lanes(pl.a, W, CIN, lane, x0_last);
#define W4_PATCH_K(i, j) (MODE == 1 ? wino4_lin_patch_k(i, j) : (MODE == 2 ? wino4_run_patch_k(i, j) : wino4_patch_k(i, j)))
  // transform read bases of this lane (run-shaped units: of its slot in the CURRENT unit, refreshed per unit)
  int pbase0 = MODE == 1 ? wino4_lin_patch_base(t, g) : wino4_patch_base(t, g, 0);
  int pbase1 = MODE == 1 ? pbase0 : wino4_patch_base(t, g, 1);
  int nxt_slot = 0;
  // (tile-linear: the kernel's `num_units` argument carries the number of TILES)
  const int last_unit = LIN ? ((num_units + 15) >> 4) - 1 : 0;
  const int ubase = wino4_u_base(t, g);
  const int lane16 = lane * 16;
  W4Const kc;
  {
    const float p4 = w4_opaque(4.f), m4 = w4_opaque(-4.f), m5 = w4_opaque(-5.f), p2 = w4_opaque(2.f),
                m2 = w4_opaque(-2.f), m1 = w4_opaque(-1.f);
    kc.p4 = f32x2{p4, p4}; kc.m4 = f32x2{m4, m4}; kc.m5 = f32x2{m5, m5};
    kc.p2 = f32x2{p2, p2}; kc.m2 = f32x2{m2, m2}; kc.m1 = f32x2{m1, m1};
  }
  const int nstages = CIN / G::CB;

  // this wave's unit of the claimed group (a group past the end / a unit past the last one is computed on the last
  // real unit's data and not stored: valid = 0)
  Wino4Work wk = wino4_decode(q, n_tiles, num_groups, xranges);
  Wino4Unit cur = Wino4Unit{0, 0, 0, 0}, nxt = cur;
  Wino4LinUnit lcur = Wino4LinUnit{0, Wino4LinTile{0, 0, 0, 0}}, lnxt = lcur;
  int cur_n0 = wk.n0, nxt_n0 = wk.n0;
  Wino4Ctx cctx;
  if constexpr (RUN) {
    const int uu = wk.unit0 + slw;
    int slot;
    lcur = wino4_run_unit(uu < last_unit ? uu : last_unit, cgroups, trows, num_units, H, W, CIN, nimg, lane, pl.a, &slot);
    lcur.tc.valid &= wk.valid & (uu <= last_unit);
    pbase0 = wino4_run_patch_base(slot, g, 0);
    pbase1 = wino4_run_patch_base(slot, g, 1);
    cctx = wino4_lin_ctx(X, H, W, CIN, nimg, lcur.b0, cur_n0);
  } else if constexpr (LIN) {
    lcur = wino4_lin_unit(wk.unit0 + slw, cgroups, trows, num_units, H, W, CIN, lane, pl.a);
    lcur.tc.valid &= wk.valid;
    cctx = wino4_lin_ctx(X, H, W, CIN, nimg, lcur.b0, cur_n0);
  } else {
    cur = wino4_unit(wk.unit0 + slw, cgroups, trows, num_units);
    cur.valid &= wk.valid;
    cctx = wino4_ctx(X, H, W, CIN, cur, cur_n0, x0_last);
  }
  nxt = cur;
  lnxt = lcur;
  Wino4Ctx nctx = cctx;
  int buf = 0;
  wino4_issue_all<PIN>(wino4_stage(cctx, U, COUT, CIN, 0, my_patch, ubufs), pl, lane, slw);
  int claim = 0;
  if (tid == 0) claim = tq_claim_own(tq);
  int nq = -1;
  bool first_tile = true;
  f32x4 acca[W4_AGPR_POINTS][2];        // points 0 .. 31: AccVGPRs
  f32x4 accv[36 - W4_AGPR_POINTS][2];   // points 32 .. 35: architectural registers

#if PA_W4_STAMP
  unsigned long long st_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int st_iter = 0;
#endif
  while (true) {
    for (int s = 0; s < nstages; ++s) {
      W4_STAMP(0);
      // LATE BARRIER (round 5).  A stage needs its own PATCH for the input transform and everybody's U pieces only for
      // the MFMA run.  The patch pieces of a stage are issued in front of its U pieces and loads complete in order, so
      // vmcnt(9) -- this wave's 9 U pieces may still fly -- is "my patch has landed"; the full wait and the workgroup
      // barrier (everybody's U pieces have landed, everybody is done with the other U buffer) move behind the
      // transform's column pass and first rows, which hide what is left of the U flight and of the waves' skew.  A
      // tile's first stage still waits for everything here: the epilogue's stores and residual loads were issued
      // behind its staging.  (PA_W4_LATE_BARRIER=0: round 4's order, barrier in front of the transform.)
      if constexpr (LATE) {
      if (s == 0) {
        // A tile's first stage: its staging was issued from inside the PREVIOUS tile's last MFMA run, i.e. in front of
        // that tile's epilogue, whose last 32 vector memory operations per lane are 32 stores (without a residual) or
        // the 16 residual loads + 16 stores of channel group 1 (with one) -- all newer than the staging, and memory
        // operations complete in order: vmcnt(32) = "the staging has landed" without waiting for the acknowledgement of
        // the stores (~2 k cycles per tile).  (The next group's claim -- an atomic of thread 0 -- is issued in FRONT of
        // the epilogue for that reason.)  The first tile of a workgroup has nothing but its staging in flight.
        if (PA_W4_STORES_IN_FLIGHT && !first_tile) asm volatile("s_waitcnt vmcnt(" W4_STR(W4_TAIL_WAIT_LIT) ")" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(" W4_STR(W4_U_PIECES_LIT) ")" ::: "memory");   // (the U pieces are the NEWEST of the stage)
      }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this stage's images have landed (issued a stage ago)
      }
      W4_STAMP(1);
      // the claim of the NEXT group was issued at the start of this tile: its value is picked up here, behind the
      // wait above (anywhere else the compiler's own vmcnt wait for it would also wait for staging in flight), and
      // published by the barrier of the tile's last stage
      if (s == nstages - 1 && tid == 0) mail[0] = tq_resolve(tq, claim);
      // the first column of the transform's patch reads goes out in FRONT of everything else (the patch is this wave's
      // own: its DMA has landed with the wait above); the LDS wait in front of them covers the mailbox store
      f32x2 x_first[6];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 6; ++i)
        x_first[i] = w4_lds_read64(my_patch + pbase0 + W4_PATCH_K(i, 0));
      unsigned char* umine = ubufs + buf * G::USLAB_BYTES;
      unsigned char* uother = ubufs + (buf ^ 1) * G::USLAB_BYTES;
      // what the MFMA run below stages: the next stage of this unit, or the first stage of the next group's unit
      bool stage_next = true;
      Wino4Stage nst;
      // the stage barrier + everything that needs it (the mailbox of the tile's last stage)
      auto stage_barrier = [&]() {
        if constexpr (LATE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's U pieces
        __builtin_amdgcn_s_barrier();                       // ... everybody's; and everybody is done with the other buffer
        asm volatile("" ::: "memory");
        W4_STAMP(2);
        if (s + 1 < nstages) {
          nst = wino4_stage(cctx, U, COUT, CIN, s + 1, my_patch, uother);
        } else {
          nq = mail[0];   // (written by thread 0 in front of this stage's barrier)
          stage_next = nq >= 0;
          wk = wino4_decode(stage_next ? nq : 0, n_tiles, num_groups, xranges);
          nxt_n0 = wk.n0;
          if constexpr (LIN) {
            // the next unit: its lane offsets and this lane's tile of it -- here, in front of the transform
            const int uu = wk.unit0 + slw;
            if constexpr (RUN) {
              lnxt = wino4_run_unit(uu < last_unit ? uu : last_unit, cgroups, trows, num_units, H, W, CIN, nimg, lane,
                                    npl.a, &nxt_slot);
              lnxt.tc.valid &= wk.valid & (uu <= last_unit);
            } else {
              lnxt = wino4_lin_unit(uu, cgroups, trows, num_units, H, W, CIN, lane, npl.a);
              lnxt.tc.valid &= wk.valid;
            }
            nctx = wino4_lin_ctx(X, H, W, CIN, nimg, lnxt.b0, nxt_n0);
          } else {
            nxt = wino4_unit(wk.unit0 + slw, cgroups, trows, num_units);
            nxt.valid &= wk.valid;
            nctx = wino4_ctx(X, H, W, CIN, nxt, nxt_n0, x0_last);
          }
          nst = wino4_stage(nctx, U, COUT, CIN, 0, my_patch, uother);
        }
      };
      if constexpr (!LATE) stage_barrier();
      // ---- input transform V = B^T d B of this lane's tile and channel pair, in registers
      f32x2 v[6][6];
      f32x2 uf_first[2][2];   // (U fragments of the first point pair, read inside the transform)
      {
        const unsigned char* pb0 = my_patch + pbase0;
        const unsigned char* pb1 = my_patch + pbase1;
        f32x2 tt[6][6];
        // columns of d: tt[.][j] = B^T d[.][j]; column j + 1 is read while column j is combined (left alone, the
        // scheduler issues all 36 reads first and the 72 extra registers spill).  (Two columns between the barriers
        // were measured SLOWER: 2 745 instead of 1 700 cycles for setup + transform.)
        f32x2 x[2][6];
#define W4_RD(i, j) w4_lds_read64(((j) >> 2 ? pb1 : pb0) + W4_PATCH_K(i, j))
#pragma unroll
        for (int i = 0; i < 6; ++i) x[0][i] = x_first[i];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if (j + 1 < 6) {
#pragma unroll
            for (int i = 0; i < 6; ++i) x[(j + 1) & 1][i] = W4_RD(i, j + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x2 y[6];
          wino4_bt(x[j & 1], y, kc);
#pragma unroll
          for (int i = 0; i < 6; ++i) tt[i][j] = y[i];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {   // rows: v[i][.] = B^T tt[i][.] (one row between two scheduling barriers)
          if (i == 5) {
            if constexpr (LATE) {
              stage_barrier();
              __builtin_amdgcn_sched_barrier(0);
            }
            // the U fragments of the MFMA run's first point pair go out in front of the transform's last row(s):
            // their LDS latency no longer opens the run
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int cg = 0; cg < 2; ++cg) uf_first[e][cg] = w4_lds_read64(umine + ubase + wino4_u_k(e, cg));
            __builtin_amdgcn_sched_barrier(0);
          }
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
      int plm[PIN];
      if constexpr (LIN) {
#pragma unroll
        for (int i = 0; i < PIN; ++i) plm[i] = s + 1 < nstages ? pl.a[i] : npl.a[i];
      } else {
#pragma unroll
        for (int i = 0; i < PIN; ++i) plm[i] = pl.a[i] & nst.keep;
      }
      const Wino4Dma dma{w4_lds_addr(my_patch), w4_lds_addr(uother) + 1024u * slw, nst.usoff + 1024 * slw};
      auto mfma_run = [&](auto first_stage) {
        constexpr bool FIRST = decltype(first_stage)::value;
        const unsigned char* ub = umine + ubase;
        f32x2 uf[2][2][2];   // [pair parity][point of the pair][channel group]
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int cg = 0; cg < 2; ++cg) {
            uf[0][e][cg] = uf_first[e][cg];
          }
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
            if ((m == 0 || m == 2) && xp + 2 < 36) {      // U fragments of the next pair, one point per slot
              const int en = m >> 1;
#pragma unroll
              for (int c2 = 0; c2 < 2; ++c2)
                uf[par ^ 1][en][c2] = w4_lds_read64(ub + wino4_u_k(xp + 2 + en, c2));
              __builtin_amdgcn_sched_barrier(0);
            }
            if ((m == 4 || m == 6) && stage_next) {       // wave-uniform; pieces xp, xp + 1 of the next stage
              const int piece = xp + ((m - 4) >> 1);
              if (piece < W4_PIECES) wino4_piece_asm_n<PIN>(piece, nst, dma, plm, lane16);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      };
      if (s == 0) mfma_run(std::true_type{});
      else mfma_run(std::false_type{});
      W4_STAMP(4);
      buf ^= 1;
      if (s + 1 < nstages) {
        W4_STAMP(5);
        W4_STAMP_FLUSH();
      }
    }
    // (the claim of the group after the next one: an atomic -- in front of the epilogue's stores, see the first-stage wait)
    if (tid == 0 && nq >= 0) claim = tq_claim_own(tq);
    first_tile = false;
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
      // row-shaped unit: one image, rows past its end fall out of the descriptor by themselves.  Tile-linear unit: the
      // descriptor runs from the unit's first image to the end of the tensor and every lane knows how many output
      // rows of ITS tile exist (`rows_ok`; the rows below belong to the next image).
      const long oimg = (long)H * W * COUT;
      const int ob = LIN ? lcur.b0 : cur.b;
      const long long orest = (long long)(LIN ? nimg - ob : 1) * oimg * 4;
      const int onum = (int)(orest > 0x7fffffffLL ? 0x7fffffffLL : orest);
      const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(Y + (long)ob * oimg, 0, onum, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(HAS_R ? R + (long)ob * oimg : Y), 0, onum, 0x00020000);
      constexpr int OOB = (int)0x80000000;
      const int xl = LIN ? lcur.tc.x : cur.x0 + 4 * t;
      const int yl = LIN ? lcur.tc.b * H + lcur.tc.y : cur.y0;
      const int rows_ok = LIN ? H - lcur.tc.y : 4;
      const int tile_ok = LIN ? lcur.tc.valid : cur.valid;
      const int obase = ((yl * W + xl) * COUT + cur_n0 + 4 * g) * 4;
      const int srow = W * COUT * 4, spix = COUT * 4;
      int offq[4];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) offq[qq] = (tile_ok && xl + qq < W) ? obase + qq * spix : OOB;
      // byte offset of output (row p, column qq) of this lane's tile, channel group cg -- or out of bounds
      auto ooff = [&](const int p, const int qq, const int cg) {
        if constexpr (LIN) return (offq[qq] == OOB || p >= rows_ok) ? OOB : offq[qq] + p * srow + 64 * cg;
        else return offq[qq] == OOB ? OOB : offq[qq] + p * srow + 64 * cg;
      };
      const float lo = relu ? 0.f : -__builtin_inff();
      const f32x4 lo4 = {lo, lo, lo, lo};
      // Every accumulator is read ONCE, column by column of the 6x6 point grid: y = A^T M[:, b] (10 operations),
      // then its 18 non-zero contributions AT[q][b] y[p] go straight into the 4x4 outputs.  (Two passes over the
      // accumulators -- rows 0-1, then rows 2-3 -- made the compiler keep every accumulator's VGPR copy alive
      // between them: 180 registers, 70-100 spills, 17 k cycles per tile.)
#define W4_ACC(xi) ((xi) < W4_AGPR_POINTS ? acca[(xi) < W4_AGPR_POINTS ? (xi) : 0][cg] \
                                          : accv[(xi) >= W4_AGPR_POINTS ? (xi)-W4_AGPR_POINTS : 0][cg])
      // A 16-byte store of 16 scattered 64-byte segments holds its wave ~200 cycles (the write path's back-pressure,
      // not issue work).  The 16 stores of channel group 0 are therefore HELD and issued one by one from inside the
      // column arithmetic of channel group 1 (three behind each of its first four columns, two behind the last two),
      // where those cycles are filled with vector work; group 1's own stores have nothing left to hide under.  (With a
      // residual the 64 held registers spill; an epilogue in four half passes over output rows {0,1} / {2,3} -- 8 held
      // stores at a time -- fits but reads every accumulator twice: measured 3-5 % SLOWER, profiles/r4_wino4_held_stores.txt.)
      constexpr bool DEFER = PA_W4_DEFER_STORES == 2 || (PA_W4_DEFER_STORES == 1 && !HAS_R);
      f32x4 held[4][4];
      auto store_held = [&](const int k) {   // (k: compile-time after unrolling)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, held[k >> 2][k & 3]), ysrd,
                                               ooff(k >> 2, k & 3, 0), 0, PA_W4_STORE_AUX);
      };
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        // BN shift of this lane's four channels, through the SCALAR cache (uniform address, lgkmcnt): a vector load
        // here would make the epilogue wait on vmcnt -- behind the next tile's staging and this tile's stores
        f32x4 sh;
        {
          const float* sp = shift + cur_n0 + 16 * cg;
          float s16[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) s16[i] = sp[i];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sh[r] = g == 0 ? s16[r] : (g == 1 ? s16[4 + r] : (g == 2 ? s16[8 + r] : s16[12 + r]));
        }
        f32x4 rv[4][4];
        if (HAS_R) {   // rows 0-1 now, rows 2-3 half way through the columns: their latency hides under the arithmetic
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
              rv[p][qq] = __builtin_bit_cast(
                  f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrd, ooff(p, qq, cg), 0, PA_W4_RES_AUX));
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
          // (two columns between barriers: one column alone is a dependency chain a single wave cannot fill)
          if (b & 1) __builtin_amdgcn_sched_barrier(0);
          if (DEFER && cg == 1) {
            constexpr int first[7] = {0, 3, 6, 9, 12, 14, 16};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = first[b]; k < first[b + 1]; ++k) store_held(k);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (HAS_R && b == 3) {
#pragma unroll
            for (int p = 2; p < 4; ++p)
#pragma unroll
              for (int qq = 0; qq < 4; ++qq)
                rv[p][qq] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrd, ooff(p, qq, cg), 0, PA_W4_RES_AUX));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#if PA_W4_STAMP
        asm volatile("s_nop 0" ::"v"(o[3][3]), "v"(o[0][0]));
        st_[6 + 2 * cg] = __builtin_amdgcn_s_memtime();
#endif
        // all 16 outputs of the channel group are finished FIRST and then stored from 16 different register quads: a
        // store reads its data when the memory pipe gets to it, and a vector write that recycles the same registers
        // for the next output waits for that (first build: one quad for all 16 stores, 3 300 cycles per group)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            o[p][qq] = o[p][qq] + sh;
            if (HAS_R) o[p][qq] = o[p][qq] + rv[p][qq];
            o[p][qq] = __builtin_elementwise_max(o[p][qq], lo4);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (DEFER && cg == 0) {
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) held[p][qq] = o[p][qq];
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[p][qq]), ysrd, ooff(p, qq, cg), 0,
                                                     PA_W4_STORE_AUX);
        }
        __builtin_amdgcn_sched_barrier(0);
#if PA_W4_STAMP
        st_[7 + 2 * cg] = __builtin_amdgcn_s_memtime();
#endif
      }
#undef W4_ACC
    }
    W4_STAMP(5);
    W4_STAMP_FLUSH();
    if (nq < 0) break;
    cur = nxt;
    if constexpr (LIN) {
      lcur = lnxt;
      pl = npl;
      if constexpr (RUN) {
        pbase0 = wino4_run_patch_base(nxt_slot, g, 0);
        pbase1 = wino4_run_patch_base(nxt_slot, g, 1);
      }
    }
    cur_n0 = nxt_n0;
    cctx = nctx;
  }
  if (tid == 0) tq_done(tq, gridDim.x);
}

// Which unit form a launch takes: 0 row-shaped, 1 tile-linear with tile-private patches, 2 tile-linear in run shape.
// Row-shaped units pad every tile row to whole groups of 16 tiles; the tile-linear forms pad only the end of the
// launch, and cost more per unit (measured, profiles/r5_wino4_linear_units.txt and r5_wino4_lin_anatomy.txt):
//   tile-private  18 instead of 13 patch pieces and 1.45x the patch bytes per stage (no shared halos): 1.14x (256
//                 channels) to 1.24x (64 channels) per unit -- taken when it leaves at least 1.2x fewer units;
//   run-shaped    the shared halos of the row-shaped unit inside every run (15 pieces, 1.1x the bytes): 1.5-3.5 % less
//                 per unit than the tile-private form, i.e. 1.14x - 1.22x a row-shaped unit -- taken when it leaves at
//                 least 1.18x fewer units and the map has at least 5 tiles per row (a unit then spans at most 4 runs).
// The maps of 10 s chunks (125 / 63 / 32 tiles per row) keep the row-shaped units; those of 3 s segments (38 / 19 / 10)
// take the run-shaped ones.  PA_WINO4_LINEAR=0 / 1 / 2 forces a form (A/B aid).
static int wino4_unit_mode(int B, int H, int W, int rows) {
  if (rows != H) return 0;                          // (row ranges: the row-shaped kernel)
  const long tcols = cdiv(W, 4), trows = cdiv(H, 4);
  static const char* e = getenv("PA_WINO4_LINEAR");
  if (e != nullptr) {
    const int m = atoi(e);
    return m == 2 && tcols < 5 ? 1 : m;
  }
  const long lin_units = cdiv((long)B * trows * tcols, 16), row_units = (long)B * trows * cdiv(W, 64);
  if (tcols >= 5 && lin_units * 118 <= row_units * 100) return 2;
  return lin_units * 120 <= row_units * 100 ? 1 : 0;
}

// tile order: contiguous ranges per XCD (neighbouring tiles share their halos in one L2) or round-robin;
// PA_XCD_RANGES=0 / 1 forces one of them for every convolution kernel (A/B aid)
int xcd_ranges_wanted(bool by_default) {
  static const char* e = getenv("PA_XCD_RANGES");
  return e != nullptr ? (atoi(e) != 0) : (by_default ? 1 : 0);
}

template <bool HAS_R, int MODE>
static int launch_wino4(const float* X, int B, int H, int W, int CIN, const float* U, const float* shift,
                        const float* R, float* Y, int COUT, int relu, int rows, hipStream_t st) {
  constexpr bool LIN = MODE != 0;
  using G = std::conditional_t<MODE == 0, Wino4Geom, std::conditional_t<MODE == 1, Wino4LinGeom, Wino4RunGeom>>;
  // row-shaped: column groups of 64 pixels x tile rows (rows < H: the tile rows that cover them only);
  // tile-linear: tile columns x tile rows, units of 16 consecutive tiles
  const int cgroups = LIN ? cdiv(W, 4) : cdiv(W, Wino4Geom::TW), trows = cdiv(rows, 4);
  const size_t lds = (size_t)G::LDS_BYTES + 16;
  auto kernel = k_conv3x3_wino4<HAS_R, MODE>;
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
  const int n_tiles = COUT / W_BN;
  const long total_tiles = (long)cgroups * trows * B;                       // (LIN)
  const long num_units = LIN ? (total_tiles + 15) / 16 : (long)cgroups * trows * B;
  const long num_groups = (num_units + 3) / 4;
  const long total = ((num_groups + 7) / 8) * 8 * n_tiles;   // padded to whole XCD stripes
  const int resident = cus_of[dev] & ~7;                      // one workgroup per CU
  const int grid = (int)(total < resident ? total : resident);
  int* counters = tile_counters();
  if (counters == nullptr) {
    set_error("pa_conv3x3_wino4: cannot allocate the tile counters");
    return 2;
  }
  // (tile-linear: the kernel's `num_units` argument carries the number of TILES)
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, st, X, H, W, CIN, U, shift, R, Y, COUT, relu, cgroups, trows,
                     (int)(LIN ? total_tiles : num_units), (int)num_groups, n_tiles, (int)total, B,
                     xcd_ranges_wanted(n_tiles < 8), counters);
  return 0;
}

}  // namespace pa

extern "C" {

#if PA_W4_STAMP
int pa_wino4_read_stamps(unsigned long long* host) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pa::g_w4_stamps), sizeof(unsigned long long) * 8 * 4 * 64 * 10) ==
                 hipSuccess
             ? 0
             : 1;
}
#endif

// conv3x3, stride 1, pad 1, via Winograd F(4x4,3x3): Y = [relu](conv(X) + shift [+ R]).  X, R, Y: NHWC float32.
// U: G g G^T (BatchNorm scale folded) in the slab layout of weights.winograd4_pack / pa_winograd4_pack_host:
// [cout/32][cin/8][row = 32 xi + n][8], xi = 6a + b.
int pa_conv3x3_wino4_rows(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                          const float* R, float* Y, int cout, int relu, int rows, void* stream);

int pa_conv3x3_wino4(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                     const float* R, float* Y, int cout, int relu, void* stream) {
  return pa_conv3x3_wino4_rows(X, B, H, W, cin, U, shift, R, Y, cout, relu, H, stream);
}

// the same convolution for the output rows 0 .. rows - 1 only (rows == H, or a multiple of 4 below it: the rest of
// the map is somebody else's -- pa_emb_forward gives the last two rows of a map whose height is 2 (mod 4) to the
// F(2x2) kernel instead of padding them to a tile row that is half empty: 10-row maps, 12 -> 10 rows of work)
int pa_conv3x3_wino4_rows(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                          const float* R, float* Y, int cout, int relu, int rows, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  PA_REQUIRE(rows == H || (rows < H && rows % 4 == 0), "pa_conv3x3_wino4_rows: rows must be H or a multiple of 4 below it");
  PA_REQUIRE(cin % 8 == 0 && cin >= 32 && cout % pa::W_BN == 0,
             "pa_conv3x3_wino4: cin %% 8 == 0, cin >= 32 and cout %% 32 == 0 required");
  PA_REQUIRE((long)H * W * (cin > cout ? cin : cout) * 4 < (1L << 28),
             "pa_conv3x3_wino4: one image must be smaller than 256 MB");
  // `flops` = the direct convolution's (the reference's operation); the kernel executes 36/144 of them
  pa::ProfScope prof("k_conv3x3_wino4", stream, 2.0 * 9 * cin * cout * (double)B * rows * W,
                     4.0 * ((double)B * rows * W * cin + (double)B * rows * W * cout * (R ? 2 : 1) + 9.0 * cin * cout));
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const int mode = pa::wino4_unit_mode(B, H, W, rows);
  if (mode != 0) {
    // a unit may straddle images: its lane offsets count from the unit's first image
    const long per = (long)pa::cdiv(H, 4) * pa::cdiv(W, 4);
    const long span = (15 / per + 2) * (long)H * W * (cin > cout ? cin : cout) * 4;
    PA_REQUIRE(span < (1L << 31), "pa_conv3x3_wino4: the images a 16-tile unit can touch must stay below 2 GB");
  }
  if (mode == 2)
    rc = R != nullptr ? pa::launch_wino4<true, 2>(X, B, H, W, cin, U, shift, R, Y, cout, relu, rows, st)
                      : pa::launch_wino4<false, 2>(X, B, H, W, cin, U, shift, R, Y, cout, relu, rows, st);
  else if (mode == 1)
    rc = R != nullptr ? pa::launch_wino4<true, 1>(X, B, H, W, cin, U, shift, R, Y, cout, relu, rows, st)
                      : pa::launch_wino4<false, 1>(X, B, H, W, cin, U, shift, R, Y, cout, relu, rows, st);
  else
    rc = R != nullptr ? pa::launch_wino4<true, 0>(X, B, H, W, cin, U, shift, R, Y, cout, relu, rows, st)
                      : pa::launch_wino4<false, 0>(X, B, H, W, cin, U, shift, R, Y, cout, relu, rows, st);
  if (rc != 0) return rc;
  PA_CHECK_LAUNCH("pa_conv3x3_wino4");
  return 0;
}

}  // extern "C"
