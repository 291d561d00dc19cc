// Winograd F(4x4, 3x3), TWO waves per SIMD: the 36 points of a unit split between a PAIR of waves.
// (reference operation: the stride-1 3x3 convolutions of models/embedding/wespeaker/resnet.py:84-145; same C entry
//  point, same U image and same unit geometry as emb_winograd4.hip, which it replaces by default.)
//
// Why: emb_winograd4.hip runs one wave per SIMD (288 accumulators + the transformed patch = all 512 registers), so
// nothing fills the cycles a wave spends waiting -- profiles/r4_wino4_anatomy.txt: 8 600 cycles per stage against
// 4 608 of MFMA + ~1 150 of packed transform arithmetic; the rest is LDS read latency, LDS-DMA issue, the barrier and
// store acknowledgements.  Halving the OUTPUT channels per wave would halve the accumulators but every wave would
// still transform the whole patch (72 registers, twice the arithmetic per SIMD).  Halving the POINTS does both:
//
//     V = B^T d B,  M = sum_cin U .* V,  Y = A^T M A          (per 4x4 output tile)
//
//   wave h = 0 of a pair owns rows {0, 1, 2} of the 6x6 point grid, wave h = 1 rows {5, 3, 4} ("outer, lo, hi"):
//   * input transform: only the first pass depends on the rows (6 of the 12 operations per column; both halves read
//     patch rows 1..4, the outer row reads rows {0,2,4} + h) and the second pass runs on 3 rows instead of 6: 72
//     packed operations per wave and stage instead of 144, 36 transformed values instead of 72;
//   * 18 points x 32 output channels = 144 accumulators (pinned AccVGPRs) + ~110 architectural registers: 256 per
//     wave, two waves per SIMD; 72 MFMAs per wave and stage;
//   * inverse transform: each half reduces its own three rows along the columns (r = M[row, :] A), forms its partial
//     of all four output rows, keeps two of them (h = 0: rows 0-1, h = 1: rows 2-3) and hands the other two to its
//     partner through LDS (8 float4 per lane and 16-channel group); each half stores 16 of the unit's 32 rows x
//     channel groups.  Sums of two partials instead of one chain: last-bit differences against emb_winograd4.hip,
//     deterministic.
// The pair shares the unit's patch (one more barrier per stage: both halves must have read it before the next
// stage's patch is staged over it) and splits the staging: 11 LDS-DMA pieces per wave and stage instead of 22.
// LDS: 4 patches + 2 U slabs as before + 32 KB of exchange space (the other 32 KB of it are the U slab of the tile's
// last stage, free once every wave is behind its last MFMA run) = 159 760 B, one 8-wave workgroup per CU.
#include <stdlib.h>

#include <type_traits>

#include "emb_winograd4_dev.h"

namespace pa {

#ifndef PA_W4P_STAMP
#define PA_W4P_STAMP 0
#endif
#if PA_W4P_STAMP
// development instrumentation (never in the product build): s_memtime at the phases of the first 64 stages of
// workgroups 0 .. 7, per wave; read back with pa_wino4p_read_stamps
__device__ unsigned long long g_w4p_stamps[8 * 8 * 64 * 10];
#define W4P_STAMP(p) st_[p] = __builtin_amdgcn_s_memtime()
#define W4P_STAMP_FLUSH()                                                                     \
  do {                                                                                        \
    if (blockIdx.x < 8 && st_iter < 64 && lane == 0) {                                         \
      _Pragma("unroll") for (int p_ = 0; p_ < 10; ++p_)                                        \
          g_w4p_stamps[((blockIdx.x * 8 + wv) * 64 + st_iter) * 10 + p_] = st_[p_];            \
    }                                                                                         \
    ++st_iter;                                                                                \
  } while (0)
#else
#define W4P_STAMP(p)
#define W4P_STAMP_FLUSH()
#endif

struct Wino4pGeom {
  static constexpr int MAIL_OFF = Wino4Geom::LDS_BYTES;
  static constexpr int SPARE_OFF = MAIL_OFF + 16;
  static constexpr int LDS_BYTES = SPARE_OFF + 8 * 2048;   // 143 376
  static constexpr int POINTS = 18;                        // per wave
  static constexpr int APOINTS = 16;                       // ... of them in AccVGPRs
  static constexpr int SLOTS = 12;                         // LDS-DMA pieces per wave and stage (11 or 12 real ones)
};

template <bool HAS_R>
__global__ __launch_bounds__(512) void k_conv3x3_wino4p(
    const float* __restrict__ X, int H, int W, int CIN, const float* __restrict__ U,
    const float* __restrict__ shift, const float* __restrict__ R, float* __restrict__ Y, int COUT, int relu,
    int cgroups, int trows, int num_units, int num_groups, int n_tiles, int total_work,
    int* __restrict__ counters) {
  using G = Wino4Geom;
  using P = Wino4pGeom;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem4p[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int t = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // waves w and w + 4 share a SIMD: group A = waves 0..3, group B = waves 4..7 run the same program ONE TICK apart
  // (a tick = the stretch between two workgroup barriers: an input transform, an MFMA run or a quarter of the
  // epilogue), so that on every SIMD one wave's MFMA run covers the other wave's transform / epilogue waits.
  // A pair (two neighbouring waves of one group, on different SIMDs) owns one unit.
  const int grp = wv >> 2, pr = wv >> 1, h = wv & 1;
  unsigned char* my_patch = smem4p + pr * G::PATCH_BYTES;
  unsigned char* ubufs = smem4p + 4 * G::PATCH_BYTES;
  int* mail = reinterpret_cast<int*>(smem4p + P::MAIL_OFF);
  unsigned char* spare = smem4p + P::SPARE_OFF;

  const TileQueue tq{counters, (int)(blockIdx.x & 7), total_work >> 3};
  if (tid == 0) mail[0] = tq_resolve(tq, tq_claim_own(tq));
  __syncthreads();
  int q = mail[0];
  if (q < 0) {
    if (tid == 0) tq_done(tq, gridDim.x);
    return;
  }
  __syncthreads();   // (everybody has read the mail before thread 0 can write it again)
  const int x0_last = (cgroups - 1) * G::TW;
  // this wave's share of the pair's 13 patch pieces: h = 0 -> 0..5 and 12, h = 1 -> 6..11
  int pla[7];
#pragma unroll
  for (int i = 0; i < 6; ++i) pla[i] = wino4_patch_lane(6 * h + i, W, CIN, lane, x0_last);
  pla[6] = wino4_patch_lane(12, W, CIN, lane, x0_last);
  const int pbase = wino4_patch_base(t, g);
  const int ubase = wino4_u_base(t, g);
  // rows of the point grid this half owns, in the order (outer, lo, hi)
  const int ig0 = h ? 5 : 0, ig1 = h ? 3 : 1, ig2 = h ? 4 : 2;
  W4Const kc;
  {
    const float p4 = w4_opaque(4.f), m4 = w4_opaque(-4.f), m5 = w4_opaque(-5.f), p2 = w4_opaque(2.f),
                m2 = w4_opaque(-2.f), m1 = w4_opaque(-1.f);
    kc.p4 = f32x2{p4, p4}; kc.m4 = f32x2{m4, m4}; kc.m5 = f32x2{m5, m5};
    kc.p2 = f32x2{p2, p2}; kc.m2 = f32x2{m2, m2}; kc.m1 = f32x2{m1, m1};
  }
  const int nstages = CIN / G::CB;

  Wino4Work wk = wino4_decode(q, n_tiles, num_groups);
  Wino4Unit cur = wino4_unit(wk.unit0 + pr, cgroups, trows, num_units), nxt = cur;
  cur.valid &= wk.valid;
  int cur_n0 = wk.n0, nxt_n0 = wk.n0;
  Wino4Ctx cctx = wino4_ctx(X, H, W, CIN, cur, cur_n0, x0_last), nctx = cctx;
  int buf = 0;

  // One of this wave's staging pieces of a stage (SLOT is a compile-time constant at every call site):
  //   slots 0..5: patch pieces 6 h + slot;   slot 6: patch piece 12 (h = 0 only);
  //   slots 7..11: U pieces (1 KB each) number wv + 8 (slot - 7) of the slab's 36 (slot 11: waves 0..3 only).
  auto issue_patch_piece = [&](const int SLOT, const Wino4Stage& st) {
    if (SLOT < 6 || h == 0) {
      const int piece = SLOT < 6 ? 6 * h + SLOT : 12;
      const int off = pla[SLOT < 7 ? SLOT : 0] & st.keep;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(st.xsrd, (lds4_ptr_t)(st.pbuf + 1024 * piece), 16, off, 0, 0, 0);
    }
  };
  auto issue_u_piece = [&](const int SLOT, const Wino4Stage& st) {
    const int k = wv + 8 * (SLOT - 7);
    if (SLOT < 11 || wv < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(st.usrd, (lds4_ptr_t)(st.ubuf + 1024 * k), 16, lane * 16,
                                               st.usoff + 1024 * k, 0, 0);
  };
  // end of a tick: this wave's staging has landed, its LDS accesses are done; then everybody's
  auto sync_tick = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  {
    const Wino4Stage st0 = wino4_stage(cctx, U, COUT, CIN, 0, my_patch, ubufs);
#pragma unroll
    for (int i = 0; i < 7; ++i) issue_patch_piece(i, st0);
#pragma unroll
    for (int i = 7; i < P::SLOTS; ++i) issue_u_piece(i, st0);
  }
  int claim = 0;
  if (tid == 0) claim = tq_claim_own(tq);
  int nq = -1;
  // local point 6 il + j, channel group.  The register file of a 2-waves-per-SIMD kernel that uses AccVGPRs is split
  // evenly by the compiler (128 + 128): 16 points live in the AccVGPRs, 2 in architectural registers.
  f32x4 acca[P::APOINTS][2];
  f32x4 accv[P::POINTS - P::APOINTS][2];

#if PA_W4P_STAMP
  unsigned long long st_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int st_iter = 0;
#endif
  sync_tick();                 // the first stage's images have landed
  if (grp == 1) sync_tick();   // group B: one tick behind
  while (true) {
    for (int s = 0; s < nstages; ++s) {
      // ================= tick T(s): this half's three rows of V = B^T d B for the lane's tile and channel pair
      W4P_STAMP(0);
      f32x2 v[3][6];
      {
        const unsigned char* pb = my_patch + pbase;
        f32x2 tt[3][6];
        // first pass, per patch column j (5 reads, 6 operations), column j + 1 read while column j is combined:
        //   h = 0 (rows 0, 1, 2 of B^T): x = d[0..4][j]:  4 x0 - 5 x2 + x4 | (x4 - 4 x2) +- (x3 - 4 x1)
        //   h = 1 (rows 5, 3, 4):        x = d[1..5][j]:  4 x0 - 5 x2 + x4 | (x3 - x1) +- 2 (x2 - x0)
        auto first_pass = [&](auto half) {
          constexpr int HH = decltype(half)::value;
          f32x2 x[2][5];
#define W4P_RD(c, j) (*reinterpret_cast<const f32x2*>(pb + wino4_patch_k((c) + HH, j)))
#pragma unroll
          for (int c = 0; c < 5; ++c) x[0][c] = W4P_RD(c, 0);
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            if (j + 1 < 6) {
#pragma unroll
              for (int c = 0; c < 5; ++c) x[(j + 1) & 1][c] = W4P_RD(c, j + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x2(&xx)[5] = x[j & 1];
            tt[0][j] = w4fma(xx[0], kc.p4, w4fma(xx[2], kc.m5, xx[4]));
            if (HH == 0) {
              const f32x2 Pq = w4fma(xx[2], kc.m4, xx[4]), Qq = w4fma(xx[1], kc.m4, xx[3]);
              tt[1][j] = Pq + Qq;
              tt[2][j] = w4fma(Qq, kc.m1, Pq);
            } else {
              const f32x2 Pq = w4fma(xx[1], kc.m1, xx[3]), Qq = w4fma(xx[0], kc.m1, xx[2]);
              tt[1][j] = w4fma(Qq, kc.p2, Pq);
              tt[2][j] = w4fma(Qq, kc.m2, Pq);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#undef W4P_RD
        };
        if (h == 0) first_pass(std::integral_constant<int, 0>{});
        else first_pass(std::integral_constant<int, 1>{});
#pragma unroll
        for (int il = 0; il < 3; ++il) {
          wino4_bt(tt[il], v[il], kc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#if PA_W4P_STAMP
      asm volatile("s_nop 0" ::"v"(v[2][5]), "v"(v[0][0]));
#endif
      // the claim of the NEXT group (issued at the start of this tile) is published by the barrier in front of
      // group A's last MFMA run; group B reads it one tick later, nobody writes it again before the next tile's
      if (s == nstages - 1 && tid == 0) mail[0] = tq_resolve(tq, claim);
      W4P_STAMP(1);
      sync_tick();
      // ================= tick R(s): the MFMA run, with the staging of stage s + 1 issued from inside it
      W4P_STAMP(2);
      unsigned char* umine = ubufs + buf * G::USLAB_BYTES;
      unsigned char* uother = ubufs + (buf ^ 1) * G::USLAB_BYTES;
      bool stage_u = true;                  // U of the next stage / of the next group's first stage
      const bool stage_patch = s + 1 < nstages;   // (the next GROUP's first patch is staged in the last epilogue tick)
      Wino4Stage nst;
      if (s + 1 < nstages) {
        nst = wino4_stage(cctx, U, COUT, CIN, s + 1, my_patch, uother);
      } else {
        nq = mail[0];
        stage_u = nq >= 0;
        wk = wino4_decode(stage_u ? nq : 0, n_tiles, num_groups);
        nxt = wino4_unit(wk.unit0 + pr, cgroups, trows, num_units);
        nxt.valid &= wk.valid;
        nxt_n0 = wk.n0;
        nctx = wino4_ctx(X, H, W, CIN, nxt, nxt_n0, x0_last);
        nst = wino4_stage(nctx, U, COUT, CIN, 0, my_patch, uother);
      }
      // 18 points x 2 channel groups x 2 k-steps, two points at a time (a dependent MFMA is four MFMAs behind its
      // producer); U fragments of the next pair and one staging piece go behind individual MFMAs
      auto mfma_run = [&](auto first_stage) {
        constexpr bool FIRST = decltype(first_stage)::value;
        const unsigned char* ub[3] = {umine + ubase + 6 * 1024 * ig0, umine + ubase + 6 * 1024 * ig1,
                                      umine + ubase + 6 * 1024 * ig2};
        f32x2 uf[2][2][2];   // [pair parity][point of the pair][channel group]
#define W4P_U(l, cg) (*reinterpret_cast<const f32x2*>(ub[(l) / 6] + 1024 * ((l) % 6) + 512 * (cg)))
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int cg = 0; cg < 2; ++cg) uf[0][e][cg] = W4P_U(e, cg);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xp = 0; xp < P::POINTS; xp += 2) {
          const int par = (xp >> 1) & 1;
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const int ks = m >> 2, e = (m >> 1) & 1, cg = m & 1;
            const int l = xp + e;
            const f32x2 bv = v[l / 6][l % 6];
            const float a = ks ? uf[par][e][cg].y : uf[par][e][cg].x, b = ks ? bv.y : bv.x;
            if (l < P::APOINTS) {
              if (ks == 0 && FIRST) W4_MFMA_A_ZERO(acca[l < P::APOINTS ? l : 0][cg], a, b);
              else W4_MFMA_A(acca[l < P::APOINTS ? l : 0][cg], a, b);
            } else {
              if (ks == 0 && FIRST) W4_MFMA_V_ZERO(accv[l >= P::APOINTS ? l - P::APOINTS : 0][cg], a, b);
              else W4_MFMA_V(accv[l >= P::APOINTS ? l - P::APOINTS : 0][cg], a, b);
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((m == 0 || m == 2) && xp + 2 < P::POINTS) {   // U fragments of the next pair, one point per slot
              const int en = m >> 1;
#pragma unroll
              for (int c2 = 0; c2 < 2; ++c2) uf[par ^ 1][en][c2] = W4P_U(xp + 2 + en, c2);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (m == 3 || m == 5 || m == 7) {   // staging slots 0 .. 11 behind the first four point pairs
              const int slot = 3 * (xp >> 1) + ((m - 3) >> 1);
              if (slot < 7) {
                if (stage_patch) issue_patch_piece(slot < 7 ? slot : 0, nst);
              } else if (slot < P::SLOTS) {
                if (stage_u) issue_u_piece(slot >= 7 && slot < P::SLOTS ? slot : 7, nst);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
#undef W4P_U
      };
      if (s == 0) mfma_run(std::true_type{});
      else mfma_run(std::false_type{});
      buf ^= 1;
      W4P_STAMP(3);
      sync_tick();
      W4P_STAMP(4);
      if (s + 1 < nstages) W4P_STAMP_FLUSH();
    }
    // ================= ticks E1 .. E4: inverse transform.  Own rows along the columns (r = M[row, :] A), partial
    // output rows, two of them handed to the partner through LDS, BN shift (+ residual) (+ ReLU), 16-byte stores of
    // output rows 2 h, 2 h + 1.  Exchange slot of a wave: 8 rows of 1 KB (one float4 per lane each); rows 0..5 lie in
    // the half of the pair's patch block that the PARTNER stages (the partner is also the one who reads them: it
    // stages the next group's patch over them behind its own last read), rows 6..7 in the spare space.
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
      unsigned char* mine_lo = my_patch + (1 - h) * 6144 + lane * 16;        // rows 0..5 of this wave's slot
      unsigned char* mine_hi = spare + wv * 2048 + lane * 16 - 6 * 1024;     // rows 6..7
      const unsigned char* peer_lo = my_patch + h * 6144 + lane * 16;
      const unsigned char* peer_hi = spare + (wv ^ 1) * 2048 + lane * 16 - 6 * 1024;
#define W4P_MINE(row) (*reinterpret_cast<f32x4*>(((row) < 6 ? mine_lo : mine_hi) + 1024 * (row)))
#define W4P_PEER(row) (*reinterpret_cast<const f32x4*>(((row) < 6 ? peer_lo : peer_hi) + 1024 * (row)))
      const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
          Y + (long)cur.b * H * W * COUT, 0, H * W * COUT * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(HAS_R ? R + (long)cur.b * H * W * COUT : Y), 0, H * W * COUT * 4, 0x00020000);
      constexpr int OOB = (int)0x80000000;
      const int xl = cur.x0 + 4 * t;
      const int srow = W * COUT * 4, spix = COUT * 4;
      const int obase = (((cur.y0 + 2 * h) * W + xl) * COUT + cur_n0 + 4 * g) * 4;
      int offq[4];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) offq[qq] = (cur.valid && xl + qq < W) ? obase + qq * spix : OOB;
      const float lo = relu ? 0.f : -__builtin_inff();
      const f32x4 lo4 = {lo, lo, lo, lo};
#define W4P_ACC(l) ((l) < P::APOINTS ? acca[(l) < P::APOINTS ? (l) : 0][cg] \
                                    : accv[(l) >= P::APOINTS ? (l)-P::APOINTS : 0][cg])
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        // ---- tick E1 / E3: partials of channel group cg
        f32x4 sh;   // BN shift of this lane's four channels, through the scalar cache (see emb_winograd4.hip)
        {
          const float* sp = shift + cur_n0 + 16 * cg;
          float s16[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) s16[i] = sp[i];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sh[r] = g == 0 ? s16[r] : (g == 1 ? s16[4 + r] : (g == 2 ? s16[8 + r] : s16[12 + r]));
        }
        f32x4 r[3][4];   // r[il][q] = sum_j M[row il][j] A[j][q]
#pragma unroll
        for (int il = 0; il < 3; ++il) {
          // (the empty volatile asm re-defines each accumulator HERE: see emb_winograd4.hip)
#pragma unroll
          for (int j = 0; j < 6; ++j)
            if (6 * il + j < P::APOINTS) asm volatile("" : "+a"(acca[6 * il + j < P::APOINTS ? 6 * il + j : 0][cg]));
          wino4_at(W4P_ACC(6 * il + 0), W4P_ACC(6 * il + 1), W4P_ACC(6 * il + 2), W4P_ACC(6 * il + 3),
                   W4P_ACC(6 * il + 4), W4P_ACC(6 * il + 5), r[il], k4);
          __builtin_amdgcn_sched_barrier(0);
        }
        // partial output rows: Y[p][q] = sum_i A^T[p][i] r_i[q];
        //   h = 0 (rows 0, 1, 2):  Y0 += r0 + s,  Y1 += d,   Y2 += s,    Y3 += d          s = r1 + r2, d = r1 - r2
        //   h = 1 (rows 5, 3, 4):  Y0 += s,       Y1 += 2 d, Y2 += 4 s,  Y3 += 8 d + r5   s = r3 + r4, d = r3 - r4
        f32x4 own[2][4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const f32x4 sm = r[1][qq] + r[2][qq];
          const f32x4 df = w4fma4(r[2][qq], k4.m1, r[1][qq]);
          f32x4 send1;
          if (h == 0) {
            own[0][qq] = r[0][qq] + sm;
            own[1][qq] = df;
            send1 = df;
          } else {
            own[0][qq] = sm * k4.p4;
            own[1][qq] = w4fma4(df, k4.p8, r[0][qq]);
            send1 = df + df;
          }
          W4P_MINE(qq) = sm;
          W4P_MINE(4 + qq) = send1;
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 rv[2][4];
        if (HAS_R) {   // their latency hides under the tick change
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
              rv[k][qq] = __builtin_bit_cast(
                  f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                             rsrd, offq[qq] == OOB ? OOB : offq[qq] + k * srow + 64 * cg, 0, 0));
        }
#if PA_W4P_STAMP
        st_[5 + 2 * cg] = __builtin_amdgcn_s_memtime();
#endif
        sync_tick();
        // ---- tick E2 / E4: the partner's partials, finish, store
#if PA_W4P_STAMP
        st_[6 + 2 * cg] = __builtin_amdgcn_s_memtime();
#endif
        f32x4 got[2][4];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) got[k][qq] = W4P_PEER(4 * k + qq);
        if (cg == 1 && nq >= 0) {
          // the next group's first patch, over the rows just read (this wave's share of the pieces = exactly the
          // rows it reads its partner's partials from, + piece 12 which no slot touches)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const Wino4Stage st0 = wino4_stage(nctx, U, COUT, CIN, 0, my_patch, ubufs);
#pragma unroll
          for (int i = 0; i < 7; ++i) issue_patch_piece(i, st0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            own[k][qq] = own[k][qq] + got[k][qq] + sh;
            if (HAS_R) own[k][qq] = own[k][qq] + rv[k][qq];
            own[k][qq] = __builtin_elementwise_max(own[k][qq], lo4);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, own[k][qq]), ysrd,
                                                   offq[qq] == OOB ? OOB : offq[qq] + k * srow + 64 * cg, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#if PA_W4P_STAMP
        if (cg == 1) st_[9] = __builtin_amdgcn_s_memtime();
#endif
        sync_tick();
      }
#undef W4P_ACC
#undef W4P_MINE
#undef W4P_PEER
    }
    W4P_STAMP_FLUSH();
    if (nq < 0) break;
    cur = nxt;
    cur_n0 = nxt_n0;
    cctx = nctx;
    if (tid == 0) claim = tq_claim_own(tq);
  }
  if (grp == 0) sync_tick();   // group A's last tick (group B is one behind)
  if (tid == 0) tq_done(tq, gridDim.x);
}

template <bool HAS_R>
static int launch_wino4p(const float* X, int B, int H, int W, int CIN, const float* U, const float* shift,
                         const float* R, float* Y, int COUT, int relu, hipStream_t st) {
  using G = Wino4Geom;
  const int cgroups = cdiv(W, G::TW), trows = cdiv(H, G::TH);
  const size_t lds = (size_t)Wino4pGeom::LDS_BYTES;
  auto kernel = k_conv3x3_wino4p<HAS_R>;
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
  const long num_units = (long)cgroups * trows * B;
  const long num_groups = (num_units + 3) / 4;
  const long total = ((num_groups + 7) / 8) * 8 * n_tiles;   // padded to whole XCD stripes
  const int resident = cus_of[dev] & ~7;                      // one workgroup per CU
  const int grid = (int)(total < resident ? total : resident);
  int* counters = tile_counters();
  if (counters == nullptr) {
    set_error("pa_conv3x3_wino4: cannot allocate the tile counters");
    return 2;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds, st, X, H, W, CIN, U, shift, R, Y, COUT, relu, cgroups, trows,
                     (int)num_units, (int)num_groups, n_tiles, (int)total, counters);
  return 0;
}

// called by pa_conv3x3_wino4 (emb_winograd4.hip) unless PA_WINO4_PAIRED=0
int conv3x3_wino4_paired(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                         const float* R, float* Y, int cout, int relu, hipStream_t st) {
  return R != nullptr ? launch_wino4p<true>(X, B, H, W, cin, U, shift, R, Y, cout, relu, st)
                      : launch_wino4p<false>(X, B, H, W, cin, U, shift, R, Y, cout, relu, st);
}

}  // namespace pa

#if PA_W4P_STAMP
extern "C" int pa_wino4p_read_stamps(unsigned long long* host) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pa::g_w4p_stamps), sizeof(unsigned long long) * 8 * 8 * 64 * 10) ==
                 hipSuccess
             ? 0
             : 1;
}
#endif
