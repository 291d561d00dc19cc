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
  static constexpr int EXCH_BYTES = 8192;   // per wave: 8 float4 per lane
  static constexpr int MAIL_OFF = Wino4Geom::LDS_BYTES;
  static constexpr int SPARE_OFF = MAIL_OFF + 16;
  static constexpr int LDS_BYTES = SPARE_OFF + 4 * EXCH_BYTES;   // 159 760
  static constexpr int POINTS = 18;                               // per wave
  static constexpr int APOINTS = 16;                              // ... of them in AccVGPRs
  static constexpr int SLOTS = 11;                                // LDS-DMA pieces per wave and stage
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
  const int pr = wv & 3, h = wv >> 2;   // pair = unit of the group; half = rows of the point grid
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
  f32x2 kA, kB, nkB;   // first pass of the half: P = x4 + kA x2, Q = x3 + kA x1, lo = P + kB Q, hi = P - kB Q
  {
    const float p4 = w4_opaque(4.f), m4 = w4_opaque(-4.f), m5 = w4_opaque(-5.f), p2 = w4_opaque(2.f),
                m2 = w4_opaque(-2.f), m1 = w4_opaque(-1.f);
    kc.p4 = f32x2{p4, p4}; kc.m4 = f32x2{m4, m4}; kc.m5 = f32x2{m5, m5};
    kc.p2 = f32x2{p2, p2}; kc.m2 = f32x2{m2, m2}; kc.m1 = f32x2{m1, m1};
    auto uni = [](float x) {   // (a float select lands in a vector register; the constants must be scalar)
      return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
    };
    const float a = w4_opaque(uni(h ? -1.f : -4.f)), b = w4_opaque(uni(h ? 2.f : 1.f)),
                nb = w4_opaque(uni(h ? -2.f : -1.f));
    kA = f32x2{a, a}; kB = f32x2{b, b}; nkB = f32x2{nb, nb};
  }
  const int nstages = CIN / G::CB;

  Wino4Work wk = wino4_decode(q, n_tiles, num_groups);
  Wino4Unit cur = wino4_unit(wk.unit0 + pr, cgroups, trows, num_units), nxt = cur;
  cur.valid &= wk.valid;
  int cur_n0 = wk.n0, nxt_n0 = wk.n0;
  Wino4Ctx cctx = wino4_ctx(X, H, W, CIN, cur, cur_n0, x0_last), nctx = cctx;
  int buf = 0;

  // one of this wave's 11 staging pieces of a stage (SLOT is a compile-time constant at every call site):
  //   slots 0..5: patch pieces 6 h + slot;  slot 6: patch piece 12 (h = 0) / U piece 4 (h = 1);
  //   slots 7..10: U pieces slot - 7 (h = 0) / slot - 2 (h = 1).   U piece u of the pair = KB number pr + 4 u.
  auto issue_piece = [&](const int SLOT, const Wino4Stage& st) {
    if (SLOT < 6 || (SLOT == 6 && h == 0)) {
      const int piece = SLOT < 6 ? 6 * h + SLOT : 12;
#ifdef PA_W4_NOPATCH
      const int off = pla[SLOT < 7 ? SLOT : 0] | WCLS_PAD;
#else
      const int off = pla[SLOT < 7 ? SLOT : 0] & st.keep;
#endif
      __builtin_amdgcn_raw_ptr_buffer_load_lds(st.xsrd, (lds4_ptr_t)(st.pbuf + 1024 * piece), 16, off, 0, 0, 0);
    } else {
      const int u = SLOT == 6 ? 4 : (h ? SLOT - 2 : SLOT - 7);
      const int k = pr + 4 * u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(st.usrd, (lds4_ptr_t)(st.ubuf + 1024 * k), 16, lane * 16,
                                               st.usoff + 1024 * k, 0, 0);
    }
  };
  {
    const Wino4Stage st0 = wino4_stage(cctx, U, COUT, CIN, 0, my_patch, ubufs);
#pragma unroll
    for (int i = 0; i < P::SLOTS; ++i) issue_piece(i, st0);
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
  while (true) {
    for (int s = 0; s < nstages; ++s) {
      W4P_STAMP(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this stage's images have landed (issued a stage ago)
      W4P_STAMP(1);
      if (s == nstages - 1 && tid == 0) mail[0] = tq_resolve(tq, claim);
      wino4_barrier();   // A: ... everybody's; and everybody is done with the other U buffer
      W4P_STAMP(2);
      unsigned char* umine = ubufs + buf * G::USLAB_BYTES;
      unsigned char* uother = ubufs + (buf ^ 1) * G::USLAB_BYTES;
      bool stage_next = true;
      Wino4Stage nst;
      if (s + 1 < nstages) {
        nst = wino4_stage(cctx, U, COUT, CIN, s + 1, my_patch, uother);
      } else {
        nq = mail[0];
        stage_next = nq >= 0;
        wk = wino4_decode(stage_next ? nq : 0, n_tiles, num_groups);
        nxt = wino4_unit(wk.unit0 + pr, cgroups, trows, num_units);
        nxt.valid &= wk.valid;
        nxt_n0 = wk.n0;
        nctx = wino4_ctx(X, H, W, CIN, nxt, nxt_n0, x0_last);
        nst = wino4_stage(nctx, U, COUT, CIN, 0, my_patch, uother);
      }
      // ---- this half's three rows of V = B^T d B for the lane's tile and channel pair
      f32x2 v[3][6];
      {
        const unsigned char* pb = my_patch + pbase;
        const unsigned char* po = pb + h * wino4_patch_k(1, 0);   // the outer row reads patch rows {0, 2, 4} + h
        f32x2 tt[3][6];
        f32x2 x[2][7];   // patch rows 1..4 of a column, then the outer row's three
#define W4P_RD(c, j)                                                                                  \
  ((c) < 4 ? *reinterpret_cast<const f32x2*>(pb + wino4_patch_k(1 + ((c) < 4 ? (c) : 0), j))          \
           : *reinterpret_cast<const f32x2*>(po + wino4_patch_k(2 * ((c) >= 4 ? (c)-4 : 0), j)))
#pragma unroll
        for (int c = 0; c < 7; ++c) x[0][c] = W4P_RD(c, 0);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if (j + 1 < 6) {
#pragma unroll
            for (int c = 0; c < 7; ++c) x[(j + 1) & 1][c] = W4P_RD(c, j + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          const f32x2(&xx)[7] = x[j & 1];   // xx[0..3] = d[1..4][j]; xx[4..6] = d[{0,2,4} + h][j]
          const f32x2 Pq = w4fma(xx[1], kA, xx[3]);
          const f32x2 Qq = w4fma(xx[0], kA, xx[2]);
          tt[0][j] = w4fma(xx[4], kc.p4, w4fma(xx[5], kc.m5, xx[6]));
          tt[1][j] = w4fma(Qq, kB, Pq);
          tt[2][j] = w4fma(Qq, nkB, Pq);
          __builtin_amdgcn_sched_barrier(0);
        }
#undef W4P_RD
#pragma unroll
        for (int il = 0; il < 3; ++il) {
          wino4_bt(tt[il], v[il], kc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#if PA_W4P_STAMP
      asm volatile("s_nop 0" ::"v"(v[2][5]), "v"(v[0][0]));
#endif
      W4P_STAMP(3);
      wino4_barrier();   // B: both halves have read the pair's patch; the next stage's may be staged over it
      W4P_STAMP(4);
      // ---- 18 points x 2 channel groups x 2 k-steps, two points at a time
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
#ifndef PA_W4_NODMA
            if ((m == 4 || m == 6) && stage_next) {   // wave-uniform
              const int slot = xp + ((m - 4) >> 1);
              if (slot < P::SLOTS) issue_piece(slot, nst);
              __builtin_amdgcn_sched_barrier(0);
            }
#endif
          }
        }
#undef W4P_U
      };
      if (s == 0) mfma_run(std::true_type{});
      else mfma_run(std::false_type{});
      W4P_STAMP(5);
      buf ^= 1;
      if (s + 1 < nstages) W4P_STAMP_FLUSH();
    }
    // ---- inverse transform: own rows along the columns, partial output rows, exchange, BN shift (+ residual)
    // (+ ReLU), 16-byte stores of output rows 2 h, 2 h + 1
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (the compiler cannot see them)
    wino4_barrier();   // E: every wave is behind its last MFMA run: that stage's U slab is exchange space now
    {
      W4Const4 k4;
      {
        const float m1 = w4_opaque(-1.f), p2 = w4_opaque(2.f), p4 = w4_opaque(4.f), p8 = w4_opaque(8.f),
                    m2 = w4_opaque(-2.f), m8 = w4_opaque(-8.f);
        k4.m1 = f32x4{m1, m1, m1, m1}; k4.p2 = f32x4{p2, p2, p2, p2};
        k4.p4 = f32x4{p4, p4, p4, p4}; k4.p8 = f32x4{p8, p8, p8, p8};
        k4.m2 = f32x4{m2, m2, m2, m2}; k4.m8 = f32x4{m8, m8, m8, m8};
      }
      unsigned char* ulast = ubufs + (buf ^ 1) * G::USLAB_BYTES;   // (buf was flipped behind the last run)
      unsigned char* slot_mine = (h ? ulast : spare) + pr * P::EXCH_BYTES + lane * 16;
      const unsigned char* slot_peer = (h ? spare : ulast) + pr * P::EXCH_BYTES + lane * 16;
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
        // r[il][q] = sum_j M[row il][j] A[j][q]
        f32x4 r[3][4];
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
          *reinterpret_cast<f32x4*>(slot_mine + 1024 * qq) = sm;
          *reinterpret_cast<f32x4*>(slot_mine + 1024 * (4 + qq)) = send1;
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 rv[2][4];
        if (HAS_R) {   // their latency hides under the exchange
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
              rv[k][qq] = __builtin_bit_cast(
                  f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                             rsrd, offq[qq] == OOB ? OOB : offq[qq] + k * srow + 64 * cg, 0, 0));
        }
        wino4_barrier();   // X: the partner's partials are in LDS
#if PA_W4P_STAMP
        st_[6 + 2 * cg] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const f32x4 got = *reinterpret_cast<const f32x4*>(slot_peer + 1024 * (4 * k + qq));
            own[k][qq] = own[k][qq] + got + sh;
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
        if (cg == 0) wino4_barrier();   // the partner has read: the slot is free for the second channel group
#if PA_W4P_STAMP
        st_[7 + 2 * cg] = __builtin_amdgcn_s_memtime();
#endif
      }
    }
#undef W4P_ACC
    W4P_STAMP_FLUSH();
    if (nq < 0) break;
    cur = nxt;
    cur_n0 = nxt_n0;
    cctx = nctx;
    if (tid == 0) claim = tq_claim_own(tq);
  }
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
