// Centroid-linkage dendrogram, heap-free merge (round 4).  Same contract as linkage.hip -- bit-identical to
// scipy.cluster.hierarchy.linkage(y, "centroid"), i.e. what AgglomerativeClustering.cluster calls (reference:
// pipelines/clustering.py:374-382) -- but without SciPy's binary heap on the critical path, and spread over up to
// 16 workgroups.  tests/linkage_model.py is a line-by-line Python model of the protocol below (checked against
// SciPy by the CPU suite); the GPU suite checks this kernel against SciPy.
//
// SciPy's `fast_linkage` (Muellner's generic algorithm) keeps a lower bound min_dist[z] and a neighbour candidate
// per row in a min-heap and pops the root until the root's bound is exact.  The heap only decides WHICH row is the
// root; when the smallest bound is attained by exactly one row, the root is that row whatever the heap looks like
// inside.  So the merge loop runs with an arg-min over min_dist[] instead of a heap, as long as no two rows ever tie
// for the smallest bound at a pop.  Ties are detected at every pop (the two smallest candidates are compared): the
// first tie ends the kernel with status 1 and the launcher's NEXT kernel -- the exact heap replay of linkage.hip,
// gated on the status word -- recomputes the dendrogram from the untouched condensed matrix.  Real embeddings never
// tie (float64 distances of 256-dimensional vectors); duplicated rows do, and take the heap.
//
// What made the heap kernel 25 us per merge (round 3, N = 7 176: pass 29 k cycles, find 14 k, replay 12 k, waits 5 k)
// and what replaces it:
//   * column accesses D[z][x], D[z][y] for z < x in the CONDENSED matrix (one cache line per lane): the merge works
//     on a SQUARE symmetric copy (k_lf_square, N x ld doubles; 412 MB at N = 7 176, 26 GB at 57 k -- HBM3E is
//     288 GB): rows x and y are read coalesced, the new row y is written coalesced, only the mirror column
//     D[z][y] is scattered -- as stores, which nothing waits for individually.
//   * `dist == D[x][neighbor[x]]`, a dependent global load per pop: replaced by an EXACT bit per row, maintained
//     where the bound or the matrix entry changes (SciPy's comparison at the pop, made at the update).
//   * the serial heap replay (sort + ~12 sifts by one lane): gone; the owner thread of row z updates min_dist[z] and
//     neighbor[z] in LDS in place.
//   * one CU's f64 divide / square-root and address rate: G workgroups of 1 024 threads (1 up to N = 1 024, 8 up to
//     12 000, 16 above) own the rows in interleaved chunks of 1 024 (row z: workgroup (z >> 10) % G, thread
//     z & 1023); each keeps the state of ITS rows in LDS and does its share of every O(N) step.  (256-thread
//     workgroups were measured SLOWER, profiles/r4_linkage_heap_free_v2_t256.txt: the pass and wave 0's serial
//     reduction chain need the other waves of a SIMD to hide their dependent-issue latency.)
// Exchange between the workgroups (one all-gather per step, no separate barrier): wave 0 of every workgroup writes a
// 16-granule record -- its two smallest rows with neighbour / size, and its minimum of the row scan or of the new
// row y -- as 8-byte {data, sequence tag} granules with sc1 stores and polls the records of all G workgroups with
// sc1 loads until every tag carries the current sequence number (MI355X_MICROARCH.md, hand-off by tagged granules;
// double-buffered by sequence parity; correct at any workgroup placement, fastest when the G workgroups share an
// XCD, hence the launch of 8 G workgroups of which every 8th works).  Wave 0 of EVERY workgroup then holds the same
// candidate table (lane 4g: best row of workgroup g, 4g+1: its second, 4g+3: the row it repaired last, lane 2: the
// row merged last, whose new bound only now is known) and pops from it without further traffic:
//   merge k:    pass over the own rows -> exchange -> pop -> [row scan -> exchange -> pop]* -> next merge
// A lower-bound repair raises the bound of its row, so that workgroup's two smallest rows are no longer known: with
// the scan minimum it publishes its two smallest rows computed WITHOUT the repaired row, which from then on travels
// as the workgroup's third table entry -- the table always contains the two smallest bounds overall.
// Every poll is bounded (2 s of the constant-rate wall clock, s_memrealtime; the limit in ticks is a kernel argument
// computed from hipDeviceAttributeWallClockRate): a workgroup that never shows up (not resident) ends the kernel with
// status 3 and the gated heap kernel takes over -- there is no way to hang.  The workgroup that gives up first
// rewrites granule 13 of its own record with LF_ABORT, which every poll of the others looks at: they leave at their
// next poll instead of each waiting out its own limit.
// ORDERING.  All matrix / size / id traffic between workgroups goes through relaxed agent-scope atomics (sc1: served
// by L2, never by a CU's vector cache), so a reader needs no acquire fence.  The WRITER side needs its stores to
// have left the CU before the mailbox tag that announces them: every wave waits for its own outstanding stores
// (`s_waitcnt vmcnt(0)`) in front of the workgroup barrier that precedes wave 0's mailbox store.  (An agent-scope
// release FENCE would also write back the whole L2 -- 10 us per round, see ROUND_NOTES round 3 -- and is not needed
// for write-through sc1 stores.)
// hipcc-flags: -ffp-contract=off
#include <stdlib.h>

#include "common.h"

namespace pa {

typedef unsigned long long lf_u64;
typedef unsigned int lf_u32;

#ifndef PA_LF_T_LOG2   // threads per workgroup = rows per ownership chunk (development switch: 8 ... 10)
#define PA_LF_T_LOG2 10
#endif
#ifndef PA_LF_PU       // rows per thread whose matrix loads are in flight together
#define PA_LF_PU 4
#endif
constexpr int LF_SH = PA_LF_T_LOG2;
constexpr int LF_T = 1 << LF_SH;
constexpr int LF_W = LF_T / 64;
constexpr int LF_PU = PA_LF_PU;
constexpr int LF_MAXG = 16;
constexpr lf_u64 LF_INF = 0x7ff0000000000000ULL;   // key of "no bound": +inf (NaN and negatives sort above)
constexpr lf_u32 LF_ABORT = 0xAB0127u;              // granule 13 of a record: "I gave up, leave"

// distances are >= +0: the bit pattern orders like the value; -0.0 is folded into +0.0, NaN / inf into LF_INF
__device__ __forceinline__ lf_u64 lf_key(double d) {
  return d < __builtin_inf() ? (lf_u64)__double_as_longlong(d + 0.0) : LF_INF;
}
__device__ __forceinline__ double lf_val(lf_u64 k) {
  return k >= LF_INF ? __builtin_inf() : __longlong_as_double((long long)k);
}

// ---- wave-wide minima through DPP (no LDS traffic): xor 1, xor 2 inside a quad, rotate 4 / 8 inside a row of 16,
// then row 0 -> 1, 2 -> 3 and rows 0-1 -> 2-3: lane 63 holds the minimum of the wave
template <int CTRL, int RMASK>
__device__ __forceinline__ lf_u32 lf_dpp(lf_u32 v) {
  return (lf_u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, RMASK, 0xf, false);
}
template <int CTRL, int RMASK>
__device__ __forceinline__ lf_u64 lf_min_step64(lf_u64 v) {
  const lf_u32 lo = lf_dpp<CTRL, RMASK>((lf_u32)v), hi = lf_dpp<CTRL, RMASK>((lf_u32)(v >> 32));
  const lf_u64 o = ((lf_u64)hi << 32) | lo;
  return o < v ? o : v;
}
template <int CTRL, int RMASK>
__device__ __forceinline__ lf_u32 lf_min_step32(lf_u32 v) {
  const lf_u32 o = lf_dpp<CTRL, RMASK>(v);
  return o < v ? o : v;
}
__device__ __forceinline__ lf_u64 lf_wave_min64(lf_u64 v) {   // all 64 lanes must be active; uniform result
  v = lf_min_step64<0xB1, 0xf>(v);
  v = lf_min_step64<0x4E, 0xf>(v);
  v = lf_min_step64<0x124, 0xf>(v);
  v = lf_min_step64<0x128, 0xf>(v);
  v = lf_min_step64<0x142, 0xa>(v);
  v = lf_min_step64<0x143, 0xc>(v);
  const lf_u32 lo = __builtin_amdgcn_readlane((int)(lf_u32)v, 63), hi = __builtin_amdgcn_readlane((int)(lf_u32)(v >> 32), 63);
  return ((lf_u64)hi << 32) | lo;
}
__device__ __forceinline__ lf_u32 lf_wave_min32(lf_u32 v) {
  v = lf_min_step32<0xB1, 0xf>(v);
  v = lf_min_step32<0x4E, 0xf>(v);
  v = lf_min_step32<0x124, 0xf>(v);
  v = lf_min_step32<0x128, 0xf>(v);
  v = lf_min_step32<0x142, 0xa>(v);
  v = lf_min_step32<0x143, 0xc>(v);
  return (lf_u32)__builtin_amdgcn_readlane((int)v, 63);
}
// lane holding the lexicographically smallest (key, idx); *mkey = that key.  Uniform results.
__device__ __forceinline__ int lf_wave_argmin(lf_u64 key, lf_u32 idx, lf_u64* mkey) {
  const lf_u64 m = lf_wave_min64(key);
  *mkey = m;
  const lf_u64 mask = __ballot(key == m);
  if (__popcll(mask) == 1) return __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1);
  const lf_u32 mi = lf_wave_min32(key == m ? idx : 0xffffffffu);
  const lf_u64 mask2 = __ballot(key == m && idx == mi);
  return __builtin_amdgcn_readfirstlane(__ffsll((long long)mask2) - 1);
}
__device__ __forceinline__ lf_u32 lf_rl(lf_u32 v, int lane) { return (lf_u32)__builtin_amdgcn_readlane((int)v, lane); }
template <int Q>
__device__ __forceinline__ lf_u32 lf_quad(lf_u32 v) {   // value of quad lane Q
  return (lf_u32)__builtin_amdgcn_update_dpp((int)v, (int)v, Q * 0x55, 0xf, 0xf, false);
}

// condensed (SciPy pdist order) -> square symmetric, leading dimension ld (multiple of 8 doubles); 64 x 64 tiles,
// the mirror tile transposed through LDS so that both are written in whole lines.  grid = (nt, nt), bj >= bi works.
__global__ __launch_bounds__(256) void k_lf_square(const double* __restrict__ cond, int n, long ld,
                                                    double* __restrict__ S) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;
  __shared__ double tile[64][65];
  const int tid = threadIdx.x;
  for (int e = tid; e < 4096; e += 256) {
    const int r = e >> 6, c = e & 63;
    const long i = bi * 64 + r, j = bj * 64 + c;
    double v = 0.0;
    if (i < n && j < n && i != j) {
      const long a = i < j ? i : j, b = i < j ? j : i;
      v = cond[(long)n * a - (a * (a + 1) / 2) + (b - a - 1)];
    }
    tile[r][c] = v;
    if (i < n && j < n) S[i * ld + j] = v;
  }
  if (bi == bj) return;
  __syncthreads();
  for (int e = tid; e < 4096; e += 256) {
    const int r = e >> 6, c = e & 63;
    const long j = bj * 64 + r, i = bi * 64 + c;
    if (i < n && j < n) S[j * ld + i] = tile[c][r];
  }
}

// find_min_dist(n, D, size, x) of the initial state for every row: first minimum of row x over i > x, from the
// condensed matrix (contiguous rows).  One wave per row; grid = ceil((n - 1) / 4), block = 256.
__global__ __launch_bounds__(256) void k_lf_row_nearest(const double* __restrict__ cond, int n,
                                                         int* __restrict__ nb0, double* __restrict__ mind0) {
  const int lane = threadIdx.x & 63;
  const int x = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (x >= n - 1) return;
  const long base = (long)n * x - ((long)x * (x + 1) / 2) - x - 1;   // cidx(n, x, i) = base + i
  lf_u64 bk = LF_INF;
  lf_u32 bi = 0xffffffffu;
  for (int i0 = x + 1 + lane; i0 < n; i0 += 4 * 64) {
    double d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = (i0 + 64 * u < n) ? cond[base + i0 + 64 * u] : __builtin_inf();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const lf_u64 kk = lf_key(d[u]);
      if (kk < bk) {
        bk = kk;
        bi = (lf_u32)(i0 + 64 * u);
      }
    }
  }
  lf_u64 m;
  const int lw = lf_wave_argmin(bk, bi, &m);
  const lf_u32 mi = lf_rl(bi, lw);
  if (lane == 0) {
    nb0[x] = m >= LF_INF ? -1 : (int)mi;
    mind0[x] = lf_val(m);
  }
}

// ---- the merge kernel ------------------------------------------------------------------------------------
struct LfWaveOut {       // per wave, LDS: its two smallest rows and its scan / new-row minimum
  lf_u64 k1;
  lf_u32 i1, nb1, sz1, pad1;
  lf_u64 k2;
  lf_u32 i2, nb2, sz2, pad2;
  lf_u64 kb;
  lf_u32 ib, pad3;
};
struct LfFinal {         // wave 0 -> every thread of the workgroup, LDS
  int action;            // 0 merge, 1 repair (scan row x), 3 give up
  int fail;              // status code of action 3
  int x;
  lf_u32 nbx;
  lf_u32 nx;
  int ap_row;            // row whose owner thread stores (ap_key, ap_nbx) as its new bound / neighbour (-1: none)
  lf_u32 ap_nbx;
  int pad;
  lf_u64 key, ap_key;
};
enum { LF_MERGE = 0, LF_REPAIR = 1, LF_FAIL = 3 };
enum { LF_X_MAIN = 0, LF_X_SCAN = 1 };   // what an exchange carries: candidates + new row y / scan of row x

// status word: 0 = dendrogram complete, 1 = tie at a pop (take the heap), 2 = degenerate input (no finite bound, or
// more repairs than SciPy's loop allows), 3 = a workgroup never showed up -- anything but 0 lets the gated heap
// kernel run.
//   IT: row-state integer type (unsigned short when MULTI is false: N <= 10 240; unsigned int otherwise).
//   mail: 2 x 16 x 16 granules of 8 bytes, zeroed by the launcher (MULTI only).
template <typename IT, bool MULTI>
__global__ __launch_bounds__(LF_T) void k_linkage_fast(double* __restrict__ S, long ld, int n,
                                                        double* __restrict__ Z, const int* __restrict__ nb0,
                                                        const double* __restrict__ mind0,
                                                        int* __restrict__ g_size, int* __restrict__ g_cid,
                                                        lf_u64* __restrict__ mail, int G, int SL,
                                                        long long poll_limit /* wall-clock ticks */,
                                                        int* __restrict__ status, long long* __restrict__ stats) {
  if (MULTI && (blockIdx.x & 7) != 0) return;   // every 8th workgroup of the launch: observed to share one XCD
  const int wg = MULTI ? (int)(blockIdx.x >> 3) : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ LfWaveOut exch[2][LF_W];
  __shared__ LfFinal fin[2];
  constexpr IT EX = (IT)((IT)1 << (8 * sizeof(IT) - 1));   // "the bound of this row is exact"
  constexpr IT NONE = (IT)(EX - 1);                         // no neighbour candidate
  constexpr lf_u32 EX32 = 0x80000000u, NONE32 = 0x7fffffffu;   // the same in the exchanged records
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows_l = SL * LF_T;
  // state of the rows this workgroup owns; local index l = (c << LF_SH) + tid  <->  row ((c G + wg) << LF_SH) + tid
  double* mind_l = reinterpret_cast<double*>(lds_raw);   // SciPy's min_dist (+inf: no bound / row left the heap)
  IT* nb_l = reinterpret_cast<IT*>(mind_l + rows_l);     // neighbour candidate | EX
  IT* size_l = nb_l + rows_l;
  IT* cid_l = size_l + rows_l;                           // (MULTI: ids live in g_cid, this array is not allocated)

  auto row_of = [&](int c) { return ((c * G + wg) << LF_SH) + tid; };
  auto owner_wg = [&](int z) { return MULTI ? ((z >> LF_SH) % G) : 0; };
  auto local_of = [&](int z) { return (((z >> LF_SH) / G) << LF_SH) | (z & (LF_T - 1)); };
  auto to32 = [&](IT v) -> lf_u32 {     // neighbour | EX of the row state -> the 32-bit form of the records
    const lf_u32 c = (lf_u32)(v & NONE);
    return ((IT)(v & NONE) == NONE ? NONE32 : c) | ((v & EX) ? EX32 : 0u);
  };
  auto from32 = [&](lf_u32 v) -> IT {
    return (IT)(((v & NONE32) == NONE32 ? (lf_u32)NONE : (v & NONE32)) | ((v & EX32) ? (lf_u32)EX : 0u));
  };

  for (int c = 0; c < SL; ++c) {
    const int z = row_of(c), l = (c << LF_SH) + tid;
    const bool in = z < n;
    const int cnd = (in && z < n - 1) ? nb0[z] : -1;
    size_l[l] = in ? (IT)1 : (IT)0;
    nb_l[l] = cnd < 0 ? NONE : (IT)((IT)cnd | EX);
    mind_l[l] = cnd >= 0 ? mind0[z] : __builtin_inf();
    if (MULTI) {
      if (in) {
        __hip_atomic_store(g_size + z, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g_cid + z, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      cid_l[l] = (IT)z;
    }
  }
  __syncthreads();

  // ---- wave 0's candidate table, one entry per lane: 4g best row of workgroup g, 4g+1 its second (both computed
  // WITHOUT the row merged last and without the row that workgroup repaired last), 4g+3 that repaired row, lane 2 the
  // row merged last (lanes 4g+2 only carry the scan minima of an exchange)
  lf_u64 t_key = LF_INF;
  lf_u32 t_idx = 0xffffffffu, t_nb = NONE32, t_sz = 0;
  // what this workgroup published last (re-sent unchanged while its rows do not change)
  lf_u64 cb1k = LF_INF, cb2k = LF_INF;
  lf_u32 cb1i = 0xffffffffu, cb1n = NONE32, cb1s = 0, cb2i = 0xffffffffu, cb2n = NONE32, cb2s = 0;
  lf_u32 seq = 0;         // exchange counter (tags; parity selects the mail / LDS buffers)
  int tries = 0;          // lower-bound repairs of the current merge (SciPy's pop loop allows n - k - 1)

  long long st_rep = 0, st_c0 = 0, st_c1 = 0;
  int fail = 0;
  int k = 0;
  int pend_y = -1;                 // row merged last: its new bound arrives with the next exchange
  lf_u32 pend_size = 0;
  int pend_cid = 0;
  int y_excl = -1;                 // ... and stays out of its workgroup's top-2 until the next merge (table lane 2)
  int xmode = LF_X_MAIN;
  lf_u64 pb_key = LF_INF;          // this thread's minimum of the new row y / of the row scan
  lf_u32 pb_idx = 0xffffffffu;
  int scan_x = -1;                 // row being repaired (LF_X_SCAN)
  lf_u32 scan_nx = 0;              // ... its size
  long long tc = __builtin_readcyclecounter();
#ifdef PA_LF_STAMP   // development: where a round's cycles go (thread 0 of workgroup 0; printed at the end)
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = tc, rounds = 0;
#define LF_PH(i)                                          \
  do {                                                    \
    const long long now_ = __builtin_readcyclecounter();  \
    ph[i] += now_ - pt;                                   \
    pt = now_;                                            \
  } while (0)
#else
#define LF_PH(i)
#endif

  while (true) {
    LF_PH(7);
    if (xmode == LF_X_MAIN) y_excl = pend_y;
    // ================= A. this thread's two smallest own rows; in a scan exchange only the workgroup of the
    // repaired row has anything new to say
    const bool fresh = xmode == LF_X_MAIN || owner_wg(scan_x) == wg;
    lf_u64 k1 = LF_INF, k2 = LF_INF;
    lf_u32 i1 = 0xffffffffu, i2 = 0xffffffffu;
    int l1 = 0, l2 = 0;
    if (fresh) {
      for (int c = 0; c < SL; ++c) {
        const int z = row_of(c), l = (c << LF_SH) + tid;
        const bool skip = z == y_excl || (xmode == LF_X_SCAN && z == scan_x);
        const lf_u64 kk = skip ? LF_INF : lf_key(mind_l[l]);
        if (kk < k1) {
          k2 = k1; i2 = i1; l2 = l1;
          k1 = kk; i1 = (lf_u32)z; l1 = l;
        } else if (kk < k2) {
          k2 = kk; i2 = (lf_u32)z; l2 = l;
        }
      }
    }
    // ================= B. wave top-2 and wave minimum of pb -> LDS
    const int pe = seq & 1;
    ++seq;
    {
      lf_u64 m;
      if (fresh) {
        const int w1 = lf_wave_argmin(k1, i1, &m);
        if (lane == w1) {
          exch[pe][wave].k1 = k1;
          exch[pe][wave].i1 = i1;
          exch[pe][wave].nb1 = k1 < LF_INF ? to32(nb_l[l1]) : NONE32;
          exch[pe][wave].sz1 = k1 < LF_INF ? (lf_u32)size_l[l1] : 0u;
        }
        const lf_u64 kc = lane == w1 ? k2 : k1;
        const lf_u32 ic = lane == w1 ? i2 : i1;
        const int lc = lane == w1 ? l2 : l1;
        const int w2 = lf_wave_argmin(kc, ic, &m);
        if (lane == w2) {
          exch[pe][wave].k2 = kc;
          exch[pe][wave].i2 = ic;
          exch[pe][wave].nb2 = kc < LF_INF ? to32(nb_l[lc]) : NONE32;
          exch[pe][wave].sz2 = kc < LF_INF ? (lf_u32)size_l[lc] : 0u;
        }
      }
      const int wb = lf_wave_argmin(pb_key, pb_idx, &m);
      if (lane == wb) {
        exch[pe][wave].kb = pb_key;
        exch[pe][wave].ib = pb_idx;
      }
    }
    LF_PH(0);
    // every store of the last pass (matrix entries, sizes, ids) has left this CU before wave 0 announces them
    if (MULTI) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    LF_PH(1);
    // ================= C. wave 0: workgroup result -> exchange -> candidate table -> pop
    if (wave == 0) {
      lf_u64 m;
      if (fresh) {
        // workgroup top-2: lanes 0 .. W-1 hold the waves' best, 16 .. 16+W-1 their second
        lf_u64 ck = LF_INF;
        lf_u32 ci = 0xffffffffu, cn = NONE32, cs = 0;
        if (lane < LF_W) {
          ck = exch[pe][lane].k1; ci = exch[pe][lane].i1; cn = exch[pe][lane].nb1; cs = exch[pe][lane].sz1;
        } else if (lane >= 16 && lane < 16 + LF_W) {
          ck = exch[pe][lane - 16].k2; ci = exch[pe][lane - 16].i2; cn = exch[pe][lane - 16].nb2; cs = exch[pe][lane - 16].sz2;
        }
        const int w1 = lf_wave_argmin(ck, ci, &m);
        cb1k = m; cb1i = lf_rl(ci, w1); cb1n = lf_rl(cn, w1); cb1s = lf_rl(cs, w1);
        const lf_u64 ck2 = lane == w1 ? LF_INF : ck;
        const int w2 = lf_wave_argmin(ck2, ci, &m);
        cb2k = m; cb2i = lf_rl(ci, w2); cb2n = lf_rl(cn, w2); cb2s = lf_rl(cs, w2);
      }
      lf_u64 pbk;
      lf_u32 pbi;
      {
        const lf_u64 ck = lane < LF_W ? exch[pe][lane].kb : LF_INF;
        const lf_u32 ci = lane < LF_W ? exch[pe][lane].ib : 0xffffffffu;
        const int wb = lf_wave_argmin(ck, ci, &m);
        pbk = m; pbi = lf_rl(ci, wb);
      }
      LF_PH(2);
      // -- exchange: this workgroup's record out, everybody's records in
      lf_u64 c_key = LF_INF;    // this lane's part of the records (lanes 4g: best, 4g+1: second, 4g+2: scan minimum)
      lf_u32 c_idx = 0xffffffffu, c_nb = NONE32, c_sz = 0;
      int timeout = 0;
      if (MULTI) {
        lf_u64* box = mail + (size_t)(pe * LF_MAXG) * 16;
        if (lane < 16) {
          lf_u32 dta = 0;
          switch (lane) {
            case 0: dta = (lf_u32)cb1k; break;
            case 1: dta = (lf_u32)(cb1k >> 32); break;
            case 2: dta = cb1i; break;
            case 3: dta = cb1n; break;
            case 4: dta = cb1s; break;
            case 5: dta = (lf_u32)cb2k; break;
            case 6: dta = (lf_u32)(cb2k >> 32); break;
            case 7: dta = cb2i; break;
            case 8: dta = cb2n; break;
            case 9: dta = cb2s; break;
            case 10: dta = (lf_u32)pbk; break;
            case 11: dta = (lf_u32)(pbk >> 32); break;
            case 12: dta = pbi; break;
            default: break;
          }
          __hip_atomic_store(box + wg * 16 + lane, ((lf_u64)seq << 32) | dta, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
        const int slot = lane >> 2, q = lane & 3;
        const bool part = slot < G;
        lf_u64 gr[4];
        const long long t0 = (long long)wall_clock64();
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            gr[j] = part ? __hip_atomic_load(box + slot * 16 + 4 * q + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                         : ((lf_u64)seq << 32);
            ok = ok && (lf_u32)(gr[j] >> 32) == seq;
          }
          // granule 13 (q == 3, j == 1) of a record that carries this exchange's tag: 0, or LF_ABORT
          if (__any(part && q == 3 && (lf_u32)(gr[1] >> 32) == seq && (lf_u32)gr[1] == LF_ABORT)) {
            timeout = 1;
            break;
          }
          if (__all(ok)) break;
          if ((long long)wall_clock64() - t0 > poll_limit) {
            timeout = 1;
            if (lane == 13)
              __hip_atomic_store(box + wg * 16 + 13, ((lf_u64)seq << 32) | LF_ABORT, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        const lf_u32 d0 = (lf_u32)gr[0], d1 = (lf_u32)gr[1], d2 = (lf_u32)gr[2], d3 = (lf_u32)gr[3];
        // granules 0-4 best {key lo, key hi, row, nb, size}, 5-9 second, 10-12 scan minimum {key lo, key hi, row}
        const lf_u32 a0 = lf_quad<0>(d0), a1 = lf_quad<0>(d1), a2 = lf_quad<0>(d2), a3 = lf_quad<0>(d3);
        const lf_u32 a4 = lf_quad<1>(d0), a5 = lf_quad<1>(d1), a6 = lf_quad<1>(d2), a7 = lf_quad<1>(d3);
        const lf_u32 a8 = lf_quad<2>(d0), a9 = lf_quad<2>(d1), a10 = lf_quad<2>(d2), a11 = lf_quad<2>(d3);
        const lf_u32 a12 = lf_quad<3>(d0);
        if (part) {
          if (q == 0) {
            c_key = ((lf_u64)a1 << 32) | a0; c_idx = a2; c_nb = a3; c_sz = a4;
          } else if (q == 1) {
            c_key = ((lf_u64)a6 << 32) | a5; c_idx = a7; c_nb = a8; c_sz = a9;
          } else if (q == 2) {
            c_key = ((lf_u64)a11 << 32) | a10; c_idx = a12;
          }
        }
      } else {
        if (lane == 0) {
          c_key = cb1k; c_idx = cb1i; c_nb = cb1n; c_sz = cb1s;
        } else if (lane == 1) {
          c_key = cb2k; c_idx = cb2i; c_nb = cb2n; c_sz = cb2s;
        } else if (lane == 2) {
          c_key = pbk; c_idx = pbi;
        }
      }
      LF_PH(3);
      // -- table update
      const int q = lane & 3;
      int ap_row = -1;
      lf_u32 ap_nbx = NONE32;
      lf_u64 ap_key = LF_INF;
      int f = timeout ? 3 : 0;
      if (q == 0 || q == 1) {
        t_key = c_key; t_idx = c_idx; t_nb = c_nb; t_sz = c_sz;
      }
      // minimum over the workgroups of the scanned row (scan exchange) / of the new row y (main exchange)
      lf_u64 m2;
      const int wr = lf_wave_argmin(q == 2 ? c_key : LF_INF, c_idx, &m2);
      const lf_u32 ri = lf_rl(c_idx, wr);
      if (xmode == LF_X_SCAN) {
        // = the repaired bound of row scan_x, which now travels as the extra entry of its workgroup
        ap_row = scan_x;
        ap_key = m2;
        ap_nbx = m2 >= LF_INF ? NONE32 : (ri | EX32);
        if (lane == 4 * owner_wg(scan_x) + 3) {
          t_key = ap_key; t_idx = (lf_u32)scan_x; t_nb = ap_nbx; t_sz = scan_nx;
        }
      } else {
        if (q == 3) t_key = LF_INF;
        if (lane == 2) t_key = LF_INF;
        if (pend_y >= 0) {
          // = the new bound of the row merged last (SciPy's loop 4)
          ap_row = pend_y;
          if (pend_y < n - 1) {
            if (m2 >= LF_INF) f = f ? f : 2;
            ap_key = m2;
            ap_nbx = ri | EX32;
            if (lane == 2) {
              t_key = m2; t_idx = (lf_u32)pend_y; t_nb = ap_nbx; t_sz = pend_size;
            }
          }
        }
      }
      // -- pop: smallest candidate, tie check against the runner-up
      const bool is_cand = q != 2 || lane == 2;
      const lf_u64 pk = is_cand ? t_key : LF_INF;
      lf_u64 mk;
      const int lw = lf_wave_argmin(pk, t_idx, &mk);
      const lf_u64 mk2 = lf_wave_min64(lane == lw ? LF_INF : pk);
      int action = LF_MERGE;
      if (!f && mk >= LF_INF) f = 2;
      if (!f && mk2 == mk) f = 1;
      const int x = (int)lf_rl(t_idx, lw);
      const lf_u32 nbx = lf_rl(t_nb, lw), nx = lf_rl(t_sz, lw);
      if (!f && ((nbx & EX32) == 0 || (nbx & NONE32) == NONE32)) {
        action = LF_REPAIR;
        if (++tries >= n - k) f = 2;   // (SciPy's pop loop would run out: rare enough for the heap)
      }
      if (f) action = LF_FAIL;
      if (lane == 0) {
        LfFinal& o = fin[pe];
        o.action = action; o.fail = f; o.x = x; o.nbx = nbx; o.nx = nx; o.key = mk;
        o.ap_row = ap_row; o.ap_nbx = ap_nbx; o.ap_key = ap_key;
      }
      LF_PH(4);
    }
    __syncthreads();
    LF_PH(5);
#ifdef PA_LF_STAMP
    ++rounds;
#endif
    const LfFinal F = fin[pe];
    // ================= D. the owner thread stores the bound that arrived with this exchange
    if (F.ap_row >= 0 && owner_wg(F.ap_row) == wg && tid == (F.ap_row & (LF_T - 1))) {
      const int l = local_of(F.ap_row);
      if (xmode == LF_X_MAIN) {      // the row merged last: also its size and id
        size_l[l] = (IT)pend_size;
        if (MULTI) {
          __hip_atomic_store(g_size + F.ap_row, (int)pend_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(g_cid + F.ap_row, pend_cid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          cid_l[l] = (IT)pend_cid;
        }
        if (F.ap_row < n - 1) {
          mind_l[l] = lf_val(F.ap_key);
          nb_l[l] = from32(F.ap_nbx);
        }
      } else {
        mind_l[l] = lf_val(F.ap_key);
        nb_l[l] = from32(F.ap_nbx);
      }
    }
    const int was_pend_y = pend_y;          // (ids / sizes of this row are special-cased below: its stores race)
    const lf_u32 was_pend_size = pend_size;
    const int was_pend_cid = pend_cid;
    if (xmode == LF_X_MAIN) pend_y = -1;
    if (F.action == LF_FAIL) {
      fail = F.fail;
      break;
    }
    const int x = F.x;
    if (F.action == LF_REPAIR) {
      // find_min_dist(n, D, size, x): this thread's columns i > x of row x
      const double* row = S + (long)x * ld;
      pb_key = LF_INF;
      pb_idx = 0xffffffffu;
      for (int c0 = 0; c0 < SL; c0 += LF_PU) {
        double d[LF_PU];
#pragma unroll
        for (int u = 0; u < LF_PU; ++u) {
          const int c = c0 + u;
          const int i = row_of(c);
          const bool a = c < SL && i < n && i > x && size_l[(c << LF_SH) + tid] != 0;
          d[u] = !a ? __builtin_inf()
                    : (MULTI ? __hip_atomic_load(row + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : row[i]);
        }
#pragma unroll
        for (int u = 0; u < LF_PU; ++u) {
          const lf_u64 kk = lf_key(d[u]);
          if (kk < pb_key) {
            pb_key = kk;
            pb_idx = (lf_u32)row_of(c0 + u);
          }
        }
      }
      scan_x = x;
      scan_nx = F.nx;
      xmode = LF_X_SCAN;
      ++st_rep;
      LF_PH(6);
      continue;
    }
    // ================= E. merge x into y = neighbor[x]
    {
      const long long t2 = __builtin_readcyclecounter();
      st_c0 += t2 - tc;
      tc = t2;
    }
    const int y = (int)(F.nbx & NONE32);
    const double dist = lf_val(F.key);
    const int nx = (int)F.nx;
    int ny;
    if (y == was_pend_y) ny = (int)was_pend_size;
    else if (MULTI) ny = __hip_atomic_load(g_size + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else ny = (int)size_l[local_of(y)];
    if (wg == 0 && tid == 64) {   // (lane 0 of wave 1: not the wave that runs the exchanges)
      auto id_of = [&](int r) -> int {
        if (r == was_pend_y) return was_pend_cid;
        if (MULTI) return __hip_atomic_load(g_cid + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (int)cid_l[local_of(r)];
      };
      int id_x = id_of(x), id_y = id_of(y);
      if (id_x > id_y) {
        const int t = id_x;
        id_x = id_y;
        id_y = t;
      }
      Z[4 * (long)k + 0] = (double)id_x;
      Z[4 * (long)k + 1] = (double)id_y;
      Z[4 * (long)k + 2] = dist;
      Z[4 * (long)k + 3] = (double)(nx + ny);
    }
    // one pass over the own rows z: Lance-Williams (centroid) update of D[z][y] = D[y][z]; neighbour reassignment
    // x -> y and lower-bound update for z < y (SciPy's loops 2 and 3); minimum of the new row over z > y (loop 4)
    pb_key = LF_INF;
    pb_idx = 0xffffffffu;
    {
      const double* rowx = S + (long)x * ld;
      double* rowy = S + (long)y * ld;
      for (int c0 = 0; c0 < SL; c0 += LF_PU) {
        bool act[LF_PU];
        double d_xi[LF_PU], d_yi[LF_PU];
#pragma unroll
        for (int u = 0; u < LF_PU; ++u) {
          const int c = c0 + u;
          const int z = row_of(c);
          act[u] = c < SL && z < n && z != y && z != x && size_l[(c << LF_SH) + tid] != 0;
          if (MULTI) {
            d_xi[u] = act[u] ? __hip_atomic_load(rowx + z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            d_yi[u] = act[u] ? __hip_atomic_load(rowy + z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
          } else {
            d_xi[u] = act[u] ? rowx[z] : 0.0;
            d_yi[u] = act[u] ? rowy[z] : 0.0;
          }
        }
#pragma unroll
        for (int u = 0; u < LF_PU; ++u) {
          if (!act[u]) continue;
          const int c = c0 + u;
          const int z = row_of(c), l = (c << LF_SH) + tid;
          const double nd = sqrt(
              (((nx * d_xi[u] * d_xi[u]) + (ny * d_yi[u] * d_yi[u])) - ((nx * ny) * dist * dist) / (nx + ny)) /
              (nx + ny));
          if (MULTI) {
            __hip_atomic_store(rowy + z, nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(S + (long)z * ld + y, nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            rowy[z] = nd;
            S[(long)z * ld + y] = nd;
          }
          if (z < y) {
            const double m = mind_l[l];
            const IT nbz = nb_l[l];
            const int cnd = (IT)(nbz & NONE) == NONE ? -1 : (int)(nbz & NONE);
            if (nd < m) {
              mind_l[l] = nd;
              nb_l[l] = (IT)((IT)y | EX);
            } else if (cnd == x || cnd == y) {
              nb_l[l] = (IT)((IT)y | (m == nd ? EX : (IT)0));
            }
          } else {
            const lf_u64 kk = lf_key(nd);
            if (kk < pb_key) {   // z > y, ascending per thread: first minimum
              pb_key = kk;
              pb_idx = (lf_u32)z;
            }
          }
        }
      }
    }
    if (owner_wg(x) == wg && tid == (x & (LF_T - 1))) {
      const int l = local_of(x);
      mind_l[l] = __builtin_inf();   // heap.remove_min()
      size_l[l] = (IT)0;
    }
    // (size, id and bound of row y: stored by its owner thread when the bound arrives, section D)
    pend_y = y;
    pend_size = (lf_u32)(nx + ny);
    pend_cid = n + k;
    tries = 0;
    ++k;
    {
      const long long t2 = __builtin_readcyclecounter();
      st_c1 += t2 - tc;
      tc = t2;
    }
    if (k >= n - 1) break;
    xmode = LF_X_MAIN;
  }
#ifdef PA_LF_STAMP
  if (wg == 0 && tid == 0)
    printf("lf stamps n=%d G=%d rounds %lld; cycles per round: A+B %lld | sync %lld | wg top-2 %lld | exchange %lld | "
           "table+pop %lld | sync %lld | scan %lld (per repair) | merge pass etc. %lld (per merge)\n",
           n, G, rounds, ph[0] / rounds, ph[1] / rounds, ph[2] / rounds, ph[3] / rounds, ph[4] / rounds,
           ph[5] / rounds, ph[6] / (st_rep ? st_rep : 1), ph[7] / (k ? k : 1));
#endif
  if (wg == 0 && tid == 0) {
    __hip_atomic_store(status, fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stats != nullptr) {
      stats[0] = fail;
      stats[1] = st_rep;
      stats[2] = st_c0;
      stats[3] = st_c1;
      stats[4] = n;
      stats[5] = G;
      stats[6] = 0;
      stats[7] = k;
    }
  }
}

constexpr size_t LF_LDS_MAX = 160 * 1024 - 6144;   // dynamic LDS budget (static part: exchange buffers, ~2.7 KB)

inline size_t lf_align(size_t v) { return (v + 255) & ~(size_t)255; }
inline long lf_ld(int n) { return ((long)n + 7) & ~7L; }

// the fast path wants N x ld doubles more: PA_LINKAGE_FAST=0 switches it off, PA_LINKAGE_FAST_MAX_GB caps the square
// matrix (default 96 GB: N = 110 k)
bool lf_wanted(int n) {
  if (n < 3) return false;
  const char* e = getenv("PA_LINKAGE_FAST");
  if (e != nullptr && atoi(e) == 0) return false;
  const char* g = getenv("PA_LINKAGE_FAST_MAX_GB");
  const double cap = (g != nullptr && atof(g) > 0 ? atof(g) : 96.0) * 1e9;
  return 8.0 * (double)n * (double)lf_ld(n) <= cap && n <= 150000;
}

// bytes the fast path adds to the linkage workspace: square matrix + initial candidates + sizes / ids + mail + status
size_t lf_workspace_bytes(int n) {
  if (!lf_wanted(n)) return 0;
  const size_t ni = lf_align(sizeof(int) * (size_t)n), nd = lf_align(sizeof(double) * (size_t)n);
  return lf_align(8 * (size_t)n * (size_t)lf_ld(n)) + 3 * ni + nd + lf_align(2 * LF_MAXG * 16 * 8) + 256;
}

// workgroups of the merge: 1 up to N = 1 024, 8 up to 12 000, 16 above (measured on MI355X, profiles/r4_linkage_*:
// N = 7 000: 8 workgroups 109 ms, 16: 113 ms, 1: 170 ms; N = 20 000: 8: 407 ms, 16: 366 ms).
// PA_LINKAGE_FAST_WGS (1 ... 16) overrides; raised when the rows of a workgroup would not fit its LDS.
static int lf_num_workgroups(int n) {
  const int chunks = cdiv(n, LF_T);
  const char* e = getenv("PA_LINKAGE_FAST_WGS");
  int G = (e != nullptr && atoi(e) >= 1) ? atoi(e) : (n <= 1024 ? 1 : (n < 12000 ? 8 : LF_MAXG));
  if (G > LF_MAXG) G = LF_MAXG;
  if (G > chunks) G = chunks;
  if (G == 1 && ((size_t)chunks * LF_T * 14 > LF_LDS_MAX || n > 32767)) G = 2;
  while (G > 1 && (size_t)cdiv(chunks, G) * LF_T * 16 > LF_LDS_MAX && G < LF_MAXG) ++G;
  return G;
}

// bound of a mailbox poll: 2 s of the constant-rate clock wall_clock64() reads (hipDeviceAttributeWallClockRate, kHz;
// 100 MHz on this chip); PA_LINKAGE_POLL_MS overrides (tests force a give-up with it)
static long long lf_poll_limit_ticks() {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess ||
      khz <= 0)
    khz = 100000;
  const char* e = getenv("PA_LINKAGE_POLL_MS");
  const double ms = (e != nullptr && atof(e) > 0) ? atof(e) : 2000.0;
  return (long long)(ms * (double)khz);
}

// launches square conversion + initial candidates + the merge kernel on `st`; *gate_out = device address of the
// status word (0 after the kernel = Z is complete).  `stats`: 8 int64 of development counters.
int lf_launch(const double* cond, int n, double* Z, void* workspace, long long* stats, int** gate_out,
              hipStream_t st) {
  const size_t ni = lf_align(sizeof(int) * (size_t)n), nd = lf_align(sizeof(double) * (size_t)n);
  const long ld = lf_ld(n);
  unsigned char* w = (unsigned char*)workspace;
  double* S = (double*)w;
  w += lf_align(8 * (size_t)n * (size_t)ld);
  int* nb0 = (int*)w;
  int* g_size = (int*)(w + ni);
  int* g_cid = (int*)(w + 2 * ni);
  double* mind0 = (double*)(w + 3 * ni);
  lf_u64* mail = (lf_u64*)(w + 3 * ni + nd);
  int* status = (int*)(w + 3 * ni + nd + lf_align(2 * LF_MAXG * 16 * 8));
  *gate_out = status;
  if (hipMemsetAsync(mail, 0, 2 * LF_MAXG * 16 * 8, st) != hipSuccess) return 1;
  if (hipMemsetAsync(status, 0xff, sizeof(int), st) != hipSuccess) return 1;   // "not run" = take the heap
  const int nt = cdiv(n, 64);
  hipLaunchKernelGGL(k_lf_square, dim3(nt, nt), dim3(256), 0, st, cond, n, ld, S);
  hipLaunchKernelGGL(k_lf_row_nearest, dim3(cdiv(n - 1, 4)), dim3(256), 0, st, cond, n, nb0, mind0);
  const int G = lf_num_workgroups(n);
  const int SL = cdiv(cdiv(n, LF_T), G);
  const long long poll_limit = lf_poll_limit_ticks();
  if (G == 1) {
    const size_t lds = (size_t)SL * LF_T * 14;
    (void)hipFuncSetAttribute((const void*)k_linkage_fast<unsigned short, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LF_LDS_MAX);
    hipLaunchKernelGGL((k_linkage_fast<unsigned short, false>), dim3(1), dim3(LF_T), lds, st, S, ld, n, Z, nb0, mind0,
                       g_size, g_cid, mail, 1, SL, poll_limit, status, stats);
  } else {
    const size_t lds = (size_t)SL * LF_T * 16;
    if (lds > LF_LDS_MAX) return 0;   // (status stays "not run": the heap kernel does the work)
    (void)hipFuncSetAttribute((const void*)k_linkage_fast<unsigned int, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LF_LDS_MAX);
    hipLaunchKernelGGL((k_linkage_fast<unsigned int, true>), dim3(8 * G), dim3(LF_T), lds, st, S, ld, n, Z, nb0, mind0,
                       g_size, g_cid, mail, G, SL, poll_limit, status, stats);
  }
  return 0;
}

}  // namespace pa
