// Centroid-linkage dendrogram, fast path (round 4).  Same contract as linkage.hip -- bit-identical to
// scipy.cluster.hierarchy.linkage(y, "centroid"), i.e. what AgglomerativeClustering.cluster calls (reference:
// pipelines/clustering.py:374-382) -- but without SciPy's binary heap on the critical path.
//
// SciPy's `fast_linkage` (Muellner's generic algorithm) keeps a lower bound min_dist[z] and a neighbour candidate
// per row in a min-heap and pops the root until the root's bound is exact.  The heap only decides WHICH row is the
// root; when the smallest bound is attained by exactly one row, the root is that row whatever the heap looks like
// inside.  So the merge loop can be run with a parallel arg-min over min_dist[] instead of a heap, as long as no two
// rows ever tie for the smallest bound at a pop.  This kernel does that and tracks ties in the reduction: the first
// pop whose minimum is not unique ends the kernel with status 1, and the launcher's NEXT kernel -- the exact heap
// replay of linkage.hip, gated on that status word -- recomputes the dendrogram from the untouched condensed matrix.
// Real embeddings never tie (float64 distances of 256-dimensional vectors); duplicated rows do, and take the heap.
//
// What made the heap kernel 25 us per merge (round 3, N = 7 176: pass 29 k cycles, find 14 k, replay 12 k, waits 5 k)
// and what replaces it:
//   * column accesses D[z][x], D[z][y] for z < x in the CONDENSED matrix: one cache line per lane, 3.9 cycles per
//     cluster on one CU.  Here the merge works on a SQUARE symmetric copy (k_lf_square, N x ld doubles; 412 MB at
//     N = 7 176, 26 GB at 57 k -- HBM3E is 288 GB): rows x and y are read coalesced, the new row y is written
//     coalesced, and only the mirror column D[z][y] is scattered -- as stores, which nothing waits for.
//   * `dist == D[x][neighbor[x]]`, a dependent global load per pop: replaced by an EXACT bit per row, maintained
//     where the bound or the matrix entry changes (the comparison SciPy makes at the pop, made at the update).
//   * the serial heap replay (sort + ~12 sifts by one lane): gone; the owner thread of row z updates min_dist[z],
//     neighbor[z] in place.  All per-row state (bound, neighbour + exact bit, size, id) lives in LDS up to
//     N = 11 154 (14 B per row), in global memory (L2) above.
// One persistent 1 024-thread workgroup; two workgroup barriers per merge + two per lower-bound repair.
// hipcc-flags: -ffp-contract=off
#include <stdlib.h>

#include "common.h"

namespace pa {

constexpr int LF_T = 1024;
constexpr int LF_W = LF_T / 64;
constexpr int LF_PU = 8;   // rows per thread and trip (all loads of a trip are issued before the first use)

struct LfMin {
  double d;
  int i;     // -1: none
  int tie;   // the minimum is attained more than once
};

// smaller value wins, equal values -> smaller index (+ tie); +inf and NaN never win (they are never candidates)
__device__ __forceinline__ LfMin lf_combine(const LfMin a, const LfMin b) {
  if (b.i < 0) return a;
  if (a.i < 0) return b;
  if (b.d < a.d) return b;
  if (a.d < b.d) return a;
  LfMin r = a.i < b.i ? a : b;
  r.tie = 1;
  return r;
}

__device__ __forceinline__ LfMin lf_wave_reduce(LfMin v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    LfMin other;
    other.d = __shfl_xor(v.d, o, 64);
    other.i = __shfl_xor(v.i, o, 64);
    other.tie = __shfl_xor(v.tie, o, 64);
    v = lf_combine(v, other);
  }
  return v;
}

// block-wide: per-wave results through `red` (one barrier), every thread combines the LF_W entries itself
__device__ __forceinline__ LfMin lf_block_reduce(LfMin v, LfMin* red) {
  v = lf_wave_reduce(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  LfMin r = red[0];
#pragma unroll
  for (int q = 1; q < LF_W; ++q) r = lf_combine(r, red[q]);
  return r;
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
  LfMin best{__builtin_inf(), -1, 0};
  for (int i0 = x + 1 + lane; i0 < n; i0 += 4 * 64) {
    double d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = (i0 + 64 * u < n) ? cond[base + i0 + 64 * u] : __builtin_inf();
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (d[u] < best.d) {
        best.d = d[u];
        best.i = i0 + 64 * u;
      }
  }
  best = lf_wave_reduce(best);
  if (lane == 0) {
    nb0[x] = best.i;
    mind0[x] = best.i < 0 ? __builtin_inf() : best.d;
  }
}

// status word: 0 = dendrogram complete, 1 = tie at a pop (take the heap), 2 = degenerate input (all bounds
// infinite / NaN, or more repairs than SciPy's loop allows) -- anything but 0 lets the gated heap kernel run.
template <typename IT, bool LDS_STATE>
__global__ __launch_bounds__(LF_T) void k_linkage_fast(double* __restrict__ S, long ld, int n,
                                                        double* __restrict__ Z, const int* __restrict__ nb0,
                                                        const double* __restrict__ mind0,
                                                        double* __restrict__ g_mind, int* __restrict__ g_nb,
                                                        int* __restrict__ g_size, int* __restrict__ g_cid,
                                                        int* __restrict__ status, long long* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ LfMin redA[LF_W], redB[LF_W], redC[LF_W];
  constexpr IT EX = (IT)((IT)1 << (8 * sizeof(IT) - 1));   // "the bound of this row is exact"
  constexpr IT NONE = (IT)(EX - 1);                         // no neighbour candidate
  const int tid = threadIdx.x;
  double* mind;   // SciPy's min_dist (+inf once the row left the heap)
  IT* nb;         // neighbour candidate | EX
  IT* size;
  IT* cid;
  if (LDS_STATE) {
    mind = reinterpret_cast<double*>(lds_raw);
    nb = reinterpret_cast<IT*>(mind + n);
    size = nb + n;
    cid = size + n;
  } else {
    mind = g_mind;
    nb = reinterpret_cast<IT*>(g_nb);
    size = reinterpret_cast<IT*>(g_size);
    cid = reinterpret_cast<IT*>(g_cid);
  }
  for (int i = tid; i < n; i += LF_T) {
    size[i] = (IT)1;
    cid[i] = (IT)i;
    const int c = i < n - 1 ? nb0[i] : -1;
    nb[i] = c < 0 ? NONE : (IT)((IT)c | EX);
    mind[i] = (i < n - 1 && c >= 0) ? mind0[i] : __builtin_inf();
  }
  __syncthreads();

  long long st_rep = 0, st_c0 = 0, st_c1 = 0;
  int fail = 0;
  for (int k = 0; k < n - 1 && !fail; ++k) {
    long long tc = __builtin_readcyclecounter();
    // ---- pop: the row with the smallest bound; repair its bound until it is exact
    int x = 0, y = 0;
    double dist = 0.0;
    for (int it = 0;; ++it) {
      LfMin b{__builtin_inf(), -1, 0};
      for (int z0 = tid; z0 < n; z0 += LF_PU * LF_T) {
        double m[LF_PU];
#pragma unroll
        for (int u = 0; u < LF_PU; ++u) m[u] = (z0 + u * LF_T < n) ? mind[z0 + u * LF_T] : __builtin_inf();
#pragma unroll
        for (int u = 0; u < LF_PU; ++u) {
          if (m[u] < b.d) {
            b.d = m[u];
            b.i = z0 + u * LF_T;
            b.tie = 0;
          } else if (m[u] == b.d && b.i >= 0) {
            b.tie = 1;
          }
        }
      }
      const LfMin r = lf_block_reduce(b, redA);
      if (r.i < 0 || r.tie || it > n - k) {   // (uniform: every thread holds the same r)
        fail = r.i >= 0 && r.tie ? 1 : 2;
        break;
      }
      x = r.i;
      dist = r.d;
      const IT nbx = nb[x];
      y = (IT)(nbx & NONE) == NONE ? -1 : (int)(nbx & NONE);
      if ((nbx & EX) != 0 && y >= 0) break;
      // lower-bound repair: find_min_dist(n, D, size, x) over the active rows i > x (row x of the square matrix)
      LfMin s{__builtin_inf(), -1, 0};
      const double* row = S + (long)x * ld;
      for (int i0 = tid; i0 < n; i0 += LF_PU * LF_T) {
        double d[LF_PU];
#pragma unroll
        for (int u = 0; u < LF_PU; ++u) {
          const int i = i0 + u * LF_T;
          d[u] = (i < n && i > x && size[i] != 0) ? row[i] : __builtin_inf();
        }
#pragma unroll
        for (int u = 0; u < LF_PU; ++u)
          if (d[u] < s.d) {
            s.d = d[u];
            s.i = i0 + u * LF_T;
          }
      }
      const LfMin r2 = lf_block_reduce(s, redC);
      if (tid == (x & (LF_T - 1))) {
        nb[x] = r2.i < 0 ? NONE : (IT)((IT)r2.i | EX);
        mind[x] = r2.i < 0 ? __builtin_inf() : r2.d;
      }
      ++st_rep;
      // (no barrier: the owner re-reads its own row in the next arg-min, everybody else reads nb[x] behind that
      //  reduction's barrier)
    }
    if (fail) break;
    {
      const long long t2 = __builtin_readcyclecounter();
      st_c0 += t2 - tc;
      tc = t2;
    }
    // ---- record the merge (the state writes wait until every thread is past its reads: after the pass)
    const int nx = (int)size[x], ny = (int)size[y];
    if (tid == 0) {
      int id_x = (int)cid[x], id_y = (int)cid[y];
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
    // ---- one pass over the rows z: Lance-Williams (centroid) update of D[z][y] = D[y][z]; neighbour reassignment
    // x -> y and lower-bound update for z < y (SciPy's loops 2 and 3, by the owner thread of z); nearest neighbour
    // of y among z > y (loop 4)
    LfMin best{__builtin_inf(), -1, 0};
    const double* rowx = S + (long)x * ld;
    double* rowy = S + (long)y * ld;
    // (row state in LDS is read where it is used: 100 cycles, and 24 registers less than holding it across the
    //  square roots; row state in global memory is loaded with the matrix rows, PP rows per trip)
    constexpr int PP = LDS_STATE ? LF_PU : LF_PU / 2;
    for (int z0 = tid; z0 < n; z0 += PP * LF_T) {
      bool act[PP];
      double d_xi[PP], d_yi[PP], m[PP];
      IT nbz[PP];
#pragma unroll
      for (int u = 0; u < PP; ++u) {
        const int z = z0 + u * LF_T;
        act[u] = z < n && z != y && z != x && size[z] != 0;
        d_xi[u] = act[u] ? rowx[z] : 0.0;
        d_yi[u] = act[u] ? rowy[z] : 0.0;
        if (!LDS_STATE) {
          m[u] = (act[u] && z < y) ? mind[z] : 0.0;
          nbz[u] = (act[u] && z < y) ? nb[z] : (IT)0;
        }
      }
#pragma unroll
      for (int u = 0; u < PP; ++u) {
        if (!act[u]) continue;
        const int z = z0 + u * LF_T;
        const double nd = sqrt(
            (((nx * d_xi[u] * d_xi[u]) + (ny * d_yi[u] * d_yi[u])) - ((nx * ny) * dist * dist) / (nx + ny)) /
            (nx + ny));
        rowy[z] = nd;
        S[(long)z * ld + y] = nd;
        if (z < y) {
          if (LDS_STATE) {
            m[u] = mind[z];
            nbz[u] = nb[z];
          }
          const int c = (int)(nbz[u] & NONE);
          if (nd < m[u]) {
            mind[z] = nd;
            nb[z] = (IT)((IT)y | EX);
          } else if (c == x || c == y) {
            nb[z] = (IT)((IT)y | (m[u] == nd ? EX : (IT)0));
          }
        } else if (nd < best.d) {   // z > y, ascending per thread: first minimum
          best.d = nd;
          best.i = z;
        }
      }
    }
    const LfMin r = lf_block_reduce(best, redB);
    if (tid == (x & (LF_T - 1))) {
      mind[x] = __builtin_inf();   // heap.remove_min()
      size[x] = (IT)0;
    }
    if (tid == (y & (LF_T - 1))) {
      size[y] = (IT)(nx + ny);
      cid[y] = (IT)(n + k);
      if (y < n - 1 && r.i >= 0) {
        nb[y] = (IT)((IT)r.i | EX);
        mind[y] = r.d;
      }
    }
    // (no barrier here: the next pop's block reduction has one before anybody reads another thread's rows)
    st_c1 += __builtin_readcyclecounter() - tc;
  }
  if (tid == 0) {
    *status = fail;
    if (stats != nullptr) {
      stats[0] = fail;
      stats[1] = st_rep;
      stats[2] = st_c0;
      stats[3] = st_c1;
      stats[4] = n;
    }
  }
}

constexpr size_t LF_LDS_MAX = 160 * 1024 - 4096;   // dynamic LDS budget (static part: 3 x 256 B + spill of nothing)

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
  return 8.0 * (double)n * (double)lf_ld(n) <= cap;
}

// bytes the fast path adds to the linkage workspace: square matrix + initial candidates + global row state + status
size_t lf_workspace_bytes(int n) {
  if (!lf_wanted(n)) return 0;
  const size_t ni = lf_align(sizeof(int) * (size_t)n), nd = lf_align(sizeof(double) * (size_t)n);
  return lf_align(8 * (size_t)n * (size_t)lf_ld(n)) + 4 * ni + 2 * nd + 256;
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
  int* g_nb = (int*)(w + ni);
  int* g_size = (int*)(w + 2 * ni);
  int* g_cid = (int*)(w + 3 * ni);
  double* mind0 = (double*)(w + 4 * ni);
  double* g_mind = (double*)(w + 4 * ni + nd);
  int* status = (int*)(w + 4 * ni + 2 * nd);
  *gate_out = status;
  if (hipMemsetAsync(status, 0xff, sizeof(int), st) != hipSuccess) return 1;   // "not run" = take the heap
  const int nt = cdiv(n, 64);
  hipLaunchKernelGGL(k_lf_square, dim3(nt, nt), dim3(256), 0, st, cond, n, ld, S);
  hipLaunchKernelGGL(k_lf_row_nearest, dim3(cdiv(n - 1, 4)), dim3(256), 0, st, cond, n, nb0, mind0);
  const size_t lds = ((size_t)n * 14 + 15) & ~(size_t)15;
  if (n <= 32767 && lds <= LF_LDS_MAX) {
    (void)hipFuncSetAttribute((const void*)k_linkage_fast<unsigned short, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LF_LDS_MAX);
    hipLaunchKernelGGL((k_linkage_fast<unsigned short, true>), dim3(1), dim3(LF_T), lds, st, S, ld, n, Z, nb0, mind0,
                       g_mind, g_nb, g_size, g_cid, status, stats);
  } else {
    hipLaunchKernelGGL((k_linkage_fast<unsigned int, false>), dim3(1), dim3(LF_T), 0, st, S, ld, n, Z, nb0, mind0, g_mind, g_nb,
                       g_size, g_cid, status, stats);
  }
  return 0;
}

}  // namespace pa
