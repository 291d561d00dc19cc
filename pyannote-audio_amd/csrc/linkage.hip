// Centroid-linkage dendrogram on gfx950, bit-identical to scipy.cluster.hierarchy.linkage(y, "centroid")
// (SciPy 1.15.3 `_hierarchy.fast_linkage` = Muellner's "generic clustering algorithm" with a
//  nearest-neighbour candidate per row and a binary min-heap of lower bounds), which is what
//  AgglomerativeClustering.cluster calls (reference: pipelines/clustering.py:374-382).
//
// Why a kernel: the merge loop is inherently serial (N-1 dependent merges) but every merge does O(N)
// independent work -- Lance-Williams update of one row/column of the distance matrix, two
// nearest-neighbour row scans, the lower-bound refresh -- which SciPy walks on one host core
// (0.8 s for N = 7 000, the Amdahl term of the whole pipeline).  Here ONE persistent 1024-thread
// workgroup keeps the condensed matrix in HBM/L2 (it is produced there by k_pdist_f64 and never
// crosses PCIe), runs the O(N) parts data-parallel and leaves only the heap sifts (O(log N), heap in
// LDS) to lane 0.  No grid-wide synchronisation is needed: a single workgroup, __syncthreads only.
//
// Exactness contract (tests/test_pipeline_gpu.py::test_linkage_*): same merge order, same float64
// heights, same tie behaviour as SciPy, because
//   * the distance update is SciPy's expression evaluated left to right in double with separately
//     rounded operations (this file is compiled with -ffp-contract=off):
//       sqrt((((sx*dxi*dxi) + (sy*dyi*dyi)) - (sx*sy*dxy*dxy)/(sx+sy)) / (sx+sy))
//   * row scans return the FIRST index attaining the minimum (SciPy scans with a strict `<`);
//   * all heap operations (build, change_value, remove_min, sift_up/down) are executed by one lane in
//     exactly SciPy's order, including the ascending-z order of the lower-bound refresh.
// hipcc-flags: -ffp-contract=off
#include <stdlib.h>

#include "common.h"

namespace pa {

constexpr int LK_T = 1024;  // threads (16 waves)
constexpr int LK_W = LK_T / 64;

__device__ __forceinline__ long cidx(long n, long i, long j) {
  // scipy condensed_index(n, i, j)
  return i < j ? n * i - (i * (i + 1) / 2) + (j - i - 1) : n * j - (j * (j + 1) / 2) + (i - j - 1);
}

// SciPy's Heap (scipy/cluster/_structures.pxi) with "hole" sifts: the moving element is held in
// registers and written once at its final position.  The final arrays are identical to the
// swap-by-swap version (same comparisons, same order), at ~1/3 of the dependent LDS traffic.
template <typename IT>
struct Heap {
  double* v;  // values by heap position
  IT* kbi;    // key_by_index
  IT* ibk;    // index_by_key
  int size;
  __device__ __forceinline__ void place(int index, double val, IT key) {
    v[index] = val;
    kbi[index] = key;
    ibk[key] = (IT)index;
  }
  __device__ void sift_up(int index, double val, IT key) {
    while (index > 0) {
      const int parent = (index - 1) >> 1;
      const double pv = v[parent];
      const IT pk = kbi[parent];
      if (!(pv > val)) break;
      place(index, pv, pk);
      index = parent;
    }
    place(index, val, key);
  }
  __device__ void sift_down(int index, double val, IT key) {
    int child = 2 * index + 1;
    while (child < size) {
      double cv = v[child];
      if (child + 1 < size) {
        const double cv1 = v[child + 1];
        if (cv1 < cv) {
          child += 1;
          cv = cv1;
        }
      }
      if (!(val > cv)) break;
      place(index, cv, kbi[child]);
      index = child;
      child = 2 * index + 1;
    }
    place(index, val, key);
  }
  __device__ void build() {  // Heap.__init__: sift_down from the last parent to the root
    for (int i = size / 2 - 1; i >= 0; --i) sift_down(i, v[i], kbi[i]);
  }
  __device__ void change_value(int key, double value) {
    const int index = ibk[key];
    const double old = v[index];
    if (value < old) sift_up(index, value, (IT)key);
    else sift_down(index, value, (IT)key);
  }
  __device__ void remove_min() {
    const int last = size - 1;
    const double lv = v[last];
    const IT lk = kbi[last];
    place(last, v[0], kbi[0]);  // swap(0, size - 1): the removed root parks behind the heap
    size -= 1;
    if (size > 0) sift_down(0, lv, lk);
  }
};

struct MinPair {
  double d;
  int i;
};
// lexicographic "first minimum": smaller value wins, equal values -> smaller index; NaN never wins
__device__ __forceinline__ MinPair min_pair(MinPair a, MinPair b) {
  if (b.i >= 0 && (a.i < 0 || b.d < a.d || (b.d == a.d && b.i < a.i))) return b;
  return a;
}

// Accesses to data that ANOTHER workgroup of the multi-workgroup kernel writes or reads: relaxed agent-scope
// atomics = `global_load / global_store ... sc1`: loads are served by L2 (never by this CU's L1, which no other
// CU's store refreshes), stores go through to memory.  With sc1 on BOTH sides no fence is needed and the protocol
// does not depend on where the workgroups run (MI355X_MICROARCH.md, "inter-workgroup visibility"); a
// __threadfence() per barrier instead costs 3.5-10 us (L2 write-back + L1 invalidate) -- measured here: the
// fenced version of this kernel was SLOWER than one workgroup at every size.
template <bool COH, typename T>
__device__ __forceinline__ T lk_ld(const T* p) {
  if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <typename T>
__device__ __forceinline__ void lk_st(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// find_min_dist(n, D, size, x): nearest active neighbour of x among indices > x.  All threads call;
// result valid in every thread.  `red` = LK_W MinPairs of LDS.
template <bool COH = false>
__device__ MinPair block_find_min(const double* __restrict__ D, const int* __restrict__ size, int n,
                                  int x, MinPair* red) {
  MinPair best{__builtin_inf(), -1};
  const long base = (long)n * x - ((long)x * (x + 1) / 2) - x - 1;  // cidx(n, x, i) = base + i
  for (int i0 = x + 1 + threadIdx.x; i0 < n; i0 += 4 * LK_T) {
    // 4 row elements per thread in flight (ascending i, so the strict `<` keeps the first minimum)
    double d[4];
    bool act[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * LK_T;
      act[u] = i < n && lk_ld<COH>(size + i) != 0;
      d[u] = act[u] ? lk_ld<COH>(D + base + i) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (act[u] && d[u] < best.d) {
        best.d = d[u];
        best.i = i0 + u * LK_T;
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MinPair other;
    other.d = __shfl_xor(best.d, o, 64);
    other.i = __shfl_xor(best.i, o, 64);
    best = min_pair(best, other);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = best;
  __syncthreads();
  MinPair r = red[0];
#pragma unroll
  for (int k = 1; k < LK_W; ++k) r = min_pair(r, red[k]);
  if (r.i < 0) r.d = __builtin_inf();
  return r;
}

constexpr int LK_PU = 8;    // clusters per thread and trip of the single-workgroup z pass
constexpr int LK_PEND = 256;  // lower-bound drops buffered per merge (more -> re-read from D)

template <typename IT, bool LDS_HEAP>
__global__ __launch_bounds__(LK_T) void k_linkage_centroid(double* __restrict__ D, int n,
                                                            double* __restrict__ Z,
                                                            int* __restrict__ size,
                                                            int* __restrict__ cluster_id,
                                                            double* __restrict__ g_hv,
                                                            int* __restrict__ g_kbi,
                                                            int* __restrict__ g_ibk,
                                                            int* __restrict__ g_nb,
                                                            long long* __restrict__ stats,
                                                            const int* __restrict__ gate) {
  // `gate`: status word of the fast path (linkage_fast.hip) that ran in front of this launch on the same stream;
  // 0 = the dendrogram is already complete
  if (gate != nullptr && *gate == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ MinPair red[LK_W];
  __shared__ int sh_x, sh_y, sh_ok, sh_nx, sh_ny, sh_npend;
  __shared__ double sh_dist;
  __shared__ int pend_z[LK_PEND], sort_z[LK_PEND];
  __shared__ double pend_d[LK_PEND], sort_d[LK_PEND];
  const int tid = threadIdx.x;
  const int hn = n - 1;  // heap capacity = rows that own a nearest-neighbour candidate
  constexpr IT NONE = (IT)~(IT)0;  // "no neighbour" (-1)

  // per-row state: heap (values = SciPy's min_dist, kept in sync with it), neighbour candidates
  Heap<IT> heap;
  IT* nb;
  unsigned int* cand;  // bitmap of rows whose lower bound dropped in this merge
  if (LDS_HEAP) {
    heap.v = reinterpret_cast<double*>(lds_raw);
    heap.kbi = reinterpret_cast<IT*>(heap.v + hn);
    heap.ibk = heap.kbi + hn;
    nb = heap.ibk + hn;
    cand = reinterpret_cast<unsigned int*>(
        lds_raw + (((size_t)hn * (8 + 3 * sizeof(IT)) + 15) & ~(size_t)15));
  } else {
    heap.v = g_hv;
    heap.kbi = reinterpret_cast<IT*>(g_kbi);
    heap.ibk = reinterpret_cast<IT*>(g_ibk);
    nb = reinterpret_cast<IT*>(g_nb);
    cand = reinterpret_cast<unsigned int*>(lds_raw);
  }
  heap.size = hn;
  const int cand_words = (n + 31) / 32;

  for (int i = tid; i < n; i += LK_T) {
    size[i] = 1;
    cluster_id[i] = i;
  }
  for (int i = tid; i < cand_words; i += LK_T) cand[i] = 0u;
  // initial nearest-neighbour candidates (one wave per row); heap position i holds key i for now
  {
    const int lane = tid & 63, w = tid >> 6;
    for (int x = w; x < n - 1; x += LK_W) {
      MinPair best{__builtin_inf(), -1};
      const long base = (long)n * x - ((long)x * (x + 1) / 2) - x - 1;
      for (int i = x + 1 + lane; i < n; i += 64) {
        const double d = D[base + i];
        if (d < best.d) {
          best.d = d;
          best.i = i;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        MinPair other;
        other.d = __shfl_xor(best.d, o, 64);
        other.i = __shfl_xor(best.i, o, 64);
        best = min_pair(best, other);
      }
      if (lane == 0) {
        nb[x] = best.i < 0 ? NONE : (IT)best.i;
        heap.v[x] = best.i < 0 ? __builtin_inf() : best.d;
        heap.kbi[x] = (IT)x;
        heap.ibk[x] = (IT)x;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    heap.build();
    sh_npend = 0;
  }
  __syncthreads();

  // development counters (lane 0): [0] lower-bound repairs, [1] heap updates of the refresh,
  // [2] refreshes that overflowed the pending buffer, [3..6] cycles in find / record / pass / replay
  long long st_retry = 0, st_cand = 0, st_ovf = 0, st_c0 = 0, st_c1 = 0, st_c2 = 0, st_c3 = 0;
  for (int k = 0; k < n - 1; ++k) {
    // ---- find the two closest clusters: at most n - k lower-bound repairs
    int x = 0, y = 0;
    double dist = 0.0;
    long long tc = __builtin_readcyclecounter();
    for (int it = 0; it < n - k; ++it) {
      if (tid == 0) {
        const int hx = heap.kbi[0];
        const double hd = heap.v[0];
        const IT hyr = nb[hx];
        const int hy = hyr == NONE ? -1 : (int)hyr;
        sh_x = hx;
        sh_y = hy;
        sh_dist = hd;
        sh_ok = (hy >= 0 && hd == D[cidx(n, hx, hy)]) ? 1 : 0;
      }
      __syncthreads();
      x = sh_x;
      y = sh_y;
      dist = sh_dist;
      const int ok = sh_ok;
      if (ok) break;
      const MinPair p = block_find_min(D, size, n, x, red);  // (barriers inside)
      y = p.i;
      dist = p.d;
      if (tid == 0) {
        nb[x] = y < 0 ? NONE : (IT)y;
        heap.change_value(x, dist);
        ++st_retry;
      }
      __syncthreads();
    }
    {
      const long long t2 = __builtin_readcyclecounter();
      st_c0 += t2 - tc;
      tc = t2;
    }
    // ---- record the merge
    if (tid == 0) {
      heap.remove_min();
      int id_x = cluster_id[x], id_y = cluster_id[y];
      const int nx = size[x], ny = size[y];
      if (id_x > id_y) {
        const int t = id_x;
        id_x = id_y;
        id_y = t;
      }
      Z[4 * (long)k + 0] = (double)id_x;
      Z[4 * (long)k + 1] = (double)id_y;
      Z[4 * (long)k + 2] = dist;
      Z[4 * (long)k + 3] = (double)(nx + ny);
      size[x] = 0;
      size[y] = nx + ny;
      cluster_id[y] = n + k;
      sh_nx = nx;
      sh_ny = ny;
    }
    __syncthreads();
    {
      const long long t2 = __builtin_readcyclecounter();
      st_c1 += t2 - tc;
      tc = t2;
    }
    const int nx = sh_nx, ny = sh_ny;
    // ---- ONE pass over all clusters z (SciPy's four loops are independent per z except for the heap,
    // which is replayed afterwards): Lance-Williams (centroid) update of D[z,y]; neighbour
    // reassignment x -> y for z < x; lower-bound refresh for z < y; nearest neighbour of y among z > y.
    MinPair best{__builtin_inf(), -1};
    for (int z0 = tid; z0 < n; z0 += LK_PU * LK_T) {
      // LK_PU clusters per thread: all distance loads are issued before the first use (one audio-hour = 7 176
      // clusters = ONE trip: the pass is a latency chain, a second trip doubles it)
      bool act[LK_PU];
      long izy[LK_PU];
      double d_xi[LK_PU], d_yi[LK_PU];
#pragma unroll
      for (int u = 0; u < LK_PU; ++u) {
        const int z = z0 + u * LK_T;
        act[u] = z < n && z != y && size[z] != 0;
        izy[u] = act[u] ? cidx(n, z, y) : 0;
        d_xi[u] = act[u] ? D[cidx(n, z, x)] : 0.0;
        d_yi[u] = act[u] ? D[izy[u]] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < LK_PU; ++u) {
        if (!act[u]) continue;
        const int z = z0 + u * LK_T;
        const double nd = sqrt(
            (((nx * d_xi[u] * d_xi[u]) + (ny * d_yi[u] * d_yi[u])) - ((nx * ny) * dist * dist) / (nx + ny)) /
            (nx + ny));
        D[izy[u]] = nd;
        if (z < y) {
          if (z < x && nb[z] == (IT)x) nb[z] = (IT)y;
          if (nd < heap.v[heap.ibk[z]]) {  // heap value of key z == SciPy's min_dist[z]
            nb[z] = (IT)y;
            atomicOr(&cand[z >> 5], 1u << (z & 31));
            const int slot = atomicAdd(&sh_npend, 1);
            if (slot < LK_PEND) {
              pend_z[slot] = z;
              pend_d[slot] = nd;
            }
          }
        } else if (nd < best.d) {  // z > y, ascending per thread: first minimum
          best.d = nd;
          best.i = z;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      MinPair other;
      other.d = __shfl_xor(best.d, o, 64);
      other.i = __shfl_xor(best.i, o, 64);
      best = min_pair(best, other);
    }
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    {
      const long long t2 = __builtin_readcyclecounter();
      st_c2 += t2 - tc;
      tc = t2;
    }
    // ---- replay the heap updates in SciPy's order: ascending z < y, then row y.  The (few) refreshed
    // rows are rank-sorted by z in parallel; lane 0 then only sifts.
    const int np = sh_npend;  // (reset by lane 0 only after the next barrier)
    if (np <= LK_PEND && tid < np) {
      const int z = pend_z[tid];
      int rank = 0;
      for (int q = 0; q < np; ++q) rank += pend_z[q] < z ? 1 : 0;
      sort_z[rank] = z;
      sort_d[rank] = pend_d[tid];
      cand[z >> 5] = 0u;  // (racing writers all store 0)
    }
    __syncthreads();
    if (tid == 0) {
      sh_npend = 0;
      st_cand += np;
      if (np <= LK_PEND) {
        for (int q = 0; q < np; ++q) heap.change_value(sort_z[q], sort_d[q]);
      } else {
        ++st_ovf;
        const int words = (y + 31) / 32;
        for (int wi = 0; wi < words; ++wi) {
          unsigned int m = cand[wi];
          if (!m) continue;
          cand[wi] = 0u;
          while (m) {
            const int bit = __builtin_ctz(m);
            m &= m - 1;
            const int z = wi * 32 + bit;
            heap.change_value(z, D[cidx(n, z, y)]);
          }
        }
      }
      if (y < n - 1) {
        MinPair r = red[0];
#pragma unroll
        for (int q = 1; q < LK_W; ++q) r = min_pair(r, red[q]);
        if (r.i != -1) {
          nb[y] = (IT)r.i;
          heap.change_value(y, r.d);
        }
      }
    }
    __syncthreads();
    st_c3 += __builtin_readcyclecounter() - tc;
  }
  if (tid == 0 && stats != nullptr) {
    stats[0] = st_retry;
    stats[1] = st_cand;
    stats[2] = st_ovf;
    stats[3] = st_c0;
    stats[4] = st_c1;
    stats[5] = st_c2;
    stats[6] = st_c3;
    stats[7] = n;
  }
}

// =============================================================================================
// Multi-workgroup form.  The O(N) part of a merge -- the Lance-Williams update of column y, the neighbour
// fix-ups, the lower-bound candidates and the nearest-neighbour scan of row y -- reads ~2 N scattered
// distances; one CU sustains only ~60-100 GB/s of such traffic, which is what made the single workgroup take
// 240 us per merge at N = 57 k (13.9 s for the joint clustering of 8 audio-hours).  Here G workgroups, all on
// ONE XCD (workgroup w runs on XCD w mod 8: the launch has 8 G workgroups and only every 8th works, so the
// matrix stays coherent in one L2), split that pass; workgroup 0 alone keeps SciPy's heap and replays its
// updates in SciPy's order, so the dendrogram stays bit-identical.  Two grid barriers per merge (fence +
// atomic counter + generation word, ~2 us each inside an XCD):
//     WG0: find the closest pair (lower-bound repairs on its own), record the merge, publish (x, y, sizes)
//     -- barrier --   all: one slice of the z pass; refreshed rows appended to a global pending list,
//                     per-workgroup nearest neighbour of y
//     -- barrier --   WG0: rank-sort the pending rows, heap updates by lane 0, neighbour of y
// Everything two workgroups share is read and written with sc1 accesses (lk_ld / lk_st): no fences.
// `mind[z]` mirrors the heap value of key z (SciPy's min_dist[z]) in global memory for the other workgroups.
// =============================================================================================
struct LkShared {          // global memory, zero-initialised by the launcher
  int bar_count, bar_gen;
  int x, y, nx, ny;
  double dist;
  int npend;
  int pad_;
  MinPair red[32];
  int pend_z[LK_PEND];
  double pend_d[LK_PEND];
};

// Grid barrier over the G participating workgroups: every thread waits for its own (sc1) stores to be
// acknowledged, lane 0 arrives on an agent-scope counter and polls the generation word.  No fence: all shared
// data is accessed with sc1 loads / stores (see lk_ld / lk_st).
__device__ __forceinline__ void lk_grid_barrier(LkShared* sh, int G) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int gen = __hip_atomic_load(&sh->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(&sh->bar_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
      __hip_atomic_store(&sh->bar_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&sh->bar_gen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&sh->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen)
        __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

template <typename IT, bool LDS_HEAP>
__global__ __launch_bounds__(LK_T) void k_linkage_centroid_mw(double* __restrict__ D, int n,
                                                               double* __restrict__ Z,
                                                               int* __restrict__ size,
                                                               int* __restrict__ cluster_id,
                                                               double* __restrict__ g_hv,
                                                               int* __restrict__ g_kbi,
                                                               int* __restrict__ g_ibk,
                                                               int* __restrict__ g_nb,
                                                               double* __restrict__ mind,
                                                               unsigned int* __restrict__ cand,
                                                               LkShared* __restrict__ sh, int G,
                                                               long long* __restrict__ stats,
                                                               const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;   // the fast path (linkage_fast.hip) completed the dendrogram
  if ((blockIdx.x & 7) != 0) return;     // only the workgroups of one XCD take part
  const int wg = blockIdx.x >> 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ MinPair red[LK_W];
  __shared__ int sh_x, sh_y, sh_ok;
  __shared__ double sh_dist;
  __shared__ int sort_z[LK_PEND];
  __shared__ double sort_d[LK_PEND];
  const int tid = threadIdx.x;
  const int hn = n - 1;
  int* nb = g_nb;                        // neighbour candidates live in global memory (every workgroup writes them)

  Heap<IT> heap;
  if (LDS_HEAP) {
    heap.v = reinterpret_cast<double*>(lds_raw);
    heap.kbi = reinterpret_cast<IT*>(heap.v + hn);
    heap.ibk = heap.kbi + hn;
  } else {
    heap.v = g_hv;
    heap.kbi = reinterpret_cast<IT*>(g_kbi);
    heap.ibk = reinterpret_cast<IT*>(g_ibk);
  }
  heap.size = hn;
  const int cand_words = (n + 31) / 32;

  // ---- initialisation, split over the workgroups: sizes, ids, candidate bitmap, nearest neighbours
  for (int i = wg * LK_T + tid; i < n; i += G * LK_T) {
    lk_st(size + i, 1);
    lk_st(cluster_id + i, i);
  }
  for (int i = wg * LK_T + tid; i < cand_words; i += G * LK_T) lk_st(cand + i, 0u);
  {
    const int lane = tid & 63, w = tid >> 6;
    for (int x = wg * LK_W + w; x < n - 1; x += G * LK_W) {
      MinPair best{__builtin_inf(), -1};
      const long base = (long)n * x - ((long)x * (x + 1) / 2) - x - 1;
      for (int i = x + 1 + lane; i < n; i += 64) {
        const double d = D[base + i];          // (written by k_pdist_f64, a previous kernel: plain load)
        if (d < best.d) {
          best.d = d;
          best.i = i;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        MinPair other;
        other.d = __shfl_xor(best.d, o, 64);
        other.i = __shfl_xor(best.i, o, 64);
        best = min_pair(best, other);
      }
      if (lane == 0) {
        lk_st(nb + x, best.i);
        lk_st(mind + x, best.i < 0 ? __builtin_inf() : best.d);
      }
    }
  }
  lk_grid_barrier(sh, G);
  if (wg == 0) {
    for (int x = tid; x < hn; x += LK_T) {
      heap.v[x] = lk_ld<true>(mind + x);
      heap.kbi[x] = (IT)x;
      heap.ibk[x] = (IT)x;
    }
    __syncthreads();
    if (tid == 0) heap.build();
    __syncthreads();
  }

  long long st_retry = 0, st_cand = 0, st_ovf = 0, st_c0 = 0, st_c1 = 0, st_c2 = 0, st_c3 = 0;
  for (int k = 0; k < n - 1; ++k) {
    long long tc = __builtin_readcyclecounter();
    if (wg == 0) {
      // ---- find the two closest clusters (lower-bound repairs) and record the merge
      int x = 0, y = 0;
      double dist = 0.0;
      for (int it = 0; it < n - k; ++it) {
        if (tid == 0) {
          const int hx = heap.kbi[0];
          const double hd = heap.v[0];
          const int hy = lk_ld<true>(nb + hx);
          sh_x = hx;
          sh_y = hy;
          sh_dist = hd;
          sh_ok = (hy >= 0 && hd == lk_ld<true>(D + cidx(n, hx, hy))) ? 1 : 0;
        }
        __syncthreads();
        x = sh_x;
        y = sh_y;
        dist = sh_dist;
        const int ok = sh_ok;
        if (ok) break;
        const MinPair p = block_find_min<true>(D, size, n, x, red);
        y = p.i;
        dist = p.d;
        if (tid == 0) {
          lk_st(nb + x, y);
          heap.change_value(x, dist);
          lk_st(mind + x, dist);
          ++st_retry;
        }
        __syncthreads();
      }
      if (tid == 0) {
        heap.remove_min();
        int id_x = lk_ld<true>(cluster_id + x), id_y = lk_ld<true>(cluster_id + y);
        const int nx = lk_ld<true>(size + x), ny = lk_ld<true>(size + y);
        if (id_x > id_y) {
          const int t = id_x;
          id_x = id_y;
          id_y = t;
        }
        Z[4 * (long)k + 0] = (double)id_x;
        Z[4 * (long)k + 1] = (double)id_y;
        Z[4 * (long)k + 2] = dist;
        Z[4 * (long)k + 3] = (double)(nx + ny);
        lk_st(size + x, 0);
        lk_st(size + y, nx + ny);
        lk_st(cluster_id + y, n + k);
        lk_st(&sh->x, x);
        lk_st(&sh->y, y);
        lk_st(&sh->nx, nx);
        lk_st(&sh->ny, ny);
        lk_st(&sh->dist, dist);
        lk_st(&sh->npend, 0);
      }
      const long long t2 = __builtin_readcyclecounter();
      st_c0 += t2 - tc;
      tc = t2;
    }
    lk_grid_barrier(sh, G);
    {
      const long long t2 = __builtin_readcyclecounter();
      st_c1 += t2 - tc;
      tc = t2;
    }
    const int x = lk_ld<true>(&sh->x), y = lk_ld<true>(&sh->y), nx = lk_ld<true>(&sh->nx),
              ny = lk_ld<true>(&sh->ny);
    const double dist = lk_ld<true>(&sh->dist);
    // ---- this workgroup's slices of the pass over all clusters z
    MinPair best{__builtin_inf(), -1};
    // (contiguous slices of ~n / G clusters: at n = 7 k and 8 workgroups every thread owns ONE cluster -- the pass
    //  is bound by the f64 division / square root throughput of a CU, so it has to be spread evenly)
    const int slice = (((n + G - 1) / G) + 63) & ~63;
    const int z_end = min(n, (wg + 1) * slice);
    for (int z0 = wg * slice + tid; z0 < z_end; z0 += 4 * LK_T) {
      bool act[4];
      long izy[4];
      double d_xi[4], d_yi[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int z = z0 + u * LK_T;
        act[u] = z < z_end && z != y && lk_ld<true>(size + z) != 0;
        izy[u] = act[u] ? cidx(n, z, y) : 0;
        d_xi[u] = act[u] ? lk_ld<true>(D + cidx(n, z, x)) : 0.0;
        d_yi[u] = act[u] ? lk_ld<true>(D + izy[u]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!act[u]) continue;
        const int z = z0 + u * LK_T;
        const double nd = sqrt(
            (((nx * d_xi[u] * d_xi[u]) + (ny * d_yi[u] * d_yi[u])) - ((nx * ny) * dist * dist) / (nx + ny)) /
            (nx + ny));
        lk_st(D + izy[u], nd);
        if (z < y) {
          if (z < x && lk_ld<true>(nb + z) == x) lk_st(nb + z, y);
          if (nd < lk_ld<true>(mind + z)) {
            lk_st(nb + z, y);
            atomicOr(&cand[z >> 5], 1u << (z & 31));
            const int slot = atomicAdd(&sh->npend, 1);
            if (slot < LK_PEND) {
              lk_st(&sh->pend_z[slot], z);
              lk_st(&sh->pend_d[slot], nd);
            }
          }
        } else if (nd < best.d) {
          best.d = nd;
          best.i = z;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      MinPair other;
      other.d = __shfl_xor(best.d, o, 64);
      other.i = __shfl_xor(best.i, o, 64);
      best = min_pair(best, other);
    }
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
      MinPair r = red[0];
#pragma unroll
      for (int q = 1; q < LK_W; ++q) r = min_pair(r, red[q]);
      lk_st(&sh->red[wg].d, r.d);
      lk_st(&sh->red[wg].i, r.i);
    }
    {
      const long long t2 = __builtin_readcyclecounter();
      st_c2 += t2 - tc;
      tc = t2;
    }
    lk_grid_barrier(sh, G);
    if (wg == 0) {
      // ---- replay the heap updates in SciPy's order: ascending z < y, then row y
      const int np = lk_ld<true>(&sh->npend);
      if (np <= LK_PEND && tid < np) {
        const int z = lk_ld<true>(&sh->pend_z[tid]);
        int rank = 0;
        for (int q = 0; q < np; ++q) rank += lk_ld<true>(&sh->pend_z[q]) < z ? 1 : 0;
        sort_z[rank] = z;
        sort_d[rank] = lk_ld<true>(&sh->pend_d[tid]);
        lk_st(cand + (z >> 5), 0u);
      }
      __syncthreads();
      if (tid == 0) {
        st_cand += np;
        if (np <= LK_PEND) {
          for (int q = 0; q < np; ++q) {
            heap.change_value(sort_z[q], sort_d[q]);
            lk_st(mind + sort_z[q], sort_d[q]);
          }
        } else {
          ++st_ovf;
          const int words = (y + 31) / 32;
          for (int wi = 0; wi < words; ++wi) {
            unsigned int m = lk_ld<true>(cand + wi);
            if (!m) continue;
            lk_st(cand + wi, 0u);
            while (m) {
              const int bit = __builtin_ctz(m);
              m &= m - 1;
              const int z = wi * 32 + bit;
              const double dz = lk_ld<true>(D + cidx(n, z, y));
              heap.change_value(z, dz);
              lk_st(mind + z, dz);
            }
          }
        }
        if (y < n - 1) {
          MinPair r{lk_ld<true>(&sh->red[0].d), lk_ld<true>(&sh->red[0].i)};
          for (int q = 1; q < G; ++q)
            r = min_pair(r, MinPair{lk_ld<true>(&sh->red[q].d), lk_ld<true>(&sh->red[q].i)});
          if (r.i != -1) {
            lk_st(nb + y, r.i);
            heap.change_value(y, r.d);
            lk_st(mind + y, r.d);
          }
        }
      }
      __syncthreads();
    }
    st_c3 += __builtin_readcyclecounter() - tc;
  }
  if (wg == 0 && tid == 0 && stats != nullptr) {
    stats[0] = st_retry;
    stats[1] = st_cand;
    stats[2] = st_ovf;
    stats[3] = st_c0;
    stats[4] = st_c1;
    stats[5] = st_c2;
    stats[6] = st_c3;
    stats[7] = n;
  }
}

constexpr size_t LK_LDS_MAX = 160 * 1024 - 7680;  // dynamic LDS budget (static part: ~6.5 KB)

inline size_t lk_align(size_t v) { return (v + 255) & ~(size_t)255; }

// linkage_fast.hip: the heap-free merge that runs first; the kernels of this file are its gated fallback
bool lf_wanted(int n);
size_t lf_workspace_bytes(int n);
int lf_launch(const double* cond, int n, double* Z, void* workspace, long long* stats, int** gate_out,
              hipStream_t st);

// size, cluster_id, neighbour, kbi, ibk (int) + heap values, min_dist mirror (double) + candidate bitmap + the
// multi-workgroup mailbox
inline size_t lk_core_bytes(int n) {
  const size_t ni = lk_align(sizeof(int) * (size_t)n), nd = lk_align(sizeof(double) * (size_t)n);
  return 5 * ni + 2 * nd + lk_align(4 * (size_t)((n + 31) / 32) + 16) + lk_align(sizeof(LkShared));
}

}  // namespace pa

extern "C" {

// layout: [heap kernel state][fast path: square matrix + row state][16 int64 counters: 8 heap kernel, 8 fast path]
size_t pa_linkage_workspace_bytes(int n) {
  if (n < 2) return 0;
  return pa::lk_core_bytes(n) + pa::lf_workspace_bytes(n) + 128;
}

// number of workgroups of the heap kernel.  ONE unless PA_LINKAGE_WGS asks for more: the multi-workgroup form
// synchronises with a hand-rolled spin barrier between workgroups that are launched non-cooperatively and each pin a
// whole CU's LDS; it is only safe when the caller owns the GPU (nothing else resident on XCD 0), which a library
// cannot know.  Since round 4 the heap kernels are the FALLBACK (exact ties) behind linkage_fast.hip, so the
// default never needs it; the opt-in stays for experiments (tools/time_linkage.py).
static int lk_num_workgroups(int n, int alone) {
  (void)n;
  (void)alone;
  const char* e = getenv("PA_LINKAGE_WGS");
  if (e != nullptr && atoi(e) >= 1) return atoi(e) > 32 ? 32 : atoi(e);
  return 1;
}

// D: condensed distance matrix (n*(n-1)/2 doubles), OVERWRITTEN.  Z: (n-1, 4) doubles, SciPy layout.
int pa_linkage_centroid_f64_ex(double* D, int n, double* Z, void* workspace, size_t workspace_bytes, int alone,
                               void* stream);

int pa_linkage_centroid_f64(double* D, int n, double* Z, void* workspace, size_t workspace_bytes,
                            void* stream) {
  return pa_linkage_centroid_f64_ex(D, n, Z, workspace, workspace_bytes, 0, stream);
}

int pa_linkage_centroid_f64_ex(double* D, int n, double* Z, void* workspace, size_t workspace_bytes, int alone,
                               void* stream) {
  if (n < 2) return 0;
  PA_REQUIRE(workspace_bytes >= pa_linkage_workspace_bytes(n), "pa_linkage_centroid_f64: workspace too small");
  const size_t ni = pa::lk_align(sizeof(int) * (size_t)n), nd = pa::lk_align(sizeof(double) * (size_t)n);
  const size_t cand_bytes = 4 * (size_t)((n + 31) / 32) + 16;
  unsigned char* w = (unsigned char*)workspace;
  int* size = (int*)w;
  int* cid = (int*)(w + ni);
  int* nb = (int*)(w + 2 * ni);
  int* kbi = (int*)(w + 3 * ni);
  int* ibk = (int*)(w + 4 * ni);
  double* hv = (double*)(w + 5 * ni);
  double* mind = (double*)(w + 5 * ni + nd);
  unsigned int* cand = (unsigned int*)(w + 5 * ni + 2 * nd);
  pa::LkShared* shared = (pa::LkShared*)(w + 5 * ni + 2 * nd + pa::lk_align(cand_bytes));
  long long* stats = (long long*)(w + pa_linkage_workspace_bytes(n) - 128);
  hipStream_t st = (hipStream_t)stream;
  // the merge loop is O(N^2) memory traffic in total; algorithmic bytes ~ 3 rows of 8*N per merge
  pa::ProfScope prof("k_linkage_centroid", stream, 9.0 * n * (double)n, 24.0 * n * (double)n);
  if (hipMemsetAsync(stats, 0, 128, st) != hipSuccess) return 1;
  // ---- the heap-free merge first (linkage_fast.hip); its status word gates the exact heap replay below
  int* gate = nullptr;
  if (pa::lf_wanted(n)) {
    if (pa::lf_launch(D, n, Z, w + pa::lk_core_bytes(n), stats + 8, &gate, st) != 0) return 1;
    PA_CHECK_LAUNCH("pa_linkage_centroid_f64 (fast path)");
  }
  const int G = lk_num_workgroups(n, alone);
  if (G > 1) {
    if (hipMemsetAsync(shared, 0, sizeof(pa::LkShared), st) != hipSuccess) return 1;
    const size_t lds_heap = ((size_t)(n - 1) * 12 + 15) & ~(size_t)15;
    if (n <= 65535 && lds_heap <= pa::LK_LDS_MAX) {
      (void)hipFuncSetAttribute((const void*)pa::k_linkage_centroid_mw<unsigned short, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pa::LK_LDS_MAX);
      hipLaunchKernelGGL((pa::k_linkage_centroid_mw<unsigned short, true>), dim3(8 * G), dim3(pa::LK_T), lds_heap,
                         st, D, n, Z, size, cid, hv, kbi, ibk, nb, mind, cand, shared, G, stats, gate);
    } else {
      hipLaunchKernelGGL((pa::k_linkage_centroid_mw<int, false>), dim3(8 * G), dim3(pa::LK_T), 0, st, D, n, Z,
                         size, cid, hv, kbi, ibk, nb, mind, cand, shared, G, stats, gate);
    }
    PA_CHECK_LAUNCH("pa_linkage_centroid_f64");
    return 0;
  }
  const size_t lds16 = (((size_t)(n - 1) * 14 + 15) & ~(size_t)15) + cand_bytes;
  if (n <= 65535 && lds16 <= pa::LK_LDS_MAX) {
    // (set on every call: the attribute belongs to the current device, not to the process)
    (void)hipFuncSetAttribute((const void*)pa::k_linkage_centroid<unsigned short, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)pa::LK_LDS_MAX);
    hipLaunchKernelGGL((pa::k_linkage_centroid<unsigned short, true>), dim3(1), dim3(pa::LK_T), lds16, st,
                       D, n, Z, size, cid, hv, kbi, ibk, nb, stats, gate);
  } else {
    hipLaunchKernelGGL((pa::k_linkage_centroid<int, false>), dim3(1), dim3(pa::LK_T), cand_bytes, st, D, n,
                       Z, size, cid, hv, kbi, ibk, nb, stats, gate);
  }
  PA_CHECK_LAUNCH("pa_linkage_centroid_f64");
  return 0;
}

}  // extern "C"
