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
#include "common.h"

namespace pa {

constexpr int LK_T = 1024;  // threads (16 waves)
constexpr int LK_W = LK_T / 64;

__device__ __forceinline__ long cidx(long n, long i, long j) {
  // scipy condensed_index(n, i, j)
  return i < j ? n * i - (i * (i + 1) / 2) + (j - i - 1) : n * j - (j * (j + 1) / 2) + (i - j - 1);
}

template <typename IT>
struct Heap {
  double* v;  // values by heap position
  IT* kbi;    // key_by_index
  IT* ibk;    // index_by_key
  int size;
  __device__ __forceinline__ void swap(int i, int j) {
    const double t = v[i];
    v[i] = v[j];
    v[j] = t;
    const IT ki = kbi[i], kj = kbi[j];
    kbi[i] = kj;
    kbi[j] = ki;
    ibk[ki] = (IT)j;
    ibk[kj] = (IT)i;
  }
  __device__ void sift_up(int index) {
    int parent = (index - 1) >> 1;
    while (index > 0 && v[parent] > v[index]) {
      swap(index, parent);
      index = parent;
      parent = (index - 1) >> 1;
    }
  }
  __device__ void sift_down(int index) {
    int child = 2 * index + 1;
    while (child < size) {
      if (child + 1 < size && v[child + 1] < v[child]) child += 1;
      if (v[index] > v[child]) {
        swap(index, child);
        index = child;
        child = 2 * index + 1;
      } else {
        break;
      }
    }
  }
  __device__ void change_value(int key, double value) {
    const int index = ibk[key];
    const double old = v[index];
    v[index] = value;
    if (value < old) sift_up(index);
    else sift_down(index);
  }
  __device__ void remove_min() {
    swap(0, size - 1);
    size -= 1;
    sift_down(0);
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

// find_min_dist(n, D, size, x): nearest active neighbour of x among indices > x.  All threads call;
// result valid in every thread.  `red` = LK_W MinPairs of LDS.
__device__ MinPair block_find_min(const double* __restrict__ D, const int* __restrict__ size, int n,
                                  int x, MinPair* red) {
  MinPair best{__builtin_inf(), -1};
  const long base = (long)n * x - ((long)x * (x + 1) / 2) - x - 1;  // cidx(n, x, i) = base + i
  for (int i = x + 1 + threadIdx.x; i < n; i += LK_T) {
    if (size[i] == 0) continue;
    const double d = D[base + i];
    if (d < best.d) {  // strict: the first (lowest i) minimum of this thread's ascending scan
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

template <typename IT, bool LDS_HEAP>
__global__ __launch_bounds__(LK_T) void k_linkage_centroid(double* __restrict__ D, int n,
                                                            double* __restrict__ Z,
                                                            int* __restrict__ size,
                                                            int* __restrict__ cluster_id,
                                                            int* __restrict__ neighbor,
                                                            double* __restrict__ min_dist,
                                                            double* __restrict__ g_hv,
                                                            int* __restrict__ g_kbi,
                                                            int* __restrict__ g_ibk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ MinPair red[LK_W];
  __shared__ int sh_x, sh_y, sh_ok, sh_nx, sh_ny;
  __shared__ double sh_dist;
  const int tid = threadIdx.x;
  const int hn = n - 1;  // heap capacity

  Heap<IT> heap;
  unsigned int* cand;  // bitmap of rows whose lower bound dropped in this merge
  if (LDS_HEAP) {
    heap.v = reinterpret_cast<double*>(lds_raw);
    heap.kbi = reinterpret_cast<IT*>(heap.v + hn);
    heap.ibk = heap.kbi + hn;
    cand = reinterpret_cast<unsigned int*>(lds_raw + (((size_t)hn * (8 + 2 * sizeof(IT)) + 15) & ~(size_t)15));
  } else {
    heap.v = g_hv;
    heap.kbi = reinterpret_cast<IT*>(g_kbi);
    heap.ibk = reinterpret_cast<IT*>(g_ibk);
    cand = reinterpret_cast<unsigned int*>(lds_raw);
  }
  heap.size = hn;
  const int cand_words = (n + 31) / 32;

  for (int i = tid; i < n; i += LK_T) {
    size[i] = 1;
    cluster_id[i] = i;
  }
  for (int i = tid; i < cand_words; i += LK_T) cand[i] = 0u;
  __syncthreads();
  // initial nearest-neighbour candidates (one wave per row)
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
        neighbor[x] = best.i;
        min_dist[x] = best.i < 0 ? __builtin_inf() : best.d;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < hn; i += LK_T) {
    heap.v[i] = min_dist[i];
    heap.kbi[i] = (IT)i;
    heap.ibk[i] = (IT)i;
  }
  __syncthreads();
  if (tid == 0)
    for (int i = hn / 2 - 1; i >= 0; --i) heap.sift_down(i);
  __syncthreads();

  for (int k = 0; k < n - 1; ++k) {
    // ---- find the two closest clusters: at most n - k lower-bound repairs
    int x = 0, y = 0;
    double dist = 0.0;
    for (int it = 0; it < n - k; ++it) {
      if (tid == 0) {
        const int hx = heap.kbi[0];
        const double hd = heap.v[0];
        const int hy = neighbor[hx];
        sh_x = hx;
        sh_y = hy;
        sh_dist = hd;
        sh_ok = (hd == D[cidx(n, hx, hy)]) ? 1 : 0;
      }
      __syncthreads();
      x = sh_x;
      y = sh_y;
      dist = sh_dist;
      const int ok = sh_ok;
      __syncthreads();
      if (ok) break;
      const MinPair p = block_find_min(D, size, n, x, red);
      y = p.i;
      dist = p.d;
      if (tid == 0) {
        neighbor[x] = y;
        min_dist[x] = dist;
        heap.change_value(x, dist);
      }
      __syncthreads();
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
    const int nx = sh_nx, ny = sh_ny;
    // ---- Lance-Williams (centroid) update of row/column y, all z in parallel
    for (int z = tid; z < n; z += LK_T) {
      const int nz = size[z];
      if (nz == 0 || z == y) continue;
      const long izy = cidx(n, z, y);
      const double d_xi = D[cidx(n, z, x)], d_yi = D[izy];
      D[izy] = sqrt((((nx * d_xi * d_xi) + (ny * d_yi * d_yi)) - ((nx * ny) * dist * dist) / (nx + ny)) /
                    (nx + ny));
    }
    __syncthreads();
    // ---- neighbour reassignment (z < x) and lower-bound refresh (z < y): per-z independent parts in
    // parallel, the heap updates afterwards by lane 0 in ascending z (SciPy's order)
    for (int z = tid; z < n - 1; z += LK_T) {
      if (size[z] == 0) continue;
      if (z < x && neighbor[z] == x) neighbor[z] = y;
      if (z < y) {
        const double d = D[cidx(n, z, y)];
        if (d < min_dist[z]) {
          neighbor[z] = y;
          min_dist[z] = d;
          atomicOr(&cand[z >> 5], 1u << (z & 31));
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      const int words = (y + 31) / 32;
      for (int wi = 0; wi < words; ++wi) {
        unsigned int m = cand[wi];
        if (!m) continue;
        cand[wi] = 0u;
        while (m) {
          const int bit = __builtin_ctz(m);
          m &= m - 1;
          const int z = wi * 32 + bit;
          heap.change_value(z, min_dist[z]);
        }
      }
    }
    __syncthreads();
    // ---- nearest neighbour of the merged cluster
    if (y < n - 1) {
      const MinPair p = block_find_min(D, size, n, y, red);
      if (tid == 0 && p.i != -1) {
        neighbor[y] = p.i;
        min_dist[y] = p.d;
        heap.change_value(y, p.d);
      }
    }
    __syncthreads();
  }
}

constexpr size_t LK_LDS_MAX = 160 * 1024 - 2048;  // dynamic LDS budget (static part is < 1 KB)

inline size_t lk_align(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace pa

extern "C" {

size_t pa_linkage_workspace_bytes(int n) {
  if (n < 2) return 0;
  const size_t ni = pa::lk_align(sizeof(int) * (size_t)n), nd = pa::lk_align(sizeof(double) * (size_t)n);
  return 5 * ni + 2 * nd;  // size, cluster_id, neighbor, kbi, ibk (int) + min_dist, heap values (double)
}

// D: condensed distance matrix (n*(n-1)/2 doubles), OVERWRITTEN.  Z: (n-1, 4) doubles, SciPy layout.
int pa_linkage_centroid_f64(double* D, int n, double* Z, void* workspace, size_t workspace_bytes,
                            void* stream) {
  if (n < 2) return 0;
  PA_REQUIRE(workspace_bytes >= pa_linkage_workspace_bytes(n), "pa_linkage_centroid_f64: workspace too small");
  const size_t ni = pa::lk_align(sizeof(int) * (size_t)n), nd = pa::lk_align(sizeof(double) * (size_t)n);
  unsigned char* w = (unsigned char*)workspace;
  int* size = (int*)w;
  int* cid = (int*)(w + ni);
  int* nb = (int*)(w + 2 * ni);
  int* kbi = (int*)(w + 3 * ni);
  int* ibk = (int*)(w + 4 * ni);
  double* md = (double*)(w + 5 * ni);
  double* hv = (double*)(w + 5 * ni + nd);
  hipStream_t st = (hipStream_t)stream;
  const size_t cand_bytes = 4 * (size_t)((n + 31) / 32) + 16;
  const size_t lds16 = (((size_t)(n - 1) * 12 + 15) & ~(size_t)15) + cand_bytes;
  // the merge loop is O(N^2) memory traffic in total; algorithmic bytes ~ 3 rows of 8*N per merge
  pa::ProfScope prof("k_linkage_centroid", stream, 9.0 * n * (double)n, 24.0 * n * (double)n);
  if (n <= 65535 && lds16 <= pa::LK_LDS_MAX) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)pa::k_linkage_centroid<unsigned short, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pa::LK_LDS_MAX);
      attr = true;
    }
    hipLaunchKernelGGL((pa::k_linkage_centroid<unsigned short, true>), dim3(1), dim3(pa::LK_T), lds16, st,
                       D, n, Z, size, cid, nb, md, hv, kbi, ibk);
  } else {
    hipLaunchKernelGGL((pa::k_linkage_centroid<int, false>), dim3(1), dim3(pa::LK_T), cand_bytes, st, D, n,
                       Z, size, cid, nb, md, hv, kbi, ibk);
  }
  PA_CHECK_LAUNCH("pa_linkage_centroid_f64");
  return 0;
}

}  // extern "C"
