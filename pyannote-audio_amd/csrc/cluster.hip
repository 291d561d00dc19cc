// float64 distance kernels of the clustering stage on gfx950
// (reference: pipelines/clustering.py:374-382 -> scipy.cluster.hierarchy.linkage(X, "centroid",
//  "euclidean") whose first step is pdist; :190-200 -> scipy.spatial.distance.cdist(., ., "cosine")).
//
// Bit-exactness contract: SciPy accumulates sum_k (u_k - v_k)^2 (resp. sum_k u_k v_k) sequentially in
// k, in double, with separately rounded multiply and add (its x86-64 baseline build has no FMA).  The
// kernels below keep one accumulator per output pair, walk k in ascending order and use
// plain operators under `#pragma clang fp contract(off)` (HIP's __dmul_rn/__dadd_rn are ordinary
// operators defined in a header, which the default -ffp-contract=fast still fuses).  HBM-bound on the condensed
// output for large N (8 B per pair); the arithmetic (3 flop per pair and k) is far from the fp64 peak.
#include "common.h"

// hipcc's default -ffp-contract=fast fuses a*b+c to FMA and IGNORES `#pragma clang fp contract`;
// __dmul_rn/__dadd_rn are plain operators in HIP headers and fuse as well.  This translation unit is
// therefore compiled with -ffp-contract=off (the marker below is read by _build.py); verified in the
// ISA: the k-loops hold v_mul_f64 + v_add_f64 only (v_fma_f64 remains inside sqrt/div expansions).
// hipcc-flags: -ffp-contract=off

namespace pa {

constexpr int PD_T = 64;   // pairs tile edge
constexpr int PD_K = 32;   // k chunk
constexpr int PD_LD = PD_K + 1;

// grid = (ceil(N/64), ceil(N/64)) with blockIdx.x >= blockIdx.y skipped lower triangle; block = 256
__global__ __launch_bounds__(256) void k_pdist_f64(const double* __restrict__ X, int N, int D,
                                                    double* __restrict__ out) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;
  __shared__ double As[PD_T * PD_LD];
  __shared__ double Bs[PD_T * PD_LD];
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;  // 16 x 16 threads, 4 x 4 pairs each (rows ti+16a, cols tj+16b)
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int k0 = 0; k0 < D; k0 += PD_K) {
    __syncthreads();
    for (int i = tid; i < PD_T * PD_K; i += 256) {
      const int r = i / PD_K, c = i % PD_K;
      const int gi = bi * PD_T + r, gj = bj * PD_T + r, k = k0 + c;
      As[r * PD_LD + c] = (gi < N && k < D) ? X[(long)gi * D + k] : 0.0;
      Bs[r * PD_LD + c] = (gj < N && k < D) ? X[(long)gj * D + k] : 0.0;
    }
    __syncthreads();
    const int kmax = min(PD_K, D - k0);
    for (int k = 0; k < kmax; ++k) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = As[(ti + 16 * a) * PD_LD + k];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = Bs[(tj + 16 * b) * PD_LD + k];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double d = av[a] - bv[b];
          acc[a][b] = acc[a][b] + d * d;  // separately rounded (contract(off))
        }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const long i = bi * PD_T + ti + 16 * a, j = bj * PD_T + tj + 16 * b;
      if (i < j && j < N)
        out[(long)N * i - i * (i + 1) / 2 + (j - i - 1)] = __dsqrt_rn(acc[a][b]);
    }
}

// SciPy's cdist "cosine" kernels (distance_impl.h: dot_product / _row_norms) are compiled 2-way
// vectorised in the x86-64 wheel (SSE2, two doubles per register): even and odd k accumulate
// separately, the two lanes are added, then an odd tail element is added last.  Pinned against
// scipy 1.15.3 for even and odd D (tools/diag_pdist.py, tests/test_pipeline_gpu.py).
__device__ __forceinline__ double dot2way(const double* __restrict__ u, const double* __restrict__ v,
                                          int D) {
  double s0 = 0.0, s1 = 0.0;
  const int m = D & ~1;
  for (int k = 0; k < m; k += 2) {
    s0 = s0 + u[k] * v[k];
    s1 = s1 + u[k + 1] * v[k + 1];
  }
  double t = s0 + s1;
  if (D & 1) t = t + u[D - 1] * v[D - 1];
  return t;
}

__global__ void k_row_norms_f64(const double* __restrict__ X, int N, int D, double* __restrict__ nrm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  nrm[i] = __dsqrt_rn(dot2way(X + (long)i * D, X + (long)i * D, D));
}

// out[i][j] = 1 - clip(<a_i, b_j> / (|a_i| |b_j|));  grid = NA, block = 128 (threads stride over j)
__global__ __launch_bounds__(128) void k_cdist_cosine_f64(const double* __restrict__ A,
                                                           const double* __restrict__ B, int NB, int D,
                                                           const double* __restrict__ nA,
                                                           const double* __restrict__ nB,
                                                           double* __restrict__ out) {
  extern __shared__ double us[];
  const int i = blockIdx.x;
  for (int k = threadIdx.x; k < D; k += 128) us[k] = A[(long)i * D + k];
  __syncthreads();
  const double na = nA[i];
  for (int j = threadIdx.x; j < NB; j += 128) {
    const double s = dot2way(us, B + (long)j * D, D);
    double c = s / (na * nB[j]);
    if (fabs(c) > 1.0) c = copysign(1.0, c);
    out[(long)i * NB + j] = 1.0 - c;
  }
}

// Cluster centroids (reference: pipelines/clustering.py:182-187 and :462-472,
//   centroids = np.vstack([np.mean(train_embeddings[train_clusters == k], axis=0) for k in range(K)])).
// numpy reduces a C-contiguous (n, D) float32 block over axis 0 row by row -- out[d] += row[d], top to bottom, in
// float32 -- and np.mean divides by n in float32.  Here cluster k owns rows[offsets[k] .. offsets[k+1]) (row numbers
// into X, in their original order: the host groups them with ONE stable sort) and a thread owns one column d: the
// same additions in the same order; the loads of eight rows are in flight together, the adds stay sequential.
// An empty cluster yields NaN (np.mean of an empty selection).  grid = (K, ceil(D / 256)), block = 256.
__global__ __launch_bounds__(256) void k_centroid_means(const float* __restrict__ X, int D,
                                                         const int* __restrict__ rows,
                                                         const int* __restrict__ offsets,
                                                         float* __restrict__ out) {
  const int k = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
  if (d >= D) return;
  const int a = offsets[k], b = offsets[k + 1];
  if (a >= b) {
    out[(long)k * D + d] = __builtin_nanf("");
    return;
  }
  float acc = X[(long)rows[a] * D + d];   // (np.add.reduce starts from the first row, not from 0)
  int j = a + 1;
  for (; j + 8 <= b; j += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = X[(long)rows[j + u] * D + d];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = acc + v[u];
  }
  for (; j < b; ++j) acc = acc + X[(long)rows[j] * D + d];
  out[(long)k * D + d] = acc / (float)(b - a);
}

}  // namespace pa

extern "C" {

// means of the rows of X (float32, row-major, D columns) selected by rows[offsets[k] .. offsets[k+1]) for every
// cluster k < K, in that order; bit-identical to np.mean(X[rows of k], axis=0).  out: (K, D) float32.
int pa_centroid_means(const float* X, int D, const int* rows, const int* offsets, int K, float* out, void* stream) {
  if (K <= 0 || D <= 0) return 0;
  pa::ProfScope prof("k_centroid_means", stream, 0.0, 0.0);
  hipLaunchKernelGGL(pa::k_centroid_means, dim3(K, pa::cdiv(D, 256)), dim3(256), 0, (hipStream_t)stream, X, D, rows,
                     offsets, out);
  PA_CHECK_LAUNCH("pa_centroid_means");
  return 0;
}

// scipy.spatial.distance.pdist(X, "euclidean"): condensed upper triangle, row-major pair order
int pa_pdist_f64(const double* X, int N, int D, double* out, void* stream) {
  if (N < 2) return 0;
  const int nt = pa::cdiv(N, pa::PD_T);
  pa::ProfScope prof("k_pdist_f64", stream, 3.0 * D * ((double)N * (N - 1) / 2), 8.0 * ((double)N * D + (double)N * (N - 1) / 2));
  hipLaunchKernelGGL(pa::k_pdist_f64, dim3(nt, nt), dim3(256), 0, (hipStream_t)stream, X, N, D, out);
  PA_CHECK_LAUNCH("pa_pdist_f64");
  return 0;
}

// scipy.spatial.distance.cdist(A, B, "cosine"); `norms` is scratch for NA + NB doubles
int pa_cdist_cosine_f64(const double* A, int NA, const double* B, int NB, int D, double* out,
                        double* norms, void* stream) {
  if (NA <= 0 || NB <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  pa::ProfScope prof("k_cdist_cosine_f64", stream, 2.0 * D * (double)NA * NB, 8.0 * ((double)NA * D + (double)NB * D + (double)NA * NB));
  hipLaunchKernelGGL(pa::k_row_norms_f64, dim3(pa::cdiv(NA, 128)), dim3(128), 0, st, A, NA, D, norms);
  hipLaunchKernelGGL(pa::k_row_norms_f64, dim3(pa::cdiv(NB, 128)), dim3(128), 0, st, B, NB, D,
                     norms + NA);
  hipLaunchKernelGGL(pa::k_cdist_cosine_f64, dim3(NA), dim3(128), (size_t)D * sizeof(double), st, A, B,
                     NB, D, norms, norms + NA, out);
  PA_CHECK_LAUNCH("pa_cdist_cosine_f64");
  return 0;
}

}  // extern "C"
