// VBx clustering (variational Bayes HMM-free x-vector clustering) and the PLDA projection on gfx950, fp64.
// Replaces the numpy loops of utils/vbx.py:27-140 (VBx), core/plda.py:47-60 + utils/vbx.py:205-217 (the
// x-vector -> PLDA transform), called from pipelines/clustering.py:606-617.
//
// Sizes: N = training embeddings of a file (<= ~10^4; 6 x 10^4 for a joint clustering), D = 128 PLDA
// dimensions, S = clusters of the AHC initialisation (tens).  One VB iteration is O(N S D) fp64 FMAs --
// far below any roofline: the kernels are written for a short dependent chain (3 launches per
// iteration, one 8-byte read-back for the convergence test) and for DETERMINISTIC reductions (fixed
// trees, no atomics), so that a run is reproducible bit for bit.  Data layout: gamma [N][S], rho [N][D],
// speaker models alpha / invL [S][D]; all double.
// hipcc-flags: -ffp-contract=off
#include "common.h"

namespace pa {

constexpr int VB_T = 256;

__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < VB_T / 64; ++i) s += red[i];
  return s;
}
__device__ __forceinline__ double block_max_d(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int i = 1; i < VB_T / 64; ++i) s = fmax(s, red[i]);
  return s;
}

// x (N, DIN) fp32 -> fea (N, DOUT) fp64:
//   y = sqrt(DMID) * l2( lda^T ( sqrt(DIN) * l2(x - mean1) ) - mean2 );  fea = (y - mu) tr^T [:, :DOUT]
// one workgroup per embedding; lda [DIN][DMID], trT [DMID][DOUT] (column k of trT = row k of tr).
__global__ __launch_bounds__(VB_T) void k_plda_transform(const float* __restrict__ X, int DIN, int DMID,
                                                        int DOUT, const double* __restrict__ mean1,
                                                        const double* __restrict__ lda,
                                                        const double* __restrict__ mean2,
                                                        const double* __restrict__ mu,
                                                        const double* __restrict__ trT,
                                                        double* __restrict__ fea) {
  extern __shared__ double sm[];  // DIN + DMID doubles
  __shared__ double red[VB_T / 64];
  double* xs = sm;
  double* ys = sm + DIN;
  const long n = blockIdx.x;
  double ss = 0.0;
  for (int i = threadIdx.x; i < DIN; i += VB_T) {
    const double v = (double)X[n * DIN + i] - mean1[i];
    xs[i] = v;
    ss += v * v;
  }
  const double sc1 = sqrt((double)DIN) / sqrt(block_sum_d(ss, red));
  double ss2 = 0.0;
  for (int j = threadIdx.x; j < DMID; j += VB_T) {
    double a = 0.0;
    for (int i = 0; i < DIN; ++i) a += lda[(long)i * DMID + j] * (xs[i] * sc1);
    a -= mean2[j];
    ys[j] = a;
    ss2 += a * a;
  }
  const double sc2 = sqrt((double)DMID) / sqrt(block_sum_d(ss2, red));
  __syncthreads();
  for (int k = threadIdx.x; k < DOUT; k += VB_T) {
    double a = 0.0;
    for (int j = 0; j < DMID; ++j) a += (ys[j] * sc2 - mu[j]) * trT[(long)j * DOUT + k];
    fea[n * DOUT + k] = a;
  }
}

// rho = fea * sqrt(Phi) (18);  G[n] = -0.5 (sum_d fea^2 + D log 2 pi)  (constant term of (23))
__global__ __launch_bounds__(VB_T) void k_vbx_prepare(const double* __restrict__ fea, int N, int D,
                                                     const double* __restrict__ Phi,
                                                     double* __restrict__ rho, double* __restrict__ G) {
  __shared__ double red[VB_T / 64];
  const long n = blockIdx.x;
  double ss = 0.0;
  for (int d = threadIdx.x; d < D; d += VB_T) {
    const double v = fea[n * D + d];
    rho[n * D + d] = v * sqrt(Phi[d]);
    ss += v * v;
  }
  ss = block_sum_d(ss, red);
  if (threadIdx.x == 0) G[n] = -0.5 * (ss + D * 1.8378770664093453 /* log(2 pi) */);
}

// M step for speaker s = blockIdx.x: Nk = sum_n gamma[n,s]; invL (17), alpha (16);
// cterm[s] = 0.5 sum_d (invL + alpha^2) Phi;  eterm[s] = sum_d (log invL - invL - alpha^2 + 1)  (25)
__global__ __launch_bounds__(VB_T) void k_vbx_mstep(const double* __restrict__ gamma,
                                                   const double* __restrict__ rho, int N, int S, int D,
                                                   const double* __restrict__ Phi, double FaFb,
                                                   double* __restrict__ alpha, double* __restrict__ invL,
                                                   double* __restrict__ Nk, double* __restrict__ cterm,
                                                   double* __restrict__ eterm) {
  __shared__ double red[VB_T / 64];
  const int s = blockIdx.x;
  double part = 0.0;
  for (int n = threadIdx.x; n < N; n += VB_T) part += gamma[(long)n * S + s];
  const double nk = block_sum_d(part, red);
  // (gamma^T rho)[s, :]: the N rows are split over the VB_T / 128 thread groups, 4 independent partial
  // sums per thread; partials are combined in a fixed order (deterministic, not numpy's BLAS order)
  extern __shared__ double part_sm[];  // [VB_T / 128][D]
  constexpr int DP = 128, NP = VB_T / DP;
  const int p = threadIdx.x / DP, dl = threadIdx.x % DP;
  const int n0 = (int)((long)N * p / NP), n1 = (int)((long)N * (p + 1) / NP);
  for (int d = dl; d < D; d += DP) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int n = n0;
    for (; n + 3 < n1; n += 4) {
      a0 += gamma[(long)n * S + s] * rho[(long)n * D + d];
      a1 += gamma[(long)(n + 1) * S + s] * rho[(long)(n + 1) * D + d];
      a2 += gamma[(long)(n + 2) * S + s] * rho[(long)(n + 2) * D + d];
      a3 += gamma[(long)(n + 3) * S + s] * rho[(long)(n + 3) * D + d];
    }
    for (; n < n1; ++n) a0 += gamma[(long)n * S + s] * rho[(long)n * D + d];
    part_sm[p * D + d] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  double c = 0.0, e = 0.0;
  for (int d = threadIdx.x; d < D; d += VB_T) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NP; ++q) acc += part_sm[q * D + d];
    const double il = 1.0 / (1.0 + FaFb * nk * Phi[d]);
    const double al = FaFb * il * acc;
    invL[(long)s * D + d] = il;
    alpha[(long)s * D + d] = al;
    c += (il + al * al) * Phi[d];
    e += log(il) - il - al * al + 1.0;
  }
  c = block_sum_d(c, red);
  e = block_sum_d(e, red);
  if (threadIdx.x == 0) {
    Nk[s] = nk;
    cterm[s] = 0.5 * c;
    eterm[s] = e;
  }
}

// E step for frame n = blockIdx.x: log_p[s] = Fa (rho_n . alpha_s - cterm[s] + G[n]) (23);
// lpi[s] = log(pi[s] + 1e-8) with pi = Nk / sum Nk (uniform in the first iteration);
// log_p_x = logsumexp_s(log_p + lpi);  gamma[n,s] = exp(log_p + lpi - log_p_x)
__global__ __launch_bounds__(VB_T) void k_vbx_estep(const double* __restrict__ rho,
                                                   const double* __restrict__ alpha,
                                                   const double* __restrict__ cterm,
                                                   const double* __restrict__ G,
                                                   const double* __restrict__ Nk, int uniform_pi, int N,
                                                   int S, int D, double Fa, double* __restrict__ gamma,
                                                   double* __restrict__ logpx) {
  extern __shared__ double sm[];  // D (rho_n) + S (scores)
  __shared__ double red[VB_T / 64];
  double* r = sm;
  double* sc = sm + D;
  const long n = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += VB_T) r[d] = rho[n * D + d];
  double tot = 0.0;
  for (int s = threadIdx.x; s < S; s += VB_T) tot += Nk[s];
  tot = block_sum_d(tot, red);  // (also the barrier that publishes r[])
  double mx = -__builtin_inf();
  for (int s = threadIdx.x; s < S; s += VB_T) {
    double dot = 0.0;
    for (int d = 0; d < D; ++d) dot += r[d] * alpha[(long)s * D + d];
    const double pi = uniform_pi ? 1.0 / S : Nk[s] / tot;
    const double v = Fa * (dot - cterm[s] + G[n]) + log(pi + 1e-8);
    sc[s] = v;
    mx = fmax(mx, v);
  }
  mx = block_max_d(mx, red);
  double se = 0.0;
  for (int s = threadIdx.x; s < S; s += VB_T) se += exp(sc[s] - mx);
  se = block_sum_d(se, red);
  const double lse = mx + log(se);
  for (int s = threadIdx.x; s < S; s += VB_T) gamma[n * S + s] = exp(sc[s] - lse);
  if (threadIdx.x == 0) logpx[n] = lse;
}

// ELBO (25) = sum_n log_p_x[n] + 0.5 Fb sum_s eterm[s]: one workgroup, fixed summation tree
__global__ __launch_bounds__(VB_T) void k_vbx_elbo(const double* __restrict__ logpx, int N,
                                                  const double* __restrict__ eterm, int S, double Fb,
                                                  double* __restrict__ out) {
  __shared__ double red[VB_T / 64];
  double a = 0.0, b = 0.0;
  for (int n = threadIdx.x; n < N; n += VB_T) a += logpx[n];
  for (int s = threadIdx.x; s < S; s += VB_T) b += eterm[s];
  a = block_sum_d(a, red);
  b = block_sum_d(b, red);
  if (threadIdx.x == 0) out[0] = a + Fb * 0.5 * b;
}

}  // namespace pa

extern "C" {

int pa_plda_transform(const float* X, int n, int din, int dmid, int dout, const double* mean1,
                      const double* lda, const double* mean2, const double* mu, const double* trT,
                      double* fea, void* stream) {
  if (n <= 0) return 0;
  PA_REQUIRE(dout <= dmid && (size_t)(din + dmid) * 8 <= 64 * 1024, "pa_plda_transform: bad dimensions");
  pa::ProfScope prof("k_plda_transform", stream, 2.0 * n * ((double)din * dmid + (double)dmid * dout),
                     4.0 * n * din + 8.0 * n * dout);
  hipLaunchKernelGGL(pa::k_plda_transform, dim3(n), dim3(pa::VB_T), (size_t)(din + dmid) * 8,
                     (hipStream_t)stream, X, din, dmid, dout, mean1, lda, mean2, mu, trT, fea);
  PA_CHECK_LAUNCH("pa_plda_transform");
  return 0;
}

size_t pa_vbx_workspace_bytes(int n, int s, int d) {
  // rho (n d) + G (n) + logpx (n) + alpha, invL (s d each) + Nk, cterm, eterm (s each)
  return 8 * ((size_t)n * d + 2 * (size_t)n + 2 * (size_t)s * d + 3 * (size_t)s) + 64;
}

// One VB iteration (utils/vbx.py:106-133) on device buffers.  `gamma` (n, s) is read by the M step and
// overwritten by the E step; `elbo_out` receives the value of (25) for the convergence test on the host.
// `first` != 0: first iteration (rho / G are derived from `fea`, speaker priors are uniform).
int pa_vbx_iteration(const double* fea, const double* Phi, int n, int s, int d, double Fa, double Fb,
                     int first, double* gamma, double* elbo_out, void* workspace, size_t workspace_bytes,
                     void* stream) {
  if (n <= 0 || s <= 0) return 0;
  PA_REQUIRE(workspace_bytes >= pa_vbx_workspace_bytes(n, s, d), "pa_vbx_iteration: workspace too small");
  PA_REQUIRE((size_t)(d + s) * 8 <= 64 * 1024, "pa_vbx_iteration: too many clusters for one workgroup");
  double* w = (double*)workspace;
  double* rho = w;
  double* G = rho + (size_t)n * d;
  double* logpx = G + n;
  double* alpha = logpx + n;
  double* invL = alpha + (size_t)s * d;
  double* Nk = invL + (size_t)s * d;
  double* cterm = Nk + s;
  double* eterm = cterm + s;
  hipStream_t st = (hipStream_t)stream;
  pa::ProfScope prof("k_vbx_iteration", stream, 4.0 * n * (double)s * d, 8.0 * (2.0 * n * s + 2.0 * n * d));
  if (first) hipLaunchKernelGGL(pa::k_vbx_prepare, dim3(n), dim3(pa::VB_T), 0, st, fea, n, d, Phi, rho, G);
  hipLaunchKernelGGL(pa::k_vbx_mstep, dim3(s), dim3(pa::VB_T), (size_t)(pa::VB_T / 128) * d * 8, st, gamma, rho, n,
                     s, d, Phi, Fa / Fb, alpha,
                     invL, Nk, cterm, eterm);
  hipLaunchKernelGGL(pa::k_vbx_estep, dim3(n), dim3(pa::VB_T), (size_t)(d + s) * 8, st, rho, alpha, cterm, G, Nk,
                     first, n, s, d, Fa, gamma, logpx);
  hipLaunchKernelGGL(pa::k_vbx_elbo, dim3(1), dim3(pa::VB_T), 0, st, logpx, n, eterm, s, Fb, elbo_out);
  PA_CHECK_LAUNCH("pa_vbx_iteration");
  return 0;
}

}  // extern "C"
