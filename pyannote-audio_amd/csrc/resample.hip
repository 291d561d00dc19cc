// Polyphase windowed-sinc resampler of the audio front door on gfx950: replaces
// torchaudio.functional.resample as called by Audio.downmix_and_resample (core/io.py:258-262).
//   out[q * P + p] = sum_k taps[p][k] * x[q * L - width + k],  0 <= k < K = 2 * width + L
// with L = orig / gcd input samples per block, P = new / gcd phases; samples outside [0, n) are zero
// (torchaudio pads `width` zeros in front and `width + L` behind).  Streaming, HBM-bound: 4 B in per
// input sample, 4 B out per output sample; every input sample is re-read K / L times from L1/L2.
#include "common.h"

namespace pa {

__global__ __launch_bounds__(256) void k_resample_poly(const float* __restrict__ x, long n,
                                                      const float* __restrict__ taps, int L, int P, int K,
                                                      int width, int taps_in_lds, float* __restrict__ out,
                                                      long n_out) {
  extern __shared__ float sm_taps[];  // P * K floats when they fit (the host decides)
  const float* tp = taps;
  if (taps_in_lds) {
    for (int i = threadIdx.x; i < P * K; i += blockDim.x) sm_taps[i] = taps[i];
    __syncthreads();
    tp = sm_taps;
  }
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  const long q = o / P;
  const int p = (int)(o % P);
  const long base = q * L - width;
  const float* t = tp + (long)p * K;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const long i = base + k;
    const float v = (i >= 0 && i < n) ? x[i] : 0.f;
    acc = fmaf(t[k], v, acc);
  }
  out[o] = acc;
}

}  // namespace pa

extern "C" {

int pa_resample_poly(const float* x, long n, const float* taps, int L, int P, int K, int width, float* out,
                     long n_out, void* stream) {
  if (n_out <= 0) return 0;
  PA_REQUIRE(L > 0 && P > 0 && K == 2 * width + L, "pa_resample_poly: inconsistent filter bank shape");
  pa::ProfScope prof("k_resample_poly", stream, 2.0 * K * (double)n_out, 4.0 * ((double)n + (double)n_out));
  const size_t lds = (size_t)P * K * sizeof(float) <= 48 * 1024 ? (size_t)P * K * sizeof(float) : 0;
  // (the tap bank of 44.1 kHz -> 16 kHz is 160 x 475 floats = 304 KB: read through L1/L2 instead)
  hipLaunchKernelGGL(pa::k_resample_poly, dim3(pa::cdiv(n_out, 256)), dim3(256), lds, (hipStream_t)stream, x, n,
                     taps, L, P, K, width, lds ? 1 : 0, out, n_out);
  PA_CHECK_LAUNCH("pa_resample_poly");
  return 0;
}

}  // extern "C"
