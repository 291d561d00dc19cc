// Kaldi-compatible log-mel filterbank on gfx950
// (reference call site: models/embedding/wespeaker/__init__.py:113-139, which calls
//  torchaudio.compliance.kaldi.fbank with snip_edges, dither 0, remove_dc_offset, preemphasis 0.97,
//  hamming window, round_to_power_of_two (400 -> 512), power spectrum, 80 mel bins from 20 Hz, log).
//
//   k_fbank          one wave per frame: scale by 2^15, DC removal, pre-emphasis, Hamming, 512-point
//                    real FFT as a 256-point complex radix-4 Stockham FFT in LDS + real unpack, power,
//                    sparse mel projection, log(max(., eps)).
//   k_fbank_center   subtract the per-chunk mean over frames (wespeaker/__init__.py:138-139).
//   k_fbank_center_span   subtract the running mean of K frames (wespeaker/__init__.py:141-157).
#include "common.h"

namespace pa {

constexpr int FB_WIN = 400, FB_SHIFT = 160, FB_NFFT = 512, FB_HALF = 256, FB_NBIN = 257;
constexpr int FB_FPB = 4;  // frames (waves) per block

struct cplx {
  float x, y;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// tables (device): window[400]; tw256[256] = exp(-2 pi i m / 256); tw512[257] = exp(-2 pi i k / 512);
// mel_w[nmel][257] dense fp32 (torchaudio layout, right column zero), mel_lo/hi[nmel] non-zero range.
__global__ __launch_bounds__(256) void k_fbank(const float* __restrict__ wav, long wav_len,
                                               long chunk_stride, int T, float scale, float preemph,
                                               const float* __restrict__ window,
                                               const float2* __restrict__ tw256,
                                               const float2* __restrict__ tw512,
                                               const float* __restrict__ mel_w,
                                               const int* __restrict__ mel_lo,
                                               const int* __restrict__ mel_hi, int nmel, float eps,
                                               float* __restrict__ out) {
  __shared__ float2 bufA[FB_FPB][FB_HALF];
  __shared__ float2 bufB[FB_FPB][FB_HALF];
  __shared__ float2 s_tw256[FB_HALF];
  __shared__ float xs[FB_FPB][FB_NFFT + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y;
  const int t = blockIdx.x * FB_FPB + wv;
  for (int i = tid; i < FB_HALF; i += 256) s_tw256[i] = tw256[i];
  const bool active = t < T;
  float2 w512[5];     // twiddles of the real unpack: fetched now, used after the FFT
#pragma unroll
  for (int i = 0; i < 5; ++i) w512[i] = lane + 64 * i <= FB_HALF ? tw512[lane + 64 * i] : make_float2(0.f, 0.f);

  // ---- load + scale, DC removal
  float x[7];
  float s = 0.f;
  const long base = (long)b * chunk_stride + (long)t * FB_SHIFT;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int j = lane + 64 * i;
    float v = 0.f;
    if (active && j < FB_WIN) {
      const long g = base + j;
      v = (g < wav_len ? wav[g] : 0.f) * scale;
    }
    x[i] = v;
    s += v;
  }
  const float mean = wave_sum(s) / (float)FB_WIN;
  float* xw = xs[wv];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int j = lane + 64 * i;
    if (j < FB_WIN) xw[j + 1] = x[i] - mean;
  }
  if (lane == 0) xw[0] = x[0] - mean;  // replicate padding for the pre-emphasis of sample 0
  __syncthreads();
  // ---- pre-emphasis + window, zero pad to 512, pack as 256 complex
  float y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + 64 * i;
    float v = 0.f;
    if (j < FB_WIN) v = (xw[j + 1] - preemph * xw[j]) * window[j];
    y[i] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) xw[lane + 64 * i] = y[i];
  __syncthreads();
  float2* A = bufA[wv];
  float2* Bf = bufB[wv];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = lane + 64 * i;
    A[n] = make_float2(xw[2 * n], xw[2 * n + 1]);
  }
  __syncthreads();

  // ---- 256-point complex FFT: 4 radix-4 Stockham passes, one butterfly per lane per pass
  float2* src = A;
  float2* dst = Bf;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int Ns = 1 << (2 * pass);
    const int j = lane;
    const int k = j & (Ns - 1);
    const int tstep = 64 >> (2 * pass);  // 256 / (Ns * 4)
    cplx v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float2 d = src[j + 64 * r];
      const float2 w = s_tw256[(r * k * tstep) & 255];
      v[r] = cmul({d.x, d.y}, {w.x, w.y});
    }
    const cplx t0 = {v[0].x + v[2].x, v[0].y + v[2].y};
    const cplx t1 = {v[0].x - v[2].x, v[0].y - v[2].y};
    const cplx t2 = {v[1].x + v[3].x, v[1].y + v[3].y};
    const cplx d13 = {v[1].x - v[3].x, v[1].y - v[3].y};
    const cplx t3 = {d13.y, -d13.x};  // -i * (v1 - v3)
    const int j0 = ((j - k) << 2) + k;  // (j / Ns) * Ns * 4 + k
    dst[j0] = make_float2(t0.x + t2.x, t0.y + t2.y);
    dst[j0 + Ns] = make_float2(t1.x + t3.x, t1.y + t3.y);
    dst[j0 + 2 * Ns] = make_float2(t0.x - t2.x, t0.y - t2.y);
    dst[j0 + 3 * Ns] = make_float2(t1.x - t3.x, t1.y - t3.y);
    __syncthreads();
    float2* tmp = src;
    src = dst;
    dst = tmp;
  }
  // result Z[0..255] is in `src`
  // ---- real unpack -> power spectrum P[0..256] (into xw)
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int k = lane + 64 * i;
    if (k <= FB_HALF) {
      const float2 zk = src[k & 255];
      const float2 zc0 = src[(FB_HALF - k) & 255];
      const cplx zc = {zc0.x, -zc0.y};
      const cplx e = {0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y)};
      const cplx dd = {0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y)};
      const cplx o = {dd.y, -dd.x};  // -i * dd
      const float2 w = w512[i];
      const cplx wo = cmul(o, {w.x, w.y});
      const float re = e.x + wo.x, im = e.y + wo.y;
      xw[k] = re * re + im * im;
    }
  }
  __syncthreads();
  // ---- mel projection + log
  if (active) {
    for (int m = lane; m < nmel; m += 64) {
      const float* wrow = mel_w + (long)m * FB_NBIN;
      float acc = 0.f;
      // eight weights in flight per trip (a plain loop waits for every weight in turn: ~30 L2 round trips per
      // mel bin, which was most of this kernel's time); the fma chain keeps its order in k
      const int lo = mel_lo[m], hi = mel_hi[m];
      for (int k0 = lo; k0 <= hi; k0 += 8) {
        float wk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wk[u] = k0 + u <= hi ? wrow[k0 + u] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u <= hi) acc = fmaf(xw[k0 + u], wk[u], acc);
      }
      out[((long)b * T + t) * nmel + m] = logf(fmaxf(acc, eps));
    }
  }
}

// mean over frames per (chunk, mel) then subtract in place.  grid = B, block = 320 (80 mel x 4 parts)
__global__ __launch_bounds__(320) void k_fbank_center(float* __restrict__ fb, int T, int nmel) {
  __shared__ float part[4][80];
  const int b = blockIdx.x;
  const int m = threadIdx.x % 80, p = threadIdx.x / 80;
  float* x = fb + (long)b * T * nmel;
  float s = 0.f;
  if (m < nmel)
    for (int t = p; t < T; t += 4) s += x[(long)t * nmel + m];
  part[p][m] = s;
  __syncthreads();
  const float mean = (part[0][m] + part[1][m] + part[2][m] + part[3][m]) / (float)T;
  if (m < nmel)
    for (int t = p; t < T; t += 4) x[(long)t * nmel + m] -= mean;
}

// Running-mean centring (wespeaker/__init__.py:141-157, fbank_centering_span given): out = x - avg_pool1d(x, K, stride 1,
// padding K / 2, count_include_pad=False) along the frames.  One workgroup per (chunk, group of 16 mel bins, tile of
// FC_TT frames): the tile and its K / 2 frames of halo on either side go to LDS once ([frame][16 bins]: the 64 lanes of a
// wave read 4 consecutive frames x 16 bins = 64 consecutive words, conflict-free), then every thread adds up the K
// frames of its outputs in ASCENDING order in float32 and divides by the number of frames inside the chunk -- the
// reference's arithmetic, not a sliding sum.  Out of place: the neighbouring tiles read this tile's frames.
constexpr int FC_TT = 240, FC_MG = 16, FC_KMAX = 2049;   // (240 + 2048) x 64 B = 143 KB of LDS
__global__ __launch_bounds__(256) void k_fbank_center_span(const float* __restrict__ fb, int T, int nmel, int K,
                                                           float* __restrict__ out) {
  extern __shared__ float fc_tile[];   // [(FC_TT + K - 1)][FC_MG]
  const int b = blockIdx.z, m0 = blockIdx.y * FC_MG, t0 = blockIdx.x * FC_TT;
  const int half = K >> 1;
  const int m = threadIdx.x & (FC_MG - 1), tl = threadIdx.x >> 4;     // 16 bins x 16 frame lanes
  const float* x = fb + (long)b * T * nmel;
  const int rows = FC_TT + K - 1;
  for (int r = tl; r < rows; r += 16) {
    const int t = t0 - half + r;
    fc_tile[r * FC_MG + m] = (t >= 0 && t < T && m0 + m < nmel) ? x[(long)t * nmel + m0 + m] : 0.f;
  }
  __syncthreads();
  if (m0 + m >= nmel) return;
  for (int i = tl; i < FC_TT; i += 16) {
    const int t = t0 + i;
    if (t >= T) break;
    const int lo = t - half < 0 ? 0 : t - half, hi = t + half > T - 1 ? T - 1 : t + half;
    float s = 0.f;
    for (int u = lo; u <= hi; ++u) s += fc_tile[(u - t0 + half) * FC_MG + m];
    out[((long)b * T + t) * nmel + m0 + m] = fc_tile[(i + half) * FC_MG + m] - s / (float)(hi - lo + 1);
  }
}

}  // namespace pa

extern "C" {

// wespeaker/__init__.py:141-157 (compute_fbank with fbank_centering_span): see include/pyannote_amd.h
int pa_fbank_center_span(const float* fb, int B, int T, int nmel, int kernel, float* out, void* stream) {
  PA_REQUIRE(kernel >= 1 && (kernel & 1), "pa_fbank_center_span: the window must be an odd number of frames (got %d)",
             kernel);
  PA_REQUIRE(fb != out, "pa_fbank_center_span: works out of place");
  if (B <= 0 || T <= 0 || nmel <= 0) return 0;
  // a window that reaches past both ends of the chunk from every frame is clipped to [0, T) anyway: 2 T - 1 frames
  // give the same sums in the same order
  if (kernel > 2 * T - 1) kernel = 2 * T - 1;
  PA_REQUIRE(kernel <= pa::FC_KMAX,
             "pa_fbank_center_span: a running mean over %d frames of a %d-frame chunk needs more than the LDS holds "
             "(at most %d frames)", kernel, T, pa::FC_KMAX);
  const size_t lds = (size_t)(pa::FC_TT + kernel - 1) * pa::FC_MG * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)pa::k_fbank_center_span, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((pa::FC_TT + pa::FC_KMAX - 1) * pa::FC_MG * sizeof(float)));
    attr_set = true;
  }
  hipLaunchKernelGGL(pa::k_fbank_center_span, dim3(pa::cdiv(T, pa::FC_TT), pa::cdiv(nmel, pa::FC_MG), B), dim3(256), lds,
                     (hipStream_t)stream, fb, T, nmel, kernel, out);
  PA_CHECK_LAUNCH("pa_fbank_center_span");
  return 0;
}

// wespeaker/__init__.py:113-139 (compute_fbank, fbank_centering_span=None)
int pa_fbank(const float* wav, long wav_len, long chunk_stride, int B, int N, const float* window,
             const float* tw256, const float* tw512, const float* mel_w, const int* mel_lo,
             const int* mel_hi, int nmel, float* out, int center, void* stream) {
  PA_REQUIRE(nmel <= 80, "pa_fbank: at most 80 mel bins are built (got %d)", nmel);
  if (B <= 0) return 0;
  PA_REQUIRE(N >= pa::FB_WIN, "pa_fbank: %d samples is shorter than one 25 ms frame", N);
  const int T = 1 + (N - pa::FB_WIN) / pa::FB_SHIFT;
  hipStream_t st = (hipStream_t)stream;
  pa::ProfScope prof("k_fbank", stream, (double)B * T * (5.0 * 512 * 9 + 2.0 * 257 * 3 + 2.0 * 257 * 2),
                     4.0 * B * N + 4.0 * B * T * nmel * (center ? 3 : 1));
  hipLaunchKernelGGL(pa::k_fbank, dim3(pa::cdiv(T, pa::FB_FPB), B), dim3(256), 0, st, wav, wav_len,
                     chunk_stride, T, 32768.0f, 0.97f, window, (const float2*)tw256,
                     (const float2*)tw512, mel_w, mel_lo, mel_hi, nmel, 1.1920928955078125e-07f, out);
  PA_CHECK_LAUNCH("pa_fbank");
  if (center) {
    hipLaunchKernelGGL(pa::k_fbank_center, dim3(B), dim3(320), 0, st, out, T, nmel);
    PA_CHECK_LAUNCH("pa_fbank_center");
  }
  return 0;
}

}  // extern "C"
