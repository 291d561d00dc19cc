// SincNet front-end of PyanNet on gfx950 (reference: models/blocks/sincnet.py:163-184).
//
//   wav --InstanceNorm1d(1)--> sinc FIR(80x251, stride 10) --|.|--> maxpool3 --IN(80)+lrelu-->
//       conv1d(80->60,k5) --> maxpool3 --IN(60)+lrelu--> conv1d(60->60,k5) --> maxpool3 --IN(60)+lrelu
//
// Kernel split (one launch each, all on the caller's stream):
//   k_row_stats        two-pass mean / rstd of a row (waveform chunk or one (chunk,channel) row)
//   k_sinc_fir_pool    normalise-on-load -> FIR as f32 MFMA GEMM -> |.| -> maxpool3   (un-normalised out)
//   k_conv5_pool<CIN>  IN+lrelu-on-load  -> conv1d k=5 as f32 MFMA GEMM -> +bias -> maxpool3
//   k_norm_transpose   IN+lrelu of the last stage, written as LSTM input rows [(tile,t,b16)][64]
//
// The max-pool over 3 consecutive conv positions is lane-local: MFMA row i of accumulator j
// (j = 0,1,2) is mapped to conv position 3*i + j, so the three accumulators of a lane hold the
// three members of one pooling window.
#include "common.h"

namespace pa {

// ---------------------------------------------------------------------------------------------
// Row statistics: mean and 1/sqrt(var_biased + eps) over `len` values; values beyond `valid`
// (zero padding of the last chunk, inference.py:270-278) count as zeros.
// grid = rows, block = 256.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_stats(const float* __restrict__ x, long row_stride,
                                                    long total_len, int len, float eps,
                                                    float* __restrict__ mean_out,
                                                    float* __restrict__ rstd_out) {
  __shared__ float red[4];
  const long base = (long)blockIdx.x * row_stride;
  long avail = total_len - base;  // number of real samples in this row
  int valid = avail >= len ? len : (avail > 0 ? (int)avail : 0);
  const float* row = x + base;
  float s = 0.f;
  for (int i = threadIdx.x; i < valid; i += 256) s += row[i];
  const float mean = block_sum<4>(s, red) / (float)len;
  float q = 0.f;
  for (int i = threadIdx.x; i < valid; i += 256) {
    float d = row[i] - mean;
    q += d * d;
  }
  // zero-padded tail contributes (0 - mean)^2 each
  float var = block_sum<4>(q, red);
  var += (float)(len - valid) * mean * mean;
  var /= (float)len;
  if (threadIdx.x == 0) {
    mean_out[blockIdx.x] = mean;
    rstd_out[blockIdx.x] = 1.0f / sqrtf(var + eps);
  }
}

// ---------------------------------------------------------------------------------------------
// Sinc FIR + abs + maxpool3.
//   grid = (ceil(P / PT), B), block = 640: wave w owns filters 16 (w % 5) .. + 15 and HALF of the workgroup's
//   positions (w / 5).  Ten waves, not five: 80 filters are five 16-wide MFMA tiles, and five waves put two on
//   SIMD 0 and one on each other SIMD (the launch then runs at the pace of SIMD 0, 5/8 of the CU); ten waves sit
//   3-3-2-2.
//   filt : B-operand image [5][63][64] : filt[(w*63+kt)*64 + lane] = h[16w + (lane&15)][4kt + (lane>>4)]
//          (tap 251 is zero padding).
//   out  : (B, 80, P) un-normalised pooled magnitudes.
// ---------------------------------------------------------------------------------------------
constexpr int SINC_PT = 128;                       // pooled outputs per workgroup
constexpr int SINC_XS = 30 * SINC_PT + 256;        // staged samples (>= 30*PT + 242)
constexpr int SINC_OS = SINC_PT + 4;               // out-tile row stride in LDS
constexpr int SINC_T = 640;                        // threads

// RAW = true (the sinc layer ONCE for a whole span of overlapping chunks, see k_sinc_fix_pool): no normalisation on
// load, no magnitude, no pooling -- MFMA row i is convolution position i, `P` counts convolution positions and `out`
// is (80, P) raw filter outputs of the span [wav, wav + N).
// STR: the SincNet stride (models/blocks/sincnet.py:58-69 accepts any; the released checkpoints use 10).  The staged
// sample window of a workgroup is 3 STR SINC_PT + 256 floats (30 KB at STR = 20).
template <bool RAW, int STR = 10>
__global__ __launch_bounds__(SINC_T) void k_sinc_fir_pool(
    const float* __restrict__ wav, long wav_len, long chunk_stride, int N, int stride, int P,
    const float* __restrict__ mean, const float* __restrict__ rstd, float gamma, float beta,
    const float* __restrict__ filt, float* __restrict__ out) {
  constexpr int XS_MAX = 3 * STR * SINC_PT + 256;   // (= SINC_XS at STR = 10)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                // [XS_MAX]
  float* os = smem + XS_MAX;       // [80][SINC_OS]
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * SINC_PT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = (tid >> 6) % 5, half = (tid >> 6) / 5;   // filter tile, half of the positions
  constexpr int PS = RAW ? STR : 3 * STR;   // sample advance per output row (pooled: three positions per row)
  constexpr int XS = RAW ? STR * SINC_PT + 256 : XS_MAX;    // staged samples

  // stage normalised samples [PS*p0, PS*p0 + nstage)
  const long cbase = (long)b * chunk_stride;
  const float mu = RAW ? 0.f : mean[b], rs = RAW ? 1.f : rstd[b] * gamma;
  if (RAW) beta = 0.f;
  // (all loads of a thread are issued before the first use: as a plain loop the compiler waits for every load
  //  in turn)
  {
    constexpr int NS = (XS_MAX + SINC_T - 1) / SINC_T;
    float raw[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int i = tid + SINC_T * k;
      const long g = cbase + PS * p0 + i;
      raw[k] = (i < XS && PS * p0 + i < N && g < wav_len) ? wav[g] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int i = tid + SINC_T * k;
      if (i < XS) xs[i] = PS * p0 + i < N ? (raw[k] - mu) * rs + beta : 0.f;
    }
  }
  // filter taps -> registers (B operand)
  float fb[63];
#pragma unroll
  for (int kt = 0; kt < 63; ++kt) fb[kt] = filt[(w * 63 + kt) * 64 + lane];
  __syncthreads();

  const int i16 = lane & 15, kq = lane >> 4;
#pragma unroll 1
  for (int grp = half * (SINC_PT / 32); grp < (half + 1) * (SINC_PT / 32); ++grp) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    const float* xp = xs + PS * (16 * grp + i16) + kq;
#pragma unroll
    for (int kt = 0; kt < 63; ++kt) {
      a0 = MFMA16(xp[4 * kt], fb[kt], a0);
      if (!RAW) {
        a1 = MFMA16(xp[4 * kt + STR], fb[kt], a1);
        a2 = MFMA16(xp[4 * kt + 2 * STR], fb[kt], a2);
      }
    }
    // lane holds filter 16w + i16 (column), output rows 16*grp + 4*kq + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = RAW ? a0[r] : fmaxf(fmaxf(fabsf(a0[r]), fabsf(a1[r])), fabsf(a2[r]));
      os[(16 * w + i16) * SINC_OS + 16 * grp + 4 * kq + r] = v;
    }
  }
  __syncthreads();
  const int np = min(SINC_PT, P - p0);
  for (int i = tid; i < 80 * SINC_PT; i += SINC_T) {
    const int c = i / SINC_PT, p = i % SINC_PT;
    if (p < np) out[((long)b * 80 + c) * P + p0 + p] = os[c * SINC_OS + p];
  }
}

// ---------------------------------------------------------------------------------------------
// The sinc layer once per SPAN of overlapping chunks (the default of pa_seg_forward for overlapping chunks whose
// stride is a multiple of 10 samples; PA_SEG_SHARED_SINC=0 selects the per-chunk kernel for an A/B.  Numerics:
// tools/probes/shared_sinc_numerics.py on the CPU, tests/test_seg_gpu.py::test_shared_sinc_layer_* on MI355X;
// measured per audio-hour: 21.2 ms -> 2.6 + 3.2 ms).  Chunks of the sliding window start at multiples of Q = chunk_stride /
// 10 convolution positions and the waveform InstanceNorm is affine, so with S = sinc(raw span), S1[f] = sum of
// the taps of filter f and g = rstd * gamma:
//     sinc((x - mu) g + beta)[f][q] = g (S[f][c Q + q] - mu S1[f]) + beta S1[f]
// k_sinc_tapsum: S1 from the packed B-operand image.  k_sinc_fix_pool: the fix-up + |.| + maxpool3 of chunk b,
// (B, 80, P) out, as k_sinc_fir_pool<false> writes it.  grid = (ceil(P / 256), B), block = 256; a thread walks
// the 80 filters of its pooled position (reads 12 contiguous bytes per filter, coalesced across the wave).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_sinc_tapsum(const float* __restrict__ filt, float* __restrict__ S1) {
  const int f = threadIdx.x;
  if (f >= 80) return;
  double acc = 0.0;
  for (int kt = 0; kt < 63; ++kt)
    for (int q = 0; q < 4; ++q) acc += (double)filt[((f >> 4) * 63 + kt) * 64 + (f & 15) + 16 * q];
  S1[f] = (float)acc;
}

__global__ __launch_bounds__(256) void k_sinc_fix_pool(const float* __restrict__ S, long Pc, int Q, int P,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float gamma, float beta,
                                                        const float* __restrict__ S1, float* __restrict__ out) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const float mu = mean[b], g = rstd[b] * gamma;
  const float* src = S + (long)b * Q + 3 * p;
  float* dst = out + (long)b * 80 * P + p;
#pragma unroll 8
  for (int f = 0; f < 80; ++f) {
    const float s1 = S1[f], off = beta * s1 - g * mu * s1;
    const float* row = src + (long)f * Pc;
    const float v0 = fabsf(fmaf(g, row[0], off)), v1 = fabsf(fmaf(g, row[1], off)), v2 = fabsf(fmaf(g, row[2], off));
    dst[(long)f * P] = fmaxf(fmaxf(v0, v1), v2);
  }
}

// ---------------------------------------------------------------------------------------------
// conv1d(k=5, stride 1) + bias + maxpool3 with InstanceNorm+leaky_relu applied to the input on load.
//   grid = (ceil(P / PT), B), block = 256 (4 waves; wave w owns output channels 16w..16w+15, 60 real).
//   wp : B-operand image [4][KT][64], K order k = tap*CIN + cin, KT = 5*CIN/4:
//        wp[(w*KT+kt)*64 + lane] = W[16w + (lane&15)][cin][tap],  4kt + (lane>>4) = tap*CIN + cin
//   xin: (B, CIN, Lin) un-normalised; in_mean/in_rstd: (B*CIN); gam/bet: (CIN)
//   out: (B, 60, P) un-normalised pooled conv outputs.
// ---------------------------------------------------------------------------------------------
constexpr int CV_PT = 32;
constexpr int CV_XW = 112;   // staged positions per channel row (>= 3*PT+4), 112 % 32 == 16: see DESIGN.md
constexpr int CV_OS = CV_PT + 1;

template <int CIN>
__global__ __launch_bounds__(256) void k_conv5_pool(const float* __restrict__ xin, int Lin, int P,
                                                     const float* __restrict__ in_mean,
                                                     const float* __restrict__ in_rstd,
                                                     const float* __restrict__ gam,
                                                     const float* __restrict__ bet,
                                                     const float* __restrict__ wp,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ out) {
  constexpr int KT = 5 * CIN / 4;
  constexpr int KPT = CIN / 4;  // k-steps per tap
  __shared__ __attribute__((aligned(16))) float xs[CIN * CV_XW];
  __shared__ float os[64 * CV_OS];
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * CV_PT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

  // stage: wave w owns channel rows w, w + 4, ...; a lane covers positions lane and lane + 64 of the row.  Five
  // rows (10 loads + their statistics) are in flight per trip: the element-wise loop this replaces waited for
  // each of its 35 loads in turn (+ a division and four statistic loads per element).
  {
    static_assert(CIN % 20 == 0, "five channel rows per wave and trip");
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int q1 = lane + 64;
    const bool ok0 = 3 * p0 + lane < Lin;                        // (lane < 3 * CV_PT + 4 always)
    const bool ok1 = q1 < 3 * CV_PT + 4 && 3 * p0 + q1 < Lin;
#pragma unroll
    for (int j0 = 0; j0 < CIN / 4; j0 += 5) {
      float x0[5], x1[5], mu[5], sc[5], sh[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int c = wu + 4 * (j0 + u), row = b * CIN + c;
        const float* src = xin + (long)row * Lin + 3 * p0;
        x0[u] = ok0 ? src[lane] : 0.f;
        x1[u] = ok1 ? src[q1] : 0.f;
        mu[u] = in_mean[row];
        sc[u] = in_rstd[row] * gam[c];
        sh[u] = bet[c];
      }
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        float* dst = xs + (wu + 4 * (j0 + u)) * CV_XW;
        dst[lane] = ok0 ? leaky_relu((x0[u] - mu[u]) * sc[u] + sh[u]) : 0.f;
        if (q1 < CV_XW) dst[q1] = ok1 ? leaky_relu((x1[u] - mu[u]) * sc[u] + sh[u]) : 0.f;
      }
    }
  }
  float wb[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) wb[kt] = wp[(w * KT + kt) * 64 + lane];
  __syncthreads();

  const int i16 = lane & 15, kq = lane >> 4;
  const float bv = bias[16 * w + i16];
#pragma unroll 1
  for (int grp = 0; grp < CV_PT / 16; ++grp) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    const float* xp = xs + kq * CV_XW + 3 * (16 * grp + i16);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int tap = kt / KPT, c4 = (kt % KPT) * 4;
      const float* q = xp + c4 * CV_XW + tap;
      a0 = MFMA16(q[0], wb[kt], a0);
      a1 = MFMA16(q[1], wb[kt], a1);
      a2 = MFMA16(q[2], wb[kt], a2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = fmaxf(fmaxf(a0[r], a1[r]), a2[r]) + bv;
      os[(16 * w + i16) * CV_OS + 16 * grp + 4 * kq + r] = v;
    }
  }
  __syncthreads();
  const int np = min(CV_PT, P - p0);
  for (int i = tid; i < 60 * CV_PT; i += 256) {
    const int c = i / CV_PT, p = i % CV_PT;
    if (p < np) out[((long)b * 60 + c) * P + p0 + p] = os[c * CV_OS + p];
  }
}

// ---------------------------------------------------------------------------------------------
// Last InstanceNorm + leaky_relu, transposed into LSTM input rows:
//   X0[((tile*T + t)*16 + b16)][64], b = 16*tile + b16, channels 60..63 and chunks b >= B are zero.
// grid = (ceil(T/64), ntiles*16), block = 256.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_norm_transpose(const float* __restrict__ xin, int B, int T,
                                                         const float* __restrict__ in_mean,
                                                         const float* __restrict__ in_rstd,
                                                         const float* __restrict__ gam,
                                                         const float* __restrict__ bet,
                                                         float* __restrict__ X0) {
  __shared__ float tile[64][65];
  const int b = blockIdx.y, t0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int c = i >> 6, tt = i & 63;
    float v = 0.f;
    if (b < B && c < 60 && t0 + tt < T) {
      const int row = b * 60 + c;
      v = (xin[(long)row * T + t0 + tt] - in_mean[row]) * (in_rstd[row] * gam[c]) + bet[c];
      v = leaky_relu(v);
    }
    tile[c][tt] = v;
  }
  __syncthreads();
  const int bt = b >> 4, b16 = b & 15;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int tt = i >> 6, c = i & 63;
    if (t0 + tt < T) X0[(((long)bt * T + t0 + tt) * 16 + b16) * 64 + c] = tile[c][tt];
  }
}

}  // namespace pa

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int pa_row_stats(const float* x, long row_stride, long total_len, int rows, int len, float eps,
                 float* mean, float* rstd, void* stream) {
  if (rows <= 0) return 0;
  pa::ProfScope prof("k_row_stats", stream, 5.0 * rows * len, 8.0 * rows * len);
  hipLaunchKernelGGL(pa::k_row_stats, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, row_stride,
                     total_len, len, eps, mean, rstd);
  PA_CHECK_LAUNCH("pa_row_stats");
  return 0;
}

int pa_sinc_fir_pool(const float* wav, long wav_len, long chunk_stride, int B, int N, int stride,
                     const float* mean, const float* rstd, float gamma, float beta,
                     const float* filt_packed, float* out, void* stream) {
  PA_REQUIRE(stride >= 1 && N >= 251, "pa_sinc_fir_pool: stride %d / %d samples", stride, N);
  const int L = (N - 251) / stride + 1;
  const int P = L / 3;
  if (B <= 0 || P <= 0) return 0;
  pa::ProfScope prof("k_sinc_fir_pool", stream, 2.0 * B * 80 * 251 * (3.0 * P),
                     4.0 * B * N + 4.0 * B * 80 * P);
#define PA_SINC_LAUNCH(S)                                                                                          \
  do {                                                                                                             \
    const size_t lds = (size_t)(3 * (S) * pa::SINC_PT + 256 + 80 * pa::SINC_OS) * sizeof(float);                     \
    /* (set on every call: the attribute belongs to the current device, not to the process) */                     \
    (void)hipFuncSetAttribute((const void*)pa::k_sinc_fir_pool<false, S>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                           \
    hipLaunchKernelGGL((pa::k_sinc_fir_pool<false, S>), dim3(pa::cdiv(P, pa::SINC_PT), B), dim3(pa::SINC_T), lds,  \
                       (hipStream_t)stream, wav, wav_len, chunk_stride, N, stride, P, mean, rstd, gamma, beta,      \
                       filt_packed, out);                                                                          \
  } while (0)
  // the strides that are built (the kernel's staging window and MFMA row pitch are compile-time)
  switch (stride) {
    case 10: PA_SINC_LAUNCH(10); break;
    case 1: PA_SINC_LAUNCH(1); break;
    case 2: PA_SINC_LAUNCH(2); break;
    case 4: PA_SINC_LAUNCH(4); break;
    case 5: PA_SINC_LAUNCH(5); break;
    case 8: PA_SINC_LAUNCH(8); break;
    case 16: PA_SINC_LAUNCH(16); break;
    case 20: PA_SINC_LAUNCH(20); break;
    default:
      PA_REQUIRE(false, "pa_sinc_fir_pool: SincNet stride %d is not built (1, 2, 4, 5, 8, 10, 16, 20 are)", stride);
  }
#undef PA_SINC_LAUNCH
  PA_CHECK_LAUNCH("pa_sinc_fir_pool");
  return 0;
}

// EXPERIMENTAL (see k_sinc_fix_pool).  S: (80, Pc) raw sinc outputs of the span wav[0, span) (zeros past wav_len),
// Pc = (span - 251) / 10 + 1.
int pa_sinc_fir_span(const float* wav, long wav_len, long span, const float* filt_packed, float* S,
                     void* stream) {
  if (span < 251) return 0;
  PA_REQUIRE(span <= 0x7fffffffL, "pa_sinc_fir_span: span of %ld samples is too long", span);
  const int Pc = (int)((span - 251) / 10 + 1);
  const size_t lds = (pa::SINC_XS + 80 * pa::SINC_OS) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)pa::k_sinc_fir_pool<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  pa::ProfScope prof("k_sinc_fir_span", stream, 2.0 * 80 * 251 * (double)Pc, 4.0 * span + 4.0 * 80 * Pc);
  hipLaunchKernelGGL(pa::k_sinc_fir_pool<true>, dim3(pa::cdiv(Pc, pa::SINC_PT), 1), dim3(pa::SINC_T), lds,
                     (hipStream_t)stream, wav, wav_len, 0L, (int)span, 10, Pc, (const float*)nullptr,
                     (const float*)nullptr, 1.f, 0.f, filt_packed, S);
  PA_CHECK_LAUNCH("pa_sinc_fir_span");
  return 0;
}

int pa_sinc_fix_pool(const float* S, long Pc, int positions_per_chunk_step, int B, int P, const float* mean,
                     const float* rstd, float gamma, float beta, const float* filt_packed, float* tap_sums,
                     float* out, void* stream) {
  if (B <= 0 || P <= 0) return 0;
  pa::ProfScope prof("k_sinc_fix_pool", stream, 8.0 * B * 80 * 3.0 * P, 4.0 * B * 80 * 4.0 * P);
  hipLaunchKernelGGL(pa::k_sinc_tapsum, dim3(1), dim3(128), 0, (hipStream_t)stream, filt_packed, tap_sums);
  hipLaunchKernelGGL(pa::k_sinc_fix_pool, dim3(pa::cdiv(P, 256), B), dim3(256), 0, (hipStream_t)stream, S, Pc,
                     positions_per_chunk_step, P, mean, rstd, gamma, beta, (const float*)tap_sums, out);
  PA_CHECK_LAUNCH("pa_sinc_fix_pool");
  return 0;
}

int pa_conv5_pool(const float* xin, int B, int cin, int Lin, const float* in_mean,
                  const float* in_rstd, const float* gam, const float* bet, const float* w_packed,
                  const float* bias64, float* out, void* stream) {
  const int P = (Lin - 4) / 3;
  if (B <= 0 || P <= 0) return 0;
  dim3 grid(pa::cdiv(P, pa::CV_PT), B);
  pa::ProfScope prof("k_conv5_pool", stream, 2.0 * B * 60 * cin * 5 * (3.0 * P),
                     4.0 * B * cin * Lin + 4.0 * B * 60 * P);
  if (cin == 80)
    hipLaunchKernelGGL(pa::k_conv5_pool<80>, grid, dim3(256), 0, (hipStream_t)stream, xin, Lin, P,
                       in_mean, in_rstd, gam, bet, w_packed, bias64, out);
  else if (cin == 60)
    hipLaunchKernelGGL(pa::k_conv5_pool<60>, grid, dim3(256), 0, (hipStream_t)stream, xin, Lin, P,
                       in_mean, in_rstd, gam, bet, w_packed, bias64, out);
  else
    PA_REQUIRE(false, "pa_conv5_pool: unsupported cin %d", cin);
  PA_CHECK_LAUNCH("pa_conv5_pool");
  return 0;
}

int pa_norm_transpose(const float* xin, int B, int T, const float* in_mean, const float* in_rstd,
                      const float* gam, const float* bet, float* X0, void* stream) {
  if (B <= 0) return 0;
  const int ntiles = (B + 15) / 16;
  pa::ProfScope prof("k_norm_transpose", stream, 4.0 * B * 60 * T, 4.0 * B * 60 * T + 4.0 * ntiles * 16 * 64 * T);
  hipLaunchKernelGGL(pa::k_norm_transpose, dim3(pa::cdiv(T, 64), ntiles * 16), dim3(256), 0,
                     (hipStream_t)stream, xin, B, T, in_mean, in_rstd, gam, bet, X0);
  PA_CHECK_LAUNCH("pa_norm_transpose");
  return 0;
}

}  // extern "C"
