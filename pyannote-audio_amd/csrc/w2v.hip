// Kernels of the wav2vec 2.0 / WavLM encoder under SSeRiouSS (reference: models/segmentation/SSeRiouSS.py:
// 289-313 -> torchaudio.models.wav2vec2 `extract_features`).  Everything GEMM-shaped (the strided feature
// extractor convolutions, every Linear, Q K^T and P V) runs on k_gemm_tn (seg_lstm.hip); this file holds what
// is left: the first convolution, the norms, the grouped positional convolution, the attention soft-max with
// WavLM's gated relative position bias, the layer mix and the re-ordering into LSTM rows.
//
// Layout: channels-last rows.  Row (b, t) of stage l lives at row b * P_l + t; the per-chunk pitches satisfy
// P_l = stride_{l+1} * P_{l+1}, so that the window of output row m of a stride-s, kernel-k convolution is
// the k * C CONTIGUOUS floats starting at input row s * m: the convolution is one GEMM with lda = s * C and
// K = k * C.  Rows t >= T_l of a chunk hold garbage that never reaches a valid row (every later stage is
// row-wise, windowed inside the chunk, or explicitly bounded by T).
#include "common.h"

namespace pa {

// ---- first convolution: (B chunks of a strided waveform) -> rows [b * P + t][C], kernel K0 <= 16 ----------
// grid = (ceil(T / 32), B), block = 256.  Rows T <= t < P are zeroed.
__global__ __launch_bounds__(256) void k_w2v_conv0(const float* __restrict__ wav, long wav_len, long chunk_stride,
                                                    int N, int T, int P, int C, int K0, int S0,
                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                    float* __restrict__ out) {
  __shared__ float xs[32 * 16 + 16];
  const int b = blockIdx.y, t0 = blockIdx.x * 32;
  const long base = (long)b * chunk_stride;
  const int span = 31 * S0 + K0;
  for (int i = threadIdx.x; i < span; i += 256) {
    const long p = (long)t0 * S0 + i;
    xs[i] = (p < N && base + p < wav_len) ? wav[base + p] : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float wk[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) wk[j] = j < K0 ? w[c * K0 + j] : 0.f;
    const float bv = bias != nullptr ? bias[c] : 0.f;
    for (int tt = 0; tt < 32; ++tt) {
      const int t = t0 + tt;
      if (t >= P) break;
      float acc = bv;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < K0) acc = fmaf(wk[j], xs[tt * S0 + j], acc);
      out[((long)b * P + t) * C + c] = t < T ? acc : 0.f;
    }
  }
}

// ---- GroupNorm(num_groups = C): per (chunk, channel) statistics over the T valid rows -----------------------
// grid = (C / 64, B), block = 256 = 64 channels x 4 row lanes; two passes (mean, then biased variance)
__global__ __launch_bounds__(256) void k_w2v_colstats(const float* __restrict__ x, int T, int P, int C, float eps,
                                                       float* __restrict__ mean, float* __restrict__ rstd) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const float* p = x + (long)b * P * C + c;
  float s = 0.f;
  if (c < C)
    for (int t = r; t < T; t += 4) s += p[(long)t * C];
  red[r][threadIdx.x & 63] = s;
  __syncthreads();
  const float m = (red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63] + red[2][threadIdx.x & 63] +
                   red[3][threadIdx.x & 63]) / (float)T;
  __syncthreads();
  float q = 0.f;
  if (c < C)
    for (int t = r; t < T; t += 4) {
      const float d = p[(long)t * C] - m;
      q += d * d;
    }
  red[r][threadIdx.x & 63] = q;
  __syncthreads();
  if (r == 0 && c < C) {
    const float var = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)T;
    mean[b * C + c] = m;
    rstd[b * C + c] = 1.f / sqrtf(var + eps);
  }
}

// y = gelu((x - mean) * rstd * gamma + beta) in place over the valid rows.  grid = (ceil(T / 16), B)
__global__ __launch_bounds__(256) void k_w2v_gn_gelu(float* __restrict__ x, int T, int P, int C,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gam, const float* __restrict__ bet) {
  const int b = blockIdx.y, t0 = blockIdx.x * 16;
  for (int i = threadIdx.x; i < 16 * C; i += 256) {
    const int t = t0 + i / C, c = i % C;
    if (t >= T) break;
    float* p = x + ((long)b * P + t) * C + c;
    *p = gelu_erf((*p - mean[b * C + c]) * (rstd[b * C + c] * gam[c]) + bet[c]);
  }
}

// ---- LayerNorm over the C channels of each row (eps 1e-5, biased variance), optional GELU -----------------
// one wave per row, C <= 64 * 16; in / out may alias.  grid = ceil(rows / 4), block = 256
template <bool GELU>
__global__ __launch_bounds__(256) void k_w2v_layernorm(const float* __restrict__ in, float* __restrict__ out,
                                                        long rows, int C, const float* __restrict__ gam,
                                                        const float* __restrict__ bet) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = in + row * C;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < C ? p[c] : 0.f;
    s += v[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float m = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + 64 * i;
    const float d = c < C ? v[i] - m : 0.f;
    q += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rs = 1.f / sqrtf(q / (float)C + 1e-5f);
  float* o_ = out + row * C;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + 64 * i;
    if (c < C) {
      const float y = (v[i] - m) * rs * gam[c] + bet[c];
      o_[c] = GELU ? gelu_erf(y) : y;
    }
  }
}

// ---- convolutional positional embedding: out = x + gelu(conv1d_grouped(x) + bias) --------------------------
// x: rows [b * P + t][D]; weight w3[g][j][ci][co] (co fastest), CG = D / groups channels per group,
// zero padding outside [0, T), output frames 0..T-1 (torchaudio drops the last frame of an even kernel).
// grid = (ceil(T / 16), groups, B), block = 256; dynamic LDS = (16 + KW - 1) * CG floats
__global__ __launch_bounds__(256) void k_w2v_posconv(const float* __restrict__ x, int T, int P, int D, int CG,
                                                      int KW, int pad, const float* __restrict__ w3,
                                                      const float* __restrict__ bias, float* __restrict__ out) {
  extern __shared__ float xs[];
  const int b = blockIdx.z, g = blockIdx.y, t0 = blockIdx.x * 16;
  const int rows = 16 + KW - 1;
  for (int i = threadIdx.x; i < rows * CG; i += 256) {
    const int r = i / CG, ci = i % CG;
    const int t = t0 + r - pad;
    xs[i] = (t >= 0 && t < T) ? x[((long)b * P + t) * D + g * CG + ci] : 0.f;
  }
  __syncthreads();
  const float* wg = w3 + (long)g * KW * CG * CG;
  for (int o = threadIdx.x; o < 16 * CG; o += 256) {
    const int tt = o / CG, co = o % CG;
    const int t = t0 + tt;
    if (t >= T) continue;
    float acc = bias[g * CG + co];
    for (int j = 0; j < KW; ++j) {
      const float* xr = xs + (tt + j) * CG;
      const float* wr = wg + ((long)j * CG) * CG + co;
      for (int ci = 0; ci < CG; ++ci) acc = fmaf(xr[ci], wr[(long)ci * CG], acc);
    }
    const long idx = ((long)b * P + t) * D + g * CG + co;
    out[idx] = x[idx] + gelu_erf(acc);
  }
}

// ---- attention soft-max, in place: P = softmax(S * scale + gate(b,h,t) * bias[h][t][:]) over k < T ----------
// S: [B][H][T][Tp] (row pitch Tp >= T, columns >= T are set to 0).  bias: [H][T][T] or NULL (wav2vec 2.0).
// gate (WavLM, wavlm_attention.py: gated relative position bias): from the LAYER INPUT xin (rows [b*P+t][D]),
// head slice q = xin[b, t, h*hd : (h+1)*hd]: u = Wg q + bg (8 values), ga = sigmoid(u0+u1+u2+u3),
// gb = sigmoid(u4+..+u7), gate = ga * (gb * const[h] - 1) + 2.
// one wave per row (b, h, t); hd <= 128.  grid = ceil(B*H*T / 4), block = 256
__global__ __launch_bounds__(256) void k_w2v_softmax(float* __restrict__ S, int B, int H, int T, int Tp, float scale,
                                                      const float* __restrict__ bias,
                                                      const float* __restrict__ xin, int P, int D, int hd,
                                                      const float* __restrict__ gw, const float* __restrict__ gb_,
                                                      const float* __restrict__ gconst) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * H * T) return;
  const int lane = threadIdx.x & 63;
  const int t = (int)(row % T), h = (int)((row / T) % H), b = (int)(row / ((long)T * H));
  float gate = 0.f;
  if (bias != nullptr) {
    const float* q = xin + ((long)b * P + t) * D + h * hd;
    const float q0 = lane < hd ? q[lane] : 0.f, q1 = lane + 64 < hd ? q[lane + 64] : 0.f;
    float u[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float d = (lane < hd ? gw[e * hd + lane] * q0 : 0.f) + (lane + 64 < hd ? gw[e * hd + lane + 64] * q1 : 0.f);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
      u[e] = d + gb_[e];
    }
    const float ga = 1.f / (1.f + expf(-(u[0] + u[1] + u[2] + u[3])));
    const float gbv = 1.f / (1.f + expf(-(u[4] + u[5] + u[6] + u[7])));
    gate = ga * (gbv * gconst[h] - 1.f) + 2.f;
  }
  float* s = S + row * Tp;
  const float* br = bias != nullptr ? bias + ((long)h * T + t) * T : nullptr;
  float mx = -__builtin_inff();
  for (int k = lane; k < T; k += 64) {
    float v = s[k] * scale;
    if (br != nullptr) v = fmaf(gate, br[k], v);
    s[k] = v;
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int k = lane; k < T; k += 64) {
    const float e = expf(s[k] - mx);
    s[k] = e;
    sum += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.f / sum;
  for (int k = lane; k < Tp; k += 64) s[k] = k < T ? s[k] * inv : 0.f;
}

// acc = (first ? 0 : acc) + w * x
__global__ __launch_bounds__(256) void k_w2v_axpy(float* __restrict__ acc, const float* __restrict__ x, float w,
                                                   long n, int first) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 xv = *reinterpret_cast<const float4*>(x + i);
  float4 a = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(acc + i);
  a.x = fmaf(w, xv.x, a.x);
  a.y = fmaf(w, xv.y, a.y);
  a.z = fmaf(w, xv.z, a.z);
  a.w = fmaf(w, xv.w, a.w);
  *reinterpret_cast<float4*>(acc + i) = a;
}

// rows [b * P + t][D] -> LSTM input rows [((b >> 4) * T + t) * 16 + (b & 15)][D]; chunks b >= B are zero
// grid = (T, ntiles * 16), block = 256
__global__ __launch_bounds__(256) void k_w2v_to_tiles(const float* __restrict__ x, int B, int T, int P, int D,
                                                       float* __restrict__ out) {
  const int t = blockIdx.x, b = blockIdx.y;
  float* o = out + (((long)(b >> 4) * T + t) * 16 + (b & 15)) * D;
  const float* p = x + ((long)b * P + t) * D;
  for (int c = threadIdx.x; c < D; c += 256) o[c] = b < B ? p[c] : 0.f;
}

}  // namespace pa

extern "C" {

PA_INTERNAL int pa_w2v_conv0(const float* wav, long wav_len, long chunk_stride, int B, int N, int T, int P, int C, int K0,
                 int S0, const float* w, const float* bias, float* out, void* stream) {
  if (B <= 0) return 0;
  PA_REQUIRE(K0 >= 1 && K0 <= 16 && S0 >= 1 && S0 <= 16, "pa_w2v_conv0: kernel and stride <= 16 required");
  pa::ProfScope prof("k_w2v_conv0", stream, 2.0 * B * T * C * K0, 4.0 * B * (N + (double)P * C));
  hipLaunchKernelGGL(pa::k_w2v_conv0, dim3(pa::cdiv(P, 32), B), dim3(256), 0, (hipStream_t)stream, wav, wav_len,
                     chunk_stride, N, T, P, C, K0, S0, w, bias, out);
  PA_CHECK_LAUNCH("pa_w2v_conv0");
  return 0;
}

PA_INTERNAL int pa_w2v_group_norm_gelu(float* x, int B, int T, int P, int C, const float* gamma, const float* beta,
                           float* mean_scratch, float* rstd_scratch, void* stream) {
  if (B <= 0) return 0;
  pa::ProfScope prof("k_w2v_group_norm", stream, 8.0 * B * T * C, 16.0 * B * T * C);
  hipLaunchKernelGGL(pa::k_w2v_colstats, dim3(pa::cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream, x, T, P, C,
                     1e-5f, mean_scratch, rstd_scratch);
  hipLaunchKernelGGL(pa::k_w2v_gn_gelu, dim3(pa::cdiv(T, 16), B), dim3(256), 0, (hipStream_t)stream, x, T, P, C,
                     mean_scratch, rstd_scratch, gamma, beta);
  PA_CHECK_LAUNCH("pa_w2v_group_norm_gelu");
  return 0;
}

PA_INTERNAL int pa_w2v_layernorm(const float* in, float* out, long rows, int C, const float* gamma, const float* beta,
                     int gelu, void* stream) {
  if (rows <= 0) return 0;
  PA_REQUIRE(C >= 1 && C <= 1024, "pa_w2v_layernorm: C <= 1024 required (got %d)", C);
  pa::ProfScope prof("k_w2v_layernorm", stream, 8.0 * rows * C, 8.0 * rows * C);
  if (gelu)
    hipLaunchKernelGGL(pa::k_w2v_layernorm<true>, dim3(pa::cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, in,
                       out, rows, C, gamma, beta);
  else
    hipLaunchKernelGGL(pa::k_w2v_layernorm<false>, dim3(pa::cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, in,
                       out, rows, C, gamma, beta);
  PA_CHECK_LAUNCH("pa_w2v_layernorm");
  return 0;
}

PA_INTERNAL int pa_w2v_posconv(const float* x, int B, int T, int P, int D, int groups, int KW, const float* w3,
                   const float* bias, float* out, void* stream) {
  if (B <= 0) return 0;
  const int CG = D / groups;
  const size_t lds = (size_t)(16 + KW - 1) * CG * sizeof(float);
  PA_REQUIRE(D % groups == 0 && lds <= 64 * 1024, "pa_w2v_posconv: (16 + kernel - 1) * D / groups floats of LDS");
  pa::ProfScope prof("k_w2v_posconv", stream, 2.0 * B * T * D * CG * KW, 8.0 * B * T * D);
  hipLaunchKernelGGL(pa::k_w2v_posconv, dim3(pa::cdiv(T, 16), groups, B), dim3(256), lds, (hipStream_t)stream, x, T,
                     P, D, CG, KW, KW / 2, w3, bias, out);
  PA_CHECK_LAUNCH("pa_w2v_posconv");
  return 0;
}

PA_INTERNAL int pa_w2v_softmax(float* S, int B, int H, int T, int Tp, float scale, const float* bias, const float* xin, int P,
                   int D, const float* gate_w, const float* gate_b, const float* gate_const, void* stream) {
  if (B <= 0) return 0;
  PA_REQUIRE(D % H == 0 && D / H <= 128, "pa_w2v_softmax: head dimension <= 128 required");
  const long rows = (long)B * H * T;
  pa::ProfScope prof("k_w2v_softmax", stream, 6.0 * rows * T, 8.0 * rows * T);
  hipLaunchKernelGGL(pa::k_w2v_softmax, dim3(pa::cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, S, B, H, T, Tp,
                     scale, bias, xin, P, D, D / H, gate_w, gate_b, gate_const);
  PA_CHECK_LAUNCH("pa_w2v_softmax");
  return 0;
}

PA_INTERNAL int pa_w2v_axpy(float* acc, const float* x, float w, long n, int first, void* stream) {
  if (n <= 0) return 0;
  PA_REQUIRE(n % 4 == 0, "pa_w2v_axpy: n %% 4 == 0 required");
  pa::ProfScope prof("k_w2v_axpy", stream, 2.0 * n, 12.0 * n);
  hipLaunchKernelGGL(pa::k_w2v_axpy, dim3(pa::cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, acc, x, w, n,
                     first);
  PA_CHECK_LAUNCH("pa_w2v_axpy");
  return 0;
}

PA_INTERNAL int pa_w2v_to_tiles(const float* x, int B, int T, int P, int D, float* out, void* stream) {
  if (B <= 0) return 0;
  const int ntiles = (B + 15) / 16;
  pa::ProfScope prof("k_w2v_to_tiles", stream, 0.0, 8.0 * B * T * D);
  hipLaunchKernelGGL(pa::k_w2v_to_tiles, dim3(T, ntiles * 16), dim3(256), 0, (hipStream_t)stream, x, B, T, P, D, out);
  PA_CHECK_LAUNCH("pa_w2v_to_tiles");
  return 0;
}

}  // extern "C"
