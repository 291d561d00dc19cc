// Weighted temporal statistics pooling (TSTP) on gfx950
// (reference: models/embedding/wespeaker/resnet.py:49-66 + models/blocks/pooling.py:30-61, 76-130).
//
// Input  feat[b][f][t][c]   (NHWC output of layer4: f = 10 mel rows, t = T' frames, c = 256)
// Masks  w[b][s][Fm]        per (chunk, local speaker) frame weights at the segmentation resolution;
//                           resized to T' with F.interpolate(mode="nearest") = idx[t] (host-computed
//                           with torch's float formula floor(t * (float)Fm / T'))
// Output stats[b][s][2*D]   D = c*10 + f (TSTP's channel-major flattening): mean | std
//   v1 = sum w + 1e-8 ; mean = sum(x w)/v1 ; var = sum((x-mean)^2 w) / (v1 - sum(w^2)/v1 + 1e-8)
// The backbone runs ONCE per chunk and is pooled for all S speakers here (the reference runs the whole
// network once per (chunk, speaker), pipelines/speaker_diarization.py:417-425).
#include "common.h"

namespace pa {

constexpr int POOL_MAXS = 4;
constexpr int POOL_MAXT = 512;
constexpr int POOL_LD = 16;   // time steps whose loads are in flight together

__global__ __launch_bounds__(256) void k_stats_pool(const float* __restrict__ feat, int Fh, int Tp,
                                                    int Cc, const float* __restrict__ masks, int S,
                                                    int Fm, const int* __restrict__ idx,
                                                    float* __restrict__ stats) {
  __shared__ float ws[POOL_MAXS][POOL_MAXT];
  __shared__ float v1s[POOL_MAXS], dens[POOL_MAXS];
  const int b = blockIdx.z, f = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  for (int i = threadIdx.x; i < S * Tp; i += 256) {
    const int s = i / Tp, t = i % Tp;
    ws[s][t] = masks != nullptr ? masks[((long)b * S + s) * Fm + idx[t]] : 1.f;
  }
  __syncthreads();
  if (threadIdx.x < S) {
    const int s = threadIdx.x;
    float a = 0.f, q = 0.f;
    for (int t = 0; t < Tp; ++t) {
      a += ws[s][t];
      q += ws[s][t] * ws[s][t];
    }
    const float v1 = a + 1e-8f;
    v1s[s] = v1;
    dens[s] = v1 - q / v1 + 1e-8f;
  }
  __syncthreads();
  if (c >= Cc) return;
  const float* x = feat + (((long)b * Fh + f) * Tp) * Cc + c;
  float m[POOL_MAXS];
#pragma unroll
  for (int s = 0; s < POOL_MAXS; ++s) m[s] = 0.f;
  // (loads in batches of POOL_LD: one load + a full wait per trip otherwise -- the same operations in the same order)
  for (int t0 = 0; t0 < Tp; t0 += POOL_LD) {
    float xb[POOL_LD];
#pragma unroll
    for (int u = 0; u < POOL_LD; ++u) xb[u] = t0 + u < Tp ? x[(long)(t0 + u) * Cc] : 0.f;
#pragma unroll
    for (int u = 0; u < POOL_LD; ++u) {
      if (t0 + u < Tp) {
#pragma unroll
        for (int s = 0; s < POOL_MAXS; ++s)
          if (s < S) m[s] = fmaf(xb[u], ws[s][t0 + u], m[s]);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < POOL_MAXS; ++s)
    if (s < S) m[s] /= v1s[s];
  float v[POOL_MAXS];
#pragma unroll
  for (int s = 0; s < POOL_MAXS; ++s) v[s] = 0.f;
  for (int t0 = 0; t0 < Tp; t0 += POOL_LD) {
    float xb[POOL_LD];
#pragma unroll
    for (int u = 0; u < POOL_LD; ++u) xb[u] = t0 + u < Tp ? x[(long)(t0 + u) * Cc] : 0.f;
#pragma unroll
    for (int u = 0; u < POOL_LD; ++u) {
      if (t0 + u < Tp) {
#pragma unroll
        for (int s = 0; s < POOL_MAXS; ++s)
          if (s < S) {
            const float d = xb[u] - m[s];
            v[s] = fmaf(d * d, ws[s][t0 + u], v[s]);
          }
      }
    }
  }
  const int D = Cc * Fh;
  const int d = c * Fh + f;
#pragma unroll
  for (int s = 0; s < POOL_MAXS; ++s)
    if (s < S) {
      float* o = stats + ((long)b * S + s) * 2 * D;
      o[d] = m[s];
      o[D + d] = sqrtf(v[s] / dens[s]);
    }
}

// The same pooling over the rows of a (tile, t, b16)-ordered activation matrix (the row order of the
// segmentation / x-vector stacks: row(b, t) = ((b >> 4) * T0 + t) * 16 + (b & 15), `ld` floats per row):
// StatsPool of XVectorSincNet (models/embedding/xvector.py:343-348, models/blocks/pooling.py:64-130).
// stats[b][s][ld_stats]: mean (C) | std (C) | zero padding up to ld_stats.  masks == NULL: the unweighted
// form, mean and std(correction = 1) (pooling.py:101-104).
constexpr int POOLR_MAXT = 640;

__global__ __launch_bounds__(256) void k_stats_pool_rows(const float* __restrict__ feat, int T0, int Tp, int Cc,
                                                         int ld, const float* __restrict__ masks, int S, int Fm,
                                                         const int* __restrict__ idx, float* __restrict__ stats,
                                                         int ld_stats, const float* __restrict__ aff_scale,
                                                         const float* __restrict__ aff_shift) {
  __shared__ float ws[POOL_MAXS][POOLR_MAXT];
  __shared__ float v1s[POOL_MAXS], dens[POOL_MAXS];
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  for (int i = threadIdx.x; i < S * Tp; i += 256) {
    const int s = i / Tp, t = i % Tp;
    ws[s][t] = masks != nullptr ? masks[((long)b * S + s) * Fm + idx[t]] : 1.f;
  }
  __syncthreads();
  if (threadIdx.x < S) {
    const int s = threadIdx.x;
    float a = 0.f, q = 0.f;
    for (int t = 0; t < Tp; ++t) {
      a += ws[s][t];
      q += ws[s][t] * ws[s][t];
    }
    if (masks != nullptr) {
      const float v1 = a + 1e-8f;
      v1s[s] = v1;
      dens[s] = v1 - q / v1 + 1e-8f;
    } else {
      v1s[s] = (float)Tp;
      dens[s] = (float)(Tp - 1);   // 0 for a single frame: std = NaN, as torch.std(correction=1)
    }
  }
  __syncthreads();
  // padding columns of the output row (the embedding GEMM reads K rounded up to 32)
  for (int i = 2 * Cc + threadIdx.x + blockIdx.x * 256; i < ld_stats; i += 256 * gridDim.x)
    for (int s = 0; s < S; ++s) stats[((long)b * S + s) * ld_stats + i] = 0.f;
  if (c >= Cc) return;
  const float* x = feat + (((long)(b >> 4) * T0) * 16 + (b & 15)) * ld + c;
  const long ts = 16L * ld;
  // per-channel affine map applied on load (the eval-mode BatchNorm1d that precedes the pooling,
  // xvector.py:245: it cannot be folded past the pooling because an all-zero mask pools to 0, not to `shift`)
  const float sc = aff_scale != nullptr ? aff_scale[c] : 1.f, sh = aff_shift != nullptr ? aff_shift[c] : 0.f;
  float m[POOL_MAXS];
#pragma unroll
  for (int s = 0; s < POOL_MAXS; ++s) m[s] = 0.f;
  for (int t = 0; t < Tp; ++t) {
    const float xv = fmaf(x[t * ts], sc, sh);
#pragma unroll
    for (int s = 0; s < POOL_MAXS; ++s)
      if (s < S) m[s] = fmaf(xv, ws[s][t], m[s]);
  }
#pragma unroll
  for (int s = 0; s < POOL_MAXS; ++s)
    if (s < S) m[s] /= v1s[s];
  float v[POOL_MAXS];
#pragma unroll
  for (int s = 0; s < POOL_MAXS; ++s) v[s] = 0.f;
  for (int t = 0; t < Tp; ++t) {
    const float xv = fmaf(x[t * ts], sc, sh);
#pragma unroll
    for (int s = 0; s < POOL_MAXS; ++s)
      if (s < S) {
        const float d = xv - m[s];
        v[s] = fmaf(d * d, ws[s][t], v[s]);
      }
  }
#pragma unroll
  for (int s = 0; s < POOL_MAXS; ++s)
    if (s < S) {
      float* o = stats + ((long)b * S + s) * ld_stats;
      o[c] = m[s];
      o[Cc + c] = sqrtf(v[s] / dens[s]);
    }
}

// calibration of the Winograd paths (pa_emb_calibrate_winograd): out[0] = max(out[0], max |ref|),
// out[1] = max(out[1], max |got - ref|) over n floats.  Non-negative floats order like their bit patterns, so the
// cross-workgroup maximum is an integer atomicMax; NaN in `got` counts as +inf.
__global__ __launch_bounds__(256) void k_absmax_diff(const float* __restrict__ got, const float* __restrict__ ref,
                                                     long n, float* __restrict__ out) {
  float mr = 0.f, md = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float r = ref[i], d = fabsf(got[i] - r);
    mr = fmaxf(mr, fabsf(r));
    md = d == d ? fmaxf(md, d) : __builtin_inff();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mr = fmaxf(mr, __shfl_xor(mr, o, 64));
    md = fmaxf(md, __shfl_xor(md, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(mr));
    atomicMax(reinterpret_cast<unsigned int*>(out) + 1, __float_as_uint(md));
  }
}

}  // namespace pa

extern "C" {

int pa_stats_pool_rows(const float* feat, int B, int T0, int Tp, int C, int ld, const float* masks, int S,
                       int Fm, const int* nearest_idx, float* stats, int ld_stats, const float* aff_scale,
                       const float* aff_shift, void* stream) {
  if (B <= 0) return 0;
  PA_REQUIRE(S >= 1 && S <= pa::POOL_MAXS && Tp >= 1 && Tp <= pa::POOLR_MAXT && ld_stats >= 2 * C,
             "pa_stats_pool_rows: S <= %d, 1 <= T' <= %d and ld_stats >= 2 C required (got %d, %d)",
             pa::POOL_MAXS, pa::POOLR_MAXT, S, Tp);
  pa::ProfScope prof("k_stats_pool_rows", stream, 6.0 * B * S * C * Tp, 4.0 * B * C * Tp + 8.0 * B * S * C);
  hipLaunchKernelGGL(pa::k_stats_pool_rows, dim3(pa::cdiv(C, 256), B), dim3(256), 0, (hipStream_t)stream, feat,
                     T0, Tp, C, ld, masks, S, Fm, nearest_idx, stats, ld_stats, aff_scale, aff_shift);
  PA_CHECK_LAUNCH("pa_stats_pool_rows");
  return 0;
}

int pa_stats_pool(const float* feat, int B, int Fh, int Tp, int C, const float* masks, int S, int Fm,
                  const int* nearest_idx, float* stats, void* stream) {
  if (B <= 0) return 0;
  PA_REQUIRE(S >= 1 && S <= pa::POOL_MAXS && Tp <= pa::POOL_MAXT,
             "pa_stats_pool: S <= %d and T' <= %d required (got %d, %d)", pa::POOL_MAXS,
             pa::POOL_MAXT, S, Tp);
  pa::ProfScope prof("k_stats_pool", stream, 6.0 * B * S * C * Fh * Tp, 4.0 * B * C * Fh * Tp + 8.0 * B * S * C * Fh);
  hipLaunchKernelGGL(pa::k_stats_pool, dim3(pa::cdiv(C, 256), Fh, B), dim3(256), 0,
                     (hipStream_t)stream, feat, Fh, Tp, C, masks, S, Fm, nearest_idx, stats);
  PA_CHECK_LAUNCH("pa_stats_pool");
  return 0;
}

int pa_absmax_diff(const float* got, const float* ref, long n, float* out2, void* stream) {
  if (n <= 0) return 0;
  const int grid = pa::cdiv(n, 256 * 8) > 2048 ? 2048 : pa::cdiv(n, 256 * 8);
  hipLaunchKernelGGL(pa::k_absmax_diff, dim3(grid), dim3(256), 0, (hipStream_t)stream, got, ref, n, out2);
  PA_CHECK_LAUNCH("pa_absmax_diff");
  return 0;
}

}  // extern "C"
