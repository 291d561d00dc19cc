// hipcc-flags: -ffp-contract=off
// Frame-domain stages of the diarization pipeline on gfx950: everything between the per-chunk hard
// segmentations and the per-frame speaker decisions.  These are the reference's Python hot loops
// #2, #4 and #5: ~213 k global frames per audio-hour walked in the interpreter.
// All arithmetic here is on small integers (0/1 activities summed over <= ~10 overlapping chunks), so
// the float32 sums of the reference are reproduced exactly with int32 atomics in any order; the one
// float32 division of speaker counting is performed as the reference does (IEEE divide, rint-to-even).
//
//   k_chunk_stats      per (chunk, speaker): #active frames, #frames where it speaks alone
//                      (clustering.py:109-116 filter, speaker_diarization.py:385-391 overlap exclusion,
//                       :681-685 inactive speakers)
//   k_embedding_masks  mask selection of get_embeddings (speaker_diarization.py:375-427) -> (C,S,F) fp32
//   k_count_scatter / k_count_finish
//                      speaker_count = aggregate(sum_s seg, hamming=False, missing=0) -> rint -> uint8
//                      (pipelines/utils/diarization.py:150-185, core/inference.py:498-620)
//   k_cluster_scatter  reconstruct (speaker_diarization.py:480-528: per chunk max over the local speakers
//                      assigned to cluster k, NaN otherwise) + aggregate(skip_average=True) (overlap SUM)
//   k_topk_binarize    to_diarization (diarization.py:250-266): per frame the count[t] most active
//                      clusters.  The reference selects with np.argsort(-act), whose order among EQUAL
//                      activations is unspecified (numpy's default sort is not stable: on AVX-512/AVX2
//                      hosts it is a SIMD sorting network).  The kernel picks ties by lowest index and
//                      FLAGS every frame where a tie straddles the selection boundary; the caller
//                      re-decides exactly those frames with numpy's own argsort (frames.py), so the
//                      result equals the reference's on whatever host it runs.
//
// HBM-bound streaming kernels; the inputs are 6.3 MB (uint8 segmentations) per audio-hour.
#include "common.h"

namespace pa {

// grid = C, block = 256.  seg: (C, F, S) uint8, S <= 8.
__global__ __launch_bounds__(256) void k_chunk_stats(const uint8_t* __restrict__ seg, int F, int S,
                                                      int* __restrict__ active,
                                                      int* __restrict__ clean) {
  __shared__ int red[2][8][4];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint8_t* p = seg + (long)c * F * S;
  int a[8], q[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) a[s] = q[s] = 0;
  for (int f = tid; f < F; f += 256) {
    int v[8], tot = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      v[s] = s < S ? p[f * S + s] : 0;
      tot += v[s];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      a[s] += v[s];
      q[s] += (tot == 1) ? v[s] : 0;
    }
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a[s] += __shfl_xor(a[s], o, 64);
      q[s] += __shfl_xor(q[s], o, 64);
    }
    if (lane == 0) {
      red[0][s][w] = a[s];
      red[1][s][w] = q[s];
    }
  }
  __syncthreads();
  if (tid < S) {
    active[c * S + tid] = red[0][tid][0] + red[0][tid][1] + red[0][tid][2] + red[0][tid][3];
    clean[c * S + tid] = red[1][tid][0] + red[1][tid][1] + red[1][tid][2] + red[1][tid][3];
  }
}

// masks[c][s][f] = use_clean(c,s) ? seg*[sum_s seg < 2] : seg ;  use_clean = exclude && clean > min_frames
// grid = (ceil(F/256), C), block = 256
__global__ __launch_bounds__(256) void k_embedding_masks(const uint8_t* __restrict__ seg, int F, int S,
                                                          const int* __restrict__ clean,
                                                          int exclude_overlap, int min_num_frames,
                                                          float* __restrict__ masks) {
  const int c = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const uint8_t* p = seg + ((long)c * F + f) * S;
  int tot = 0;
  for (int s = 0; s < S; ++s) tot += p[s];
  for (int s = 0; s < S; ++s) {
    const bool use_clean = exclude_overlap && clean[c * S + s] > min_num_frames;
    const int v = p[s];
    masks[((long)c * S + s) * F + f] = (float)((use_clean && tot >= 2) ? 0 : v);
  }
}

// acc[2t] += sum_s seg[c][f][s], acc[2t+1] += 1  for t = start[c] + f.  grid = (ceil(F/256), C)
__global__ __launch_bounds__(256) void k_count_scatter(const uint8_t* __restrict__ seg, int F, int S,
                                                        const int* __restrict__ start, int T,
                                                        int* __restrict__ acc) {
  const int c = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const int t = start[c] + f;
  if (t < 0 || t >= T) return;
  const uint8_t* p = seg + ((long)c * F + f) * S;
  int tot = 0;
  for (int s = 0; s < S; ++s) tot += p[s];
  if (tot) atomicAdd(acc + 2 * t, tot);
  atomicAdd(acc + 2 * t + 1, 1);
}

__global__ __launch_bounds__(256) void k_count_finish(const int* __restrict__ acc, int T,
                                                       uint8_t* __restrict__ count) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int n = acc[2 * t + 1];
  // float32: aggregated / max(overlapping_chunk_count, 1e-12), missing -> 0, np.rint (half to even)
  const float avg = n > 0 ? __fdiv_rn((float)acc[2 * t], (float)n) : 0.f;
  count[t] = (uint8_t)rintf(avg);
}

// act[t][k] += max_{s : hard[c][s] == k} seg[c][f][s].  grid = (ceil(F/256), C)
__global__ __launch_bounds__(256) void k_cluster_scatter(const uint8_t* __restrict__ seg, int F, int S,
                                                          const int* __restrict__ start,
                                                          const int* __restrict__ hard, int K, int T,
                                                          int* __restrict__ act) {
  const int c = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const int t = start[c] + f;
  if (t < 0 || t >= T) return;
  const uint8_t* p = seg + ((long)c * F + f) * S;
  const int* h = hard + c * S;
  for (int s = 0; s < S; ++s) {
    const int k = h[s];
    if (k < 0 || k >= K || !p[s]) continue;
    // one contribution per (chunk, frame, cluster): only the first active local speaker of k adds
    bool first = true;
    for (int s2 = 0; s2 < s; ++s2) first = first && !(h[s2] == k && p[s2]);
    if (first) atomicAdd(act + (long)t * K + k, 1);
  }
}

// out[t][k] = 1 for the min(count[t], cap, K) largest act[t][.], ties -> lowest k.  One thread per frame.
// VT = int (hard {0,1} segmentations: sums of small integers) or float (soft scores of non-powerset models:
// activations are >= 0 sums of sigmoid scores, never NaN after aggregate(missing=0)).
template <typename VT>
struct TopkTop;
template <>
struct TopkTop<int> {
  static __device__ __forceinline__ int value() { return 0x7fffffff; }
};
template <>
struct TopkTop<float> {
  static __device__ __forceinline__ float value() { return __builtin_inff(); }
};

template <typename VT>
__global__ __launch_bounds__(256) void k_topk_binarize(const VT* __restrict__ act,
                                                        const uint8_t* __restrict__ count, int T, int K,
                                                        int cap, uint8_t* __restrict__ out,
                                                        uint8_t* __restrict__ tie) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const VT* a = act + (long)t * K;
  uint8_t* o = out + (long)t * K;
  for (int k = 0; k < K; ++k) o[k] = 0;
  int n = count[t];
  n = n < cap ? n : cap;
  n = n < K ? n : K;
  VT last_v = TopkTop<VT>::value();
  int last_k = -1;  // previously selected (value, index): next pick is "after" it
  for (int i = 0; i < n; ++i) {
    VT best_v = (VT)-1;
    int best_k = -1;
    for (int k = 0; k < K; ++k) {
      const VT v = a[k];
      const bool after = v < last_v || (v == last_v && k > last_k);
      if (after && v > best_v) {
        best_v = v;
        best_k = k;
      }
    }
    if (best_k < 0) break;
    o[best_k] = 1;
    last_v = best_v;
    last_k = best_k;
  }
  // ambiguous iff an unselected cluster has the same activation as the last selected one
  bool amb = false;
  if (n > 0 && n < K)
    for (int k = 0; k < K; ++k) amb = amb || (!o[k] && a[k] == last_v);
  tie[t] = amb ? 1 : 0;
}

// Hysteresis thresholding, replaces `binarize` (utils/signal.py:78-140) for non-powerset segmentation
// models (pipelines/speaker_diarization.py:599-606): per (chunk, class) a state machine over the frames --
// on where score > onset, off where score < offset, unchanged in between; NaN counts as 0 (nan_to_num).
// initial: 0 / 1 = given state, -1 = `scores[0] >= (onset + offset) / 2`.  One thread per (chunk, class).
__global__ __launch_bounds__(256) void k_hysteresis(const float* __restrict__ scores, long rows, int F, int K,
                                                     float onset, float offset, int initial,
                                                     uint8_t* __restrict__ out) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;  // r = c * K + k
  if (r >= rows) return;
  const long c = r / K;
  const int k = (int)(r % K);
  const float* p = scores + c * F * K + k;
  uint8_t* q = out + c * F * K + k;
  float s0 = p[0];
  s0 = s0 != s0 ? 0.f : s0;
  bool state = initial < 0 ? (s0 >= 0.5f * (onset + offset)) : (initial != 0);
  for (int f = 0; f < F; ++f) {
    float s = p[(long)f * K];
    s = s != s ? 0.f : s;
    if (s > onset) state = true;
    else if (s < offset) state = false;
    q[(long)f * K] = state ? 1 : 0;
  }
}

// clustered[c][f][k] = max over the local speakers s with hard[c][s] == k of scores[c][f][s], NaN when the
// chunk has none (pipelines/speaker_diarization.py:506-522) -- the soft-score form of k_cluster_scatter's
// first half; the overlap-add SUM is pa_aggregate(skip_average) on the result.
__global__ __launch_bounds__(256) void k_cluster_max(const float* __restrict__ scores, long CF, int F, int S,
                                                      const int* __restrict__ hard, int K,
                                                      float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (c, f)
  if (i >= CF) return;
  const long c = i / F;
  const float* p = scores + i * S;
  for (int k = 0; k < K; ++k) {
    float best = __builtin_nanf("");
    bool any = false;
    for (int s = 0; s < S; ++s)
      if (hard[c * S + s] == k) {
        const float v = p[s];
        // np.max propagates NaN; otherwise the larger value
        if (!any) best = v;
        else if (v != v || best != best) best = __builtin_nanf("");
        else best = v > best ? v : best;
        any = true;
      }
    out[i * K + k] = best;
  }
}

}  // namespace pa

extern "C" {

int pa_binarize_hysteresis(const float* scores, int C, int F, int K, float onset, float offset,
                           int initial_state, uint8_t* out, void* stream) {
  if (C <= 0 || F <= 0 || K <= 0) return 0;
  const long rows = (long)C * K;
  pa::ProfScope prof("k_hysteresis", stream, 2.0 * rows * F, 5.0 * rows * F);
  hipLaunchKernelGGL(pa::k_hysteresis, dim3(pa::cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, scores,
                     rows, F, K, onset, offset, initial_state, out);
  PA_CHECK_LAUNCH("pa_binarize_hysteresis");
  return 0;
}

int pa_cluster_max(const float* scores, int C, int F, int S, const int32_t* hard, int K, float* out,
                   void* stream) {
  if (C <= 0 || F <= 0 || K <= 0) return 0;
  const long CF = (long)C * F;
  pa::ProfScope prof("k_cluster_max", stream, 1.0 * CF * S * K, 4.0 * CF * (S + K));
  hipLaunchKernelGGL(pa::k_cluster_max, dim3(pa::cdiv(CF, 256)), dim3(256), 0, (hipStream_t)stream, scores, CF,
                     F, S, hard, K, out);
  PA_CHECK_LAUNCH("pa_cluster_max");
  return 0;
}

int pa_topk_binarize_f32(const float* act, const uint8_t* count, int T, int K, int cap, uint8_t* out,
                         uint8_t* tie, void* stream) {
  if (T <= 0 || K <= 0) return 0;
  pa::ProfScope prof("k_topk_binarize", stream, 1.0 * T * K, 5.0 * T * K + T);
  hipLaunchKernelGGL(pa::k_topk_binarize<float>, dim3(pa::cdiv(T, 256)), dim3(256), 0, (hipStream_t)stream,
                     act, count, T, K, cap, out, tie);
  PA_CHECK_LAUNCH("pa_topk_binarize_f32");
  return 0;
}

int pa_seg_chunk_stats(const uint8_t* seg, int C, int F, int S, int32_t* active, int32_t* clean,
                       void* stream) {
  if (C <= 0) return 0;
  PA_REQUIRE(S >= 1 && S <= 8, "pa_seg_chunk_stats: 1 <= S <= 8 required (got %d)", S);
  pa::ProfScope prof("k_chunk_stats", stream, 2.0 * C * F * S, (double)C * F * S + 8.0 * C * S);
  hipLaunchKernelGGL(pa::k_chunk_stats, dim3(C), dim3(256), 0, (hipStream_t)stream, seg, F, S, active,
                     clean);
  PA_CHECK_LAUNCH("pa_seg_chunk_stats");
  return 0;
}

int pa_embedding_masks(const uint8_t* seg, int C, int F, int S, const int32_t* clean,
                       int exclude_overlap, int min_num_frames, float* masks, void* stream) {
  if (C <= 0) return 0;
  pa::ProfScope prof("k_embedding_masks", stream, 1.0 * C * F * S, 5.0 * C * F * S);
  hipLaunchKernelGGL(pa::k_embedding_masks, dim3(pa::cdiv(F, 256), C), dim3(256), 0,
                     (hipStream_t)stream, seg, F, S, clean, exclude_overlap, min_num_frames, masks);
  PA_CHECK_LAUNCH("pa_embedding_masks");
  return 0;
}

int pa_speaker_count(const uint8_t* seg, int C, int F, int S, const int32_t* start_frame, int T,
                     uint8_t* count, int32_t* scratch, void* stream) {
  if (T <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  pa::ProfScope prof("k_speaker_count", stream, 1.0 * C * F * S, (double)C * F * S + 9.0 * T);
  if (hipMemsetAsync(scratch, 0, sizeof(int32_t) * 2 * (size_t)T, st) != hipSuccess) {
    pa::set_error("pa_speaker_count: memset failed");
    return 1;
  }
  if (C > 0)
    hipLaunchKernelGGL(pa::k_count_scatter, dim3(pa::cdiv(F, 256), C), dim3(256), 0, st, seg, F, S,
                       start_frame, T, scratch);
  hipLaunchKernelGGL(pa::k_count_finish, dim3(pa::cdiv(T, 256)), dim3(256), 0, st, scratch, T, count);
  PA_CHECK_LAUNCH("pa_speaker_count");
  return 0;
}

int pa_cluster_activations(const uint8_t* seg, int C, int F, int S, const int32_t* start_frame,
                           const int32_t* hard, int K, int T, int32_t* act, void* stream) {
  if (T <= 0 || K <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  pa::ProfScope prof("k_cluster_scatter", stream, 1.0 * C * F * S, (double)C * F * S + 4.0 * T * K);
  if (hipMemsetAsync(act, 0, sizeof(int32_t) * (size_t)T * K, st) != hipSuccess) {
    pa::set_error("pa_cluster_activations: memset failed");
    return 1;
  }
  if (C > 0)
    hipLaunchKernelGGL(pa::k_cluster_scatter, dim3(pa::cdiv(F, 256), C), dim3(256), 0, st, seg, F, S,
                       start_frame, hard, K, T, act);
  PA_CHECK_LAUNCH("pa_cluster_activations");
  return 0;
}

int pa_topk_binarize(const int32_t* act, const uint8_t* count, int T, int K, int cap, uint8_t* out,
                     uint8_t* tie, void* stream) {
  if (T <= 0 || K <= 0) return 0;
  pa::ProfScope prof("k_topk_binarize", stream, 1.0 * T * K, 5.0 * T * K + T);
  hipLaunchKernelGGL(pa::k_topk_binarize<int>, dim3(pa::cdiv(T, 256)), dim3(256), 0, (hipStream_t)stream, act,
                     count, T, K, cap, out, tie);
  PA_CHECK_LAUNCH("pa_topk_binarize");
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// General overlap-add aggregation (core/inference.py:498-620): float scores (NaN = missing), a
// per-frame weight (Hamming x warm-up), average or plain sum.  Gather form: one thread per (global
// frame t, class k) walks the chunks that cover t in ASCENDING chunk order and accumulates exactly like
// the reference's chunk loop does -- float32 accumulators, each addition performed in float64 and
// rounded back (numpy's `float32_array += float64_array`) -- so the result is bit-identical to it.
// ---------------------------------------------------------------------------------------------
namespace pa {

__global__ __launch_bounds__(256) void k_aggregate(const float* __restrict__ scores, int C, int F, int K,
                                                   const int* __restrict__ start, int T,
                                                   const double* __restrict__ window,
                                                   const double* __restrict__ warm, float epsilon,
                                                   float missing, int skip_average,
                                                   float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)T * K) return;
  const int t = (int)(idx / K), k = (int)(idx % K);
  // first chunk c with start[c] + F > t (start is non-decreasing)
  int lo = 0, hi = C;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (start[mid] + F > t) hi = mid;
    else lo = mid + 1;
  }
  float agg = 0.f, cnt = 0.f, msk = 0.f;
  for (int c = lo; c < C && start[c] <= t; ++c) {
    const int f = t - start[c];
    const float x = scores[((long)c * F + f) * K + k];
    const bool nan = x != x;
    const double m = nan ? 0.0 : 1.0;
    // ((score * mask) * hamming) * warm_up and ((mask * hamming) * warm_up), as numpy evaluates them
    agg = (float)((double)agg + (((double)(nan ? 0.f : x) * m) * window[f]) * warm[f]);
    cnt = (float)((double)cnt + (m * window[f]) * warm[f]);
    msk = fmaxf(msk, nan ? 0.f : 1.f);
  }
  float v = skip_average ? agg : agg / fmaxf(cnt, epsilon);
  if (msk == 0.f) v = missing;
  out[idx] = v;
}

}  // namespace pa

extern "C" {

int pa_aggregate(const float* scores, int C, int F, int K, const int32_t* start_frame, int T,
                 const double* window, const double* warm, float epsilon, float missing, int skip_average,
                 float* out, void* stream) {
  if (T <= 0 || K <= 0) return 0;
  pa::ProfScope prof("k_aggregate", stream, 3.0 * C * (double)F * K, 4.0 * ((double)C * F * K + (double)T * K));
  hipLaunchKernelGGL(pa::k_aggregate, dim3(pa::cdiv((long)T * K, 256)), dim3(256), 0, (hipStream_t)stream,
                     scores, C, F, K, start_frame, T, window, warm, epsilon, missing, skip_average, out);
  PA_CHECK_LAUNCH("pa_aggregate");
  return 0;
}

}  // extern "C"
