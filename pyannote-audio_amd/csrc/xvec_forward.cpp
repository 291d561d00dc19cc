// pa_xvec_forward: XVectorSincNet (models/embedding/xvector.py:205-349) sequenced on one stream:
//   SincNet front end (the segmentation model's kernels: row stats, sinc FIR + pool, conv5 + pool x2,
//   last InstanceNorm + leaky_relu written as rows [(tile, t, b16)][64])
//   -> 5 TDNN layers = Conv1d(k, dilation d) + LeakyReLU + BatchNorm1d (xvector.py:232-247).  In the
//      (tile, t, b16) row order one time step is 16 rows, so tap j of a dilated convolution is the SAME
//      activation matrix shifted by 16 j d rows: a layer is k chained GEMMs C += A(shift j) W_j^T on
//      pa_gemm_tn_ex (bias with the first, LeakyReLU with the last); rows whose taps run past the end of
//      their chunk hold garbage and are never read (valid frames shrink 589 -> 585 -> 581 -> 575).
//      Each BatchNorm (eval: an affine map) is folded on the host into the NEXT layer's weights / bias; the
//      last one is applied on load inside the pooling kernel (an all-zero mask must pool to 0, not to its shift).
//   -> weighted statistics pooling over the valid frames for all masks of a chunk (k_stats_pool_rows)
//   -> Linear(3000 -> dimension).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pyannote_amd.h"

namespace pa {
void set_error(const char* fmt, ...);
}

namespace {

struct XvecPlan {
  int B, N, L1, P1, P2, T, ntiles, Tp, S, ldstats;
  long M;
  size_t wav_mean, wav_rstd, s1, st1m, st1r, s2, st2m, st2r, s3, st3m, st3r, x0, a0, a1, stats, total;
};
inline size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }
constexpr int SLACK_ROWS = 128;   // >= 16 * (k - 1) * d of every layer (96)

bool make_plan(const pa_xvec_weights* w, int B, int N, int S, XvecPlan* p) {
  p->B = B;
  p->N = N;
  p->S = S < 1 ? 1 : S;
  p->L1 = (N - 251) / w->sinc_stride + 1;
  if (N < 251 || p->L1 < 3) return false;
  p->P1 = p->L1 / 3;
  if (p->P1 < 5) return false;
  p->P2 = (p->P1 - 4) / 3;
  if (p->P2 < 5) return false;
  p->T = (p->P2 - 4) / 3;
  p->Tp = p->T;
  for (int l = 0; l < PA_XVEC_TDNN; ++l) p->Tp -= (w->tdnn_kernel[l] - 1) * w->tdnn_dilation[l];
  if (p->Tp < 1) return false;
  p->ntiles = (B + 15) / 16;
  p->M = (long)p->ntiles * p->T * 16;
  int cmax = 64;
  for (int l = 0; l < PA_XVEC_TDNN; ++l) cmax = w->tdnn_channels[l] > cmax ? w->tdnn_channels[l] : cmax;
  p->ldstats = (2 * w->tdnn_channels[PA_XVEC_TDNN - 1] + 31) & ~31;
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t r = o;
    o += align64(n);
    return r;
  };
  p->wav_mean = take(B);
  p->wav_rstd = take(B);
  p->s1 = take((size_t)B * 80 * p->P1);
  p->st1m = take((size_t)B * 80);
  p->st1r = take((size_t)B * 80);
  p->s2 = take((size_t)B * 60 * p->P2);
  p->st2m = take((size_t)B * 60);
  p->st2r = take((size_t)B * 60);
  p->s3 = take((size_t)B * 60 * p->T);
  p->st3m = take((size_t)B * 60);
  p->st3r = take((size_t)B * 60);
  p->x0 = take((size_t)(p->M + SLACK_ROWS) * 64);
  p->a0 = take((size_t)(p->M + SLACK_ROWS) * cmax);
  p->a1 = take((size_t)(p->M + SLACK_ROWS) * cmax);
  p->stats = take((size_t)B * p->S * p->ldstats);
  p->total = o;
  return true;
}

}  // namespace

extern "C" {

int pa_xvec_num_frames(const pa_xvec_weights* w, int num_samples) {
  XvecPlan p;
  return make_plan(w, 1, num_samples, 1, &p) ? p.Tp : 0;
}

size_t pa_xvec_workspace_bytes(const pa_xvec_weights* w, int num_chunks, int num_samples, int num_masks) {
  XvecPlan p;
  if (!make_plan(w, num_chunks, num_samples, num_masks, &p)) return 0;
  return p.total * sizeof(float);
}

int pa_xvec_forward(const pa_xvec_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                    int num_chunks, int num_samples, const float* masks, int num_masks, int mask_frames,
                    const int32_t* nearest_idx, float* emb, void* workspace, size_t workspace_bytes,
                    void* stream) {
  if (num_chunks <= 0) return 0;
  XvecPlan p;
  if (!make_plan(w, num_chunks, num_samples, masks ? num_masks : 1, &p)) {
    pa::set_error("pa_xvec_forward: %d samples leave no frame after SincNet + the TDNN stack", num_samples);
    return 3;
  }
  if (workspace_bytes < p.total * sizeof(float)) {
    pa::set_error("pa_xvec_forward: workspace too small (%zu < %zu bytes)", workspace_bytes,
                  p.total * sizeof(float));
    return 3;
  }
  float* ws = (float*)workspace;
  const int B = p.B;
  hipStream_t st = (hipStream_t)stream;
  int rc;
#define RUN(call)           \
  do {                      \
    rc = (call);            \
    if (rc != 0) return rc; \
  } while (0)

  // SincNet (models/blocks/sincnet.py:163-184), as in pa_seg_forward
  RUN(pa_row_stats(wav, chunk_stride, wav_len, B, p.N, 1e-5f, ws + p.wav_mean, ws + p.wav_rstd, stream));
  RUN(pa_sinc_fir_pool(wav, wav_len, chunk_stride, B, p.N, w->sinc_stride, ws + p.wav_mean, ws + p.wav_rstd,
                       w->wav_gamma, w->wav_beta, w->sinc_filt, ws + p.s1, stream));
  RUN(pa_row_stats(ws + p.s1, p.P1, (long)B * 80 * p.P1, B * 80, p.P1, 1e-5f, ws + p.st1m, ws + p.st1r, stream));
  RUN(pa_conv5_pool(ws + p.s1, B, 80, p.P1, ws + p.st1m, ws + p.st1r, w->norm0, w->norm0 + 80, w->conv1_w,
                    w->conv1_b, ws + p.s2, stream));
  RUN(pa_row_stats(ws + p.s2, p.P2, (long)B * 60 * p.P2, B * 60, p.P2, 1e-5f, ws + p.st2m, ws + p.st2r, stream));
  RUN(pa_conv5_pool(ws + p.s2, B, 60, p.P2, ws + p.st2m, ws + p.st2r, w->norm1, w->norm1 + 60, w->conv2_w,
                    w->conv2_b, ws + p.s3, stream));
  RUN(pa_row_stats(ws + p.s3, p.T, (long)B * 60 * p.T, B * 60, p.T, 1e-5f, ws + p.st3m, ws + p.st3r, stream));
  if (hipMemsetAsync(ws + p.x0 + (size_t)p.M * 64, 0, sizeof(float) * SLACK_ROWS * 64, st) != hipSuccess) return 1;
  RUN(pa_norm_transpose(ws + p.s3, B, p.T, ws + p.st3m, ws + p.st3r, w->norm2, w->norm2 + 60, ws + p.x0, stream));

  // TDNN stack
  const float* in = ws + p.x0;
  int cin = 64;
  float* buf[2] = {ws + p.a0, ws + p.a1};
  for (int l = 0; l < PA_XVEC_TDNN; ++l) {
    const int cout = w->tdnn_channels[l], k = w->tdnn_kernel[l], d = w->tdnn_dilation[l];
    float* out = buf[l & 1];
    if (hipMemsetAsync(out + (size_t)p.M * cout, 0, sizeof(float) * SLACK_ROWS * cout, st) != hipSuccess) return 1;
    for (int j = 0; j < k; ++j) {
      // tap j: rows shifted by j * d time steps = 16 j d rows; W_j = tdnn_w[l] + j * cout * cin
      RUN(pa_gemm_tn_ex(in + (size_t)16 * j * d * cin, cin, w->tdnn_w[l] + (size_t)j * cout * cin, cin,
                        j == 0 ? w->tdnn_b[l] : nullptr, j == 0 ? nullptr : out, out, cout, (int)p.M, cout, cin,
                        j == k - 1 ? 1 : 0, 0, stream));
    }
    in = out;
    cin = cout;
  }
  // statistics pooling over the Tp valid frames, for every mask of a chunk at once
  const int S = masks ? num_masks : 1;
  RUN(pa_stats_pool_rows(in, B, p.T, p.Tp, cin, cin, masks, S, mask_frames, nearest_idx, ws + p.stats,
                         p.ldstats, w->bn_scale, w->bn_shift, stream));
  // embedding Linear(2 C -> dimension) (xvector.py:250, 348); K padded to a multiple of 32 with zeros
  RUN(pa_gemm_tn_ex(ws + p.stats, p.ldstats, w->emb_w, p.ldstats, w->emb_b, nullptr, emb, w->dimension, B * S,
                    w->dimension, p.ldstats, 0, 0, stream));
#undef RUN
  return 0;
}

}  // extern "C"
