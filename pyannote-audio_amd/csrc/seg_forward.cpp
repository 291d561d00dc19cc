// pa_seg_forward: sequences the segmentation kernels on one stream out of a caller workspace.
// Replaces PyanNet.forward + hard Powerset conversion (PyanNet.py:211-240, powerset.py:115-140).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/pyannote_amd.h"

namespace pa {
void set_error(const char* fmt, ...);
}

namespace {

struct SegPlan {
  int B, N, L1, P1, P2, T, ntiles;
  long M;
  // offsets in floats
  size_t wav_mean, wav_rstd, s1, st1m, st1r, s2, st2m, st2r, s3, st3m, st3r, x0, xproj, h0, h1, l0, l1,
      total;
  // the sinc layer once per span of overlapping chunks (default; PA_SEG_SHARED_SINC=0 selects the per-chunk layer
  // for an A/B): raw filter outputs of the whole span + the tap sums
  long span, span_pos;
  size_t span_s, tap_sums;
};

// see seg_frontend.hip (k_sinc_fix_pool).  Measured on MI355X (round 4, one audio-hour = 3 591 chunks): k_sinc_fir_pool
// 21.2 ms -> k_sinc_fir_span 2.6 ms + k_sinc_fix_pool 3.2 ms, pipeline step 914.9 -> 903.7 ms; parity test
// tests/test_seg_gpu.py::test_shared_sinc_layer_matches_the_per_chunk_layer.
inline bool shared_sinc_wanted(const pa_seg_weights* w, int B, int N, int64_t chunk_stride) {
  const char* e = getenv("PA_SEG_SHARED_SINC");
  return (e == nullptr || atoi(e) != 0) && w->sinc_stride == 10 && B >= 2 && chunk_stride > 0 && chunk_stride < N &&
         chunk_stride % 10 == 0 && (int64_t)(B - 1) * chunk_stride + N <= 0x7fffffffLL;
}

inline size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }

bool make_plan(const pa_seg_weights* w, int B, int N, int64_t chunk_stride, SegPlan* p) {
  p->B = B;
  p->N = N;
  p->L1 = (N - 251) / w->sinc_stride + 1;
  if (N < 251 || p->L1 < 3) return false;
  p->P1 = p->L1 / 3;
  if (p->P1 < 5) return false;
  p->P2 = (p->P1 - 4) / 3;
  if (p->P2 < 5) return false;
  p->T = (p->P2 - 4) / 3;
  if (p->T < 1) return false;
  p->ntiles = (B + 15) / 16;
  p->M = (long)p->ntiles * p->T * 16;
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
  p->x0 = take((size_t)p->M * 64);
  {
    // gate pre-activations (ndir * 4H columns), two layer outputs (ndir * H), two head activations
    const size_t nd = w->lstm_bidir ? 2 : 1, Hh = (size_t)w->lstm_hidden;
    const size_t lw = w->num_linear > 0 ? (size_t)w->linear_hidden : 0;
    p->xproj = take((size_t)p->M * nd * 4 * Hh);
    p->h0 = take((size_t)p->M * nd * Hh);
    p->h1 = take((size_t)p->M * nd * Hh);
    p->l0 = take((size_t)p->M * lw);
    p->l1 = take((size_t)p->M * lw);
  }
  p->span = p->span_pos = 0;
  p->span_s = p->tap_sums = 0;
  if (shared_sinc_wanted(w, B, N, chunk_stride)) {
    p->span = (long)(B - 1) * chunk_stride + N;
    p->span_pos = (p->span - 251) / 10 + 1;
    p->span_s = take((size_t)80 * p->span_pos);
    p->tap_sums = take(80);
  }
  p->total = o;
  return true;
}

}  // namespace

extern "C" {

int pa_seg_num_frames(int num_samples, int sinc_stride) {
  int n = num_samples;
  const int ks[6] = {251, 3, 5, 3, 5, 3};
  const int ss[6] = {sinc_stride, 3, 1, 3, 1, 3};
  for (int i = 0; i < 6; ++i) {
    if (n < ks[i]) return 0;
    n = 1 + (n - ks[i]) / ss[i];
  }
  return n;
}

size_t pa_seg_workspace_bytes_strided(const pa_seg_weights* w, int num_chunks, int num_samples,
                                      int64_t chunk_stride) {
  SegPlan p;
  if (!make_plan(w, num_chunks, num_samples, chunk_stride, &p)) return 0;
  return p.total * sizeof(float);
}

size_t pa_seg_workspace_bytes(const pa_seg_weights* w, int num_chunks, int num_samples) {
  return pa_seg_workspace_bytes_strided(w, num_chunks, num_samples, num_samples);
}

int pa_seg_forward(const pa_seg_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                   int num_chunks, int num_samples, float* logp, uint8_t* multilabel, void* workspace,
                   size_t workspace_bytes, void* stream) {
  if (num_chunks <= 0) return 0;
  SegPlan p;
  if (!make_plan(w, num_chunks, num_samples, chunk_stride, &p)) {
    pa::set_error("pa_seg_forward: chunk of %d samples is too short for SincNet", num_samples);
    return 3;
  }
  if (p.span_pos > 0 && workspace_bytes < p.total * sizeof(float)) {
    // a workspace sized without the stride (pa_seg_workspace_bytes): the per-chunk sinc layer
    make_plan(w, num_chunks, num_samples, num_samples, &p);
  }
  if (w->lstm_hidden < 16 || w->lstm_hidden % 16 != 0 || w->lstm_hidden > 512 ||
      (!w->lstm_bidir && w->lstm_hidden % 32 != 0) || w->lstm_layers < 1 ||
      w->lstm_layers > PA_MAX_LSTM_LAYERS || w->num_linear > PA_MAX_LINEAR ||
      (w->num_linear > 0 && (w->linear_hidden < 32 || w->linear_hidden % 32 != 0))) {
    pa::set_error("pa_seg_forward: LSTM hidden size must be a multiple of 16 (32 when unidirectional) up to 512, "
                  "Linear widths multiples of 32 (got %d, %d)", w->lstm_hidden, w->linear_hidden);
    return 3;
  }
  if (workspace_bytes < p.total * sizeof(float)) {
    pa::set_error("pa_seg_forward: workspace too small (%zu < %zu bytes)", workspace_bytes,
                  p.total * sizeof(float));
    return 3;
  }
  float* ws = (float*)workspace;
  const int B = p.B;
  int rc;
#define RUN(call)          \
  do {                     \
    rc = (call);           \
    if (rc != 0) return rc; \
  } while (0)

  // SincNet (models/blocks/sincnet.py:163-184)
  RUN(pa_row_stats(wav, chunk_stride, wav_len, B, p.N, 1e-5f, ws + p.wav_mean, ws + p.wav_rstd, stream));
  if (p.span_pos > 0) {
    RUN(pa_sinc_fir_span(wav, wav_len, p.span, w->sinc_filt, ws + p.span_s, stream));
    RUN(pa_sinc_fix_pool(ws + p.span_s, p.span_pos, (int)(chunk_stride / 10), B, p.P1, ws + p.wav_mean,
                         ws + p.wav_rstd, w->wav_gamma, w->wav_beta, w->sinc_filt, ws + p.tap_sums, ws + p.s1,
                         stream));
  } else {
    RUN(pa_sinc_fir_pool(wav, wav_len, chunk_stride, B, p.N, w->sinc_stride, ws + p.wav_mean,
                         ws + p.wav_rstd, w->wav_gamma, w->wav_beta, w->sinc_filt, ws + p.s1, stream));
  }
  RUN(pa_row_stats(ws + p.s1, p.P1, (long)B * 80 * p.P1, B * 80, p.P1, 1e-5f, ws + p.st1m, ws + p.st1r,
                   stream));
  RUN(pa_conv5_pool(ws + p.s1, B, 80, p.P1, ws + p.st1m, ws + p.st1r, w->norm0, w->norm0 + 80,
                    w->conv1_w, w->conv1_b, ws + p.s2, stream));
  RUN(pa_row_stats(ws + p.s2, p.P2, (long)B * 60 * p.P2, B * 60, p.P2, 1e-5f, ws + p.st2m, ws + p.st2r,
                   stream));
  RUN(pa_conv5_pool(ws + p.s2, B, 60, p.P2, ws + p.st2m, ws + p.st2r, w->norm1, w->norm1 + 60,
                    w->conv2_w, w->conv2_b, ws + p.s3, stream));
  RUN(pa_row_stats(ws + p.s3, p.T, (long)B * 60 * p.T, B * 60, p.T, 1e-5f, ws + p.st3m, ws + p.st3r,
                   stream));
  RUN(pa_norm_transpose(ws + p.s3, B, p.T, ws + p.st3m, ws + p.st3r, w->norm2, w->norm2 + 60,
                        ws + p.x0, stream));

  // LSTM stack (PyanNet.py:226-234)
  const float* in = ws + p.x0;
  int kin = 64;
  float* hbuf[2] = {ws + p.h0, ws + p.h1};
  const int ndir = w->lstm_bidir ? 2 : 1, Hh = w->lstm_hidden;
  for (int l = 0; l < w->lstm_layers; ++l) {
    RUN(pa_gemm_tn(in, kin, w->lstm_wih[l], kin, w->lstm_bias[l], ws + p.xproj, 0, (int)p.M, ndir * 4 * Hh, kin,
                   0, 1, stream));
    RUN(pa_lstm_rec_h(ws + p.xproj, w->lstm_whh[l], hbuf[l & 1], p.ntiles, ndir, p.T, Hh, stream));
    in = hbuf[l & 1];
    kin = ndir * Hh;
  }
  // feed-forward head (PyanNet.py:236-240)
  float* lbuf[2] = {ws + p.l0, ws + p.l1};
  for (int l = 0; l < w->num_linear; ++l) {
    RUN(pa_gemm_tn(in, kin, w->lin_w[l], kin, w->lin_b[l], lbuf[l & 1], w->linear_hidden, (int)p.M,
                   w->linear_hidden, kin, 1, 0, stream));
    in = lbuf[l & 1];
    kin = w->linear_hidden;
  }
  RUN(pa_classifier(in, kin, kin, p.ntiles, p.T, B, w->cls_w, w->cls_b, w->num_classes,
                    w->powerset_map, w->num_speakers, logp, multilabel, stream));
#undef RUN
  return 0;
}

}  // extern "C"
