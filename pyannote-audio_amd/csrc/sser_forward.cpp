// pa_sser_forward: SSeRiouSS (models/segmentation/SSeRiouSS.py:289-328) sequenced on one stream:
//   wav2vec 2.0 / WavLM feature extractor + transformer encoder (torchaudio `extract_features`; kernels and the
//   row layout are described in w2v.hip) -> softmax-weighted mix of the layer outputs (or one layer)
//   -> the bi-LSTM stack, feed-forward head and classifier of the PyanNet path (seg_lstm.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pyannote_amd.h"

namespace pa {
void set_error(const char* fmt, ...);
}

// (internal launchers of w2v.hip: not part of the C ABI, hidden in the shared library -- csrc/common.h)
#define PA_INTERNAL __attribute__((visibility("hidden")))
extern "C" {
PA_INTERNAL int pa_w2v_conv0(const float* wav, long wav_len, long chunk_stride, int B, int N, int T, int P, int C, int K0,
                 int S0, const float* w, const float* bias, float* out, void* stream);
PA_INTERNAL int pa_w2v_group_norm_gelu(float* x, int B, int T, int P, int C, const float* gamma, const float* beta,
                           float* mean_scratch, float* rstd_scratch, void* stream);
PA_INTERNAL int pa_w2v_layernorm(const float* in, float* out, long rows, int C, const float* gamma, const float* beta,
                     int gelu, void* stream);
PA_INTERNAL int pa_w2v_posconv(const float* x, int B, int T, int P, int D, int groups, int KW, const float* w3,
                   const float* bias, float* out, void* stream);
PA_INTERNAL int pa_w2v_softmax(float* S, int B, int H, int T, int Tp, float scale, const float* bias, const float* xin, int P,
                   int D, const float* gate_w, const float* gate_b, const float* gate_const, void* stream);
PA_INTERNAL int pa_w2v_axpy(float* acc, const float* x, float w, long n, int first, void* stream);
PA_INTERNAL int pa_w2v_to_tiles(const float* x, int B, int T, int P, int D, float* out, void* stream);
int pa_gemm_tn_batched(const float* A, int lda, long sAo, long sAi, const float* W, int ldw, long sWo, long sWi,
                       const float* bias, float* C, long ldc, long sCo, long sCi, int M, int N, int K,
                       int outer, int inner, int act, void* stream);
}

namespace {

struct SserPlan {
  int B, N, nconv, T, P, Tp, ntiles;
  int Tl[PA_W2V_MAX_CONV], Pl[PA_W2V_MAX_CONV];
  long M, Ml;
  size_t stat_m, stat_r, c0, c1, x, x2, a, qk, vt, S, o, h, acc, x0, xproj, h0, h1, l0, l1, total;
};
inline size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }
constexpr int SLACK = 64;  // rows past the last chunk that a garbage row's window may touch

bool make_plan(const pa_sser_weights* w, int B, int N, SserPlan* p) {
  p->B = B;
  p->N = N;
  p->nconv = w->num_conv;
  int t = N;
  for (int l = 0; l < w->num_conv; ++l) {
    if (t < w->conv_kernel[l]) return false;
    t = (t - w->conv_kernel[l]) / w->conv_stride[l] + 1;
    p->Tl[l] = t;
  }
  p->T = t;
  // per-chunk row pitches with P_l = stride_{l+1} * P_{l+1} and P_l >= T_l
  int pitch = 1;
  for (int l = 0; l < w->num_conv; ++l) {
    long prod = 1;
    for (int j = l + 1; j < w->num_conv; ++j) prod *= w->conv_stride[j];
    const int need = (int)((p->Tl[l] + prod - 1) / prod);
    pitch = need > pitch ? need : pitch;
  }
  for (int l = w->num_conv - 1, q = pitch; l >= 0; --l) {
    p->Pl[l] = q;
    q *= w->conv_stride[l];
  }
  p->P = pitch;
  p->Tp = (p->T + 31) & ~31;
  p->M = (long)B * p->P;
  p->ntiles = (B + 15) / 16;
  p->Ml = (long)p->ntiles * p->T * 16;
  const int D = w->embed_dim, F = w->ff_dim, H = w->num_heads;
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t r = o;
    o += align64(n);
    return r;
  };
  size_t cmax = 0;
  for (int l = 0; l < w->num_conv; ++l) {
    const size_t n = ((size_t)B * p->Pl[l] + SLACK) * w->conv_channels[l];
    cmax = n > cmax ? n : cmax;
  }
  p->stat_m = take((size_t)B * w->conv_channels[0]);
  p->stat_r = take((size_t)B * w->conv_channels[0]);
  p->c0 = take(cmax);
  p->c1 = take(cmax);
  p->x = take((size_t)p->M * D);
  p->x2 = take((size_t)p->M * D);
  p->a = take((size_t)p->M * D);
  p->qk = take((size_t)p->M * 2 * D);
  p->vt = take((size_t)B * D * p->Tp);
  p->S = take((size_t)B * H * p->T * p->Tp);
  p->o = take((size_t)p->M * D);
  p->h = take((size_t)p->M * F);
  p->acc = take((size_t)p->M * D);
  p->x0 = take((size_t)p->Ml * D);
  {
    // gate pre-activations (ndir * 4H columns), two layer outputs (ndir * H), two head activations
    const size_t nd = w->lstm_bidir ? 2 : 1, Hh = (size_t)w->lstm_hidden;
    const size_t lw = w->num_linear > 0 ? (size_t)w->linear_hidden : 0;
    p->xproj = take((size_t)p->Ml * nd * 4 * Hh);
    p->h0 = take((size_t)p->Ml * nd * Hh);
    p->h1 = take((size_t)p->Ml * nd * Hh);
    p->l0 = take((size_t)p->Ml * lw);
    p->l1 = take((size_t)p->Ml * lw);
  }
  p->total = o;
  return true;
}

}  // namespace

extern "C" {

int pa_sser_num_frames(const pa_sser_weights* w, int num_samples) {
  SserPlan p;
  return make_plan(w, 1, num_samples, &p) ? p.T : 0;
}

size_t pa_sser_workspace_bytes(const pa_sser_weights* w, int num_chunks, int num_samples) {
  SserPlan p;
  if (!make_plan(w, num_chunks, num_samples, &p)) return 0;
  return p.total * sizeof(float);
}

int pa_sser_forward(const pa_sser_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                    int num_chunks, int num_samples, const float* rel_bias, float* logp, uint8_t* multilabel,
                    void* workspace, size_t workspace_bytes, void* stream) {
  if (num_chunks <= 0) return 0;
  SserPlan p;
  if (w->num_conv < 1 || w->num_conv > PA_W2V_MAX_CONV || w->num_layers < 1 || w->num_layers > PA_W2V_MAX_LAYERS ||
      !make_plan(w, num_chunks, num_samples, &p)) {
    pa::set_error("pa_sser_forward: bad configuration or a chunk of %d samples is too short", num_samples);
    return 3;
  }
  const int D = w->embed_dim, F = w->ff_dim, H = w->num_heads;
  if (D % H != 0 || (D / H) % 32 != 0 || D % 32 != 0 || F % 32 != 0 || (w->wavlm != 0) != (rel_bias != nullptr)) {
    pa::set_error("pa_sser_forward: embed_dim / ff_dim / head size must be multiples of 32; a WavLM encoder "
                  "needs its relative position table");
    return 3;
  }
  if (w->lstm_hidden < 16 || w->lstm_hidden % 16 != 0 || w->lstm_hidden > 512 ||
      (!w->lstm_bidir && w->lstm_hidden % 32 != 0) || w->lstm_layers < 1 ||
      w->lstm_layers > PA_MAX_LSTM_LAYERS || w->num_linear > PA_MAX_LINEAR ||
      (w->num_linear > 0 && (w->linear_hidden < 32 || w->linear_hidden % 32 != 0))) {
    pa::set_error("pa_sser_forward: LSTM hidden size must be a multiple of 16 (32 when unidirectional) up to 512, "
                  "Linear widths multiples of 32 (got %d, %d)", w->lstm_hidden, w->linear_hidden);
    return 3;
  }
  if (workspace_bytes < p.total * sizeof(float)) {
    pa::set_error("pa_sser_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, p.total * sizeof(float));
    return 3;
  }
  float* ws = (float*)workspace;
  const int B = p.B, T = p.T, P = p.P, Tp = p.Tp, hd = D / H;
  const int M = (int)p.M;
  hipStream_t st = (hipStream_t)stream;
  int rc;
#define RUN(call)           \
  do {                      \
    rc = (call);            \
    if (rc != 0) return rc; \
  } while (0)

  // ---- feature extractor (components.FeatureExtractor): conv -> [norm] -> gelu, 7 times
  float* cb[2] = {ws + p.c0, ws + p.c1};
  const int ln_mode = w->extractor_layer_norm;
  RUN(pa_w2v_conv0(wav, wav_len, chunk_stride, B, p.N, p.Tl[0], p.Pl[0], w->conv_channels[0], w->conv_kernel[0],
                   w->conv_stride[0], w->conv_w[0], w->conv_b[0], cb[0], stream));
  if (hipMemsetAsync(cb[0] + (size_t)B * p.Pl[0] * w->conv_channels[0], 0,
                     sizeof(float) * SLACK * w->conv_channels[0], st) != hipSuccess) return 1;
  if (ln_mode)
    RUN(pa_w2v_layernorm(cb[0], cb[0], (long)B * p.Pl[0], w->conv_channels[0], w->conv_norm_g[0], w->conv_norm_b[0],
                         1, stream));
  else
    RUN(pa_w2v_group_norm_gelu(cb[0], B, p.Tl[0], p.Pl[0], w->conv_channels[0], w->conv_norm_g[0],
                               w->conv_norm_b[0], ws + p.stat_m, ws + p.stat_r, stream));
  for (int l = 1; l < w->num_conv; ++l) {
    const int cin = w->conv_channels[l - 1], cout = w->conv_channels[l], k = w->conv_kernel[l], s = w->conv_stride[l];
    const float* in = cb[(l - 1) & 1];
    float* out = cb[l & 1];
    const long rows = (long)B * p.Pl[l];
    if (hipMemsetAsync(out + (size_t)rows * cout, 0, sizeof(float) * SLACK * cout, st) != hipSuccess) return 1;
    RUN(pa_gemm_tn_ex(in, s * cin, w->conv_w[l], k * cin, w->conv_b[l], nullptr, out, cout, (int)rows, cout, k * cin,
                      ln_mode ? 0 : 3, 0, stream));
    if (ln_mode) RUN(pa_w2v_layernorm(out, out, rows, cout, w->conv_norm_g[l], w->conv_norm_b[l], 1, stream));
  }
  const float* feat = cb[(w->num_conv - 1) & 1];
  const int C = w->conv_channels[w->num_conv - 1];

  // ---- encoder: feature projection, positional convolution (components.Encoder / Transformer._preprocess)
  float* x = ws + p.x;
  float* x2 = ws + p.x2;
  float* a = ws + p.a;
  RUN(pa_w2v_layernorm(feat, cb[w->num_conv & 1], M, C, w->proj_ln_g, w->proj_ln_b, 0, stream));
  RUN(pa_gemm_tn_ex(cb[w->num_conv & 1], C, w->proj_w, C, w->proj_b, nullptr, x2, D, M, D, C, 0, 0, stream));
  RUN(pa_w2v_posconv(x2, B, T, P, D, w->pos_groups, w->pos_kernel, w->pos_w, w->pos_b, x, stream));
  // the encoder-level LayerNorm precedes the layers of a POST-LN model (torchaudio's `_get_encoder` builds the
  // Transformer with `not layer_norm_first`; fairseq / HuggingFace agree -- tests/test_oracle_wav2vec2_pin.py).  In a
  // pre-LN model it follows the last layer in `forward`, which `extract_features` (the call of SSeRiouSS.py:289-296)
  // never reaches.
  if (!w->layer_norm_first) RUN(pa_w2v_layernorm(x, x, M, D, w->enc_ln_g, w->enc_ln_b, 0, stream));
  if (hipMemsetAsync(ws + p.vt, 0, sizeof(float) * (size_t)B * D * Tp, st) != hipSuccess) return 1;
  if (hipMemsetAsync(ws + p.o, 0, sizeof(float) * (size_t)M * D, st) != hipSuccess) return 1;

  // ---- transformer layers (components.EncoderLayer, SelfAttention / WavLMSelfAttention, FeedForward)
  const int use_layer = w->use_layer < 0 ? -1 : (w->use_layer < 1 ? 1 : w->use_layer);
  const float* result = nullptr;
  const float scale = 1.0f / sqrtf((float)hd);
  for (int l = 0; l < w->num_layers; ++l) {
    const pa_w2v_layer* L = &w->layers[l];
    const float* att_in = x;
    if (w->layer_norm_first) {
      RUN(pa_w2v_layernorm(x, a, M, D, L->ln1_g, L->ln1_b, 0, stream));
      att_in = a;
    }
    RUN(pa_gemm_tn_ex(att_in, D, L->qk_w, D, L->qk_b, nullptr, ws + p.qk, 2 * D, M, 2 * D, D, 0, 0, stream));
    // V^T per chunk: vt[b][d][t] = sum_c Wv[d][c] x[b][t][c]   (its bias is folded into the output projection)
    RUN(pa_gemm_tn_batched(L->v_w, D, 0, 0, att_in, D, (long)P * D, 0, nullptr, ws + p.vt, Tp, (long)D * Tp, 0, D, T,
                           D, B, 1, 0, stream));
    // S[b][h] = Q_bh K_bh^T
    RUN(pa_gemm_tn_batched(ws + p.qk, 2 * D, (long)P * 2 * D, hd, ws + p.qk + D, 2 * D, (long)P * 2 * D, hd, nullptr,
                           ws + p.S, Tp, (long)H * T * Tp, (long)T * Tp, T, T, hd, B, H, 0, stream));
    RUN(pa_w2v_softmax(ws + p.S, B, H, T, Tp, scale, rel_bias, att_in, P, D, L->gate_w, L->gate_b, L->gate_const,
                       stream));
    // O[b][:, h] = P_bh V_bh
    RUN(pa_gemm_tn_batched(ws + p.S, Tp, (long)H * T * Tp, (long)T * Tp, ws + p.vt, Tp, (long)D * Tp, (long)hd * Tp,
                           nullptr, ws + p.o, D, (long)P * D, hd, T, hd, Tp, B, H, 0, stream));
    // x1 = x + out_proj(O)
    RUN(pa_gemm_tn_ex(ws + p.o, D, L->out_w, D, L->out_b, x, x2, D, M, D, D, 0, 0, stream));
    if (w->layer_norm_first) {
      RUN(pa_w2v_layernorm(x2, a, M, D, L->ln2_g, L->ln2_b, 0, stream));
      RUN(pa_gemm_tn_ex(a, D, L->ff1_w, D, L->ff1_b, nullptr, ws + p.h, F, M, F, D, 3, 0, stream));
      RUN(pa_gemm_tn_ex(ws + p.h, F, L->ff2_w, F, L->ff2_b, x2, x, D, M, D, F, 0, 0, stream));
    } else {
      RUN(pa_w2v_layernorm(x2, a, M, D, L->ln1_g, L->ln1_b, 0, stream));
      RUN(pa_gemm_tn_ex(a, D, L->ff1_w, D, L->ff1_b, nullptr, ws + p.h, F, M, F, D, 3, 0, stream));
      RUN(pa_gemm_tn_ex(ws + p.h, F, L->ff2_w, F, L->ff2_b, a, x2, D, M, D, F, 0, 0, stream));
      RUN(pa_w2v_layernorm(x2, x, M, D, L->ln2_g, L->ln2_b, 0, stream));
    }
    if (use_layer < 0) {
      RUN(pa_w2v_axpy(ws + p.acc, x, w->layer_mix[l], (long)M * D, l == 0, stream));
      result = ws + p.acc;
    } else if (l + 1 == use_layer) {
      result = x;
      break;
    }
  }
  if (result == nullptr) result = x;

  // ---- LSTM stack + head (SSeRiouSS.py:315-328), as in pa_seg_forward
  RUN(pa_w2v_to_tiles(result, B, T, P, D, ws + p.x0, stream));
  const float* in = ws + p.x0;
  int kin = D;
  float* hbuf[2] = {ws + p.h0, ws + p.h1};
  const int ndir = w->lstm_bidir ? 2 : 1, Hh = w->lstm_hidden;
  for (int l = 0; l < w->lstm_layers; ++l) {
    RUN(pa_gemm_tn(in, kin, w->lstm_wih[l], kin, w->lstm_bias[l], ws + p.xproj, 0, (int)p.Ml, ndir * 4 * Hh, kin, 0,
                   1, stream));
    RUN(pa_lstm_rec_h(ws + p.xproj, w->lstm_whh[l], hbuf[l & 1], p.ntiles, ndir, T, Hh, stream));
    in = hbuf[l & 1];
    kin = ndir * Hh;
  }
  float* lbuf[2] = {ws + p.l0, ws + p.l1};
  for (int l = 0; l < w->num_linear; ++l) {
    RUN(pa_gemm_tn(in, kin, w->lin_w[l], kin, w->lin_b[l], lbuf[l & 1], w->linear_hidden, (int)p.Ml,
                   w->linear_hidden, kin, 1, 0, stream));
    in = lbuf[l & 1];
    kin = w->linear_hidden;
  }
  RUN(pa_classifier(in, kin, kin, p.ntiles, T, B, w->cls_w, w->cls_b, w->num_classes, w->powerset_map,
                    w->num_speakers, logp, multilabel, stream));
#undef RUN
  return 0;
}

}  // extern "C"
