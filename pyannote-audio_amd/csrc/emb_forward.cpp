// pa_emb_forward: fbank -> ResNet34 -> weighted stats pooling -> Linear, sequenced on one stream.
// Replaces WeSpeakerResNet34.forward (wespeaker/__init__.py:324-343, resnet.py:399-430).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/pyannote_amd.h"

namespace pa {
void set_error(const char* fmt, ...);
}

namespace {

struct EmbPlan {
  int B, N, T, F, S;
  int Hs[5], Ws[5];  // spatial dims after stem (index 0) and after each layer
  size_t fbank, act[4], stats, total, act_elems;
  size_t t2_off, gather_off;  // Bottleneck: sub-regions of act[3] (t1 at 0)
};
inline size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }

// Winograd F(4x4,3x3) (pa_conv3x3_wino4: units of 4 x 64 output pixels) or F(2x2,3x3) (pa_conv3x3_wino: 8 x 32,
// 4 x 64 or 2 x 128 per workgroup) for an H x W map, both weight images being available.  Per USEFUL pixel F(4x4) is
// 1.02x / 1.12x / 1.29x as fast on the 64 / 128 / 256-channel layers (B = 512 launches on MI355X,
// profiles/r4_wino4_anatomy.txt: 3.09 vs 3.14, 2.58 vs 2.88 ms at equal padding, 2.77 vs 2.98 ms with 20 % more
// padding), but it pads to whole units -- on the narrow maps of short chunks (3 s: 149 / 75 / 38 columns) that can
// cost more than it gains.
inline bool prefer_wino4(int H, int W, int cin) {
  auto up = [](int a, int b) { return (long)((a + b - 1) / b) * b; };
  // (the F(4x4) launcher takes tile-linear units when that leaves at least 1.2x fewer units: they pad the map to whole
  //  4x4 tiles only, at ~1.2x the cost per unit -- emb_winograd4.hip: wino4_linear_wanted)
  long p4 = up(H, 4) * up(W, 64);
  const long p4_lin = up(H, 4) * up(W, 4);
  if (p4_lin * 120 <= p4 * 100) p4 = p4_lin * 120 / 100;
  long p2 = up(H, 8) * up(W, 32);
  if (up(H, 4) * up(W, 64) < p2) p2 = up(H, 4) * up(W, 64);
  if (up(H, 2) * up(W, 128) < p2) p2 = up(H, 2) * up(W, 128);
  // percent.  Measured at 3 s (profiles/r4_emb3s_dispatch.json): F(2x2) everywhere 844.7 ms per 10 000 segments,
  // F(4x4) on layer 4 only 815.9, on layers 2-4 799.8 -- F(4x4) also wins the 40 x 149 (1.20x the padded pixels)
  // and 20 x 75 (1.11x) maps: the F(2x2) kernel does not reach its large-map rate on them either.
  const long gain = cin >= 256 ? 130 : 122;
  return p4 * 100 <= p2 * gain;
}

// OFF by default: measured neutral on MI355X (profiles/r5_row_split_ab.txt: per audio-hour k_conv3x3_wino4 417.0 ->
// 395.7 ms, k_conv3x3_wino 161.5 -> 184.1 ms; step 793.8 / 802.2 vs 795.7 / 794.8 ms) -- the 2 x 128 tiles of the
// F(2x2) kernel run the two-row strips at 2.25 ms per launch, exactly what the F(4x4) kernel saves.
inline bool split_rows_wanted() {   // PA_EMB_SPLIT_ROWS=1 switches it on
  static const bool on = getenv("PA_EMB_SPLIT_ROWS") != nullptr && atoi(getenv("PA_EMB_SPLIT_ROWS")) != 0;
  return on;
}

bool make_plan(const pa_emb_weights* w, int B, int N, int S, EmbPlan* p, bool calib = false) {
  if (N < 400) return false;
  p->B = B;
  p->N = N;
  p->S = S < 1 ? 1 : S;
  p->T = 1 + (N - 400) / 160;
  p->F = w->num_mel;
  p->Hs[0] = p->F;
  p->Ws[0] = p->T;
  for (int l = 0; l < w->num_layers; ++l) {
    const int s = l == 0 ? 1 : 2;
    p->Hs[l + 1] = (p->Hs[l] - 1) / s + 1;
    p->Ws[l + 1] = (p->Ws[l] - 1) / s + 1;
  }
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t r = o;
    o += align64(n);
    return r;
  };
  p->fbank = take((size_t)B * p->T * p->F);
  // largest activation: the output of layer 1 (Hs[0] x Ws[0] pixels, planes[0] x expansion channels)
  const int ex = w->bottleneck ? 4 : 1;
  p->act_elems = (size_t)B * p->Hs[0] * p->Ws[0] * w->planes[0] * ex;
  // BasicBlock: 3 ping-pong buffers; Bottleneck: block input / output / shortcut at 4 x planes channels
  // (3 buffers) + the two planes-wide intermediates and the stride-2 gather (a 4th buffer, split in 3)
  for (int i = 0; i < 3; ++i) p->act[i] = take(p->act_elems);
  p->t2_off = p->gather_off = 0;
  if (w->bottleneck) {
    // sub-regions sized from the real block dimensions (odd maps: Ho = ceil(H / 2), so Ho * Wo > H * W / 4)
    size_t t1 = 0, t2 = 0, g = 0;
    int cin = w->planes[0];
    for (int l = 0; l < w->num_layers; ++l) {
      const size_t planes = w->planes[l];
      const size_t in_px = (size_t)B * p->Hs[l == 0 ? 1 : l] * p->Ws[l == 0 ? 1 : l];
      const size_t out_px = (size_t)B * p->Hs[l + 1] * p->Ws[l + 1];
      t1 = t1 > in_px * planes ? t1 : in_px * planes;  // first block of the layer runs its 1x1 at input size
      t2 = t2 > out_px * planes ? t2 : out_px * planes;
      if (l > 0) g = g > out_px * cin ? g : out_px * cin;
      cin = 4 * (int)planes;
    }
    p->t2_off = align64(t1);
    p->gather_off = p->t2_off + align64(t2);
    p->act[3] = take(p->gather_off + align64(g));
  } else if (calib) {
    p->act[3] = take(p->act_elems);   // where the Winograd result of a convolution waits for its comparison
  }
  const int L = w->num_layers;
  p->stats = take((size_t)B * p->S * 2 * w->planes[L - 1] * ex * p->Hs[L]);
  p->total = o;
  return true;
}

}  // namespace

extern "C" {

int pa_emb_num_fbank_frames(int num_samples) {
  return num_samples < 400 ? 0 : 1 + (num_samples - 400) / 160;
}

int pa_emb_num_pool_frames(const pa_emb_weights* w, int num_samples) {
  EmbPlan p;
  if (!make_plan(w, 1, num_samples, 1, &p)) return 0;
  return p.Ws[w->num_layers];
}

size_t pa_emb_workspace_bytes(const pa_emb_weights* w, int num_chunks, int num_samples, int num_masks) {
  EmbPlan p;
  if (!make_plan(w, num_chunks, num_samples, num_masks, &p)) return 0;
  return p.total * sizeof(float);
}

}  // extern "C"

extern "C" int pa_absmax_diff(const float* got, const float* ref, long n, float* out2, void* stream);   // emb_pool.hip

// `calib` (pa_emb_calibrate_winograd): every stride-1 3x3 convolution of a BasicBlock network runs through the
// DIRECT kernel -- whose result the following layers see -- and, beside it, through each Winograd image its block
// carries; calib[4 (2 blk + j) ...] = {max |direct|, max |F(4x4) - direct|, max |direct|, max |F(2x2) - direct|}.
static int emb_forward_impl(const pa_emb_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                            int num_chunks, int num_samples, const float* masks, int num_masks, int mask_frames,
                            const int32_t* nearest_idx, float* emb, void* workspace, size_t workspace_bytes,
                            void* stream, float* calib) {
  if (num_chunks <= 0) return 0;
  EmbPlan p;
  if (w->num_layers != 4 || !make_plan(w, num_chunks, num_samples, masks ? num_masks : 1, &p, calib != nullptr)) {
    pa::set_error("pa_emb_forward: %d samples is too short (fbank needs >= 400) or bad layer count",
                  num_samples);
    return 3;
  }
  if (workspace_bytes < p.total * sizeof(float)) {
    pa::set_error("pa_emb_forward: workspace too small (%zu < %zu bytes)", workspace_bytes,
                  p.total * sizeof(float));
    return 3;
  }
  float* ws = (float*)workspace;
  const int B = p.B;
  int rc;
#define RUN(call)           \
  do {                      \
    rc = (call);            \
    if (rc != 0) return rc; \
  } while (0)

  // fbank + centring: the mean over the whole chunk (fused behind the fbank kernel), or -- fbank_centering_span, a
  // checkpoint hyper-parameter -- the running mean of fb_center_kernel frames, out of place into the second
  // activation buffer (free until the first residual block), which the stem then reads
  const int span = w->fb_center_kernel;
  RUN(pa_fbank(wav, wav_len, chunk_stride, B, p.N, w->fb_window, w->fb_tw256, w->fb_tw512, w->fb_mel_w,
               w->fb_mel_lo, w->fb_mel_hi, w->num_mel, ws + p.fbank, span == 0 ? 1 : 0, stream));
  float* cur = ws + p.act[0];
  float* f1 = ws + p.act[1];
  float* f2 = ws + p.act[2];
  const float* feats = ws + p.fbank;
  if (span != 0) {
    RUN(pa_fbank_center_span(ws + p.fbank, B, p.T, p.F, span, f1, stream));
    feats = f1;
  }
  RUN(pa_resnet_stem(feats, B, p.T, p.F, w->stem_w, w->stem_shift, cur, stream));

  // a stride-1 3x3 convolution (+ shift, residual R, ReLU) of a BasicBlock: F(4x4) where it pays, else F(2x2), else
  // the direct kernel -- as far as the block carries the images (the numerical guard of EmbeddingPack removes them)
  auto conv_s1 = [&](const float* X, int H, int W, int ci, const float* v, const float* u, const float* wd,
                     const float* shift, const float* R, float* Y, int co, int slot) -> int {
    if (calib != nullptr) {
      float* scratch = ws + p.act[3];
      float* rep = calib + 4 * slot;
      const long n = (long)B * H * W * co;
      int r = pa_conv3x3(X, B, H, W, ci, wd, shift, R, Y, co, 1, 1, stream);
      if (r == 0 && v != nullptr) {
        r = pa_conv3x3_wino4(X, B, H, W, ci, v, shift, R, scratch, co, 1, stream);
        if (r == 0) r = pa_absmax_diff(scratch, Y, n, rep, stream);
      }
      if (r == 0 && u != nullptr) {
        r = pa_conv3x3_wino(X, B, H, W, ci, u, shift, R, scratch, co, 1, stream);
        if (r == 0) r = pa_absmax_diff(scratch, Y, n, rep + 2, stream);
      }
      return r;
    }
    if (v != nullptr && u != nullptr && H % 4 == 2 && H > 4 && split_rows_wanted() && prefer_wino4(H - 2, W, ci)) {
      // a map whose height is 2 (mod 4) -- the 10-row maps of layer 4 on 10 s chunks -- would pad its last tile row
      // half empty (12 rows of F(4x4) work for 10): F(4x4) on the rows above, the last two through the 2 x 128 tiles
      // of the F(2x2) kernel (per useful pixel 1.29x the F(4x4) cost at 256 channels: 8 + 2.6 instead of 12)
      const int r = pa_conv3x3_wino4_rows(X, B, H, W, ci, v, shift, R, Y, co, 1, H - 2, stream);
      return r != 0 ? r : pa_conv3x3_wino_rows(X, B, H, W, ci, u, shift, R, Y, co, 1, H - 2, stream);
    }
    if (v != nullptr && prefer_wino4(H, W, ci)) return pa_conv3x3_wino4(X, B, H, W, ci, v, shift, R, Y, co, 1, stream);
    if (u != nullptr) return pa_conv3x3_wino(X, B, H, W, ci, u, shift, R, Y, co, 1, stream);
    return pa_conv3x3(X, B, H, W, ci, wd, shift, R, Y, co, 1, 1, stream);
  };

  int blk = 0;
  int cin = w->planes[0];
  if (w->bottleneck) {
    // Bottleneck blocks (resnet.py:148-212): 1x1 (GEMM over the pixels of the NHWC map) -> 3x3 (Winograd /
    // strided direct kernel) -> 1x1 + shortcut + ReLU fused into the last GEMM's epilogue
    float* nxt = f1;                 // block output
    float* sc = f2;                  // shortcut branch
    // 4th buffer = t1 | t2 | stride-2 gather, each sized in make_plan from the largest block that uses it
    float* tmp = ws + p.act[3];
    for (int l = 0; l < w->num_layers; ++l) {
      const int planes = w->planes[l], cout = 4 * planes;
      for (int i = 0; i < w->num_blocks[l]; ++i, ++blk) {
        if (blk >= PA_MAX_RES_BLOCKS) {
          pa::set_error("pa_emb_forward: more than %d residual blocks", PA_MAX_RES_BLOCKS);
          return 3;
        }
        const int stride = (i == 0 && l > 0) ? 2 : 1;
        const int H = stride == 2 ? p.Hs[l] : p.Hs[l + 1], W = stride == 2 ? p.Ws[l] : p.Ws[l + 1];
        const int Ho = p.Hs[l + 1], Wo = p.Ws[l + 1];
        float* t1 = tmp;
        float* t2 = tmp + p.t2_off;
        RUN(pa_gemm_tn_ex(cur, cin, w->blk_w1[blk], cin, w->blk_shift1[blk], nullptr, t1, planes, B * H * W,
                          planes, cin, 2, 0, stream));
        if (stride == 1 && w->blk_u2[blk] != nullptr)
          RUN(pa_conv3x3_wino(t1, B, H, W, planes, w->blk_u2[blk], w->blk_shift2[blk], nullptr, t2, planes, 1,
                              stream));
        else
          RUN(pa_conv3x3(t1, B, H, W, planes, w->blk_w2[blk], w->blk_shift2[blk], nullptr, t2, planes, stride,
                         1, stream));
        const float* res = cur;      // identity shortcut
        if (w->blk_wsc[blk] != nullptr) {
          if (stride == 2)
            RUN(pa_gemm_tn_s2(cur, B, H, W, cin, w->blk_wsc[blk], cin, w->blk_shiftsc[blk], sc, cout, cout, stream));
          else
            RUN(pa_gemm_tn_ex(cur, cin, w->blk_wsc[blk], cin, w->blk_shiftsc[blk], nullptr, sc, cout,
                              B * Ho * Wo, cout, cin, 0, 0, stream));
          res = sc;
        } else if (stride != 1 || cin != cout) {
          pa::set_error("pa_emb_forward: block %d needs a shortcut conv but none was given", blk);
          return 3;
        }
        RUN(pa_gemm_tn_ex(t2, planes, w->blk_w3[blk], planes, w->blk_shift3[blk], res, nxt, cout, B * Ho * Wo,
                          cout, planes, 2, 0, stream));
        float* t = cur;
        cur = nxt;
        nxt = t;
        cin = cout;
      }
    }
  } else
  for (int l = 0; l < w->num_layers; ++l) {
    const int cout = w->planes[l];
    for (int i = 0; i < w->num_blocks[l]; ++i, ++blk) {
      if (blk >= PA_MAX_RES_BLOCKS) {
        pa::set_error("pa_emb_forward: more than %d residual blocks", PA_MAX_RES_BLOCKS);
        return 3;
      }
      const int stride = (i == 0 && l > 0) ? 2 : 1;
      const int H = stride == 2 ? p.Hs[l] : p.Hs[l + 1], W = stride == 2 ? p.Ws[l] : p.Ws[l + 1];
      const int Ho = p.Hs[l + 1], Wo = p.Ws[l + 1];
      if (w->blk_wsc[blk] != nullptr) {
        // out = relu(bn2(conv2(relu(bn1(conv1_s(x))))) + bn_sc(conv1x1_s(x)))   (resnet.py:140-145)
        RUN(pa_conv3x3(cur, B, H, W, cin, w->blk_w1[blk], w->blk_shift1[blk], nullptr, f1, cout, stride,
                       1, stream));
        const size_t q = (size_t)B * Ho * Wo * cin;
        float* R = f2 + ((q + 63) & ~(size_t)63);
        // (the 1x1 stride-2 shortcut reads its pixels in place: no gathered copy)
        if (stride == 2)
          RUN(pa_gemm_tn_s2(cur, B, H, W, cin, w->blk_wsc[blk], cin, w->blk_shiftsc[blk], R, cout, cout, stream));
        else
          RUN(pa_gemm_tn(cur, cin, w->blk_wsc[blk], cin, w->blk_shiftsc[blk], R, cout, B * Ho * Wo, cout, cin, 0, 0,
                         stream));
        RUN(conv_s1(f1, Ho, Wo, cout, w->blk_v2[blk], w->blk_u2[blk], w->blk_w2[blk], w->blk_shift2[blk], R, cur, cout,
                    2 * blk + 1));
      } else {
        if (stride != 1 || cin != cout) {
          pa::set_error("pa_emb_forward: block %d needs a shortcut conv but none was given", blk);
          return 3;
        }
        RUN(conv_s1(cur, H, W, cin, w->blk_v1[blk], w->blk_u1[blk], w->blk_w1[blk], w->blk_shift1[blk], nullptr, f1,
                    cout, 2 * blk));
        RUN(conv_s1(f1, H, W, cout, w->blk_v2[blk], w->blk_u2[blk], w->blk_w2[blk], w->blk_shift2[blk], cur, f2, cout,
                    2 * blk + 1));
        float* t = cur;
        cur = f2;
        f2 = t;
      }
      cin = cout;
    }
  }
  const int L = w->num_layers;
  const int S = p.S;
  const int cfin = w->planes[L - 1] * (w->bottleneck ? 4 : 1);
  RUN(pa_stats_pool(cur, B, p.Hs[L], p.Ws[L], cfin, masks, S, mask_frames, nearest_idx, ws + p.stats, stream));
  const int D2 = 2 * cfin * p.Hs[L];
  RUN(pa_gemm_tn(ws + p.stats, D2, w->seg1_w, D2, w->seg1_b, emb, w->embed_dim, B * S, w->embed_dim, D2,
                 0, 0, stream));
#undef RUN
  return 0;
}

extern "C" {

int pa_emb_forward(const pa_emb_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                   int num_chunks, int num_samples, const float* masks, int num_masks, int mask_frames,
                   const int32_t* nearest_idx, float* emb, void* workspace, size_t workspace_bytes,
                   void* stream) {
  return emb_forward_impl(w, wav, wav_len, chunk_stride, num_chunks, num_samples, masks, num_masks, mask_frames,
                          nearest_idx, emb, workspace, workspace_bytes, stream, nullptr);
}

size_t pa_emb_calibrate_workspace_bytes(const pa_emb_weights* w, int num_chunks, int num_samples) {
  EmbPlan p;
  if (!make_plan(w, num_chunks, num_samples, 1, &p, true)) return 0;
  return p.total * sizeof(float);
}

int pa_emb_calibrate_winograd(const pa_emb_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                              int num_chunks, int num_samples, float* report, float* emb, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (w->bottleneck) {
    pa::set_error("pa_emb_calibrate_winograd: BasicBlock networks only (Bottleneck blocks run F(2x2) / direct)");
    return 3;
  }
  if (hipMemsetAsync(report, 0, sizeof(float) * 8 * PA_MAX_RES_BLOCKS, (hipStream_t)stream) != hipSuccess) return 1;
  return emb_forward_impl(w, wav, wav_len, chunk_stride, num_chunks, num_samples, nullptr, 1, 0, nullptr, emb,
                          workspace, workspace_bytes, stream, report);
}

}  // extern "C"
