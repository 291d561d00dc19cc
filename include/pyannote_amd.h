/* pyannote_amd.h -- C ABI of libpyannote_amd.so (gfx950 / MI355X kernels for the
 * speaker-diarization-3.1 hot path).
 *
 * The reference (pyannote.audio 4.0.x) is 100 % Python and has NO C/FFI boundary of its own
 * (SURVEY.md section 8b); the boundary below is what the reference's Python objects would bind with
 * `ctypes` (see INTEGRATION.md for the stubs).  Each group cites the reference interface it
 * replaces, relative to src/pyannote/audio/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless its name ends in `_host` or it is uint8;
 *     (`tensor.data_ptr()` of a contiguous torch-ROCm tensor is what callers pass)
 *   - `stream` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); nothing synchronises
 *   - no hidden allocation: scratch comes from the caller (`*_workspace_bytes`)
 *   - return 0 = OK, 1 = launch/runtime failure, 2 = out of memory, 3 = invalid argument;
 *     pa_last_error() returns a thread-local message.  Python maps 2 -> MemoryError, which is the
 *     convention of Inference.infer (core/inference.py:199-208).
 *   - all arithmetic is IEEE fp32 (f32-input MFMA; the reference disables TF32,
 *     utils/reproducibility.py:68-73); the clustering distance kernels are fp64 like SciPy.
 */
#ifndef PYANNOTE_AMD_H
#define PYANNOTE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int pa_version(void);
const char* pa_last_error(void);

/* Built-in kernel profiler (measurement aid, no reference counterpart): while enabled every launcher
 * brackets its kernel with two hipEvents on the launch stream.  pa_prof_report synchronises and writes
 * a JSON object {"kernel": {"launches", "ms", "flops", "bytes"}} (algorithmic flops/bytes) into buf,
 * clears the records and returns the size needed. */
void pa_prof_enable(int on);
size_t pa_prof_report(char* buf, size_t cap);

/* ------------------------------------------------------------------------------------------
 * Segmentation model: replaces PyanNet.forward (models/segmentation/PyanNet.py:211-240) +
 * Powerset.to_multilabel(hard) (utils/powerset.py:115-140) as called from Inference.infer
 * (core/inference.py:182-215).
 * ---------------------------------------------------------------------------------------- */
#define PA_MAX_LSTM_LAYERS 8
#define PA_MAX_LINEAR 4

typedef struct pa_seg_weights {
  int32_t sinc_stride;   /* 10 */
  int32_t lstm_layers;   /* L */
  int32_t lstm_hidden;   /* multiple of 16 (32 when unidirectional), <= 512; 128 + bidirectional = register-resident kernel */
  int32_t lstm_bidir;    /* 1 / 0 */
  int32_t num_linear;    /* 0..PA_MAX_LINEAR */
  int32_t linear_hidden; /* multiple of 32 */
  int32_t num_classes;   /* powerset classes (7) */
  int32_t num_speakers;  /* multilabel width (3) */
  float wav_gamma, wav_beta;  /* sincnet.wav_norm1d.{weight,bias} */
  const float* sinc_filt; /* [5][63][64]  MFMA B image of the 80x251 sinc taps (tap 251 = 0) */
  const float* norm0;     /* [2][80] gamma | beta  (sincnet.norm1d.0) */
  const float* conv1_w;   /* [4][100][64] MFMA B image of sincnet.conv1d.1.weight (60,80,5) */
  const float* conv1_b;   /* [64] (60 real) */
  const float* norm1;     /* [2][60] */
  const float* conv2_w;   /* [4][75][64]  image of sincnet.conv1d.2.weight (60,60,5) */
  const float* conv2_b;   /* [64] */
  const float* norm2;     /* [2][60] */
  const float* lstm_wih[PA_MAX_LSTM_LAYERS];  /* [ndir * 4H][Kin] rows permuted (pa_lstm_rec / pa_lstm_rec_h column
                                               * order), Kin = 64 (layer 0) / ndir * H */
  const float* lstm_bias[PA_MAX_LSTM_LAYERS]; /* [ndir * 4H] b_ih + b_hh, permuted */
  const float* lstm_whh[PA_MAX_LSTM_LAYERS];  /* MFMA B image of weight_hh: [2][4][8][32][64] (H = 128, bidirectional)
                                               * or the layout of pa_lstm_rec_h */
  const float* lin_w[PA_MAX_LINEAR];          /* [out][in] as torch */
  const float* lin_b[PA_MAX_LINEAR];
  const float* cls_w;                         /* [num_classes][in] */
  const float* cls_b;
  const uint8_t* powerset_map;                /* [num_classes][num_speakers] 0/1 */
} pa_seg_weights;

/* frames per chunk for `num_samples` (SincNet.num_frames, models/blocks/sincnet.py:82-107) */
int pa_seg_num_frames(int num_samples, int sinc_stride);
size_t pa_seg_workspace_bytes(const pa_seg_weights* w, int num_chunks, int num_samples);
/* The same for callers that know the chunk stride: identical to pa_seg_workspace_bytes unless the
 * shared sinc layer applies (overlapping chunks whose stride is a multiple of 10 samples; PA_SEG_SHARED_SINC=0
 * in the environment switches it off), which needs 320 B per span position more.  pa_seg_forward falls back to the per-chunk
 * layer when the workspace it is given is the smaller one. */
size_t pa_seg_workspace_bytes_strided(const pa_seg_weights* w, int num_chunks, int num_samples,
                                      int64_t chunk_stride);
/* Chunk b is wav[b*chunk_stride : b*chunk_stride + num_samples], zero beyond wav_len
 * (Inference.slide's unfold + zero-padded last chunk, core/inference.py:261-278).
 * logp: (num_chunks, F, num_classes) log-probabilities or NULL;
 * multilabel: (num_chunks, F, num_speakers) uint8 {0,1} or NULL. */
int pa_seg_forward(const pa_seg_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                   int num_chunks, int num_samples, float* logp, uint8_t* multilabel, void* workspace,
                   size_t workspace_bytes, void* stream);

/* building blocks of pa_seg_forward (exported for unit parity tests) */
int pa_row_stats(const float* x, long row_stride, long total_len, int rows, int len, float eps,
                 float* mean, float* rstd, void* stream);
int pa_sinc_fir_pool(const float* wav, long wav_len, long chunk_stride, int B, int N, int stride,
                     const float* mean, const float* rstd, float gamma, float beta,
                     const float* filt_packed, float* out, void* stream);
/* The sinc layer once per span of overlapping chunks (what pa_seg_forward runs when the chunk stride allows it).  pa_sinc_fir_span: S (80, Pc) = raw filter outputs of wav[0, span) (zeros past wav_len),
 * Pc = (span - 251) / 10 + 1.  pa_sinc_fix_pool: chunk b starts `positions_per_chunk_step` positions after chunk
 * b - 1; per-chunk affine fix-up of the waveform InstanceNorm, magnitude, maxpool3 -> out (B, 80, P) exactly what
 * pa_sinc_fir_pool writes (up to float rounding: 1e-6 of the peak); tap_sums: 80 floats of scratch. */
int pa_sinc_fir_span(const float* wav, long wav_len, long span, const float* filt_packed, float* S, void* stream);
int pa_sinc_fix_pool(const float* S, long Pc, int positions_per_chunk_step, int B, int P, const float* mean,
                     const float* rstd, float gamma, float beta, const float* filt_packed, float* tap_sums,
                     float* out, void* stream);
int pa_conv5_pool(const float* xin, int B, int cin, int Lin, const float* in_mean,
                  const float* in_rstd, const float* gam, const float* bet, const float* w_packed,
                  const float* bias64, float* out, void* stream);
int pa_norm_transpose(const float* xin, int B, int T, const float* in_mean, const float* in_rstd,
                      const float* gam, const float* bet, float* X0, void* stream);
/* + act 2 = ReLU and an optional residual laid out like C (out_mode 0): C = act(A W^T + bias + Res) */
int pa_gemm_tn_ex(const float* A, int lda, const float* W, int ldw, const float* bias, const float* Res,
                  float* C, long ldc, int M, int N, int K, int act, int out_mode, void* stream);
int pa_gemm_tn(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, long ldc,
               int M, int N, int K, int act, int out_mode, void* stream);
/* the 1x1 stride-2 shortcut convolution + BatchNorm of a BasicBlock / Bottleneck (resnet.py:109-118) as the same GEMM
 * reading pixel (2y, 2x) of the NHWC map X (B, H, W, cin) in place: C[(b, y, x)][n] = sum_c X[b][2y][2x][c] W[n][c]
 * + bias[n]; cin % 32 == 0 */
int pa_gemm_tn_s2(const float* X, int B, int H, int W, int cin, const float* Wt, int ldw, const float* bias, float* C,
                  long ldc, int N, void* stream);
int pa_lstm_rec(const float* xproj, const float* whh_packed, float* out, int ntiles, int ndir, int T,
                void* stream);
/* the recurrence for any hidden size H (multiple of 16, <= 512) and ndir in {1, 2} (PyanNet.py:64-72 accepts any
 * nn.LSTM configuration).  H == 128 && ndir == 2: pa_lstm_rec with its operand layouts.  Otherwise
 *   xproj : [tile][t][ndir * 4H][16], column dir * 4H + (4 u + q) * 16 + n  <->  torch gate row q H + 16 u + n (q: i,f,g,o)
 *   whh   : [dir][u][q][k4][lane][j] = weight_hh[q H + 16 u + (lane & 15)][16 k4 + 4 j + (lane >> 4)]
 *   out   : [m][ndir * H], m = (tile * T + t) * 16 + b16, columns dir * H + j */
int pa_lstm_rec_h(const float* xproj, const float* whh_packed, float* out, int ntiles, int ndir, int T, int H,
                  void* stream);
int pa_classifier(const float* X, int ldx, int K, int ntiles, int T, int B, const float* cw,
                  const float* cb, int NC, const unsigned char* mapping, int S, float* logp,
                  unsigned char* multilabel, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embedding model: replaces WeSpeakerResNet34.forward (models/embedding/wespeaker/__init__.py:
 * 324-343: compute_fbank :113-139 -> ResNet.forward resnet.py:399-430 -> TSTP/StatsPool
 * resnet.py:49-66, blocks/pooling.py:30-130 -> seg_1) as called from
 * PyannoteAudioPretrainedSpeakerEmbedding.__call__ (pipelines/speaker_verification.py:704-716).
 * The backbone runs once per chunk and is pooled for all S masks (forward_frames /
 * forward_embedding split, wespeaker/__init__.py:288-322).
 * ---------------------------------------------------------------------------------------- */
#define PA_MAX_RES_BLOCKS 128   /* ResNet293: 10 + 20 + 64 + 3 = 97 blocks */

typedef struct pa_emb_weights {
  int32_t num_mel;       /* 80 */
  int32_t embed_dim;     /* 256 */
  int32_t num_layers;    /* 4 */
  int32_t num_blocks[4]; /* 3,4,6,3 (ResNet34); 3,8,36,3 / 6,16,48,3 / 10,20,64,3 (ResNet152/221/293) */
  int32_t planes[4];     /* 32,64,128,256 */
  int32_t bottleneck;    /* 0: BasicBlock (resnet.py:84-145); 1: Bottleneck, expansion 4 (resnet.py:148-212) */
  /* fbank tables */
  const float* fb_window;   /* [400] hamming (periodic=False) */
  const float* fb_tw256;    /* [256][2] exp(-2 pi i m/256) */
  const float* fb_tw512;    /* [257][2] exp(-2 pi i k/512) */
  const float* fb_mel_w;    /* [num_mel][257] */
  const int32_t* fb_mel_lo; /* [num_mel] first / last FFT bin with non-zero weight */
  const int32_t* fb_mel_hi;
  /* BatchNorm folded: w *= gamma/sqrt(var+eps), shift = beta - mean*gamma/sqrt(var+eps) */
  const float* stem_w;     /* [9][32]  (tap = 3*dmel + dtime) */
  const float* stem_shift; /* [32] */
  const float* blk_w1[PA_MAX_RES_BLOCKS];     /* [9][cout][cin] */
  const float* blk_shift1[PA_MAX_RES_BLOCKS]; /* [cout] */
  const float* blk_w2[PA_MAX_RES_BLOCKS];     /* [9][cout][cout] */
  const float* blk_shift2[PA_MAX_RES_BLOCKS];
  const float* blk_u1[PA_MAX_RES_BLOCKS];     /* [16][cout][cin] Winograd F(2x2,3x3) image G g G^T of w1, or NULL */
  const float* blk_u2[PA_MAX_RES_BLOCKS];     /* same for w2; used for stride-1 convolutions when not NULL */
  const float* blk_wsc[PA_MAX_RES_BLOCKS];    /* [cout][cin] 1x1 shortcut (stride 2, or stride 1 for the first Bottleneck) or NULL */
  const float* blk_shiftsc[PA_MAX_RES_BLOCKS];
  /* Bottleneck only: w1 = [planes][cin] 1x1, w2 / u2 = the 3x3 (stride s), w3 = [4 planes][planes] 1x1 */
  const float* blk_w3[PA_MAX_RES_BLOCKS];
  const float* blk_shift3[PA_MAX_RES_BLOCKS];
  const float* seg1_w; /* [embed_dim][2 * expansion * planes[3] * num_mel/8] */
  const float* seg1_b;
  /* Winograd F(4x4,3x3) images of w1 / w2 (pa_winograd4_pack_host), or NULL: when set, a stride-1 3x3 convolution
   * runs through pa_conv3x3_wino4 instead of pa_conv3x3_wino (blk_u*) / pa_conv3x3 (blk_w*) */
  const float* blk_v1[PA_MAX_RES_BLOCKS];
  const float* blk_v2[PA_MAX_RES_BLOCKS];
  /* fbank centring (wespeaker/__init__.py:137-157): 0 = subtract the mean over all frames of the chunk
   * (fbank_centering_span=None); odd K >= 1 = subtract the running mean of K frames (pa_fbank_center_span) */
  int32_t fb_center_kernel;
} pa_emb_weights;

/* fbank frames for num_samples (25 ms / 10 ms, snip_edges) and frames after the 3 stride-2 stages */
int pa_emb_num_fbank_frames(int num_samples);
int pa_emb_num_pool_frames(const pa_emb_weights* w, int num_samples);
size_t pa_emb_workspace_bytes(const pa_emb_weights* w, int num_chunks, int num_samples, int num_masks);
/* chunk b = wav[b*chunk_stride : +num_samples] (zero past wav_len);
 * masks: (num_chunks, S, mask_frames) fp32 or NULL (S = 1, unweighted);
 * nearest_idx: (pool_frames) int32 source index of F.interpolate(mode="nearest"), ignored if !masks;
 * emb: (num_chunks, S, embed_dim). */
int pa_emb_forward(const pa_emb_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                   int num_chunks, int num_samples, const float* masks, int num_masks, int mask_frames,
                   const int32_t* nearest_idx, float* emb, void* workspace, size_t workspace_bytes,
                   void* stream);

/* Numerical guard of the Winograd paths (weights.EmbeddingPack runs it once per loaded checkpoint; the reference has
 * no counterpart: its convolutions are torch's direct fp32 ones, resnet.py:92-107).  The chunks go through a
 * BasicBlock network in which every stride-1 3x3 convolution is evaluated by the DIRECT kernel (whose output feeds
 * the next layer) and, beside it, by every Winograd image the block carries.
 * report (device, 8 * PA_MAX_RES_BLOCKS floats, zeroed by the call): for block b, convolution j in {0, 1}:
 *   report[4 (2 b + j) + 0] = max |direct|,  [+1] = max |F(4x4) - direct|   (both 0: no F(4x4) image)
 *   report[4 (2 b + j) + 2] = max |direct|,  [+3] = max |F(2x2) - direct|   (both 0: no F(2x2) image)
 * over the whole output map (after shift, residual and ReLU).  emb: (num_chunks, embed_dim) from the direct path. */
size_t pa_emb_calibrate_workspace_bytes(const pa_emb_weights* w, int num_chunks, int num_samples);
int pa_emb_calibrate_winograd(const pa_emb_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                              int num_chunks, int num_samples, float* report, float* emb, void* workspace,
                              size_t workspace_bytes, void* stream);
/* out2[0] = max(out2[0], max |ref|), out2[1] = max(out2[1], max |got - ref|) over n floats (out2 >= 0 on entry) */
int pa_absmax_diff(const float* got, const float* ref, long n, float* out2, void* stream);

int pa_fbank(const float* wav, long wav_len, long chunk_stride, int B, int N, const float* window,
             const float* tw256, const float* tw512, const float* mel_w, const int* mel_lo,
             const int* mel_hi, int nmel, float* out, int center, void* stream);
/* Running-mean centring of fbank features, out of place (wespeaker/__init__.py:141-157): for every chunk b, frame t and
 * mel bin m,  out[b][t][m] = fb[b][t][m] - mean(fb[b][t - K/2 .. t + K/2][m] inside [0, T)),  the mean being the float32
 * sum in ascending frame order divided by the number of frames inside -- F.avg_pool1d(kernel K, stride 1, padding K / 2,
 * count_include_pad=False).  fb, out: (B, T, nmel) float32, out != fb; K odd >= 1; min(K, 2 T - 1) <= 2049. */
int pa_fbank_center_span(const float* fb, int B, int T, int nmel, int kernel, float* out, void* stream);
int pa_resnet_stem(const float* fbank, int B, int T, int F, const float* w9, const float* shift,
                   float* out, void* stream);
int pa_conv3x3(const float* X, int B, int H, int W, int cin, const float* Wg, const float* shift,
               const float* R, float* Y, int cout, int stride, int relu, void* stream);
/* the same stride-1 convolution through Winograd F(2x2,3x3).  U is NOT a plain [16][cout][cin] array: it is
 * G g G^T packed as one contiguous 32-KB slab per (32-cout slice, 16-cin stage),
 * [cout/32][cin/16][row = 32 xi + (cout % 32)][slot][4] with xi = 4a + b and channel quad q of the stage at
 * slot (q + 2 ((row >> 2) & 1)) & 3 -- build it with pa_winograd_pack_host (cout % 32 == 0, cin % 16 == 0). */
int pa_winograd_pack_host(const float* conv_weight /* (cout, cin, 3, 3), resnet.py:92-107 */,
                          const float* bn_scale /* (cout) gamma / sqrt(var + eps), or NULL */, int cout, int cin,
                          float* U_slabs /* HOST buffer, 16 * cout * cin floats */);
int pa_conv3x3_wino(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                    const float* R, float* Y, int cout, int relu, void* stream);
/* the same stride-1 convolution through Winograd F(4x4,3x3) (csrc/emb_winograd4.hip: 36 instead of 64 multiplies per
 * 16 outputs; error ~1e-5 of max |Y| per convolution, tools/probes/winograd_f4_numerics.py).  U: G g G^T packed as
 * one contiguous 36-KB slab per (32-cout slice, 8-cin stage), [cout/32][cin/8][row = 32 xi + (cout % 32)][8] with
 * xi = 6a + b -- build it with pa_winograd4_pack_host (cout % 32 == 0, cin % 8 == 0, cin >= 32). */
int pa_winograd4_pack_host(const float* conv_weight /* (cout, cin, 3, 3), resnet.py:92-107 */,
                           const float* bn_scale /* (cout) gamma / sqrt(var + eps), or NULL */, int cout, int cin,
                           float* U_slabs /* HOST buffer, 36 * cout * cin floats */);
int pa_conv3x3_wino4(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                     const float* R, float* Y, int cout, int relu, void* stream);
/* row ranges of one convolution (the maps X, R, Y are whole in every call): output rows 0 .. rows - 1 through F(4x4)
 * (rows == H or a multiple of 4 below it) and rows y_first .. H - 1 through F(2x2) (y_first even).  pa_emb_forward
 * splits a map whose height is 2 (mod 4) this way instead of padding its last F(4x4) tile row. */
int pa_conv3x3_wino4_rows(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                          const float* R, float* Y, int cout, int relu, int rows, void* stream);
int pa_conv3x3_wino_rows(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                         const float* R, float* Y, int cout, int relu, int y_first, void* stream);
int pa_stats_pool(const float* feat, int B, int Fh, int Tp, int C, const float* masks, int S, int Fm,
                  const int* nearest_idx, float* stats, void* stream);

/* ------------------------------------------------------------------------------------------
 * SSeRiouSS segmentation model: replaces SSeRiouSS.forward (models/segmentation/SSeRiouSS.py:289-328) =
 * torchaudio wav2vec 2.0 / WavLM `extract_features` -> (softmax-weighted mix of | one of) the transformer
 * layer outputs -> bi-LSTM stack -> Linear head -> classifier, as called from Inference.infer.
 * Weight layouts (all [out][in] row-major like torch unless stated):
 * ---------------------------------------------------------------------------------------- */
#define PA_W2V_MAX_CONV 8
#define PA_W2V_MAX_LAYERS 24
typedef struct pa_w2v_layer {
  const float* qk_w;   /* [2D][D] q rows then k rows (in_proj_weight[:2D] / q_proj, k_proj) */
  const float* qk_b;   /* [2D] */
  const float* v_w;    /* [D][D]; its bias is folded into out_b (soft-max rows sum to 1) */
  const float* out_w;  /* [D][D] */
  const float* out_b;  /* [D] = out_proj.bias + out_proj.weight @ v_bias */
  const float* ln1_g;  /* layer_norm */
  const float* ln1_b;
  const float* ff1_w;  /* [F][D] feed_forward.intermediate_dense */
  const float* ff1_b;
  const float* ff2_w;  /* [D][F] feed_forward.output_dense */
  const float* ff2_b;
  const float* ln2_g;  /* final_layer_norm */
  const float* ln2_b;
  const float* gate_w;     /* WavLM only: gru_rel_pos_linear [8][D/H] */
  const float* gate_b;     /* [8] */
  const float* gate_const; /* gru_rel_pos_const [H] */
} pa_w2v_layer;

typedef struct pa_sser_weights {
  int32_t num_conv;                        /* feature extractor layers (7) */
  int32_t conv_channels[PA_W2V_MAX_CONV];  /* multiples of 32 */
  int32_t conv_kernel[PA_W2V_MAX_CONV];    /* 10, 3, 3, 3, 3, 2, 2 (first <= 16) */
  int32_t conv_stride[PA_W2V_MAX_CONV];    /* 5, 2, 2, 2, 2, 2, 2 */
  int32_t extractor_layer_norm;            /* 0: "group_norm" (GroupNorm on layer 0 only), 1: "layer_norm" */
  int32_t embed_dim, num_layers, num_heads, ff_dim, layer_norm_first;
  int32_t pos_kernel, pos_groups;          /* 128, 16 */
  int32_t wavlm;                           /* gated relative position bias (rel_bias argument of the call) */
  int32_t use_layer;                       /* wav2vec_layer: < 0 = weighted mix of all layers */
  int32_t lstm_layers, lstm_hidden, lstm_bidir, num_linear, linear_hidden, num_classes, num_speakers;
  const float* conv_w[PA_W2V_MAX_CONV];      /* layer 0: [C0][K0]; layer l: [C_l][k * C_{l-1}], index j * C + c */
  const float* conv_b[PA_W2V_MAX_CONV];      /* or NULL (extractor_conv_bias = False) */
  const float* conv_norm_g[PA_W2V_MAX_CONV]; /* GroupNorm (layer 0) / LayerNorm (every layer) affine */
  const float* conv_norm_b[PA_W2V_MAX_CONV];
  const float* proj_ln_g; /* encoder.feature_projection.layer_norm */
  const float* proj_ln_b;
  const float* proj_w;    /* [D][C_last] */
  const float* proj_b;
  const float* pos_w;     /* [groups][kernel][D/groups (in)][D/groups (out)], weight_norm materialised */
  const float* pos_b;     /* [D] */
  const float* enc_ln_g;  /* encoder.transformer.layer_norm: in front of the layers iff !layer_norm_first (post-LN) */
  const float* enc_ln_b;
  pa_w2v_layer layers[PA_W2V_MAX_LAYERS];
  float layer_mix[PA_W2V_MAX_LAYERS];      /* softmax(wav2vec_weights) */
  const float* lstm_wih[PA_MAX_LSTM_LAYERS]; /* as in pa_seg_weights (layer 0: Kin = embed_dim) */
  const float* lstm_bias[PA_MAX_LSTM_LAYERS];
  const float* lstm_whh[PA_MAX_LSTM_LAYERS];
  const float* lin_w[PA_MAX_LINEAR];
  const float* lin_b[PA_MAX_LINEAR];
  const float* cls_w;
  const float* cls_b;
  const uint8_t* powerset_map;             /* NULL = multi-label head (sigmoid scores) */
} pa_sser_weights;

/* frames per chunk (SSeRiouSS.num_frames, SSeRiouSS.py:217-241); 0 = too short */
int pa_sser_num_frames(const pa_sser_weights* w, int num_samples);
size_t pa_sser_workspace_bytes(const pa_sser_weights* w, int num_chunks, int num_samples);
/* chunks and outputs as pa_seg_forward; rel_bias: [H][T][T] fp32 = rel_attn_embed[bucket(k - q)] for the
 * T = pa_sser_num_frames(num_samples) frames of a chunk (WavLM), NULL for a wav2vec 2.0 encoder */
int pa_sser_forward(const pa_sser_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                    int num_chunks, int num_samples, const float* rel_bias, float* logp, uint8_t* multilabel,
                    void* workspace, size_t workspace_bytes, void* stream);
/* outer x inner independent TN GEMMs in one launch (operand z = (zo, zi) starts zo * s?o + zi * s?i floats
 * after its base pointer); act 0 or 3 (GELU) */
int pa_gemm_tn_batched(const float* A, int lda, long sAo, long sAi, const float* W, int ldw, long sWo, long sWi,
                       const float* bias, float* C, long ldc, long sCo, long sCi, int M, int N, int K,
                       int outer, int inner, int act, void* stream);

/* ------------------------------------------------------------------------------------------
 * XVectorSincNet embedding model: replaces XVectorSincNet.forward (models/embedding/xvector.py:330-349) =
 * SincNet -> 5 x (Conv1d(k, dilation) + LeakyReLU + BatchNorm1d) -> StatsPool(weights) -> Linear, as called
 * by PyannoteAudioPretrainedSpeakerEmbedding (pipelines/speaker_verification.py:704-716).
 * ---------------------------------------------------------------------------------------- */
#define PA_XVEC_TDNN 5
typedef struct pa_xvec_weights {
  int32_t sinc_stride;                 /* 10 */
  int32_t dimension;                   /* embedding size (512) */
  int32_t tdnn_channels[PA_XVEC_TDNN]; /* 512, 512, 512, 512, 1500 (multiples of 4) */
  int32_t tdnn_kernel[PA_XVEC_TDNN];   /* 5, 3, 3, 1, 1 */
  int32_t tdnn_dilation[PA_XVEC_TDNN]; /* 1, 2, 3, 1, 1 */
  float wav_gamma, wav_beta;           /* SincNet fields: exactly those of pa_seg_weights */
  const float* sinc_filt;
  const float* norm0;
  const float* conv1_w;
  const float* conv1_b;
  const float* norm1;
  const float* conv2_w;
  const float* conv2_b;
  const float* norm2;
  /* [k][cout][cin_pad] per layer (cin_pad = 64 for layer 0), the PREVIOUS layer's BatchNorm folded in */
  const float* tdnn_w[PA_XVEC_TDNN];
  const float* tdnn_b[PA_XVEC_TDNN];
  const float* bn_scale; /* [channels[4]] the LAST BatchNorm as an affine map, applied inside the pooling */
  const float* bn_shift;
  const float* emb_w; /* [dimension][ld] ld = 2 * channels[4] rounded up to 32 (zero padded) */
  const float* emb_b;
} pa_xvec_weights;

/* frames left after SincNet and the TDNN stack (XVectorSincNet.num_frames, xvector.py:264-287); 0 = too short */
int pa_xvec_num_frames(const pa_xvec_weights* w, int num_samples);
size_t pa_xvec_workspace_bytes(const pa_xvec_weights* w, int num_chunks, int num_samples, int num_masks);
/* chunks addressed like pa_seg_forward; masks (num_chunks, num_masks, mask_frames) fp32 or NULL (unweighted
 * pooling), nearest_idx (frames) = F.interpolate(mode="nearest") source index of every pooled frame;
 * emb: (num_chunks * num_masks, dimension) */
int pa_xvec_forward(const pa_xvec_weights* w, const float* wav, int64_t wav_len, int64_t chunk_stride,
                    int num_chunks, int num_samples, const float* masks, int num_masks, int mask_frames,
                    const int32_t* nearest_idx, float* emb, void* workspace, size_t workspace_bytes,
                    void* stream);
/* StatsPool over the rows of a (tile, t, b16)-ordered activation matrix (models/blocks/pooling.py:64-130) */
int pa_stats_pool_rows(const float* feat, int B, int T0, int Tp, int C, int ld, const float* masks, int S,
                       int Fm, const int* nearest_idx, float* stats, int ld_stats,
                       const float* aff_scale /* (C) or NULL: x -> scale x + shift on load */,
                       const float* aff_shift, void* stream);

/* ------------------------------------------------------------------------------------------
 * Clustering distances (fp64, bit-identical to SciPy): replace the pdist inside
 * scipy.cluster.hierarchy.linkage(X, "centroid", "euclidean") (pipelines/clustering.py:374-382) and
 * scipy.spatial.distance.cdist(E, centroids, "cosine") (pipelines/clustering.py:190-200).
 * ---------------------------------------------------------------------------------------- */
/* out: condensed (N*(N-1)/2) upper triangle, pair order (0,1),(0,2),...,(N-2,N-1) */
int pa_pdist_f64(const double* X, int N, int D, double* out, void* stream);
/* out: (NA, NB); norms: scratch of NA + NB doubles */
int pa_cdist_cosine_f64(const double* A, int NA, const double* B, int NB, int D, double* out,
                        double* norms, void* stream);

/* Cluster centroids, bit-identical to np.mean(X[rows of cluster k], axis=0) (pipelines/clustering.py:182-187,
 * :462-472: float32 row-order sums divided in float32).  X: (., D) float32 rows; rows: row numbers grouped by
 * cluster, original order inside a cluster; offsets: K + 1 group boundaries into `rows`; out: (K, D) float32, NaN
 * rows for empty clusters. */
int pa_centroid_means(const float* X, int D, const int* rows, const int* offsets, int K, float* out, void* stream);

/* Centroid-linkage dendrogram, bit-identical to scipy.cluster.hierarchy.linkage(y, "centroid")
 * (pipelines/clustering.py:374-382).  D: condensed distances (n*(n-1)/2 doubles, e.g. straight from
 * pa_pdist_f64), OVERWRITTEN when the heap kernel runs;  Z: (n-1, 4) doubles in SciPy's layout [id_a, id_b,
 * height, size].  Two kernels behind one call (csrc/linkage_fast.hip, csrc/linkage.hip): a heap-free merge on a
 * square copy of the matrix over 1 / 8 / 16 workgroups, and -- only when two rows ever tie for the smallest lower
 * bound, i.e. when SciPy's heap order would matter -- the exact replay of SciPy's heap on the condensed matrix.
 * The workspace holds the square copy (8 n^2 bytes; PA_LINKAGE_FAST_MAX_GB caps it, default 96). */
size_t pa_linkage_workspace_bytes(int n);
int pa_linkage_centroid_f64(double* D, int n, double* Z, void* workspace, size_t workspace_bytes,
                            void* stream);
/* The same with the placement hint of earlier rounds (`alone`: nothing else competes for the GPU); since round 4
 * the hint is ignored -- the number of workgroups only depends on n (PA_LINKAGE_FAST_WGS / PA_LINKAGE_WGS
 * override it for experiments).  Results are identical either way. */
int pa_linkage_centroid_f64_ex(double* D, int n, double* Z, void* workspace, size_t workspace_bytes, int alone,
                               void* stream);

/* ------------------------------------------------------------------------------------------
 * Frame-domain stages (uint8 hard segmentations in, per-frame decisions out).  Replace the Python
 * loops of Inference.aggregate (core/inference.py:589-611), speaker_count
 * (pipelines/utils/diarization.py:150-185), SpeakerDiarization.reconstruct
 * (pipelines/speaker_diarization.py:480-528), to_diarization (pipelines/utils/diarization.py:
 * 221-268) and the numpy reductions of filter_embeddings (pipelines/clustering.py:109-116) and
 * get_embeddings' mask selection (pipelines/speaker_diarization.py:375-427).
 * seg: (C, F, S) uint8 {0,1};  start_frame: (C) int32 = closest_frame(chunk.start + frame.duration/2)
 * (core/inference.py:596), computed by the caller in float64 exactly as the reference does.
 * ---------------------------------------------------------------------------------------- */
/* active[c][s] = sum_f seg;  clean[c][s] = sum_f seg * [sum_s' seg == 1]   (both (C,S) int32) */
int pa_seg_chunk_stats(const uint8_t* seg, int C, int F, int S, int32_t* active, int32_t* clean,
                       void* stream);
/* masks (C,S,F) fp32: the overlap-free mask where exclude_overlap && clean[c][s] > min_num_frames,
 * the full mask otherwise */
int pa_embedding_masks(const uint8_t* seg, int C, int F, int S, const int32_t* clean,
                       int exclude_overlap, int min_num_frames, float* masks, void* stream);
/* count (T) uint8 = rint( overlap-add average of sum_s seg );  scratch: 2*T int32 */
int pa_speaker_count(const uint8_t* seg, int C, int F, int S, const int32_t* start_frame, int T,
                     uint8_t* count, int32_t* scratch, void* stream);
/* act (T,K) int32 = overlap-add SUM over chunks of max_{s: hard[c][s]==k} seg[c][f][s];
 * hard: (C,S) int32, negative = unassigned / inactive */
int pa_cluster_activations(const uint8_t* seg, int C, int F, int S, const int32_t* start_frame,
                           const int32_t* hard, int K, int T, int32_t* act, void* stream);
/* out (T,K) uint8: the min(count[t], cap, K) most active clusters per frame, ties to the lowest index;
 * tie (T) uint8: 1 where equal activations straddle the selection boundary (the reference's
 * np.argsort order among equals is host-dependent: the caller re-decides those frames with numpy) */
int pa_topk_binarize(const int32_t* act, const uint8_t* count, int T, int K, int cap, uint8_t* out,
                     uint8_t* tie, void* stream);

/* Soft-score (non-powerset segmentation) forms of the same stages:
 * hysteresis thresholding = `binarize` (utils/signal.py:78-140; pipelines/speaker_diarization.py:599-606):
 * scores (C,F,K) fp32 -> out (C,F,K) uint8; on where score > onset, off where score < offset, unchanged in
 * between, NaN = 0; initial_state 0 / 1, or -1 for `scores[:, 0] >= (onset + offset) / 2`. */
int pa_binarize_hysteresis(const float* scores, int C, int F, int K, float onset, float offset,
                           int initial_state, uint8_t* out, void* stream);
/* out (C,F,K) fp32 = max_{s: hard[c][s]==k} scores[c][f][s], NaN when chunk c has no speaker in cluster k
 * (pipelines/speaker_diarization.py:506-522); overlap-add it with pa_aggregate(skip_average = 1). */
int pa_cluster_max(const float* scores, int C, int F, int S, const int32_t* hard, int K, float* out,
                   void* stream);
/* pa_topk_binarize on fp32 activations (>= 0, no NaN) */
int pa_topk_binarize_f32(const float* act, const uint8_t* count, int T, int K, int cap, uint8_t* out,
                         uint8_t* tie, void* stream);

/* General overlap-add aggregation, replaces Inference.aggregate (core/inference.py:498-620):
 * scores (C,F,K) fp32 (NaN = missing), window (F) fp64 = Hamming or ones, warm (F) fp64 = warm-up
 * window, start_frame (C) non-decreasing -> out (T,K) fp32 = sum / max(weight sum, epsilon) (or the plain sum), `missing` where
 * nothing voted.  Bit-identical to the reference's chunk loop (same accumulation order and dtypes). */
int pa_aggregate(const float* scores, int C, int F, int K, const int32_t* start_frame, int T,
                 const double* window, const double* warm, float epsilon, float missing, int skip_average,
                 float* out, void* stream);

/* ---- audio front door (core/io.py:223-265) ---- */

/* Polyphase windowed-sinc resampling, replaces torchaudio.functional.resample in
 * Audio.downmix_and_resample (core/io.py:258-262): x (n) fp32 -> out (n_out) fp32,
 * out[q P + p] = sum_k taps[p][k] x[q L - width + k], K = 2 width + L taps per phase, zeros outside x. */
int pa_resample_poly(const float* x, long n, const float* taps, int L, int P, int K, int width, float* out,
                     long n_out, void* stream);

/* ---- VBx clustering + PLDA (pipelines/clustering.py:550-669, utils/vbx.py:27-218, core/plda.py:33-60) ---- */

/* x-vector -> PLDA space, replaces PLDA.__call__ (core/plda.py:47-60) = plda_tf(xvec_tf(x))
 * (utils/vbx.py:205-217): X (n, din) fp32 -> fea (n, dout) fp64.  lda [din][dmid], trT [dmid][dout]. */
int pa_plda_transform(const float* X, int n, int din, int dmid, int dout, const double* mean1,
                      const double* lda, const double* mean2, const double* mu, const double* trT,
                      double* fea, void* stream);
size_t pa_vbx_workspace_bytes(int n, int s, int d);
/* one iteration of VBx (utils/vbx.py:106-133): M step (16)(17), E step (23) + GMM responsibilities,
 * ELBO (25) -> elbo_out[0].  gamma (n, s) fp64 is updated in place. */
int pa_vbx_iteration(const double* fea, const double* Phi, int n, int s, int d, double Fa, double Fb,
                     int first, double* gamma, double* elbo_out, void* workspace, size_t workspace_bytes,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYANNOTE_AMD_H */
