// LSTM stack + feed-forward head of PyanNet on gfx950
// (reference: models/segmentation/PyanNet.py:211-240; torch.nn.LSTM gate order i,f,g,o).
//
//   k_gemm_tn     C = act(A[M][K] * W[N][K]^T + bias)   f32 MFMA 32x32x2, 128x128x32 tiles.
//                 Used for the LSTM input projections of ALL time steps at once, the two Linear
//                 layers of the head and the embedding's seg_1 Linear.
//   k_lstm_rec    the recurrence: one workgroup per (16-chunk tile, direction).  W_hh (128x512 fp32,
//                 256 KB) lives in the register file of the CU for the whole sequence: 4 waves x 256
//                 VGPRs as MFMA B operands; h_t goes through a double-buffered 16x128 LDS tile; the
//                 cell state never leaves registers.  589 dependent steps, one barrier per step.
//   k_lstm_rec_gen  the recurrence for any other hidden size (multiple of 16) / a single direction: W_hh streamed
//                 from L2 as MFMA operands, two 16-chunk tiles per workgroup.
//   k_classifier  Linear(K -> NC) + log-softmax + powerset arg-max -> multilabel LUT; or, for multi-label
//                 (non-powerset) checkpoints, Linear(K -> NC) + sigmoid (core/model.py:271-299).
//
// Row order of all [m][*] activations in the LSTM stack: m = (tile*T + t)*16 + b16 (b = 16*tile+b16),
// i.e. the 16 chunks of a tile are adjacent for a fixed time step, which is exactly the MFMA M
// dimension of the recurrence.
#include "common.h"

namespace pa {

// ---------------------------------------------------------------------------------------------
// GEMM  (TN: both operands K-contiguous)
// ---------------------------------------------------------------------------------------------
constexpr int GB = 128;      // BM = BN
constexpr int GK = 32;       // BK
constexpr int GLD = GK + 4;  // LDS row stride (floats): 9 x 16-B slots -> conflict-free b128 reads

// out_mode 0: C[m*ldc + n]; out_mode 1 (LSTM gate pre-activations): C[((m>>4)*N + n)*16 + (m&15)]
// ACT: 0 none, 1 leaky_relu(0.01), 2 relu, 3 gelu (erf).  RES (out_mode 0 only): C = act(A W^T + bias + Res),
// Res laid out like C -- the 1x1 "conv3 + bn3 + shortcut + relu" of a Bottleneck block (resnet.py:205-211).
// blockIdx.y = batch index z (the attention GEMMs of the wav2vec encoder): operand z starts
// (z / inner) * s?o + (z % inner) * s?i floats after the base pointer (all zero for plain launches).
template <int ACT, int OUT_MODE>
__global__ __launch_bounds__(256) void k_gemm_tn(const float* __restrict__ A, int lda,
                                                  const float* __restrict__ W, int ldw,
                                                  const float* __restrict__ bias,
                                                  const float* __restrict__ Res,
                                                  float* __restrict__ C, long ldc, int M, int N, int K,
                                                  int nm, int nn, int inner, long sAo, long sAi, long sWo,
                                                  long sWi, long sCo, long sCi, int gHW, int gWo, int gH, int gW) {
  if (gridDim.y > 1) {
    const int zo = blockIdx.y / inner, zi = blockIdx.y % inner;
    A += zo * sAo + zi * sAi;
    W += zo * sWo + zi * sWi;
    C += zo * sCo + zi * sCi;
    if (Res != nullptr) Res += zo * sCo + zi * sCi;
  }
  __shared__ __attribute__((aligned(16))) float As[GB * GLD];
  __shared__ __attribute__((aligned(16))) float Ws[GB * GLD];
  // block -> tile mapping; with 8 column tiles the 8 blocks sharing an A panel run on one XCD
  int mt, nt;
  const int bid = blockIdx.x;
  if (nn == 8) {
    nt = (bid >> 3) & 7;
    mt = (bid & 7) + 8 * (bid >> 6);
  } else {
    nt = bid % nn;
    mt = bid / nn;
  }
  if (mt >= nm) return;
  const int m0 = mt * GB, n0 = nt * GB;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int lr = tid >> 3, lc = (tid & 7) * 4;  // loader: row lr + 32*i, 4 floats at column lc

  float4 ra[4], rw[4];
  // row pointers of this thread's four A / W rows (once per tile).  gHW > 0: row m of A is pixel (2 y, 2 x) of image b
  // of an NHWC map (b = m / gHW, y = (m % gHW) / gWo, x = m % gWo; lda = channels) -- the 1x1 stride-2 shortcut
  // convolution of a ResNet block reads its input in place instead of through a gathered copy (k_gather_s2).
  const float* ap[4];
  const float* wp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ar = m0 + lr + 32 * i, wr = n0 + lr + 32 * i;
    long arow = ar;
    if (gHW > 0) {
      const int b = ar / gHW, rem = ar - b * gHW, y = rem / gWo, x = rem - y * gWo;
      arow = ((long)b * gH + 2 * y) * gW + 2 * x;
    }
    ap[i] = ar < M ? A + arow * lda + lc : nullptr;
    wp[i] = wr < N ? W + (long)wr * ldw + lc : nullptr;
  }
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = ap[i] != nullptr ? *reinterpret_cast<const float4*>(ap[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      rw[i] = wp[i] != nullptr ? *reinterpret_cast<const float4*>(wp[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(As + (lr + 32 * i) * GLD + lc) = ra[i];
      *reinterpret_cast<float4*>(Ws + (lr + 32 * i) * GLD + lc) = rw[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fi = lane & 31, kh = lane >> 5;
  gload(0);
  for (int k0 = 0; k0 < K; k0 += GK) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (k0 + GK < K) gload(k0 + GK);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      float4 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const float4*>(As + (64 * wm + 32 * i + fi) * GLD + kh * 16 + 4 * r4);
        wf[i] = *reinterpret_cast<const float4*>(Ws + (64 * wn + 32 * i + fi) * GLD + kh * 16 + 4 * r4);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = MFMA32(af[i].x, wf[j].x, acc[i][j]);
          acc[i][j] = MFMA32(af[i].y, wf[j].y, acc[i][j]);
          acc[i][j] = MFMA32(af[i].z, wf[j].z, acc[i][j]);
          acc[i][j] = MFMA32(af[i].w, wf[j].w, acc[i][j]);
        }
    }
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + 64 * wn + 32 * j + fi;
      const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + 64 * wm + 32 * i + 8 * q + 4 * kh;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][4 * q + e] + bv;
          if (OUT_MODE == 0 && Res != nullptr && n < N && m + e < M) v[e] += Res[(long)(m + e) * ldc + n];
          if (ACT == 1) v[e] = leaky_relu(v[e]);
          if (ACT == 2) v[e] = fmaxf(v[e], 0.f);
          if (ACT == 3) v[e] = gelu_erf(v[e]);
        }
        if (n < N) {
          if (OUT_MODE == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (m + e < M) C[(long)(m + e) * ldc + n] = v[e];
          } else {
            if (m < M)
              *reinterpret_cast<float4*>(C + ((long)(m >> 4) * N + n) * 16 + (m & 15)) =
                  make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// LSTM recurrence (hidden size 128).  grid = (ntiles, ndir), block = 256, 1 workgroup per CU.
//   xproj : [tile][t][1024][16]  gate pre-activations x_t W_ih^T + b_ih + b_hh, column index
//           dir*512 + w*128 + (q*2+s)*16 + n  <->  torch gate row q*128 + 32w + 16s + n   (q: i,f,g,o)
//   whh_p : [dir][w][q*2+s][kt][lane]  =  W_hh[q*128 + 32w + 16s + (lane&15)][(lane>>4)*32 + kt]
//   out   : [m][256], m = (tile*T + t)*16 + b16, columns dir*128 + j
// ---------------------------------------------------------------------------------------------
constexpr int LSTM_HS = 162;  // LDS row stride of h (floats); k index j lives at (j/32)*33 + j%32

// Gate non-linearities on the hardware exp / rcp units: v_exp_f32 and v_rcp_f32 are accurate to ~1 ulp
// each, far inside the float contract of the path (the end-to-end log-probabilities sit at 0.002 of the
// rtol 1e-4 / atol 1e-5 bound, tests/test_seg_gpu.py), and the f32 MFMA shares the vector ALUs with
// these instructions: libm's expf / tanhf cost ~5x the issue cycles per step.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 2 sigmoid(2x) - 1; the argument is clamped so that exp2 stays finite
  const float xc = fminf(fmaxf(x, -15.f), 15.f);
  return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * xc)) - 1.0f;
}

__global__ __launch_bounds__(256, 1) void k_lstm_rec(const float* __restrict__ xproj,
                                                      const float* __restrict__ whh_p,
                                                      float* __restrict__ out, int T) {
  __shared__ float hs[2 * 16 * LSTM_HS];
  const int tile = blockIdx.x, dir = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, g = lane >> 4;

  float wr[8][32];
  {
    const float* wp = whh_p + (long)((dir * 4 + w) * 8 * 32) * 64 + lane;
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8)
#pragma unroll
      for (int kt = 0; kt < 32; ++kt) wr[q8][kt] = wp[(q8 * 32 + kt) * 64];
  }
  for (int i = tid; i < 2 * 16 * LSTM_HS; i += 256) hs[i] = 0.f;
  float c[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[s][r] = 0.f;
  __syncthreads();

  const float* xbase = xproj + ((long)tile * T * 1024 + dir * 512 + w * 128 + n) * 16 + 4 * g;
  float* obase = out + ((long)tile * T * 16 + 4 * g) * 256 + dir * 128 + 32 * w + n;

  f32x4 xin[8];
  {
    const int t = dir ? T - 1 : 0;
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8)
      xin[q8] = *reinterpret_cast<const f32x4*>(xbase + ((long)t * 1024 + q8 * 16) * 16);
  }
  int cur = 0;
#pragma unroll 1
  for (int step = 0; step < T; ++step) {
    const int t = dir ? T - 1 - step : step;
    f32x4 acc[8];
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8) acc[q8] = xin[q8];
    if (step + 1 < T) {
      const int tn = dir ? t - 1 : t + 1;
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8)
        xin[q8] = *reinterpret_cast<const f32x4*>(xbase + ((long)tn * 1024 + q8 * 16) * 16);
    }
    const float* hp = hs + (cur * 16 + n) * LSTM_HS + g * 33;
    float a[32];
#pragma unroll
    for (int kt = 0; kt < 32; ++kt) a[kt] = hp[kt];
#pragma unroll
    for (int kt = 0; kt < 32; ++kt)
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8) acc[q8] = MFMA16(a[kt], wr[q8][kt], acc[q8]);

    float* hn = hs + ((cur ^ 1) * 16 + 4 * g) * LSTM_HS + w * 33 + n;
    float* op = obase + (long)t * 16 * 256;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = sigmoidf_(acc[0 + s][r]);
        const float fg = sigmoidf_(acc[2 + s][r]);
        const float gg = tanhf_(acc[4 + s][r]);
        const float og = sigmoidf_(acc[6 + s][r]);
        const float cn = fg * c[s][r] + ig * gg;
        c[s][r] = cn;
        const float h = og * tanhf_(cn);
        hn[r * LSTM_HS + 16 * s] = h;
        op[r * 256 + 16 * s] = h;
      }
    __syncthreads();
    cur ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------
// LSTM recurrence, any hidden size H that is a multiple of 16 (<= 512), one or two directions
// (PyanNet.py:64-72, 98-123 accept any nn.LSTM configuration; the register-resident kernel above is the
// H = 128 bidirectional special case).  grid = (ceil(ntiles / MT), ndir), block = 256.
//   W_hh (4H x H) no longer fits a CU's register file in general (H = 256: 1 MB): it is STREAMED from L2 every
//   step as MFMA B operands, and a workgroup owns MT = 2 tiles of 16 chunks so that every streamed operand feeds
//   two MFMAs (H = 256: 2 048 MFMAs ~ 27 us against ~7 us of L2 traffic per step).
//   A wave owns the hidden units of the 16-unit tiles u = w, w + 4, ...; per unit tile the four gates are four
//   16 x 16 accumulators per chunk tile.
//   xproj : [tile][t][ndir * 4H][16], column dir * 4H + (4 u + q) * 16 + n  <->  torch gate row q H + 16 u + n
//   whh_g : [dir][u][q][k4][lane][j]  =  W_hh[q H + 16 u + (lane & 15)][16 k4 + 4 j + (lane >> 4)]
//   out   : [m][ndir * H], m = (tile * T + t) * 16 + b16, columns dir * H + j
// ---------------------------------------------------------------------------------------------
constexpr int LSTMG_MT = 2;
constexpr int LSTMG_MAXU = 8;    // unit tiles per wave: H <= 16 * 4 * 8 = 512

__global__ __launch_bounds__(256) void k_lstm_rec_gen(const float* __restrict__ xproj,
                                                      const float* __restrict__ whh_g,
                                                      float* __restrict__ out, int T, int H, int ntiles,
                                                      int ndir, int hstride) {
  extern __shared__ float hsg[];   // [MT][2][16][hstride], hstride = 4 (mod 64): conflict-free reads and writes
  constexpr int MT = LSTMG_MT;
  const int tile0 = blockIdx.x * MT, dir = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int U = H >> 4, K4 = H >> 4;
  const long NG = (long)ndir * 4 * H;
  const int ldo = ndir * H;
  for (int i = tid; i < MT * 2 * 16 * hstride; i += 256) hsg[i] = 0.f;
  f32x4 c[LSTMG_MAXU][MT];
#pragma unroll
  for (int ui = 0; ui < LSTMG_MAXU; ++ui)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) c[ui][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  // a tile beyond the last one (odd ntiles) is computed on the last tile's inputs and never stored
  int tl[MT];
  bool live[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    live[mt] = tile0 + mt < ntiles;
    tl[mt] = live[mt] ? tile0 + mt : ntiles - 1;
  }
  int cur = 0;
#pragma unroll 1
  for (int step = 0; step < T; ++step) {
    const int t = dir ? T - 1 - step : step;
#pragma unroll
    for (int ui = 0; ui < LSTMG_MAXU; ++ui) {
      const int u = w + 4 * ui;
      if (u >= U) break;
      f32x4 xin[MT][4], acc[MT][4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xin[mt][q] = *reinterpret_cast<const f32x4*>(
              xproj + (((long)tl[mt] * T + t) * NG + (long)dir * 4 * H + (4 * u + q) * 16 + n) * 16 + 4 * g);
          acc[mt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      const f32x4* wp = reinterpret_cast<const f32x4*>(whh_g) + ((long)(dir * U + u) * 4 * K4) * 64 + lane;
      const float* hp0 = hsg + (cur * 16 + n) * hstride + g;
#pragma unroll 2
      for (int k4 = 0; k4 < K4; ++k4) {
        f32x4 b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = wp[(long)(q * K4 + k4) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float a = hp0[mt * 2 * 16 * hstride + 16 * k4 + 4 * j];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[mt][q] = MFMA16(a, b[q][j], acc[mt][q]);
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float* hn = hsg + ((mt * 2 + (cur ^ 1)) * 16 + 4 * g) * hstride + 16 * u + n;
        float* op = out + (((long)tl[mt] * T + t) * 16 + 4 * g) * ldo + dir * H + 16 * u + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ig = sigmoidf_(acc[mt][0][r] + xin[mt][0][r]);
          const float fg = sigmoidf_(acc[mt][1][r] + xin[mt][1][r]);
          const float gg = tanhf_(acc[mt][2][r] + xin[mt][2][r]);
          const float og = sigmoidf_(acc[mt][3][r] + xin[mt][3][r]);
          const float cn = fg * c[ui][mt][r] + ig * gg;
          c[ui][mt][r] = cn;
          const float h = og * tanhf_(cn);
          hn[r * hstride] = h;
          if (live[mt]) op[(long)r * ldo] = h;
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------
// classifier Linear(K -> NC) + log-softmax + hard powerset -> multilabel
// (PyanNet.py:240, core/model.py:290-291, utils/powerset.py:115-140).  One thread per row.
// ---------------------------------------------------------------------------------------------
constexpr int CLS_MAXC = 16;

__global__ __launch_bounds__(256) void k_classifier(const float* __restrict__ X, int ldx, int K, int M,
                                                     int T, int B, const float* __restrict__ cw,
                                                     const float* __restrict__ cb, int NC,
                                                     const unsigned char* __restrict__ mapping, int S,
                                                     float* __restrict__ logp,
                                                     unsigned char* __restrict__ ml) {
  extern __shared__ float wsh[];  // [NC][K] + [NC]
  for (int i = threadIdx.x; i < NC * K; i += 256) wsh[i] = cw[i];
  for (int i = threadIdx.x; i < NC; i += 256) wsh[NC * K + i] = cb[i];
  __syncthreads();
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int b16 = (int)(m & 15);
  const long tt = m >> 4;
  const int t = (int)(tt % T), tile = (int)(tt / T);
  const int b = tile * 16 + b16;
  if (b >= B) return;
  float z[CLS_MAXC];
#pragma unroll
  for (int cidx = 0; cidx < CLS_MAXC; ++cidx) z[cidx] = cidx < NC ? wsh[NC * K + cidx] : 0.f;
  const float* x = X + m * ldx;
  for (int k = 0; k < K; k += 4) {
    const float4 xv = *reinterpret_cast<const float4*>(x + k);
#pragma unroll
    for (int cidx = 0; cidx < CLS_MAXC; ++cidx)
      if (cidx < NC) {
        const float* wrow = wsh + cidx * K + k;
        z[cidx] = fmaf(xv.x, wrow[0], z[cidx]);
        z[cidx] = fmaf(xv.y, wrow[1], z[cidx]);
        z[cidx] = fmaf(xv.z, wrow[2], z[cidx]);
        z[cidx] = fmaf(xv.w, wrow[3], z[cidx]);
      }
  }
  const long o = (long)b * T + t;
  if (mapping == nullptr) {
    // multi-label / binary problems: sigmoid scores (default_activation, core/model.py:286-294); the hard
    // decisions come from the hysteresis kernel (frames.hip: k_hysteresis), not from here
    if (logp != nullptr)
      for (int cidx = 0; cidx < NC; ++cidx) logp[o * NC + cidx] = 1.f / (1.f + expf(-z[cidx]));
    return;
  }
  float mx = z[0];
  int am = 0;
#pragma unroll
  for (int cidx = 1; cidx < CLS_MAXC; ++cidx)
    if (cidx < NC && z[cidx] > mx) {
      mx = z[cidx];
      am = cidx;
    }
  float se = 0.f;
#pragma unroll
  for (int cidx = 0; cidx < CLS_MAXC; ++cidx)
    if (cidx < NC) se += expf(z[cidx] - mx);
  const float lse = mx + logf(se);
  if (logp != nullptr)
    for (int cidx = 0; cidx < NC; ++cidx) logp[o * NC + cidx] = z[cidx] - lse;
  if (ml != nullptr)
    for (int s = 0; s < S; ++s) ml[o * S + s] = mapping[am * S + s];
}

}  // namespace pa

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int pa_gemm_tn_ex(const float* A, int lda, const float* W, int ldw, const float* bias, const float* Res,
                  float* C, long ldc, int M, int N, int K, int act, int out_mode, void* stream);

int pa_gemm_tn(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, long ldc,
               int M, int N, int K, int act, int out_mode, void* stream) {
  return pa_gemm_tn_ex(A, lda, W, ldw, bias, nullptr, C, ldc, M, N, K, act, out_mode, stream);
}

int pa_gemm_tn_batched(const float* A, int lda, long sAo, long sAi, const float* W, int ldw, long sWo, long sWi,
                       const float* bias, float* C, long ldc, long sCo, long sCi, int M, int N, int K,
                       int outer, int inner, int act, void* stream);

static int gemm_tn_launch(const float* A, int lda, const float* W, int ldw, const float* bias, const float* Res,
                          float* C, long ldc, int M, int N, int K, int act, int out_mode, int gHW, int gWo, int gH,
                          int gW, void* stream);

int pa_gemm_tn_ex(const float* A, int lda, const float* W, int ldw, const float* bias, const float* Res,
                  float* C, long ldc, int M, int N, int K, int act, int out_mode, void* stream) {
  return gemm_tn_launch(A, lda, W, ldw, bias, Res, C, ldc, M, N, K, act, out_mode, 0, 0, 0, 0, stream);
}

// C[(b, y, x)][n] = sum_c X[b][2 y][2 x][c] W[n][c] + bias[n]: the 1x1 stride-2 shortcut convolution + BatchNorm of a
// ResNet block (models/embedding/wespeaker/resnet.py:109-118) straight from the NHWC map X (B, H, W, cin)
int pa_gemm_tn_s2(const float* X, int B, int H, int W, int cin, const float* Wt, int ldw, const float* bias, float* C,
                  long ldc, int N, void* stream) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (B <= 0) return 0;
  return gemm_tn_launch(X, cin, Wt, ldw, bias, nullptr, C, ldc, B * Ho * Wo, N, cin, 0, 0, Ho * Wo, Wo, H, W, stream);
}

static int gemm_tn_launch(const float* A, int lda, const float* W, int ldw, const float* bias, const float* Res,
                          float* C, long ldc, int M, int N, int K, int act, int out_mode, int gHW, int gWo, int gH,
                          int gW, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  PA_REQUIRE(Res == nullptr || out_mode == 0, "pa_gemm_tn_ex: a residual needs out_mode 0");
  PA_REQUIRE(K % pa::GK == 0 && lda % 4 == 0 && ldw % 4 == 0,
             "pa_gemm_tn: K (%d) must be a multiple of 32 and lda/ldw multiples of 4", K);
  PA_REQUIRE(out_mode == 0 || (M % 16 == 0), "pa_gemm_tn: out_mode 1 needs M %% 16 == 0");
  const int nm = pa::cdiv(M, pa::GB), nn = pa::cdiv(N, pa::GB);
  const int grid = nn == 8 ? pa::cdiv(nm, 8) * 64 : nm * nn;
  hipStream_t st = (hipStream_t)stream;
  pa::ProfScope prof("k_gemm_tn", stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
#define PA_GEMM(ACT, OM)                                                                          \
  hipLaunchKernelGGL((pa::k_gemm_tn<ACT, OM>), dim3(grid), dim3(256), 0, st, A, lda, W, ldw, bias, \
                     Res, C, ldc, M, N, K, nm, nn, 1, 0L, 0L, 0L, 0L, 0L, 0L, gHW, gWo, gH, gW)
  if (act == 0 && out_mode == 0) PA_GEMM(0, 0);
  else if (act == 1 && out_mode == 0) PA_GEMM(1, 0);
  else if (act == 2 && out_mode == 0) PA_GEMM(2, 0);
  else if (act == 3 && out_mode == 0) PA_GEMM(3, 0);
  else if (act == 0 && out_mode == 1) PA_GEMM(0, 1);
  else PA_REQUIRE(false, "pa_gemm_tn: unsupported act/out_mode %d/%d", act, out_mode);
#undef PA_GEMM
  PA_CHECK_LAUNCH("pa_gemm_tn");
  return 0;
}

// outer x inner independent GEMMs in one launch: operand z = (zo, zi) starts zo * s?o + zi * s?i floats after
// its base pointer (bias shared).  The per-(chunk, head) Q K^T and P V products of the wav2vec encoder.
int pa_gemm_tn_batched(const float* A, int lda, long sAo, long sAi, const float* W, int ldw, long sWo, long sWi,
                       const float* bias, float* C, long ldc, long sCo, long sCi, int M, int N, int K,
                       int outer, int inner, int act, void* stream) {
  if (M <= 0 || N <= 0 || outer <= 0 || inner <= 0) return 0;
  PA_REQUIRE(K % pa::GK == 0 && lda % 4 == 0 && ldw % 4 == 0 && sAo % 4 == 0 && sAi % 4 == 0 && sWo % 4 == 0 &&
                 sWi % 4 == 0,
             "pa_gemm_tn_batched: K (%d) must be a multiple of 32 and every stride a multiple of 4", K);
  PA_REQUIRE((long)outer * inner <= 65535, "pa_gemm_tn_batched: at most 65535 products per launch");
  const int nm = pa::cdiv(M, pa::GB), nn = pa::cdiv(N, pa::GB);
  const int grid = nn == 8 ? pa::cdiv(nm, 8) * 64 : nm * nn;
  hipStream_t st = (hipStream_t)stream;
  const double z = (double)outer * inner;
  pa::ProfScope prof("k_gemm_tn", stream, 2.0 * z * M * N * K, 4.0 * z * ((double)M * K + (double)N * K + (double)M * N));
#define PA_GEMMB(ACT)                                                                                          \
  hipLaunchKernelGGL((pa::k_gemm_tn<ACT, 0>), dim3(grid, outer * inner), dim3(256), 0, st, A, lda, W, ldw, bias, \
                     (const float*)nullptr, C, ldc, M, N, K, nm, nn, inner, sAo, sAi, sWo, sWi, sCo, sCi, 0, 0, 0, 0)
  if (act == 0) PA_GEMMB(0);
  else if (act == 3) PA_GEMMB(3);
  else PA_REQUIRE(false, "pa_gemm_tn_batched: unsupported act %d", act);
#undef PA_GEMMB
  PA_CHECK_LAUNCH("pa_gemm_tn_batched");
  return 0;
}

int pa_lstm_rec(const float* xproj, const float* whh_packed, float* out, int ntiles, int ndir, int T,
                void* stream) {
  if (ntiles <= 0 || T <= 0) return 0;
  // algorithmic: h_{t-1} (16x128) x W_hh^T (128x512) per tile, direction and step + 512 gate inputs in,
  // 128 outputs out per (chunk, step, direction)
  pa::ProfScope prof("k_lstm_rec", stream, 2.0 * ntiles * 16 * ndir * T * 128 * 512,
                     4.0 * ntiles * 16 * ndir * T * (512 + 128));
  hipLaunchKernelGGL(pa::k_lstm_rec, dim3(ntiles, ndir), dim3(256), 0, (hipStream_t)stream, xproj,
                     whh_packed, out, T);
  PA_CHECK_LAUNCH("pa_lstm_rec");
  return 0;
}

int pa_lstm_rec_h(const float* xproj, const float* whh_packed, float* out, int ntiles, int ndir, int T, int H,
                  void* stream) {
  if (ntiles <= 0 || T <= 0) return 0;
  if (H == 128 && ndir == 2) return pa_lstm_rec(xproj, whh_packed, out, ntiles, ndir, T, stream);
  PA_REQUIRE(H >= 16 && H % 16 == 0 && H <= 64 * pa::LSTMG_MAXU && (ndir == 1 || ndir == 2),
             "pa_lstm_rec_h: hidden size %d must be a multiple of 16 in [16, %d], directions 1 or 2 (got %d)", H,
             64 * pa::LSTMG_MAXU, ndir);
  const int hstride = ((H + 63) / 64) * 64 + 4;
  const size_t lds = sizeof(float) * pa::LSTMG_MT * 2 * 16 * hstride;
  // 66.5 KB at H = 256, 132 KB at H = 512: above the 64 KB a launch gets without asking, and above what a part with
  // less LDS than gfx950's 160 KB has at all -- ask for it, and refuse with a message instead of a failed launch
  {
    int dev = 0, limit = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&limit, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    PA_REQUIRE(limit <= 0 || lds <= (size_t)limit,
               "pa_lstm_rec_h: hidden size %d needs %zu bytes of LDS per workgroup, this device offers %d", H, lds, limit);
    static size_t granted = 0;
    if (lds > granted) {
      PA_REQUIRE(hipFuncSetAttribute((const void*)pa::k_lstm_rec_gen, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds) == hipSuccess,
                 "pa_lstm_rec_h: cannot reserve %zu bytes of dynamic LDS for hidden size %d", lds, H);
      granted = lds;
    }
  }
  pa::ProfScope prof("k_lstm_rec_gen", stream, 2.0 * ntiles * 16 * ndir * T * H * 4.0 * H,
                     4.0 * ntiles * 16 * ndir * T * (4.0 * H + H));
  hipLaunchKernelGGL(pa::k_lstm_rec_gen, dim3(pa::cdiv(ntiles, pa::LSTMG_MT), ndir), dim3(256), lds,
                     (hipStream_t)stream, xproj, whh_packed, out, T, H, ntiles, ndir, hstride);
  PA_CHECK_LAUNCH("pa_lstm_rec_h");
  return 0;
}

int pa_classifier(const float* X, int ldx, int K, int ntiles, int T, int B, const float* cw,
                  const float* cb, int NC, const unsigned char* mapping, int S, float* logp,
                  unsigned char* multilabel, void* stream) {
  PA_REQUIRE(NC <= pa::CLS_MAXC && K % 4 == 0, "pa_classifier: NC <= 16 and K %% 4 == 0 required");
  const long M = (long)ntiles * T * 16;
  if (M <= 0) return 0;
  const size_t lds = (size_t)(NC * K + NC) * sizeof(float);
  pa::ProfScope prof("k_classifier", stream, 2.0 * M * K * NC, 4.0 * M * K + 4.0 * M * NC);
  hipLaunchKernelGGL(pa::k_classifier, dim3(pa::cdiv(M, 256)), dim3(256), lds, (hipStream_t)stream, X,
                     ldx, K, (int)M, T, B, cw, cb, NC, mapping, S, logp, multilabel);
  PA_CHECK_LAUNCH("pa_classifier");
  return 0;
}

}  // extern "C"
