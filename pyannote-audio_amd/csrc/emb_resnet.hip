// WeSpeaker ResNet34 backbone on gfx950 (reference: models/embedding/wespeaker/resnet.py:84-145,
// 399-430).  ~97 % of the pipeline's FLOPs are the 3x3 convolutions below.
//
// Activations are NHWC fp32: X[b][h][w][c], h = mel axis (80/40/20/10), w = time axis, so that the
// channel dimension -- the MFMA K/N dimension -- is contiguous.  BatchNorm (eval) is folded on the
// host: weights carry gamma/sqrt(var+eps), the epilogue adds the shift, the residual and the ReLU.
//
//   k_stem         conv3x3(1 -> 32) + BN + ReLU straight from the (B,T,80) fbank   (HBM-write bound)
//   k_conv3x3      implicit GEMM on v_mfma_f32_32x32x2_f32: a workgroup owns TH x (32*TWT) output
//                  pixels x BN output channels; per 16-channel input block the (halo'd) input patch
//                  and the 9 x BN x 16 weight slab are staged in LDS once and reused by all 9 taps.
#include "common.h"
#ifndef PA_CONV_STORE_AUX   // cache-policy bits of the output stores (2 = nt): A/B aid, see profiles/r5_xcd_ranges.txt
#define PA_CONV_STORE_AUX 0
#endif

namespace pa {

// ---------------------------------------------------------------------------------------------
// stem: out[b][f][t][c] = relu( sum_{df,dt} w[c][df][dt] * fb[b][t+dt-1][f+df-1] + shift[c] )
// grid = (ceil(T/64), B), block = 256: thread = (time step p of 64, 8-channel group); a workgroup walks ALL mel
// rows of its 64 time steps.  The 72 weights + 8 shifts of a thread stay in registers, the (66 x F) slice of the
// fbank is staged once (coalesced along F, transposed in LDS), every step writes 2 KB contiguous per wave.  (One
// workgroup per 64 pixels of ONE mel row, as before, was 4.6 M workgroups per audio-hour, each reloading the
// weights and reading the fbank with a stride of F floats: 12 ms at 3.2 TB/s for a kernel that only has to write.)
// ---------------------------------------------------------------------------------------------
constexpr int STEM_TS = 64;             // time steps per workgroup
constexpr int STEM_LD = STEM_TS + 3;    // LDS row stride (odd: the transposing stores spread over the banks)

__global__ __launch_bounds__(256) void k_stem(const float* __restrict__ fb, int T, int F,
                                              const float* __restrict__ w9,   // [9][32] tap-major
                                              const float* __restrict__ shift,  // [32]
                                              float* __restrict__ out) {
  extern __shared__ float xs[];         // [(F + 2)][STEM_LD]: row f + 1 = mel bin f, column tt = time t0 - 1 + tt
  const int b = blockIdx.y, t0 = blockIdx.x * STEM_TS;
  const int tid = threadIdx.x;
  // zero the two halo rows, then stage: consecutive threads read consecutive mel bins of one frame
  for (int i = tid; i < 2 * STEM_LD; i += 256) xs[(i < STEM_LD ? 0 : (F + 1) * STEM_LD - STEM_LD) + i] = 0.f;
  const int n = (STEM_TS + 2) * F;
  for (int i0 = 0; i0 < n; i0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {       // 8 loads in flight per thread
      const int i = i0 + tid + 256 * k;
      const int tt = i / F, ff = i - tt * F, t = t0 + tt - 1;
      v[k] = (i < n && t >= 0 && t < T) ? fb[((long)b * T + t) * F + ff] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + tid + 256 * k;
      const int tt = i / F, ff = i - tt * F;
      if (i < n) xs[(ff + 1) * STEM_LD + tt] = v[k];
    }
  }
  // thread = (channel quad q, time steps pl and pl + 32): the 8 lanes of a time step store its 32 channels as ONE full
  // 128-byte line per instruction (round 6; before: 8 channels of one time step per thread = two 16-byte stores 32 bytes
  // apart per lane, half-filled sectors: 304 instead of 157 cycles per KB on the CU's store path,
  // profiles/r6_store_stagger_probe.txt).  Same products in the same order per output: bit-identical results.
  const int q4 = (tid & 7) * 4, pl = tid >> 3;
  float w[9][4], sh[4];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) w[k][c] = w9[k * 32 + q4 + c];
#pragma unroll
  for (int c = 0; c < 4; ++c) sh[c] = shift[q4 + c];
  __syncthreads();
  // rolling 3 x 3 windows of the two time steps: rows f - 1, f, f + 1 (LDS rows f, f + 1, f + 2), columns p .. p + 2
  float x0[2][3], x1[2][3], x2[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      x0[h][dt] = xs[0 * STEM_LD + pl + 32 * h + dt];
      x1[h][dt] = xs[1 * STEM_LD + pl + 32 * h + dt];
    }
  const bool ok0 = t0 + pl < T, ok1 = t0 + pl + 32 < T;
  float* o = out + ((long)b * F * T + t0 + pl) * 32 + q4;
  for (int f = 0; f < F; ++f) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) x2[h][dt] = xs[(f + 2) * STEM_LD + pl + 32 * h + dt];
    float* of = o + (long)f * T * 32;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float acc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = 0.f;
      // (df outer, dt inner: the summation order of the first kernel, so the results stay bit-identical)
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(x0[h][dt], w[dt][c], acc[c]);
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(x1[h][dt], w[3 + dt][c], acc[c]);
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(x2[h][dt], w[6 + dt][c], acc[c]);
      float4 o4;
      o4.x = fmaxf(acc[0] + sh[0], 0.f);
      o4.y = fmaxf(acc[1] + sh[1], 0.f);
      o4.z = fmaxf(acc[2] + sh[2], 0.f);
      o4.w = fmaxf(acc[3] + sh[3], 0.f);
      if (h == 0 ? ok0 : ok1) *reinterpret_cast<float4*>(of + 32 * 32 * h) = o4;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        x0[h][dt] = x1[h][dt];
        x1[h][dt] = x2[h][dt];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// conv3x3 (pad 1, stride S) + shift (+ residual) (+ ReLU), f32 MFMA implicit GEMM.
//   X  : [B][H][W][CIN]          Wg : [9][COUT][CIN]  (tap = 3*dy+dx, BN scale folded)
//   Y  : [B][Ho][Wo][COUT]       R  : residual, same shape as Y, or nullptr
//   grid = (ceil(Wo/TW) * ceil(Ho/TH), COUT/BN, B), block = 256.
// LDS images (row stride CB+4 = 20 floats = 5 x 16-B slots, conflict-free ds_read_b128):
//   patch: S=1: [PH][PW][20];  S=2: [PH][2 (column parity)][PWH][20]  (de-interleaved columns so that
//          the 32 lanes of an M-tile read consecutive entries)
//   wts  : [9][BN][20]
// MFMA k-order inside a 16-channel block: step q uses channel (lane>>5)*8 + q  (A and B alike).
// ---------------------------------------------------------------------------------------------
constexpr int CB = 16;
constexpr int CLD = CB + 4;

template <int S, int TH, int TWT>
struct ConvGeom {
  static constexpr int TW = 32 * TWT;
  static constexpr int PH = (TH - 1) * S + 3;
  static constexpr int PW = (TW - 1) * S + 3;
  // entries per column parity (S == 2): TW + 1, padded to 4 mod 8 -- the staging ds_write_b128 of 8 neighbouring lanes
  // writes two pixels of DIFFERENT parity planes, PWH * 20 floats apart: 16 mod 32 banks apart only then (33 entries:
  // 20 banks apart, 4 banks shared -- the 2-way conflicts of profiles/r3_pipeline_pmc_sq.txt)
  static constexpr int PWH = ((TW + 1 + 3) / 8) * 8 + 4;
  static constexpr int PATCH = S == 1 ? PH * PW * CLD : PH * 2 * PWH * CLD;
};

template <int S, int TH, int TWT, int BN, bool HAS_R>
__global__ __launch_bounds__(256, 2) void k_conv3x3(const float* __restrict__ X, int H, int W, int CIN,
                                                    const float* __restrict__ Wg,
                                                    const float* __restrict__ shift,
                                                    const float* __restrict__ R, float* __restrict__ Y,
                                                    int Ho, int Wo, int COUT, int relu, int tiles_w,
                                                    int tiles_hw, int n_tiles, int total_tiles, int xranges,
                                                    int* __restrict__ counters) {
  using G = ConvGeom<S, TH, TWT>;
  constexpr int MT = TH * TWT;   // 32-pixel M-tiles per workgroup
  constexpr int MPW = MT / 4;    // M-tiles per wave
  constexpr int NT = BN / 32;    // N-tiles (every wave covers all of them)
  static_assert(MT % 4 == 0, "M-tiles must split over 4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;
  float* wts = smem + G::PATCH;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  constexpr int NPF4 = G::PH * G::PW * (CB / 4);  // float4 slots of the patch
  constexpr int NP = (NPF4 + 255) / 256;
  constexpr int NWF4 = 9 * BN * (CB / 4);         // float4 slots of the weight slab
  constexpr int NW = (NWF4 + 255) / 256;
  constexpr int OOB = (int)0x80000000;

  // Persistent workgroup: tiles t = blockIdx.x, blockIdx.x + gridDim.x, ...  (t -> image b, channel
  // tile n0, pixel tile y0/x0).  The (tile, channel-block) iteration space is ONE software pipeline:
  // the global loads of the next stage -- the next 16-channel block of this tile or the first block of
  // the NEXT tile -- are issued into registers (buffer_load_dwordx4, wave-uniform descriptor, hardware
  // bounds check = free zero padding of the halo) before the MFMAs of the current stage and land in
  // LDS after them, so neither a tile's prologue nor its epilogue leaves the matrix pipe idle.
  struct Tile {
    int b, n0, y0, x0;
  };
  // XCD-aware tile order (as wino_decode, emb_winograd_geom.h): tile t runs on XCD t % 8 (tile_queue.h) and the
  // n_tiles workgroups that read the same input patch -- same image, same pixel tile, different cout slices -- are
  // consecutive claims of ONE XCD: the patch comes from HBM once and from that XCD's L2 n_tiles - 1 times.  (Until
  // round 5 the cout slice was the SLOWEST index: the slices of a patch ran a whole image stack apart and every one
  // fetched it from HBM -- 2.07x the algorithmic bytes, profiles/r4_traffic.json.)  (pixel tile, image) pair
  // pb = (r / n_tiles) * 8 + xcd; the index space is padded to whole XCD stripes, pairs >= num_pb are holes.
  const int num_pb = total_tiles / n_tiles;
  auto pair_of = [&](int t) {
    return xranges ? (t & 7) * ((num_pb + 7) >> 3) + (t >> 3) / n_tiles : ((t >> 3) / n_tiles) * 8 + (t & 7);
  };
  auto decode = [&](int t) {
    Tile q;
    const int pb = pair_of(t);
    q.n0 = (((t >> 3) % n_tiles)) * BN;
    const int pix = pb % tiles_hw;
    q.b = pb / tiles_hw;
    q.y0 = (pix / tiles_w) * TH;
    q.x0 = (pix % tiles_w) * G::TW;
    return q;
  };
  // a claim's raw counter value -> a real tile (holes are skipped by claiming again), or -1
  auto resolve = [&](const TileQueue& tqq, int r) {
    for (;;) {
      const int t = tq_resolve(tqq, r);
      if (t < 0 || pair_of(t) < num_pb) return t;
      r = tq_claim_own(tqq);
    }
  };
  u32x4 rp[NP], rw[NW];
  // Per-lane byte offsets of a tile's staging loads (halo / out-of-image / padding lanes out of bounds: the hardware
  // bounds check writes the zeros).  They do not depend on the channel block -- that is the scalar offset of the load
  // -- so they are computed once per TILE (the patch: two integer divisions and four compares per element) or once
  // per KERNEL (the weight slab), not once per stage: ~150 vector-ALU instructions per stage that were paid in
  // matrix time (the f32 MFMA shares the vector ALUs).  Stride 2 only: the stride-1 instantiations (fallback of the
  // Winograd kernels, Bottleneck ResNets) sit at the register limit and keep computing them in place.
  constexpr bool HOIST = S == 2;
  struct Offs {
    int p[NP];
  };
  int offw[NW];
#pragma unroll
  for (int e = 0; e < NW; ++e) {
    const int i = tid + 256 * e;
    const int c4 = i & 3, rn = i >> 2;  // rn = tap*BN + n
    const int tap = rn / BN, n = rn % BN;
    offw[e] = (NWF4 % 256 == 0 || i < NWF4) ? ((tap * COUT + n) * CIN + 4 * c4) * 4 : OOB;
  }
  auto tile_offsets = [&](const Tile& q, Offs& o) {
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int i = tid + 256 * e;
      const int c4 = i & 3, pp = i >> 2;
      const int py = pp / G::PW, px = pp % G::PW;
      const int iy = q.y0 * S - 1 + py, ix = q.x0 * S - 1 + px;
      o.p[e] = ((NPF4 % 256 == 0 || i < NPF4) && iy >= 0 && iy < H && ix >= 0 && ix < W)
                   ? ((iy * W + ix) * CIN + 4 * c4) * 4
                   : OOB;
    }
  };
  auto gload = [&](const Tile& q, const Offs& o, int c0) {
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X + (long)q.b * H * W * CIN), 0, H * W * CIN * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wg + (long)q.n0 * CIN), 0, (9 * COUT - q.n0) * CIN * 4, 0x00020000);
    Offs here;
    if (!HOIST) tile_offsets(q, here);
#pragma unroll
    for (int e = 0; e < NP; ++e)
      rp[e] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, HOIST ? o.p[e] : here.p[e], c0 * 4, 0);
#pragma unroll
    for (int e = 0; e < NW; ++e) rw[e] = __builtin_amdgcn_raw_buffer_load_b128(wsrd, offw[e], c0 * 4, 0);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int i = tid + 256 * e;
      const int c4 = i & 3, pp = i >> 2;
      const int py = pp / G::PW, px = pp % G::PW;
      int off;
      if (S == 1) off = pp * CLD + 4 * c4;
      else off = ((py * 2 + (px & 1)) * G::PWH + (px >> 1)) * CLD + 4 * c4;
      if (NPF4 % 256 == 0 || i < NPF4) *reinterpret_cast<u32x4*>(patch + off) = rp[e];
    }
#pragma unroll
    for (int e = 0; e < NW; ++e) {
      const int i = tid + 256 * e;
      if (NWF4 % 256 == 0 || i < NWF4)
        *reinterpret_cast<u32x4*>(wts + (i >> 2) * CLD + 4 * (i & 3)) = rw[e];
    }
  };

  // tiles are CLAIMED at run time (common.h: TileQueue), one tile ahead: thread 0 claims tile k+1 when
  // tile k starts and publishes it through LDS during tile k's first stage, in time for the prefetch of
  // its first channel block in tile k's last stage
  __shared__ int s_next;
  const TileQueue tq{counters, (int)(blockIdx.x & 7), ((num_pb + 7) >> 3) * n_tiles};
  if (tid == 0) s_next = resolve(tq, tq_claim_own(tq));
  __syncthreads();
  int t = s_next;
  if (t < 0) {
    if (tid == 0) tq_done(tq, gridDim.x);
    return;
  }
  Tile cur = decode(t);
  Offs oc, on;
  if (HOIST) tile_offsets(cur, oc);
  on = oc;
  gload(cur, oc, 0);
  for (;;) {
    int ahead = 0;
    if (tid == 0) ahead = tq_claim_own(tq);
    int tn = -1;
    f32x16 acc[MPW][NT];
#pragma unroll
    for (int i = 0; i < MPW; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Tile nxt = cur;

    for (int c0 = 0; c0 < CIN; c0 += CB) {
      __syncthreads();  // every wave is done reading the previous stage from LDS
      lstore();
      if (c0 == 0 && tid == 0) s_next = resolve(tq, ahead);
      __syncthreads();
      if (c0 == 0) {     // CIN >= 2 channel blocks: the next tile is known before the last stage
        tn = s_next;
        if (tn >= 0) {
          nxt = decode(tn);
          if (HOIST) tile_offsets(nxt, on);
        }
      }
      if (c0 + CB < CIN) gload(cur, oc, c0 + CB);
      else if (tn >= 0) gload(nxt, on, 0);
      // ---- 9 taps x 8 k-steps (dy stays a real loop: unrolling all 9 taps only buys register pressure).
      // Prefetching the fragments of tap+1 under the MFMAs of tap with a pinned schedule -- what the
      // Winograd kernel needs -- was measured neutral here (79.4 vs 79.5 ms per audio-hour): two
      // workgroups per CU already cover the LDS latency of this loop.
      float4 af[1][MPW][2], bf[1][NT][2];
      auto load_tap = [&](int tap, float4 (&a)[MPW][2], float4 (&bq)[NT][2]) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
          const int mt = wv * MPW + i;
          const int yy = mt / TWT, xt = mt % TWT;
          int off;
          if (S == 1) off = ((yy + dy) * G::PW + 32 * xt + li + dx) * CLD;
          else off = (((yy * 2 + dy) * 2 + (dx & 1)) * G::PWH + 32 * xt + li + (dx >> 1)) * CLD;
          a[i][0] = *reinterpret_cast<const float4*>(patch + off + kh * 8);
          a[i][1] = *reinterpret_cast<const float4*>(patch + off + kh * 8 + 4);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int off = (tap * BN + 32 * j + li) * CLD + kh * 8;
          bq[j][0] = *reinterpret_cast<const float4*>(wts + off);
          bq[j][1] = *reinterpret_cast<const float4*>(wts + off + 4);
        }
      };
      auto mfma_tap = [&](const float4 (&a)[MPW][2], const float4 (&bq)[NT][2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < MPW; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              acc[i][j] = MFMA32(a[i][h].x, bq[j][h].x, acc[i][j]);
              acc[i][j] = MFMA32(a[i][h].y, bq[j][h].y, acc[i][j]);
              acc[i][j] = MFMA32(a[i][h].z, bq[j][h].z, acc[i][j]);
              acc[i][j] = MFMA32(a[i][h].w, bq[j][h].w, acc[i][j]);
            }
      };
#pragma unroll 1
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          load_tap(dy * 3 + dx, af[0], bf[0]);
          mfma_tap(af[0], bf[0]);
        }
    }
    // ---- epilogue: lane holds channel n0 + 32j + li, pixels x = x0 + 32xt + (r&3) + 8(r>>2) + 4kh.
    // Branch-free buffer loads/stores (out-of-range pixels get an out-of-bounds offset: loads return
    // 0, stores are dropped), so the 16 residual loads of a tile are in flight together.
    {
      const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
          Y + (long)cur.b * Ho * Wo * COUT, 0, Ho * Wo * COUT * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(HAS_R ? R + (long)cur.b * Ho * Wo * COUT : Y), 0,
          Ho * Wo * COUT * 4, 0x00020000);
      // groups g = (M-tile i, N-tile j); the residual loads of group g+1 are issued before the stores
      // of group g (vmcnt retires in order), so the epilogue is one stream instead of a load->store
      // round trip per group
      constexpr int NG = MPW * NT;
      int off[2][16];
      float rv[2][16];
      auto goffs = [&](int g, int* o) {
        const int i = g / NT, j = g % NT;
        const int mt = wv * MPW + i;
        const int y = cur.y0 + mt / TWT, xbase = cur.x0 + 32 * (mt % TWT) + 4 * kh;
        const int n = cur.n0 + 32 * j + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = xbase + (r & 3) + 8 * (r >> 2);
          o[r] = (y < Ho && x < Wo) ? ((y * Wo + x) * COUT + n) * 4 : OOB;
        }
      };
      auto gres = [&](const int* o, float* v) {
        if (HAS_R) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrd, o[r], 0, 0));
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = 0.f;
        }
      };
      goffs(0, off[0]);
      gres(off[0], rv[0]);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
          goffs(g + 1, off[(g + 1) & 1]);
          gres(off[(g + 1) & 1], rv[(g + 1) & 1]);
        }
        const int i = g / NT, j = g % NT;
        const float sh = shift[cur.n0 + 32 * j + li];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + sh + rv[g & 1][r];
          if (relu) v = fmaxf(v, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), ysrd,
                                                off[g & 1][r], 0, PA_CONV_STORE_AUX);
        }
      }
    }
    if (tn < 0) break;
    cur = nxt;
    oc = on;
  }
  if (tid == 0) tq_done(tq, gridDim.x);
}

int xcd_ranges_wanted(bool by_default);   // emb_winograd4.hip

template <int S, int TH, int TWT, int BN, bool HAS_R>
static int launch_conv_r(const float* X, int B, int H, int W, int CIN, const float* Wg,
                         const float* shift, const float* R, float* Y, int COUT, int relu,
                         hipStream_t st) {
  using G = ConvGeom<S, TH, TWT>;
  const int Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
  const int tiles_w = cdiv(Wo, G::TW), tiles_h = cdiv(Ho, TH);
  const size_t lds = (size_t)(G::PATCH + 9 * BN * CLD) * sizeof(float);
  // workgroups the chip holds at once (2 per CU: LDS- and VGPR-bound); per DEVICE, like the attribute
  constexpr int MAXDEV = 16;
  static int resident_of[MAXDEV] = {0}, per_cu_of[MAXDEV] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= MAXDEV) dev = 0;
  if (!resident_of[dev]) {
    (void)hipFuncSetAttribute((const void*)k_conv3x3<S, TH, TWT, BN, HAS_R>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int cus = 256, per_cu = 2;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_conv3x3<S, TH, TWT, BN, HAS_R>, 256, lds) !=
            hipSuccess || per_cu < 1)
      per_cu = 2;
    resident_of[dev] = cus * per_cu;
    per_cu_of[dev] = per_cu;
  }
  // every resident workgroup is launched; tiles are claimed at run time (common.h: TileQueue)
  const int resident = resident_of[dev] & ~7;
  const int tiles_hw = tiles_w * tiles_h, n_tiles = COUT / BN;
  const long total = (long)tiles_hw * n_tiles * B;
  const int grid = (int)(total < resident ? total : resident);
  int* counters = tile_counters();
  if (counters == nullptr) {
    set_error("pa_conv3x3: cannot allocate the tile counters");
    return 2;
  }
  hipLaunchKernelGGL((k_conv3x3<S, TH, TWT, BN, HAS_R>), dim3(grid), dim3(256), lds, st, X, H, W, CIN, Wg,
                     shift, R, Y, Ho, Wo, COUT, relu, tiles_w, tiles_hw, n_tiles, (int)total, xcd_ranges_wanted(true),
                     counters);
  return 0;
}

template <int S, int TH, int TWT, int BN>
static int launch_conv(const float* X, int B, int H, int W, int CIN, const float* Wg,
                       const float* shift, const float* R, float* Y, int COUT, int relu,
                       hipStream_t st) {
  return R != nullptr ? launch_conv_r<S, TH, TWT, BN, true>(X, B, H, W, CIN, Wg, shift, R, Y, COUT, relu, st)
                      : launch_conv_r<S, TH, TWT, BN, false>(X, B, H, W, CIN, Wg, shift, R, Y, COUT, relu, st);
}

}  // namespace pa

extern "C" {

// conv1 + bn1 + relu of ResNet.forward (resnet.py:413-415), reading the (B,T,F) fbank directly
int pa_resnet_stem(const float* fbank, int B, int T, int F, const float* w9, const float* shift,
                   float* out, void* stream) {
  if (B <= 0) return 0;
  pa::ProfScope prof("k_stem", stream, 2.0 * B * T * F * 9 * 32, 4.0 * B * T * F * 33);
  PA_REQUIRE(F >= 1 && (size_t)(F + 2) * pa::STEM_LD * 4 <= 64 * 1024, "pa_resnet_stem: %d mel bins do not fit LDS", F);
  hipLaunchKernelGGL(pa::k_stem, dim3(pa::cdiv(T, pa::STEM_TS), B), dim3(256), (size_t)(F + 2) * pa::STEM_LD * 4,
                     (hipStream_t)stream, fbank, T, F, w9, shift, out);
  PA_CHECK_LAUNCH("pa_resnet_stem");
  return 0;
}

// BasicBlock convolutions (resnet.py:84-145): Y = [relu]( conv3x3_s(X) + shift [+ R] )
int pa_conv3x3(const float* X, int B, int H, int W, int cin, const float* Wg, const float* shift,
               const float* R, float* Y, int cout, int stride, int relu, void* stream) {
  if (B <= 0) return 0;
  PA_REQUIRE(cin % pa::CB == 0 && cout % 32 == 0, "pa_conv3x3: cin %% 16 and cout %% 32 required");
  hipStream_t st = (hipStream_t)stream;
  const int Ho = (H - 1) / stride + 1, Wo_ = (W - 1) / stride + 1;
  // algorithmic work: 2*9*cin*cout per output pixel; bytes: input + output (+ residual) + weights once
  pa::ProfScope prof("k_conv3x3", stream, 2.0 * 9 * cin * cout * (double)B * Ho * Wo_,
                     4.0 * ((double)B * H * W * cin + (double)B * Ho * Wo_ * cout * (R ? 2 : 1) + 9.0 * cin * cout));
  if (stride == 1) {
    if (cout == 32) pa::launch_conv<1, 8, 1, 32>(X, B, H, W, cin, Wg, shift, R, Y, cout, relu, st);
    else if (Ho >= 32) pa::launch_conv<1, 8, 1, 64>(X, B, H, W, cin, Wg, shift, R, Y, cout, relu, st);
    else if (Ho >= 16) pa::launch_conv<1, 4, 2, 64>(X, B, H, W, cin, Wg, shift, R, Y, cout, relu, st);
    else pa::launch_conv<1, 2, 2, 64>(X, B, H, W, cin, Wg, shift, R, Y, cout, relu, st);
  } else if (stride == 2) {
    PA_REQUIRE(cout % 64 == 0, "pa_conv3x3: stride 2 needs cout %% 64 == 0");
    // 32-cout tiles: two workgroups per CU (+4-7 % over 64-cout tiles, measured in round 1)
    if (Ho >= 16) pa::launch_conv<2, 4, 1, 32>(X, B, H, W, cin, Wg, shift, R, Y, cout, relu, st);
    else pa::launch_conv<2, 2, 2, 32>(X, B, H, W, cin, Wg, shift, R, Y, cout, relu, st);
  } else {
    PA_REQUIRE(false, "pa_conv3x3: stride %d not supported", stride);
  }
  PA_CHECK_LAUNCH("pa_conv3x3");
  return 0;
}

}  // extern "C"
