// Winograd F(2x2, 3x3) for the stride-1 BasicBlock convolutions of the WeSpeaker ResNet on gfx950
// (reference: models/embedding/wespeaker/resnet.py:84-145; what MIOpen, the reference's backend, also
//  selects for fp32 3x3 convolutions).  2.25x fewer multiplies than the direct form at true fp32:
//
//     Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A         per 2x2 output tile, 4x4 input patch d
//
//   * U = G g G^T (BatchNorm scale folded into g) is precomputed on the host in float64 -> [16][cout][cin]
//   * a workgroup (4 waves, 2 workgroups per CU) owns TR x 16*TCG Winograd tiles x 32 output channels;
//     wave w owns 16 consecutive tiles of one tile row: lane (t = lane & 15, g = lane >> 4) transforms
//     the 4x4 patch of tile t for the channel quad g IN REGISTERS (V = B^T d B, 32 adds per channel) --
//     V never touches LDS -- and feeds it straight to v_mfma_f32_16x16x4_f32 as the A operand: for each
//     of the 16 transform points xi, D[tile][cout] += V_xi[tile][cin] * U_xi[cin][cout];
//   * the accumulators of the 16 points (16 x 2 cout groups x 4 VGPRs) stay in registers over the whole
//     cin loop; the inverse transform A^T M A is lane-local because a lane holds all 16 points of its
//     (tile, cout) entries; epilogue = + BN shift (+ residual) (+ ReLU), branch-free buffer stores;
//   * staging is LDS-DMA (buffer_load_dwordx4 ... lds: global -> LDS without passing through VGPRs, the
//     hardware bounds check zero-fills the halo): the input patch, de-interleaved by column parity so
//     that the stride-2 tile walk reads consecutive LDS rows, and the U slab [16][32][16 cin].  Rows are
//     64 B (16 channels) unpadded; the 16-B quad of row r sits at slot (g + 2*((r>>2)&1)) & 3, which makes
//     every ds_read_b128 of 16 consecutive rows bank-conflict free (brute-forced over all alignments).
// Numerics: fp32 throughout; the transforms only add/subtract and the 1/2 factors of G are applied in
// float64 on the host; error ~3x the direct form's (tests: |err| <= 1e-4 max|ref| per conv, end to end).
#include <stdlib.h>

#include "common.h"

namespace pa {

constexpr int WCB = 16;   // input channels per stage (one 64-B LDS row)
constexpr int W_BN = 32;  // output channels per workgroup
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int TR, int TCG>
struct WinoGeom {
  static constexpr int NWV = TR * TCG;            // waves per workgroup: one tile row x 16 tile cols each
  static_assert(NWV == 4 || NWV == 8, "4 or 8 waves");
  static constexpr int THREADS = 64 * NWV;
  static constexpr int PH = 2 * TR + 2;          // patch rows
  static constexpr int PW = 2 * 16 * TCG + 2;    // patch cols
  static constexpr int PWH = PW / 2;             // entries per column parity
  static constexpr int PROWS = PH * 2 * PWH;     // LDS rows of the patch
  static constexpr int PINSTR = (PROWS + 15) / 16;  // 1-KB DMA pieces
  static constexpr int PATCH = PINSTR * 16 * WCB;   // floats
  static constexpr int UINSTR = 16 * W_BN / 16;     // 32 pieces
  static constexpr int USLAB = 16 * W_BN * WCB;     // floats
};

// physical 16-B slot of logical channel quad g in LDS row r
__device__ __forceinline__ int wslot(int r, int g) { return (g + 2 * ((r >> 2) & 1)) & 3; }

struct WinoTile {
  int b, n0, y0, x0, valid;
};

// LDS-DMA of one stage (16 input channels from c0) of tile q:
// piece k fills LDS rows 16k .. 16k+15; lane l -> row 16k + (l>>2), physical slot l&3.
// BLK (measurement aid for the round-2 layout study): read X as channel-blocked [c/16][h][w][16]
template <int TR, int TCG, bool BLK = false>
__device__ __forceinline__ void wino_issue_patch(const float* __restrict__ X, int H, int W, int CIN,
                                                 const WinoTile& q, int c0, float* patch, int lane,
                                                 int wv) {
  using G = WinoGeom<TR, TCG>;
  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(X + (long)q.b * H * W * CIN), 0, H * W * CIN * 4, 0x00020000);
  const int swv = __builtin_amdgcn_readfirstlane(wv);
#pragma unroll
  for (int i = 0; i < (G::PINSTR + G::NWV - 1) / G::NWV; ++i) {
    const int k = swv + G::NWV * i;
    if (k >= G::PINSTR) break;  // wave-uniform
    const int row = 16 * k + (lane >> 2);
    const int gq = ((lane & 3) - 2 * ((row >> 2) & 1)) & 3;  // logical quad stored in this slot
    const int pr = row / G::PWH, idx = row % G::PWH;         // pr = py*2 + parity
    const int py = pr >> 1, px = 2 * idx + (pr & 1);
    const int iy = q.y0 - 1 + py, ix = q.x0 - 1 + px;
    const bool inb = row < G::PROWS && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const int off = !inb ? OOB : BLK ? ((iy * W + ix) * 16 + 4 * gq) * 4 : ((iy * W + ix) * CIN + 4 * gq) * 4;
    const int soff = BLK ? (c0 / 16) * H * W * 64 : c0 * 4;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)(patch + 256 * k), 16, off, soff, 0, 0);
  }
}

// Patch DMA with the per-lane address arithmetic hoisted out of the stage loop: for every piece this
// wave issues, `prel` = byte offset of the lane's (patch row, quad) relative to the patch origin and
// `pyx` = (patch y << 16) | patch x are computed ONCE per kernel (they contain the divisions by the
// patch width); a stage only adds the tile origin and checks the image bounds (2 unsigned compares).
template <int TR, int TCG>
struct WinoPatchLanes {
  static constexpr int NPP = (WinoGeom<TR, TCG>::PINSTR + WinoGeom<TR, TCG>::NWV - 1) / WinoGeom<TR, TCG>::NWV;
  int prel[NPP], pyx[NPP];
};

template <int TR, int TCG>
__device__ __forceinline__ void wino_patch_lanes(WinoPatchLanes<TR, TCG>& pl, int W, int CIN, int lane,
                                                 int swv) {
  using G = WinoGeom<TR, TCG>;
#pragma unroll
  for (int i = 0; i < WinoPatchLanes<TR, TCG>::NPP; ++i) {
    const int k = swv + G::NWV * i;
    const int row = 16 * k + (lane >> 2);
    const int gq = ((lane & 3) - 2 * ((row >> 2) & 1)) & 3;  // logical quad stored in this slot
    const int pr = row / G::PWH, idx = row % G::PWH;         // pr = py*2 + parity
    const int py = pr >> 1, px = 2 * idx + (pr & 1);
    const bool real = k < G::PINSTR && row < G::PROWS;
    pl.prel[i] = ((py * W + px) * CIN + 4 * gq) * 4;
    pl.pyx[i] = real ? ((py << 16) | px) : 0x7fff7fff;       // (never inside an image)
  }
}

template <int TR, int TCG>
__device__ __forceinline__ void wino_issue_patch_fast(const float* __restrict__ X, int H, int W, int CIN,
                                                      const WinoTile& q, int c0, float* patch,
                                                      const int* prel, const int* pyx, int swv) {
  using G = WinoGeom<TR, TCG>;
  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(X + (long)q.b * H * W * CIN), 0, H * W * CIN * 4, 0x00020000);
  const int tile_rel = ((q.y0 - 1) * W + (q.x0 - 1)) * CIN * 4;
#pragma unroll
  for (int i = 0; i < WinoPatchLanes<TR, TCG>::NPP; ++i) {
    const int k = swv + G::NWV * i;
    if (k >= G::PINSTR) break;  // wave-uniform
    const int iy = q.y0 - 1 + (pyx[i] >> 16), ix = q.x0 - 1 + (pyx[i] & 0xffff);
    const bool inb = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)(patch + 256 * k), 16,
                                             inb ? prel[i] + tile_rel : OOB, c0 * 4, 0, 0);
  }
}

template <int TR, int TCG>
__device__ __forceinline__ void wino_issue_u(const float* __restrict__ U, int CIN, int COUT,
                                             const WinoTile& q, int c0, float* uslab, int lane, int wv) {
  using G = WinoGeom<TR, TCG>;
  const __amdgpu_buffer_rsrc_t usrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(U + (long)q.n0 * CIN), 0, (16 * COUT - q.n0) * CIN * 4, 0x00020000);
  // row = 16k + r (r = lane >> 2) = xi * 32 + n  ->  xi = k >> 1, n = 16 (k & 1) + r: the k-dependent part of
  // the address is wave-uniform and goes into the scalar offset; one VGPR serves all pieces
  const int r = lane >> 2;
  const int gq = ((lane & 3) - 2 * ((r >> 2) & 1)) & 3;  // logical quad stored in this slot
  const int voff = (r * CIN + 4 * gq) * 4;
  const int swv = __builtin_amdgcn_readfirstlane(wv);
#pragma unroll
  for (int i = 0; i < G::UINSTR / G::NWV; ++i) {
    const int k = swv + G::NWV * i;
    const int soff = (c0 + ((k >> 1) * COUT + 16 * (k & 1)) * CIN) * 4;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(usrd, (lds_ptr_t)(uslab + 256 * k), 16, voff, soff, 0, 0);
  }
}

template <int TR, int TCG>
__device__ __forceinline__ void wino_issue(const float* __restrict__ X, const float* __restrict__ U,
                                           int H, int W, int CIN, int COUT, const WinoTile& q, int c0,
                                           float* patch, float* uslab, int lane, int wv) {
  wino_issue_patch<TR, TCG>(X, H, W, CIN, q, c0, patch, lane, wv);
  wino_issue_u<TR, TCG>(U, CIN, COUT, q, c0, uslab, lane, wv);
}

// input transform V = B^T d B of tile (wr, 16*wc + t) for channels 4g..4g+3, in registers
template <int TR, int TCG>
__device__ __forceinline__ void wino_transform(const float* patch, f32x4 (&v)[4][4], int t, int g, int wr,
                                               int wc) {
  using G = WinoGeom<TR, TCG>;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = ((2 * wr + i) * 2 + (j & 1)) * G::PWH + 16 * wc + t + (j >> 1);
      d[j] = *reinterpret_cast<const f32x4*>(patch + row * WCB + 4 * wslot(row, g));
    }
    v[i][0] = d[0] - d[2];
    v[i][1] = d[1] + d[2];
    v[i][2] = d[2] - d[1];
    v[i][3] = d[1] - d[3];
  }
#pragma unroll
  for (int bb = 0; bb < 4; ++bb) {
    const f32x4 r0 = v[0][bb], r1 = v[1][bb], r2 = v[2][bb], r3 = v[3][bb];
    v[0][bb] = r0 - r2;
    v[1][bb] = r1 + r2;
    v[2][bb] = r2 - r1;
    v[3][bb] = r1 - r3;
  }
}

// 16 transform points x 2 cout groups x 4 k-steps; B fragments PF points ahead of their MFMAs.
// TUNE bit 0: PF = 2 instead of 1; bit 1: raise the wave priority for the MFMA section.
template <int TUNE>
__device__ __forceinline__ void wino_mfma(const float* uslab, const f32x4 (&v)[4][4], f32x4 (&acc)[16][2],
                                          int t, int g) {
  constexpr int PF = (TUNE & 1) ? 2 : 1;
  f32x4 bf[PF + 1][2];
  // row = 32 xi + 16 cg + t: its swizzle bit ((row >> 2) & 1) = (t >> 2) & 1 does not depend on xi / cg,
  // so every read is (lane base) + compile-time offset -> folded into the ds_read immediate
  const float* ub = uslab + t * WCB + 4 * wslot(t, g);
  auto load = [&](int xi) {
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
      bf[xi % (PF + 1)][cg] = *reinterpret_cast<const f32x4*>(ub + (xi * W_BN + 16 * cg) * WCB);
  };
#pragma unroll
  for (int xi = 0; xi < PF; ++xi) load(xi);
  if (TUNE & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) {
    if (xi + PF < 16 && !(TUNE & 32)) load(xi + PF);
    const f32x4 av = v[xi >> 2][xi & 3];
    // alternate the two accumulators: a 16x16x4 MFMA issues every 32 cycles but its result is ready
    // after 40, so back-to-back MFMAs on ONE accumulator would stall 8 cycles each
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      acc[xi][0] = MFMA16(av[ks], bf[(TUNE & 32) ? 0 : xi % (PF + 1)][0][ks], acc[xi][0]);
      acc[xi][1] = MFMA16(av[ks], bf[(TUNE & 32) ? 0 : xi % (PF + 1)][1][ks], acc[xi][1]);
    }
  }
  if (TUNE & 2) __builtin_amdgcn_s_setprio(0);
}

template <int TR, int TCG, int TUNE>
__device__ __forceinline__ void wino_compute(const float* patch, const float* uslab, f32x4 (&acc)[16][2],
                                             int t, int g, int wr, int wc) {
  f32x4 v[4][4];
  wino_transform<TR, TCG>(patch, v, t, g, wr, wc);
  wino_mfma<TUNE>(uslab, v, acc, t, g);
}

// inverse transform + epilogue.  acc[xi][cg][r]: cout n0 + 16cg + t, tile column 16wc + 4g + r of tile
// row wr: outputs (y0 + 2wr + p, x0 + 2(16wc + 4g + r) + qq), p, qq in {0, 1}.
template <bool HAS_R>
__device__ __forceinline__ void wino_epilogue(const f32x4 (&acc)[16][2], const WinoTile& q, int H, int W,
                                              int COUT, const float* __restrict__ shift,
                                              const float* __restrict__ R, float* __restrict__ Y,
                                              int relu, int t, int g, int wr, int wc) {
  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
      Y + (long)q.b * H * W * COUT, 0, H * W * COUT * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(HAS_R ? R + (long)q.b * H * W * COUT : Y), 0, H * W * COUT * 4, 0x00020000);
  constexpr int NG = 8;  // groups (cg, r) of 4 outputs
  int off[2][4];
  float rv[2][4];
  auto goffs = [&](int gi, int* o) {
    const int cg = gi >> 2, r = gi & 3;
    const int n = q.n0 + 16 * cg + t;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int y = q.y0 + 2 * wr + (e >> 1), x = q.x0 + 2 * (16 * wc + 4 * g + r) + (e & 1);
      o[e] = (y < H && x < W) ? ((y * W + x) * COUT + n) * 4 : OOB;
    }
  };
  auto gres = [&](const int* o, float* vv) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      vv[e] = HAS_R ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrd, o[e], 0, 0)) : 0.f;
  };
  goffs(0, off[0]);
  gres(off[0], rv[0]);
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    if (gi + 1 < NG) {
      goffs(gi + 1, off[(gi + 1) & 1]);
      gres(off[(gi + 1) & 1], rv[(gi + 1) & 1]);
    }
    const int cg = gi >> 2, r = gi & 3;
    const float sh = shift[q.n0 + 16 * cg + t];
    float s[4], dd[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      s[a] = acc[4 * a + 0][cg][r] + acc[4 * a + 1][cg][r] + acc[4 * a + 2][cg][r];
      dd[a] = acc[4 * a + 1][cg][r] - acc[4 * a + 2][cg][r] - acc[4 * a + 3][cg][r];
    }
    float o4[4];
    o4[0] = s[0] + s[1] + s[2];
    o4[1] = dd[0] + dd[1] + dd[2];
    o4[2] = s[1] - s[2] - s[3];
    o4[3] = dd[1] - dd[2] - dd[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float vv = o4[e] + sh + rv[gi & 1][e];
      if (relu) vv = fmaxf(vv, 0.f);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, vv), ysrd, off[gi & 1][e], 0, 0);
    }
  }
}

// Tile order is XCD-aware: workgroup w runs on XCD w % 8 (each XCD has its own L2), so tile q is decoded
// as xcd = q % 8, r = q / 8 with the cout tile FASTEST in r: the n_tiles workgroups that read the same
// input patch (same image, same pixel tile, different 32-cout slices) run side by side on ONE XCD and
// the patch comes from HBM once instead of n_tiles times.  (pixel tile, image) = (r / n_tiles) * 8 + xcd;
// the index space is padded to a multiple of 8 (pixel tile, image) pairs: padding tiles are computed on
// the last real tile's data and not stored (valid = 0).
__device__ __forceinline__ WinoTile wino_decode(int q, int tiles_w, int tiles_hw, int n_tiles, int th,
                                                int tw, int num_pb) {
  WinoTile o;
  const int xcd = q & 7, r = q >> 3;
  int pb = (r / n_tiles) * 8 + xcd;
  o.n0 = (r % n_tiles) * W_BN;
  o.valid = pb < num_pb;
  pb = pb < num_pb ? pb : num_pb - 1;
  const int pix = pb % tiles_hw;
  o.b = pb / tiles_hw;
  o.y0 = (pix / tiles_w) * th;
  o.x0 = (pix % tiles_w) * tw;
  return o;
}

// MODE 0: one LDS buffer, two workgroups per CU cover each other's DMA waits.
// MODE 1: two LDS buffers, one workgroup per CU; the DMA of stage s+1 (next channel block, or the
//         first block of the next tile) flies under the MFMAs of stage s; one barrier per stage.
// MODE 2: one LDS buffer, two workgroups per CU, but the two halves of the buffer are recycled at
//         different times: the patch is only read by the input transform, so the patch DMA of stage s+1
//         is issued right after the transform of stage s and flies under its MFMAs; the (L2-resident) U
//         slab of stage s is issued at the top of the stage and lands while the transform runs.
// (measured on the ResNet34 shapes: MODE 0 beats MODE 1 by 10-25 % -- two waves per SIMD hide the LDS
//  and DMA latencies better than one wave with a prefetch)
template <int TR, int TCG, bool HAS_R, int MODE_TUNE>
__global__ __launch_bounds__(64 * TR * TCG, ((MODE_TUNE & 3) == 1 && TR * TCG == 4) ? 1 : 2) void k_conv3x3_wino(
    const float* __restrict__ X, int H, int W, int CIN, const float* __restrict__ U,
    const float* __restrict__ shift, const float* __restrict__ R, float* __restrict__ Y, int COUT,
    int relu, int tiles_w, int tiles_hw, int n_tiles, int total_tiles, int num_pb) {
  using G = WinoGeom<TR, TCG>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF = G::PATCH + G::USLAB;
  constexpr int MODE = MODE_TUNE & 3, TUNE = MODE_TUNE >> 2;
  constexpr bool DB = MODE == 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int t = lane & 15, g = lane >> 4;
  const int wr = wv / TCG, wc = wv % TCG;  // tile row / 16-tile column group of this wave

  int q = blockIdx.x;
  if (q >= total_tiles) return;
  WinoTile cur = wino_decode(q, tiles_w, tiles_hw, n_tiles, 2 * TR, 32 * TCG, num_pb);
  int stage = 0;
  const int swv = __builtin_amdgcn_readfirstlane(wv);
  WinoPatchLanes<TR, TCG> plan;
  if constexpr (MODE == 0) wino_patch_lanes<TR, TCG>(plan, W, CIN, lane, swv);
  f32x4 dbg_v[4][4];  // (measurement aids only)
  if (TUNE & 48)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) dbg_v[i][j] = f32x4{1.f, 1.f, 1.f, 1.f};
  if (DB) wino_issue<TR, TCG>(X, U, H, W, CIN, COUT, cur, 0, smem, smem + G::PATCH, lane, wv);
  if (MODE == 2) wino_issue_patch<TR, TCG>(X, H, W, CIN, cur, 0, smem, lane, wv);
  for (; q < total_tiles; q += gridDim.x) {
    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[xi][cg][r] = 0.f;
    const int qn = q + gridDim.x;
    WinoTile nxt = cur;
    if (qn < total_tiles) nxt = wino_decode(qn, tiles_w, tiles_hw, n_tiles, 2 * TR, 32 * TCG, num_pb);

    for (int c0 = 0; c0 < CIN; c0 += (MODE == 3 ? 2 * WCB : WCB), ++stage) {
      if (MODE == 3) {
        // 32-channel patch stage: both 16-channel halves of every pixel (one full 128-byte line in
        // NHWC) are fetched back to back into two LDS planes; the U slab is staged per half.
        float* p0 = smem;
        float* p1 = smem + G::PATCH;
        float* us = smem + 2 * G::PATCH;
        __syncthreads();  // every wave is done reading the previous stage
        wino_issue_patch<TR, TCG>(X, H, W, CIN, cur, c0, p0, lane, wv);
        wino_issue_patch<TR, TCG>(X, H, W, CIN, cur, c0 + WCB, p1, lane, wv);
        wino_issue_u<TR, TCG>(U, CIN, COUT, cur, c0, us, lane, wv);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        wino_compute<TR, TCG, TUNE>(p0, us, acc, t, g, wr, wc);
        __syncthreads();  // every wave is done reading the U slab
        wino_issue_u<TR, TCG>(U, CIN, COUT, cur, c0 + WCB, us, lane, wv);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        wino_compute<TR, TCG, TUNE>(p1, us, acc, t, g, wr, wc);
      } else if (MODE == 2) {
        float* us = smem + G::PATCH;
        // (all waves passed the barrier that ends the previous stage: the U slab is free)
        wino_issue_u<TR, TCG>(U, CIN, COUT, cur, c0, us, lane, wv);
        // vmcnt retires in order and every wave issues exactly UINSTR/4 U pieces: everything older --
        // this wave's patch pieces of this stage -- has landed once only those remain outstanding
        __builtin_amdgcn_s_waitcnt(0x0F70 | (G::UINSTR / G::NWV));
        __syncthreads();  // the whole patch has landed
        f32x4 v[4][4];
        wino_transform<TR, TCG>(smem, v, t, g, wr, wc);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // this wave's U pieces landed (they flew under the transform)
        __syncthreads();  // whole U slab landed; every wave is done reading the patch
        if (c0 + WCB < CIN) wino_issue_patch<TR, TCG>(X, H, W, CIN, cur, c0 + WCB, smem, lane, wv);
        else if (qn < total_tiles) wino_issue_patch<TR, TCG>(X, H, W, CIN, nxt, 0, smem, lane, wv);
        wino_mfma<TUNE>(us, v, acc, t, g);
        __syncthreads();  // every wave is done reading the U slab
      } else if (DB) {
        float* pb = smem + (stage & 1) * BUF;
        float* nb = smem + ((stage + 1) & 1) * BUF;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of stage `stage` landed
        __syncthreads();  // ... everybody's did, and everybody finished reading the other buffer
        if (c0 + WCB < CIN) wino_issue<TR, TCG>(X, U, H, W, CIN, COUT, cur, c0 + WCB, nb, nb + G::PATCH, lane, wv);
        else if (qn < total_tiles) wino_issue<TR, TCG>(X, U, H, W, CIN, COUT, nxt, 0, nb, nb + G::PATCH, lane, wv);
        wino_compute<TR, TCG, TUNE>(pb, pb + G::PATCH, acc, t, g, wr, wc);
      } else {
        // TUNE bits 2..5 are measurement aids (results invalid): 4 = stage LDS only once, 8 = no barriers,
        // 16 = input transform only once, 32 = B fragments read only once
        if (!(TUNE & 8) || stage == 0) __syncthreads();  // every wave is done reading the previous stage
        // (64 = patch staged only once, 128 = U slab staged only once)
        if ((!(TUNE & 4) && !(TUNE & 64)) || stage == 0) {
          if constexpr (MODE != 0 || (TUNE & 256) != 0)
            wino_issue_patch<TR, TCG, (TUNE & 256) != 0>(X, H, W, CIN, cur, c0, smem, lane, wv);
          else
            wino_issue_patch_fast<TR, TCG>(X, H, W, CIN, cur, c0, smem, plan.prel, plan.pyx, swv);
        }
        if ((!(TUNE & 4) && !(TUNE & 128)) || stage == 0)
          wino_issue_u<TR, TCG>(U, CIN, COUT, cur, c0, smem + G::PATCH, lane, wv);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (!(TUNE & 8) || stage == 0) __syncthreads();
        if (TUNE & 48) {
          if (!(TUNE & 16) || stage == 0) wino_transform<TR, TCG>(smem, dbg_v, t, g, wr, wc);
          wino_mfma<TUNE>(smem + G::PATCH, dbg_v, acc, t, g);
        } else {
          wino_compute<TR, TCG, TUNE>(smem, smem + G::PATCH, acc, t, g, wr, wc);
        }
      }
    }
    if (cur.valid) wino_epilogue<HAS_R>(acc, cur, H, W, COUT, shift, R, Y, relu, t, g, wr, wc);
    cur = nxt;
  }
}

template <int TR, int TCG, bool HAS_R, int MODE_TUNE>
static int launch_wino_r(const float* X, int B, int H, int W, int CIN, const float* U, const float* shift,
                         const float* R, float* Y, int COUT, int relu, hipStream_t st) {
  using G = WinoGeom<TR, TCG>;
  const int tiles_w = cdiv(W, 32 * TCG), tiles_h = cdiv(H, 2 * TR);
  constexpr int MODE = MODE_TUNE & 3;
  const size_t lds = (size_t)(MODE == 3 ? 2 * G::PATCH + G::USLAB : (G::PATCH + G::USLAB) * (MODE == 1 ? 2 : 1)) *
                     sizeof(float);
  static int resident = 0;
  if (!resident) {
    (void)hipFuncSetAttribute((const void*)k_conv3x3_wino<TR, TCG, HAS_R, MODE_TUNE>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_conv3x3_wino<TR, TCG, HAS_R, MODE_TUNE>, G::THREADS,
                                                     lds) != hipSuccess || per_cu < 1)
      per_cu = MODE == 1 ? 1 : 2;
    resident = cus * per_cu;
  }
  const int tiles_hw = tiles_w * tiles_h, n_tiles = COUT / W_BN;
  const long num_pb = (long)tiles_hw * B;                    // (pixel tile, image) pairs
  const long total = ((num_pb + 7) / 8) * 8 * n_tiles;       // padded to whole XCD stripes
  const int grid = (int)(total < resident ? total : resident);
  hipLaunchKernelGGL((k_conv3x3_wino<TR, TCG, HAS_R, MODE_TUNE>), dim3(grid), dim3(G::THREADS), lds, st, X, H, W, CIN,
                     U, shift, R, Y, COUT, relu, tiles_w, tiles_hw, n_tiles, (int)total, (int)num_pb);
  return 0;
}

template <int TR, int TCG>
static int launch_wino(const float* X, int B, int H, int W, int CIN, const float* U, const float* shift,
                       const float* R, float* Y, int COUT, int relu, hipStream_t st) {
  // tuning aid: PA_WINO_MODE = 0 single LDS buffer, 2 workgroups per CU | 1 double buffer, 1 WG/CU |
  //             2 = 0 + split recycling of patch / U.  Measured on the full pipeline: 0 is the fastest
  //             (0.755 vs 0.733 audio-h/s for 2 and 0.627 for 1), hence the default.
  static const int mode = getenv("PA_WINO_MODE") ? atoi(getenv("PA_WINO_MODE")) : 0;
  using G = WinoGeom<TR, TCG>;
  if (mode == 3 && (2 * G::PATCH + G::USLAB) * sizeof(float) * 2 <= 160 * 1024 && CIN % 32 == 0)
    return R != nullptr ? launch_wino_r<TR, TCG, true, 3>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, st)
                        : launch_wino_r<TR, TCG, false, 3>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, st);
  if (mode == 2)
    return R != nullptr ? launch_wino_r<TR, TCG, true, 2>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, st)
                        : launch_wino_r<TR, TCG, false, 2>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, st);
  if (mode == 1)
    return R != nullptr ? launch_wino_r<TR, TCG, true, 1>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, st)
                        : launch_wino_r<TR, TCG, false, 1>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, st);
  static const int tune = getenv("PA_WINO_TUNE") ? atoi(getenv("PA_WINO_TUNE")) : 0;
#define PA_WINO_GO(MT)                                                                                      \
  return R != nullptr ? launch_wino_r<TR, TCG, true, MT>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, st)  \
                      : launch_wino_r<TR, TCG, false, MT>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, st)
  if (tune == 1) PA_WINO_GO(4);
  if (tune == 2) PA_WINO_GO(8);
  if (tune == 3) PA_WINO_GO(12);
  if (tune == 4) PA_WINO_GO(16);
  if (tune == 12) PA_WINO_GO(48);
  if (tune == 28) PA_WINO_GO(112);
  if (tune == 60) PA_WINO_GO(240);
  if (tune == 64) PA_WINO_GO(256);
  if (tune == 128) PA_WINO_GO(512);
  if (tune == 256) PA_WINO_GO(1024);
  PA_WINO_GO(0);
#undef PA_WINO_GO
}

template <int TR, int TCG>
static int launch_wino8(const float* X, int B, int H, int W, int CIN, const float* U, const float* shift,
                        const float* R, float* Y, int COUT, int relu, hipStream_t st) {
  return R != nullptr ? launch_wino_r<TR, TCG, true, 1>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, st)
                      : launch_wino_r<TR, TCG, false, 1>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, st);
}

}  // namespace pa

extern "C" {

// conv3x3, stride 1, pad 1, via Winograd F(2x2,3x3): Y = [relu](conv(X) + shift [+ R]).
// U: [16][cout][cin] = G g G^T (xi = 4a + b), BatchNorm scale folded.
int pa_conv3x3_wino(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                    const float* R, float* Y, int cout, int relu, void* stream) {
  if (B <= 0) return 0;
  PA_REQUIRE(cin % pa::WCB == 0 && cout % pa::W_BN == 0, "pa_conv3x3_wino: cin %% 16 and cout %% 32 required");
  // algorithmic work = the direct convolution's (the reference's operation), not the reduced multiply count
  pa::ProfScope prof("k_conv3x3_wino", stream, 2.0 * 9 * cin * cout * (double)B * H * W,
                     4.0 * ((double)B * H * W * cin + (double)B * H * W * cout * (R ? 2 : 1) + 9.0 * cin * cout));
  hipStream_t st = (hipStream_t)stream;
  // workgroup tile = 2*TR x 32*TCG output pixels: pick the shape that wastes the fewest rows.
  // PA_WINO_WG8=1 (tuning aid): tall maps use 8-wave workgroups with double-buffered LDS (one per CU, two
  // waves per SIMD, DMA of the next stage under the MFMAs, U slab shared by twice as many tiles).
  static const int wg8 = getenv("PA_WINO_WG8") ? atoi(getenv("PA_WINO_WG8")) : 0;
  if (wg8 && H % 16 == 0) pa::launch_wino8<8, 1>(X, B, H, W, cin, U, shift, R, Y, cout, relu, st);
  else if (wg8 && H % 8 == 0) pa::launch_wino8<4, 2>(X, B, H, W, cin, U, shift, R, Y, cout, relu, st);
  else if (H % 8 == 0 || H > 24) pa::launch_wino<4, 1>(X, B, H, W, cin, U, shift, R, Y, cout, relu, st);
  else if (H % 4 == 0 || H > 12) pa::launch_wino<2, 2>(X, B, H, W, cin, U, shift, R, Y, cout, relu, st);
  else pa::launch_wino<1, 4>(X, B, H, W, cin, U, shift, R, Y, cout, relu, st);
  PA_CHECK_LAUNCH("pa_conv3x3_wino");
  return 0;
}

}  // extern "C"
