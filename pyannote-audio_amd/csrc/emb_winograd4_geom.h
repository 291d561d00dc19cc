// Integer geometry of the Winograd F(4x4, 3x3) convolution kernel (csrc/emb_winograd4.hip): LDS layouts, the
// per-lane DMA offsets with their halo class bits, the transform's read addresses.  No HIP types: the header is
// also compiled for the HOST by tests/test_winograd4_geometry_cpu.py, which replays the DMA and the transform reads
// of every wave and lane and checks that they agree (and that the reads are bank-conflict free).
// Needs: __device__, __forceinline__, and emb_winograd_geom.h (WinoTile, the class bits, wino_decode).
#pragma once

namespace pa {

// Workgroup tile: 8 x 128 output pixels x 32 output channels = 2 rows x 32 columns of 4x4 Winograd tiles; wave w
// owns tile row wr = w >> 1 and the 16 tile columns 16 wc .., wc = w & 1.  A stage stages 8 input channels:
//   patch  10 x 130 input pixels x 8 channels (one pixel = one 32-B LDS row), de-interleaved by column mod 4 so
//          that the 16 lanes of a tile row, 4 pixels apart, read CONSECUTIVE rows: row = (py*4 + px%4)*33 + px/4;
//   U slab 36 points x 32 output channels x 8 input channels: row = 32 xi + n (32 B = the 8 input channels).
struct Wino4Geom {
  static constexpr int CB = 8;                       // input channels per stage
  static constexpr int TH = 8, TW = 128;             // output pixels per workgroup tile
  static constexpr int PH = TH + 2, PW = TW + 2;     // patch
  static constexpr int PWQ = (PW + 3) / 4;           // 33 entries per column residue
  static constexpr int PROWS = PH * 4 * PWQ;         // 1320 LDS rows of 32 B
  static constexpr int PINSTR = (PROWS + 31) / 32;   // 42 DMA pieces of 1 KB (32 rows)
  static constexpr int NPP = (PINSTR + 3) / 4;       // 11 per wave
  static constexpr int PATCH_BYTES = PINSTR * 1024;  // 43 008
  static constexpr int UINSTR = 36;                  // 36 x 1 KB
  static constexpr int USLAB_BYTES = 36 * 32 * CB * 4;   // 36 864
  static constexpr int BUF_BYTES = PATCH_BYTES + USLAB_BYTES;   // one stage: 79 872 (two buffers: 159 744)
};

// Patch DMA: piece k fills LDS rows 32k .. 32k+31; lane l -> row 32k + (l >> 1), channel quad l & 1.
// `prel` = byte offset of the lane's (patch pixel, quad) from the patch origin (y0 - 1, x0 - 1) + class bits (see
// emb_winograd_geom.h): top halo row, left halo column, every column at or right of the image border in the LAST
// column tile (F(4x4) mixes all six patch columns into every output of a tile: columns past the border must be
// zeros, not the next row's pixels), padding lanes.
__device__ __forceinline__ void wino4_patch_lanes(int* prel, int W, int CIN, int lane, int slw, int x0_last) {
  using G = Wino4Geom;
#pragma unroll
  for (int i = 0; i < G::NPP; ++i) {
    const int k = slw + 4 * i;
    const int row = 32 * k + (lane >> 1);
    const int pr = row / G::PWQ, idx = row % G::PWQ;   // pr = py*4 + residue
    const int py = pr >> 2, px = 4 * idx + (pr & 3);
    const bool real = k < G::PINSTR && row < G::PROWS && px < G::PW;
    int v = ((py * W + px) * CIN + 4 * (lane & 1)) * 4;
    if (py == 0) v |= WCLS_TOP;
    if (px == 0) v |= WCLS_LEFT;
    if (x0_last - 1 + px >= W) v |= WCLS_RIGHT;
    prel[i] = real ? v : WCLS_PAD;
  }
}

// Patch reads of the input transform: tile (wr, 16 wc + t), patch element (i, j), channel pair g: byte address
// base(lane) + K_ij with base = 32 (16 wr * 33 + 16 wc + t) + 8 g and K_ij = 32 ((4 i + (j & 3)) * 33 + (j >> 2))
// (compile time: a ds_read_b64 immediate).  The 64 lanes of one read cover 512 contiguous bytes.
__device__ __forceinline__ int wino4_patch_base(int t, int g, int wr, int wc) {
  return 32 * (16 * wr * Wino4Geom::PWQ + 16 * wc + t) + 8 * g;
}
constexpr int wino4_patch_k(int i, int j) { return 32 * ((4 * i + (j & 3)) * Wino4Geom::PWQ + (j >> 2)); }

// U reads of the MFMA A operand: lane (m = lane & 15, g = lane >> 4) reads the input-channel pair g of output
// channel 16 cg + m at point xi: base 32 m + 8 g, offset 1024 xi + 512 cg.
__device__ __forceinline__ int wino4_u_base(int m, int g) { return 32 * m + 8 * g; }
constexpr int wino4_u_k(int xi, int cg) { return 1024 * xi + 512 * cg; }

}  // namespace pa
