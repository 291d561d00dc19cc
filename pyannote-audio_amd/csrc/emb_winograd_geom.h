// Integer geometry of the Winograd convolution kernel (csrc/emb_winograd.hip): LDS layouts, the per-lane
// DMA offsets with their halo class bits, the transform's read addresses, the tile order.  No HIP types: the
// header is also compiled for the HOST by tests/test_winograd_geometry_cpu.py, which replays the DMA and the
// transform reads of every wave and lane and checks that they agree (and that the reads are bank-conflict
// free).  Needs: __device__, __forceinline__.
#pragma once

namespace pa {

constexpr int WCB = 16;   // input channels per stage (one 64-B LDS row)
constexpr int W_BN = 32;  // output channels per workgroup
constexpr int HW = 4;     // waves per workgroup

template <int TR, int TCG>
struct WinoGeom {
  static_assert(TR * TCG == HW, "4 waves: one tile row x 16 tile columns each");
  static constexpr int PH = 2 * TR + 2;          // patch rows
  static constexpr int PW = 2 * 16 * TCG + 2;    // patch cols
  static constexpr int PWH = PW / 2;             // entries per column parity
  static constexpr int PROWS = PH * 2 * PWH;     // LDS rows of the patch
  static constexpr int PINSTR = (PROWS + 15) / 16;  // 1-KB DMA pieces
  static constexpr int PATCH = PINSTR * 16 * WCB;   // floats
  static constexpr int NPP = (PINSTR + HW - 1) / HW;  // patch pieces per wave
  static constexpr int UINSTR = 16 * W_BN / 16;     // 32 pieces per U slab
  static constexpr int USLAB = 16 * W_BN * WCB;     // floats
  static constexpr int LDS_FLOATS = PATCH + USLAB;
};

// physical 16-B slot of logical channel quad g in LDS row r
__device__ __forceinline__ int wslot(int r, int g) { return (g + 2 * ((r >> 2) & 1)) & 3; }

struct WinoTile {
  int b, n0, y0, x0, valid;
};

// Patch DMA: piece k fills LDS rows 16k .. 16k+15; lane l -> row 16k + (l>>2), physical slot l&3.
// All per-lane address arithmetic is done ONCE per kernel (VALU work cannot hide under f32 MFMAs on this
// chip: they execute on the vector ALUs, see tools/probes/pingpong_probe.py, so every vector instruction
// of the staging path is paid in matrix time).  `prel` = byte offset of the lane's (patch row, quad) from
// the patch origin (y0 - 1, x0 - 1), plus CLASS bits above the largest image: bit 28 = top halo row,
// 29 = left halo column, 30 = the column right of the image in the LAST column tile, 31 = padding lane.
// A stage moves the buffer descriptor to the patch origin (scalar arithmetic) and issues
// `prel & keep`, where the scalar `keep` clears the class bits that are inside the image for this tile: a
// surviving class bit pushes the offset past num_records and the hardware bounds check writes zeros.
// Rows below the image and the wrap of the last image row are past num_records by themselves; columns
// further right than W only feed output tiles that are never stored.
constexpr int WCLS_TOP = 1 << 28, WCLS_LEFT = 1 << 29, WCLS_RIGHT = 1 << 30, WCLS_PAD = (int)0x80000000;

template <int TR, int TCG>
__device__ __forceinline__ void wino_patch_lanes(int* prel, int W, int CIN, int lane, int slw, int x0_last) {
  using G = WinoGeom<TR, TCG>;
#pragma unroll
  for (int i = 0; i < G::NPP; ++i) {
    const int k = slw + HW * i;
    const int row = 16 * k + (lane >> 2);
    const int gq = ((lane & 3) - 2 * ((row >> 2) & 1)) & 3;  // logical quad stored in this slot
    const int pr = row / G::PWH, idx = row % G::PWH;         // pr = py*2 + parity
    const int py = pr >> 1, px = 2 * idx + (pr & 1);
    const bool real = k < G::PINSTR && row < G::PROWS;
    int v = ((py * W + px) * CIN + 4 * gq) * 4;
    if (py == 0) v |= WCLS_TOP;
    if (px == 0) v |= WCLS_LEFT;
    if (x0_last - 1 + px == W) v |= WCLS_RIGHT;
    prel[i] = real ? v : WCLS_PAD;
  }
}

// Class bits that stay SET for tile q (the lane's offset is then past num_records: the DMA writes zeros):
// the top halo row / left halo column / right halo column exist only when the tile touches that border of
// the image; padding lanes are always out.
__device__ __forceinline__ int wino_patch_keep(const WinoTile& q, int x0_last) {
  int keep = 0x0fffffff;
  if (q.y0 == 0) keep |= WCLS_TOP;
  if (q.x0 == 0) keep |= WCLS_LEFT;
  if (q.x0 == x0_last) keep |= WCLS_RIGHT;
  return keep | WCLS_PAD;
}

// Patch reads of the input transform: tile (wr, 16 wc + t), patch element (i, j) lives in LDS row
// R + K_ij with R = 4 wr PWH + 16 wc + t (per lane) and K_ij = (2i + (j & 1)) PWH + (j >> 1) (compile time).
// The quad swizzle of a row only looks at bit 2 of the row number, i.e. at (R + K_ij mod 8): 8 per-lane
// byte offsets (one per residue) computed once per kernel, everything else is a ds_read immediate.
template <int TR, int TCG>
__device__ __forceinline__ void wino_patch_bases(int (&pbase)[8], int t, int g, int wr, int wc) {
  using G = WinoGeom<TR, TCG>;
  const int R = 4 * wr * G::PWH + 16 * wc + t;
#pragma unroll
  for (int c = 0; c < 8; ++c) pbase[c] = (R + c) * (WCB * 4) + 16 * wslot(R + c, g);
}

// Tile order is XCD-aware: workgroup w runs on XCD w % 8 (each XCD has its own L2), so tile q is decoded
// as xcd = q % 8, r = q / 8 with the cout tile FASTEST in r: the n_tiles workgroups that read the same
// input patch (same image, same pixel tile, different 32-cout slices) run side by side on ONE XCD and
// the patch comes from HBM once instead of n_tiles times.  (pixel tile, image) = (r / n_tiles) * 8 + xcd;
// the index space is padded to a multiple of 8 (pixel tile, image) pairs: padding tiles are computed on
// the last real tile's data and not stored (valid = 0).
// `xranges` (round 5): an XCD owns a CONTIGUOUS range of (pixel tile, image) pairs, pb = xcd * ceil(num_pb / 8) +
// r / n_tiles, instead of every eighth one: neighbouring tiles share their halo rows / columns, and dealt round-robin
// every XCD fetched its own copy of them from HBM.
__device__ __forceinline__ WinoTile wino_decode(int q, int tiles_w, int tiles_hw, int n_tiles, int th,
                                                int tw, int num_pb, int y_first = 0, int xranges = 0) {
  WinoTile o;
  const int xcd = q & 7, r = q >> 3;
  int pb = xranges ? xcd * ((num_pb + 7) >> 3) + r / n_tiles : (r / n_tiles) * 8 + xcd;
  o.n0 = (r % n_tiles) * W_BN;
  o.valid = pb < num_pb;
  pb = pb < num_pb ? pb : num_pb - 1;
  const int pix = pb % tiles_hw;
  o.b = pb / tiles_hw;
  o.y0 = (pix / tiles_w) * th + y_first;   // (y_first: the launch covers the rows from there on, pa_conv3x3_wino_rows)
  o.x0 = (pix % tiles_w) * tw;
  return o;
}

}  // namespace pa
