// Integer geometry of the Winograd F(4x4, 3x3) convolution kernel (csrc/emb_winograd4.hip): LDS layouts, the
// per-lane DMA offsets with their halo class bits, the transform's read addresses, the unit order.  No HIP types: the
// header is also compiled for the HOST by tests/test_winograd_geometry_cpu.py (tests/native/
// winograd4_geom_harness.cpp), which replays the DMA and the transform reads of every lane and checks that they agree
// (and that the reads are bank-conflict free).
// Needs: __device__, __forceinline__, and emb_winograd_geom.h (the class bits, W_BN).
#pragma once

namespace pa {

// A WAVE owns one "unit" = 16 consecutive 4x4 Winograd tiles of one tile row of one image: 4 x 64 output pixels; a
// workgroup = 4 waves = 4 consecutive units (any rows, any images) x the same 32 output channels.  A stage stages 8
// input channels:
//   patch  (PRIVATE to the wave: no other wave reads it, so it needs no barrier and no second buffer -- the wave
//          issues the next stage's patch right behind its own input transform) 6 x 66 input pixels x 8 channels, one
//          pixel = one 32-B LDS row, de-interleaved by column mod 4 so that the 16 lanes of the tile row, 4 pixels
//          apart, read CONSECUTIVE rows: row = (py*4 + px%4)*17 + px/4;
//   U slab (shared by the 4 waves, double-buffered) 36 points x 32 output channels x 8 input channels:
//          row = 32 xi + n (32 B = the 8 input channels).
// Bank swizzle.  Both images are read with ds_read_b64 by lane (t = lane & 15, g = lane >> 4) -- tile / output
// channel t, channel pair g -- i.e. row r0 + t, pair g.  The hardware serves a ds_read_b64 in the lane groups {0-31}
// and {32-63}, bank = (byte / 4) mod 64 (MI355X_MICROARCH.md, LDS table): with the pairs in their natural order the
// 32 lanes (t, g in {0,1}) touch dwords 8 (r0 + t) + 2 g + {0,1}, rows 8 apart collide -- every read a 2-way
// conflict (and the ds_read2_b64 the compiler fused pairs of them into, 16-lane groups mod 32 banks, 4-way: the
// first PMC pass over this kernel counted 31 % of its cycles as LDS bank conflicts, profiles/r4_pipeline_pmc_sq.txt).
// So rows whose index within their group of 16 / 17 has bit 3 set hold their two channel QUADS swapped: pair g sits
// in slot g ^ 2.  Then the 32 lanes of a group cover dwords 8 t' + {0..3} (t' < 8) and 8 t' + {4..7}: 64 banks once.
struct Wino4Geom {
  static constexpr int CB = 8;                       // input channels per stage
  static constexpr int TH = 4, TW = 64;              // output pixels per unit
  static constexpr int PH = TH + 2, PW = TW + 2;     // patch
  static constexpr int PWQ = (PW + 3) / 4;           // 17 entries per column residue
  static constexpr int PROWS = PH * 4 * PWQ;         // 408 LDS rows of 32 B
  static constexpr int PINSTR = (PROWS + 31) / 32;   // 13 DMA pieces of 1 KB (32 rows) per wave and stage
  static constexpr int PATCH_BYTES = PINSTR * 1024;  // 13 312 per wave
  static constexpr int UINSTR = 36;                  // 36 x 1 KB per workgroup and stage: 9 per wave
  static constexpr int USLAB_BYTES = 36 * 32 * CB * 4;   // 36 864
  static constexpr int LDS_BYTES = 4 * PATCH_BYTES + 2 * USLAB_BYTES;   // 126 976
};

struct Wino4Unit {   // wave-uniform
  int b, y0, x0, valid;
};

// Patch DMA of one wave: piece i fills LDS rows 32i .. 32i+31 of the wave's block; lane l -> row 32i + (l >> 1),
// channel quad l & 1.  `prel` = byte offset of the lane's (patch pixel, quad) from the patch origin (y0 - 1, x0 - 1) +
// class bits (emb_winograd_geom.h): top halo row, left halo column, every column at or right of the image border in
// the LAST column group (F(4x4) mixes all six patch columns into every output of a tile: columns past the border
// must be zeros, not the next row's pixels), padding lanes.
// rows that hold their channel quads swapped: index within the row group (tile number + column group) has bit 3 set
__device__ __forceinline__ int wino4_swz(int idx) { return (idx >> 3) & 1; }
__device__ __forceinline__ int wino4_patch_lane(int piece, int W, int CIN, int lane, int x0_last) {
  using G = Wino4Geom;
  const int row = 32 * piece + (lane >> 1);
  const int pr = row / G::PWQ, idx = row % G::PWQ;   // pr = py*4 + residue
  const int py = pr >> 2, px = 4 * idx + (pr & 3);
  const bool real = piece < G::PINSTR && row < G::PROWS && px < G::PW;
  // (this lane fills the 16-byte half (lane & 1) of the row: the channel quad stored there, see the bank swizzle)
  int v = ((py * W + px) * CIN + 4 * ((lane & 1) ^ wino4_swz(idx))) * 4;
  if (py == 0) v |= WCLS_TOP;
  if (px == 0) v |= WCLS_LEFT;
  if (x0_last - 1 + px >= W) v |= WCLS_RIGHT;
  return real ? v : WCLS_PAD;
}
__device__ __forceinline__ void wino4_patch_lanes(int* prel, int W, int CIN, int lane, int x0_last) {
#pragma unroll
  for (int i = 0; i < Wino4Geom::PINSTR; ++i) prel[i] = wino4_patch_lane(i, W, CIN, lane, x0_last);
}
// class bits that stay SET for a unit (offset past num_records -> the DMA writes zeros)
__device__ __forceinline__ int wino4_patch_keep(const Wino4Unit& u, int x0_last) {
  int keep = 0x0fffffff;
  if (u.y0 == 0) keep |= WCLS_TOP;
  if (u.x0 == 0) keep |= WCLS_LEFT;
  if (u.x0 == x0_last) keep |= WCLS_RIGHT;
  return keep | WCLS_PAD;
}

// Patch reads of the input transform: tile t, patch element (i, j), channel pair g: byte address (within the wave's
// block) base(t, g, j >> 2) + K_ij, K_ij = 32 ((4 i + (j & 3)) * 17 + (j >> 2)) (compile time: a ds_read_b64
// immediate); the row read is number t + (j >> 2) of its group, which decides the slot of pair g.
__device__ __forceinline__ int wino4_patch_base(int t, int g, int jq) {
  return 32 * t + 8 * (g ^ (2 * wino4_swz(t + jq)));
}
constexpr int wino4_patch_k(int i, int j) { return 32 * ((4 * i + (j & 3)) * Wino4Geom::PWQ + (j >> 2)); }

// U reads of the MFMA A operand: lane (m = lane & 15, g = lane >> 4) reads the input-channel pair g of output
// channel 16 cg + m at point xi: base 32 m + 8 (g ^ 2 swz(m)), offset 1024 xi + 512 cg (rows n with bit 3 of n set
// hold their quads swapped: weights.winograd4_pack / pa_winograd4_pack_host write them that way).
__device__ __forceinline__ int wino4_u_base(int m, int g) { return 32 * m + 8 * (g ^ (2 * wino4_swz(m))); }
constexpr int wino4_u_k(int xi, int cg) { return 1024 * xi + 512 * cg; }

// Unit u of an (B, H, W) map: column group fastest, then tile row, then image -- the 4 units of a workgroup are
// neighbours.  Work item q of the launch (XCD-aware like wino_decode: q % 8 = XCD, the n_tiles cout slices of one
// group of 4 units side by side on one XCD): -> first unit of the group and the cout slice.
struct Wino4Work {
  int unit0, n0, valid;
};
// Round 5: an XCD owns a CONTIGUOUS range of groups (group = xcd * ceil(num_groups / 8) + r / n_tiles) instead of every
// eighth one: vertically adjacent tile rows share two of their six patch rows, and with the groups dealt round-robin
// over the XCDs every XCD fetched its own copy of them from HBM (the 2.2x algorithmic bytes of round 4's PMC pass).
// `xranges` = 0 restores the round-robin order (the launcher: PA_XCD_RANGES, and the 256-channel layers, where the
// ranges measured 4 % slower without a residual).
__device__ __forceinline__ int wino4_group_of(int q, int n_tiles, int num_groups, int xranges) {
  const int xcd = q & 7, r = q >> 3;
  return xranges ? xcd * ((num_groups + 7) >> 3) + r / n_tiles : (r / n_tiles) * 8 + xcd;
}
__device__ __forceinline__ Wino4Work wino4_decode(int q, int n_tiles, int num_groups, int xranges = 1) {
  Wino4Work o;
  const int r = q >> 3;
  const int grp = wino4_group_of(q, n_tiles, num_groups, xranges);
  o.n0 = (r % n_tiles) * W_BN;
  o.valid = grp < num_groups;
  o.unit0 = 4 * (grp < num_groups ? grp : num_groups - 1);
  return o;
}
__device__ __forceinline__ Wino4Unit wino4_unit(int u, int cgroups, int trows, int num_units) {
  Wino4Unit o;
  o.valid = u < num_units;
  const int uu = u < num_units ? u : num_units - 1;
  const int c = uu % cgroups, r = (uu / cgroups) % trows;
  o.b = uu / (cgroups * trows);
  o.y0 = 4 * r;
  o.x0 = 64 * c;
  return o;
}

// ---------------------------------------------------------------------------------------------------------------
// TILE-LINEAR units (round 5): a wave's unit = 16 CONSECUTIVE tiles of the raster order (image, tile row, tile
// column) -- it may straddle tile rows and images.  The row-shaped unit above pads every tile row to whole groups of 16
// tiles: fine for the maps of 10 s chunks (125 / 63 / 32 tiles per row), ruinous for the narrow maps of short chunks
// (3 s: 38 / 19 / 10 tiles per row -> 1.26x / 1.68x / 1.60x the useful tiles).  Linear units pad once per launch.
// The tiles of a unit no longer share their halo, so each tile's 6 x 6 patch is staged on its own:
//   patch  (private to the wave) 36 elements x 16 tiles, one (element, tile) = one 32-B LDS row: row = 16 e + t,
//          e = 6 i + j; lane (t, g) of the transform reads row 16 e + t -- consecutive lanes, consecutive rows, with
//          the same quad swizzle on bit 3 of t.  576 rows = 18 DMA pieces per stage (13 for the row-shaped unit).
//   Every halo pixel outside the image is zero-filled through an out-of-bounds LANE offset (there are no class bits:
//   the neighbours in memory are real pixels of the next row / image).
struct Wino4LinGeom {
  static constexpr int CB = 8;
  static constexpr int PROWS = 36 * 16;              // 576 LDS rows of 32 B
  static constexpr int PINSTR = PROWS / 32;          // 18 DMA pieces of 1 KB per wave and stage
  static constexpr int PATCH_BYTES = PINSTR * 1024;  // 18 432 per wave
  static constexpr int USLAB_BYTES = Wino4Geom::USLAB_BYTES;
  static constexpr int LDS_BYTES = 4 * PATCH_BYTES + 2 * USLAB_BYTES;   // 147 456
};
struct Wino4LinTile {   // per LANE
  int b;      // image, relative to the first image of the unit
  int y, x;   // top-left output pixel
  int valid;
};
// tile T of the raster order; tiles past the end repeat the last real one (computed, never stored)
__device__ __forceinline__ Wino4LinTile wino4_lin_tile(int T, int tcols, int trows, int total_tiles, int b0) {
  Wino4LinTile o;
  o.valid = T < total_tiles;
  const int TT = T < total_tiles ? T : total_tiles - 1;
  const int per = tcols * trows, b = TT / per, r = TT - b * per;
  const int ty = r / tcols;
  o.b = b - b0;
  o.y = 4 * ty;
  o.x = 4 * (r - ty * tcols);
  return o;
}
// first image of unit u (wave-uniform): the image of its first tile
__device__ __forceinline__ int wino4_lin_b0(int u, int tcols, int trows, int total_tiles) {
  const int T = 16 * u < total_tiles ? 16 * u : total_tiles - 1;
  return T / (tcols * trows);
}
// DMA: piece i fills LDS rows 32 i .. 32 i + 31; lane l -> row 32 i + (l >> 1) = element e = 2 i + (l >> 5) of tile
// td = (l >> 1) & 15, 16-byte half l & 1, which holds channel quad (l & 1) ^ swz(td).  -> byte offset from the first
// pixel of the unit's first image (stage 0), or the out-of-bounds marker.  `td_tile` = wino4_lin_tile(16 u + td, ...).
__device__ __forceinline__ int wino4_lin_patch_lane(int piece, const Wino4LinTile& td_tile, int H, int W, int CIN,
                                                    int lane) {
  const int e = 2 * piece + (lane >> 5), i = e / 6, j = e - 6 * i;
  const int iy = td_tile.y - 1 + i, ix = td_tile.x - 1 + j;
  const int td = (lane >> 1) & 15;
  const int quad = (lane & 1) ^ wino4_swz(td);
  const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
  return in ? (((td_tile.b * H + iy) * W + ix) * CIN + 4 * quad) * 4 : WCLS_PAD;
}
// transform reads: lane (t, g), element (i, j): base(t, g) + K_ij
__device__ __forceinline__ int wino4_lin_patch_base(int t, int g) { return 32 * t + 8 * (g ^ (2 * wino4_swz(t))); }
constexpr int wino4_lin_patch_k(int i, int j) { return 512 * (6 * i + j); }

// ---------------------------------------------------------------------------------------------------------------
// RUN-shaped units (round 5, second form of the tile-linear units): the 16 consecutive tiles of a unit fall into at most
// RMAX runs of tiles of ONE tile row; inside a run neighbouring tiles share their halo columns exactly as in the
// row-shaped unit.  The patch is the row-shaped layout with the runs side by side: run r occupies the virtual columns
// [4 C_r, 4 C_r + 4 n_r + 2) with C_0 = 0, C_{r+1} = C_r + n_r + 1 (n_r tiles: one 4-column slot of slack per run), i.e.
// tile t of run r(t) sits at slot T' = t + r(t) and reads LDS row (4 py + column % 4) * PWQ + T' + column / 4 --
// consecutive tiles, consecutive rows.  64 + 4 RMAX columns: 6 x 4 x 20 = 480 rows = 15 DMA pieces per stage (row-shaped
// 13, tile-private 18) and 1.1x instead of 1.45x the patch bytes.  Maps with fewer than 6 tiles per row could need more
// than RMAX runs: they keep the tile-private form.
struct Wino4RunGeom {
  static constexpr int CB = 8;
  static constexpr int RMAX = 4;
  static constexpr int PWQ = 16 + RMAX;              // slots per column residue
  static constexpr int PROWS = 6 * 4 * PWQ;          // 480 LDS rows of 32 B
  static constexpr int PINSTR = PROWS / 32;          // 15 DMA pieces of 1 KB per wave and stage
  static constexpr int PATCH_BYTES = PINSTR * 1024;
  static constexpr int USLAB_BYTES = Wino4Geom::USLAB_BYTES;
  static constexpr int LDS_BYTES = 4 * PATCH_BYTES + 2 * USLAB_BYTES;   // 135 168
};
// the runs of unit u (wave-uniform).  The raster order continues into a VIRTUAL image behind the last one: tiles past
// the end of the launch lie there (every pixel out of bounds: zeros), so no tile is ever repeated.
struct Wino4Runs {   // (scalar members, not arrays: the compiler kept the array form in scratch memory)
  int n;                                       // runs (1 .. RMAX)
  int first0, first1, first2, first3, first4;  // first tile (0 .. 16) of run r; 16 from run n on
  int b0, b1, b2, b3;                          // image of the run, relative to the unit's first image
  int y0, y1, y2, y3;                          // top output row of the run's tiles
  int x0, x1, x2, x3;                          // left output column of the run's first tile
};
__device__ __forceinline__ Wino4Runs wino4_runs(int u, int tcols, int trows, int* b0_out) {
  static_assert(Wino4RunGeom::RMAX == 4, "four runs are spelled out");
  Wino4Runs o;
  const int per = tcols * trows;
  int T = 16 * u;
  const int img0 = T / per;
  *b0_out = img0;
  int t = 0;
  o.n = 0;
#define W4_RUN_STEP(r)                                                                    \
  {                                                                                       \
    o.first##r = t < 16 ? t : 16;                                                         \
    const int b = T / per, rem = T - b * per, ty = rem / tcols, tx = rem - ty * tcols;   \
    o.b##r = b - img0;                                                                    \
    o.y##r = 4 * ty;                                                                      \
    o.x##r = 4 * tx;                                                                      \
    if (t < 16) {                                                                         \
      int n = tcols - tx;                                                                 \
      if (n > 16 - t) n = 16 - t;                                                         \
      t += n;                                                                             \
      T += n;                                                                             \
      o.n = r + 1;                                                                        \
    }                                                                                     \
  }
  W4_RUN_STEP(0) W4_RUN_STEP(1) W4_RUN_STEP(2) W4_RUN_STEP(3)
#undef W4_RUN_STEP
  o.first4 = 16;
  return o;
}
// field f of run r for a per-LANE r: a chain of selects over the wave-uniform values (indexing the arrays with a
// per-lane r would move them to scratch memory)
// (as a sum of selected DIFFERENCES: a chain of selects on r == 0, 1, 2 is turned into a look-up table in scratch memory
//  by the compiler, which costs a scratch load per field and lane)
#define W4_RUN_SEL(R, f, r)                                                                             \
  ((R).f##0 + ((r) >= 1 ? (R).f##1 - (R).f##0 : 0) + ((r) >= 2 ? (R).f##2 - (R).f##1 : 0) + \
   ((r) >= 3 ? (R).f##3 - (R).f##2 : 0))
#define W4_RUN_SEL_NEXT(R, f, r)                                                                        \
  ((R).f##1 + ((r) >= 1 ? (R).f##2 - (R).f##1 : 0) + ((r) >= 2 ? (R).f##3 - (R).f##2 : 0) + \
   ((r) >= 3 ? (R).f##4 - (R).f##3 : 0))
// run of tile t / of patch slot `idx` (slot C_r + j holds tile j of run r and, for j = n_r, its right halo)
__device__ __forceinline__ int wino4_run_of_tile(const Wino4Runs& R, int t) {
  return (t >= R.first1) + (t >= R.first2) + (t >= R.first3);
}
__device__ __forceinline__ int wino4_run_of_slot(const Wino4Runs& R, int idx) {   // C_r = first[r] + r
  return (idx >= R.first1 + 1) + (idx >= R.first2 + 2) + (idx >= R.first3 + 3);
}
// the tile of lane t (compute role)
__device__ __forceinline__ Wino4LinTile wino4_run_tile(const Wino4Runs& R, int t, int u, int total_tiles) {
  const int r = wino4_run_of_tile(R, t);
  Wino4LinTile o;
  o.valid = 16 * u + t < total_tiles;
  o.b = W4_RUN_SEL(R, b, r);
  o.y = W4_RUN_SEL(R, y, r);
  o.x = W4_RUN_SEL(R, x, r) + 4 * (t - W4_RUN_SEL(R, first, r));
  return o;
}
// DMA: piece i fills LDS rows 32 i .. 32 i + 31; lane l -> row 32 i + (l >> 1), 16-byte half l & 1 holding channel
// quad (l & 1) ^ swz(slot).  -> byte offset from the first pixel of the unit's first image, or the out-of-bounds marker
// (outside the image, in the slack of a run, in an unused slot, or in the virtual image behind the last one).
__device__ __forceinline__ int wino4_run_patch_lane(int piece, const Wino4Runs& R, int H, int W, int CIN, int nimg_left,
                                                   int lane) {
  using G = Wino4RunGeom;
  const int row = 32 * piece + (lane >> 1);
  const int pr = row / G::PWQ, idx = row - pr * G::PWQ;   // pr = 4 py + residue
  const int py = pr >> 2, v = 4 * idx + (pr & 3);          // virtual column
  const int r = wino4_run_of_slot(R, idx);
  const int first = W4_RUN_SEL(R, first, r), rb = W4_RUN_SEL(R, b, r);
  const int nr = W4_RUN_SEL_NEXT(R, first, r) - first;     // tiles of the run (0: unused)
  const int cx = v - 4 * (first + r);                      // column within the run's strip
  const int iy = W4_RUN_SEL(R, y, r) - 1 + py, ix = W4_RUN_SEL(R, x, r) - 1 + cx;
  const int quad = (lane & 1) ^ wino4_swz(idx);
  const bool in = nr > 0 && cx < 4 * nr + 2 && iy >= 0 && iy < H && ix >= 0 && ix < W && rb < nimg_left;
  return in ? (((rb * H + iy) * W + ix) * CIN + 4 * quad) * 4 : WCLS_PAD;
}
// The same 15 offsets, the way the kernel computes them (the function above is the definition the host harness checks
// this one against).  The LDS row of piece i is 32 i + (l >> 1) and 32 x 5 = 8 x PWQ: pieces i and i + 5 of a lane hit
// the SAME slot and column residue, two patch rows further down -- so a lane looks up its run only five times per unit
// (one chain of selects from the wave-uniform run table each), and every piece is an add, a row check and a select.
__device__ __forceinline__ void wino4_run_patch_lanes(int (&off)[15], const Wino4Runs& R, int H, int W, int CIN,
                                                      int nimg_left, int lane) {
  using G = Wino4RunGeom;
  static_assert(G::PINSTR == 15 && 32 * 5 == 8 * G::PWQ, "five-piece period of the lane -> slot map");
  const int half = lane >> 1;
  const int srow = W * CIN * 4;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int row = 32 * k + half;
    const int pr = row / G::PWQ, idx = row - pr * G::PWQ;
    const int py0 = pr >> 2, res = pr & 3;
    const int r = wino4_run_of_slot(R, idx);
    const int first = W4_RUN_SEL(R, first, r), nr = W4_RUN_SEL_NEXT(R, first, r) - first, rb = W4_RUN_SEL(R, b, r);
    const int cx = 4 * (idx - first - r) + res;
    const int ix = W4_RUN_SEL(R, x, r) - 1 + cx, ytop = W4_RUN_SEL(R, y, r) - 1;
    const int quad = (lane & 1) ^ wino4_swz(idx);
    const bool colok = nr > 0 && cx < 4 * nr + 2 && ix >= 0 && ix < W && rb < nimg_left;
    const int base = (((rb * H + ytop) * W + ix) * CIN + 4 * quad) * 4;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int py = py0 + 2 * m;                  // (pieces k, k + 5, k + 10)
      const int iy = ytop + py;
      off[k + 5 * m] = (colok && iy >= 0 && iy < H) ? base + py * srow : WCLS_PAD;
    }
  }
}
// transform reads: lane (t, g) at slot T' = t + r(t): base(T', g, j >> 2) + K_ij
__device__ __forceinline__ int wino4_run_patch_base(int slot, int g, int jq) {
  return 32 * slot + 8 * (g ^ (2 * wino4_swz(slot + jq)));
}
constexpr int wino4_run_patch_k(int i, int j) { return 32 * ((4 * i + (j & 3)) * Wino4RunGeom::PWQ + (j >> 2)); }

}  // namespace pa
