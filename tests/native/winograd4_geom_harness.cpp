// Host harness for csrc/emb_winograd4_geom.h (the integer geometry of k_conv3x3_wino4, compiled unchanged):
// replays, for every lane of a wave, (1) the LDS-DMA of the wave's input patch -- which global element, or a hardware
// zero, lands in which 16-byte LDS slot -- and (2) the ds_read_b64 addresses of the input transform, and checks that
// every read returns exactly the patch element (pixel, channel pair) the Winograd tile needs, zeros wherever the
// pixel lies outside the image (F(4x4) mixes all six patch columns into every output of a tile, so EVERY column
// past the border must be a zero, not only the first), that no read touches an LDS location the DMA did not write,
// and that every ds_read_b64 is bank-conflict free under the hardware's rule (MI355X_MICROARCH.md, LDS table: lane
// groups {0-31} and {32-63}, bank = (byte / 4) mod 64: the 64 dwords of a group must be 64 different banks -- "512
// contiguous bytes per wave", the first version of this check, is NOT that rule and let a 2-way conflict through).
// Same for the U-slab reads; and the unit order hands out every (unit, cout slice) exactly once, the slices of a group of four
// units on one XCD.  Exit code 0 = all good.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define __device__
#define __forceinline__ inline
#include "emb_winograd_geom.h"
#include "emb_winograd4_geom.h"
using namespace pa;

struct Cell {
  int kind;      // 0 = never written, 1 = zero fill, 2 = data
  long gpix;     // linear pixel index (iy * W + ix) of the image
  int quad;      // channel quad 0..1 of the 8-channel stage
};

static int check_unit(int H, int W, int CIN, int y0, int x0, int x0_last, int c0) {
  using G = Wino4Geom;
  Wino4Unit u{0, y0, x0, 1};
  const long img = (long)H * W * CIN;
  const long org = ((long)(y0 - 1) * W + (x0 - 1)) * CIN + c0;
  const unsigned num_records = (unsigned)((img - org) * 4);
  std::vector<Cell> lds((size_t)G::PINSTR * 64, Cell{0, 0, 0});   // 16-byte slots of the wave's block
  const int keep = wino4_patch_keep(u, x0_last);
  for (int lane = 0; lane < 64; ++lane) {
    int prel[G::PINSTR];
    wino4_patch_lanes(prel, W, CIN, lane, x0_last);
    for (int i = 0; i < G::PINSTR; ++i) {
      const unsigned off = (unsigned)(prel[i] & keep);
      Cell& c = lds[(size_t)64 * i + lane];
      if (c.kind != 0) return printf("LDS location written twice\n"), 1;
      if (off >= num_records) {
        c = Cell{1, 0, 0};
      } else {
        const long fl = (long)(off / 4) + org - c0;
        if (fl < 0 || (off % 16) != 0) return printf("bad offset\n"), 1;
        c = Cell{2, fl / CIN, (int)(fl % CIN) / 4};
        if ((fl % CIN) % 4 != 0 || (fl % CIN) >= G::CB) return printf("channel quad out of the stage\n"), 1;
      }
    }
  }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      std::set<int> banks[2];
      for (int lane = 0; lane < 64; ++lane) {
        const int t = lane & 15, g = lane >> 4;
        const int addr = wino4_patch_base(t, g, j >> 2) + wino4_patch_k(i, j);
        if (addr % 8 != 0 || addr / 16 >= (int)lds.size()) return printf("read outside the patch block\n"), 1;
        banks[lane >> 5].insert((addr / 4) % 64);
        banks[lane >> 5].insert((addr / 4 + 1) % 64);
        const Cell& c = lds[addr / 16];
        const int iy = y0 - 1 + i, ix = x0 - 1 + 4 * t + j;
        if (c.kind == 0) return printf("read of an LDS location the DMA never wrote (i=%d j=%d lane=%d)\n", i, j, lane), 1;
        const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
        if (inside) {
          // the 8-byte read takes channels 2g, 2g+1: quad g >> 1 (wherever the swizzle put it), second half of the
          // 16-byte slot when g is odd
          if (c.kind != 2 || c.gpix != (long)iy * W + ix || c.quad != (g >> 1) || ((addr % 16) / 8) != (g & 1))
            return printf("unit (%d,%d) lane %d (i=%d,j=%d): wrong element\n", y0, x0, lane, i, j), 1;
        } else {
          if (c.kind != 1) return printf("element (%d,%d) outside the image is not a hardware zero\n", iy, ix), 1;
        }
      }
      if (banks[0].size() != 64 || banks[1].size() != 64)
        return printf("transform read (i=%d,j=%d): bank conflict in a 32-lane group\n", i, j), 1;
    }
  return 0;
}

static int check_image(int H, int W, int CIN) {
  using G = Wino4Geom;
  const int trows = (H + G::TH - 1) / G::TH, cgroups = (W + G::TW - 1) / G::TW;
  const int x0_last = (cgroups - 1) * G::TW;
  for (int r = 0; r < trows; ++r)
    for (int c = 0; c < cgroups; ++c) {
      if (r > 1 && r < trows - 2 && c > 1 && c < cgroups - 2 && (r * 7 + c) % 5) continue;
      for (int c0 = 0; c0 < CIN; c0 += CIN - G::CB > 0 ? CIN - G::CB : G::CB)
        if (check_unit(H, W, CIN, r * G::TH, c * G::TW, x0_last, c0)) {
          printf("  image %dx%dx%d unit (%d,%d) stage %d\n", H, W, CIN, r, c, c0);
          return 1;
        }
    }
  return 0;
}

static int check_u_reads() {
  for (int xi = 0; xi < 36; ++xi)
    for (int cg = 0; cg < 2; ++cg) {
      std::set<int> banks[2];
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, g = lane >> 4;
        const int addr = wino4_u_base(m, g) + wino4_u_k(xi, cg);
        // slab image: row = 32 xi + n (n = 16 cg + m), 8 input channels of 4 bytes; the pair g in slot g, or g ^ 2
        // in the rows with bit 3 of n set (weights.winograd4_pack writes them that way)
        const int n = 16 * cg + m, slot = g ^ (2 * ((n >> 3) & 1));
        if (addr != ((32 * xi + n) * 8 + 2 * slot) * 4) return printf("U read address\n"), 1;
        if (addr + 8 > Wino4Geom::USLAB_BYTES) return printf("U read outside the slab\n"), 1;
        banks[lane >> 5].insert((addr / 4) % 64);
        banks[lane >> 5].insert((addr / 4 + 1) % 64);
      }
      if (banks[0].size() != 64 || banks[1].size() != 64) return printf("U read: bank conflict in a 32-lane group\n"), 1;
    }
  return 0;
}

static int check_order(int B, int H, int W, int n_tiles) {
  using G = Wino4Geom;
  const int trows = (H + G::TH - 1) / G::TH, cgroups = (W + G::TW - 1) / G::TW;
  const int num_units = B * trows * cgroups, num_groups = (num_units + 3) / 4;
  const int total = ((num_groups + 7) / 8) * 8 * n_tiles;
  std::vector<int> seen((size_t)num_units * n_tiles, 0);
  for (int q = 0; q < total; ++q) {
    const Wino4Work w = wino4_decode(q, n_tiles, num_groups);
    if (w.n0 % W_BN || w.n0 / W_BN >= n_tiles || w.unit0 % 4 || w.unit0 / 4 >= num_groups) return printf("decode out of range\n"), 1;
    // an XCD owns a contiguous range of groups (xranges = 1, the default); the cout slices of a group are consecutive claims
    if (w.valid && (w.unit0 / 4) / ((num_groups + 7) / 8) != (q & 7)) return printf("group %d is not on XCD %d\n", w.unit0 / 4, q & 7), 1;
    if (w.valid && w.n0 / W_BN != (q >> 3) % n_tiles) return printf("cout slices of a group are not consecutive claims\n"), 1;
    for (int s = 0; s < 4; ++s) {
      Wino4Unit u = wino4_unit(w.unit0 + s, cgroups, trows, num_units);
      u.valid &= w.valid;
      if (u.b < 0 || u.b >= B || u.y0 % 4 || u.y0 >= H || u.x0 % 64 || u.x0 >= W) return printf("unit out of the map\n"), 1;
      if (!u.valid) continue;
      const int idx = (u.b * trows + u.y0 / 4) * cgroups + u.x0 / 64;
      if (idx != w.unit0 + s) return printf("unit decode is not the inverse of the unit order\n"), 1;
      seen[(size_t)idx * n_tiles + w.n0 / W_BN]++;
    }
  }
  for (int v : seen)
    if (v != 1) return printf("a (unit, cout slice) pair is handed out %d times\n", v), 1;
  return 0;
}

// ---- tile-linear units (Wino4LinGeom): DMA and transform reads of unit u of a (B, H, W) stack of maps, stage c0
struct LinCell {
  int kind;      // 0 = never written, 1 = zero fill, 2 = data
  int b;         // image
  long gpix;     // iy * W + ix
  int quad;
};
static int check_lin_unit(int B, int H, int W, int CIN, int u, int c0) {
  using G = Wino4LinGeom;
  const int tcols = (W + 3) / 4, trows = (H + 3) / 4, total = B * trows * tcols;
  const int b0 = wino4_lin_b0(u, tcols, trows, total);
  const long img = (long)H * W * CIN;
  // descriptor of the stage: base = first pixel of image b0 + c0 channels, records to the end of the tensor
  const long long rec = (long long)(B - b0) * img * 4 - 4LL * c0;
  const unsigned num_records = (unsigned)(rec > 0x7fffffffLL ? 0x7fffffffLL : rec);
  std::vector<LinCell> lds((size_t)G::PINSTR * 64, LinCell{0, 0, 0, 0});
  for (int lane = 0; lane < 64; ++lane) {
    const int td = (lane >> 1) & 15;
    const Wino4LinTile tile = wino4_lin_tile(16 * u + td, tcols, trows, total, b0);
    for (int i = 0; i < G::PINSTR; ++i) {
      const unsigned off = (unsigned)wino4_lin_patch_lane(i, tile, H, W, CIN, lane);
      LinCell& c = lds[(size_t)64 * i + lane];
      if (c.kind != 0) return printf("lin: LDS location written twice\n"), 1;
      if (off >= num_records) {
        c = LinCell{1, 0, 0, 0};
      } else {
        const long fl = (long)(off / 4) + c0;   // float index from the first pixel of image b0
        if ((off % 16) != 0) return printf("lin: bad offset\n"), 1;
        const long within = fl % img;
        c = LinCell{2, b0 + (int)(fl / img), within / CIN, (int)((within % CIN) - c0) / 4};
        if (((within % CIN) - c0) % 4 != 0 || (within % CIN) - c0 < 0 || (within % CIN) - c0 >= G::CB)
          return printf("lin: channel quad out of the stage\n"), 1;
      }
    }
  }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      std::set<int> banks[2];
      for (int lane = 0; lane < 64; ++lane) {
        const int t = lane & 15, g = lane >> 4;
        const int addr = wino4_lin_patch_base(t, g) + wino4_lin_patch_k(i, j);
        if (addr % 8 != 0 || addr / 16 >= (int)lds.size()) return printf("lin: read outside the patch block\n"), 1;
        banks[lane >> 5].insert((addr / 4) % 64);
        banks[lane >> 5].insert((addr / 4 + 1) % 64);
        const LinCell& c = lds[addr / 16];
        const Wino4LinTile tile = wino4_lin_tile(16 * u + t, tcols, trows, total, b0);
        const int iy = tile.y - 1 + i, ix = tile.x - 1 + j;
        if (c.kind == 0) return printf("lin: read of an LDS location the DMA never wrote\n"), 1;
        const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
        if (inside) {
          if (c.kind != 2 || c.b != b0 + tile.b || c.gpix != (long)iy * W + ix || c.quad != (g >> 1) ||
              ((addr % 16) / 8) != (g & 1))
            return printf("lin: unit %d lane %d (i=%d,j=%d): wrong element\n", u, lane, i, j), 1;
        } else if (c.kind != 1) {
          return printf("lin: element (%d,%d) outside the image is not a hardware zero\n", iy, ix), 1;
        }
      }
      if (banks[0].size() != 64 || banks[1].size() != 64)
        return printf("lin: transform read (i=%d,j=%d): bank conflict in a 32-lane group\n", i, j), 1;
    }
  return 0;
}
static int check_lin_stack(int B, int H, int W, int CIN) {
  const int tcols = (W + 3) / 4, trows = (H + 3) / 4, total = B * trows * tcols, units = (total + 15) / 16;
  // every tile is in exactly one unit, in raster order; tiles of a unit lie in images b0, b0 + 1, ...
  for (int u = 0; u < units; ++u) {
    if (u > 2 && u < units - 3 && u % 7) continue;
    for (int c0 = 0; c0 < CIN; c0 += CIN - 8 > 0 ? CIN - 8 : 8)
      if (check_lin_unit(B, H, W, CIN, u, c0)) return printf("  stack %dx%dx%dx%d unit %d stage %d\n", B, H, W, CIN, u, c0), 1;
  }
  for (int T = 0; T < total; ++T) {
    const int b0 = wino4_lin_b0(T / 16, tcols, trows, total);
    const Wino4LinTile t = wino4_lin_tile(T, tcols, trows, total, b0);
    const int b = b0 + t.b;
    if (!t.valid || t.b < 0 || b >= B || t.y % 4 || t.y >= H || t.x % 4 || t.x >= W) return printf("lin: tile out of the map\n"), 1;
    if ((b * trows + t.y / 4) * tcols + t.x / 4 != T) return printf("lin: tile decode is not the raster order\n"), 1;
  }
  return 0;
}

// ---- run-shaped units (Wino4RunGeom): DMA and transform reads of unit u of a (B, H, W) stack of maps, stage c0
static int check_run_unit(int B, int H, int W, int CIN, int u, int c0) {
  using G = Wino4RunGeom;
  const int tcols = (W + 3) / 4, trows = (H + 3) / 4, total = B * trows * tcols;
  int b0 = 0;
  const Wino4Runs R = wino4_runs(u, tcols, trows, &b0);
  if (R.n < 1 || R.n > G::RMAX) return printf("run: bad run count\n"), 1;
  {   // the runs cover the 16 tiles of the unit, in raster order (continuing into the virtual image behind the last)
    int t = 0;
    const int first[5] = {R.first0, R.first1, R.first2, R.first3, R.first4};
    const int rb[4] = {R.b0, R.b1, R.b2, R.b3}, ry[4] = {R.y0, R.y1, R.y2, R.y3}, rx[4] = {R.x0, R.x1, R.x2, R.x3};
    for (int r = 0; r < R.n; ++r) {
      if (first[r] != t) return printf("run: first[] is not the prefix sum\n"), 1;
      const int n = first[r + 1] - first[r];
      if (n < 1) return printf("run: empty run\n"), 1;
      const int T = 16 * u + t, b = T / (tcols * trows), rem = T % (tcols * trows);
      if (b != b0 + rb[r] || 4 * (rem / tcols) != ry[r] || 4 * (rem % tcols) != rx[r]) return printf("run: origin\n"), 1;
      if (rem % tcols + n > tcols) return printf("run: crosses a tile row\n"), 1;
      t += n;
    }
    if (t != 16) return printf("run: %d runs do not cover the unit (tcols %d)\n", R.n, tcols), 1;
  }
  const long img = (long)H * W * CIN;
  const long long rec = (long long)(B - b0) * img * 4 - 4LL * c0;
  const unsigned num_records = (unsigned)(rec > 0x7fffffffLL ? 0x7fffffffLL : (rec < 0 ? 0 : rec));
  std::vector<LinCell> lds((size_t)G::PINSTR * 64, LinCell{0, 0, 0, 0});
  for (int lane = 0; lane < 64; ++lane) {
    int fast[15];
    wino4_run_patch_lanes(fast, R, H, W, CIN, B - b0, lane);     // what the kernel computes == the definition
    for (int i = 0; i < G::PINSTR; ++i)
      if (fast[i] != wino4_run_patch_lane(i, R, H, W, CIN, B - b0, lane))
        return printf("run: the five-slot form of the lane offsets disagrees (piece %d lane %d)\n", i, lane), 1;
  }
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < G::PINSTR; ++i) {
      const unsigned off = (unsigned)wino4_run_patch_lane(i, R, H, W, CIN, B - b0, lane);
      LinCell& c = lds[(size_t)64 * i + lane];
      if (c.kind != 0) return printf("run: LDS location written twice\n"), 1;
      if (off >= num_records) {
        c = LinCell{1, 0, 0, 0};
      } else {
        const long fl = (long)(off / 4) + c0;
        if ((off % 16) != 0) return printf("run: bad offset\n"), 1;
        const long within = fl % img;
        c = LinCell{2, b0 + (int)(fl / img), within / CIN, (int)((within % CIN) - c0) / 4};
        if (((within % CIN) - c0) % 4 != 0 || (within % CIN) - c0 < 0 || (within % CIN) - c0 >= G::CB)
          return printf("run: channel quad out of the stage\n"), 1;
      }
    }
  int conflicts = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      std::multiset<int> banks[2];
      for (int lane = 0; lane < 64; ++lane) {
        const int t = lane & 15, g = lane >> 4;
        const int slot = t + wino4_run_of_tile(R, t);
        const int addr = wino4_run_patch_base(slot, g, j >> 2) + wino4_run_patch_k(i, j);
        if (addr % 8 != 0 || addr / 16 >= (int)lds.size()) return printf("run: read outside the patch block\n"), 1;
        banks[lane >> 5].insert((addr / 4) % 64);
        banks[lane >> 5].insert((addr / 4 + 1) % 64);
        const LinCell& c = lds[addr / 16];
        const Wino4LinTile tile = wino4_run_tile(R, t, u, total);
        const int iy = tile.y - 1 + i, ix = tile.x - 1 + j, b = b0 + tile.b;
        if (c.kind == 0) return printf("run: read of an LDS location the DMA never wrote\n"), 1;
        if ((16 * u + t < total) != (tile.valid != 0)) return printf("run: tile validity\n"), 1;
        const bool inside = b < B && iy >= 0 && iy < H && ix >= 0 && ix < W;
        if (inside) {
          if (c.kind != 2 || c.b != b || c.gpix != (long)iy * W + ix || c.quad != (g >> 1) || ((addr % 16) / 8) != (g & 1))
            return printf("run: unit %d lane %d (i=%d,j=%d): wrong element\n", u, lane, i, j), 1;
        } else if (c.kind != 1) {
          return printf("run: element (%d,%d) of image %d outside the stack is not a hardware zero\n", iy, ix, b), 1;
        }
      }
      // the slots of a unit span 16 + (runs - 1) rows: two lanes 16 slots apart share their banks (a 2-way conflict on at
      // most runs - 1 lane pairs per 32-lane group); everything else must be conflict free
      for (int h = 0; h < 2; ++h) {
        std::set<int> distinct(banks[h].begin(), banks[h].end());
        conflicts += 64 - (int)distinct.size();
        if ((int)distinct.size() < 64 - 4 * (R.n - 1)) return printf("run: more bank conflicts than the slack slots explain\n"), 1;
      }
    }
  (void)conflicts;
  return 0;
}
static int check_run_stack(int B, int H, int W, int CIN) {
  const int tcols = (W + 3) / 4, trows = (H + 3) / 4, total = B * trows * tcols, units = (total + 15) / 16;
  if (tcols < 5) return printf("run-shaped units need >= 5 tiles per row\n"), 1;
  for (int u = 0; u < units; ++u) {
    if (u > 3 && u < units - 4 && u % 5) continue;
    for (int c0 = 0; c0 < CIN; c0 += CIN - 8 > 0 ? CIN - 8 : 8)
      if (check_run_unit(B, H, W, CIN, u, c0)) return printf("  stack %dx%dx%dx%d unit %d stage %d\n", B, H, W, CIN, u, c0), 1;
  }
  return 0;
}

int main() {
  static_assert(Wino4RunGeom::LDS_BYTES + 16 <= 160 * 1024, "run-shaped units: patch blocks + two U buffers must fit LDS");
  {
    const int stacks[][4] = {{3, 10, 38, 256}, {5, 20, 75, 128}, {2, 40, 149, 64}, {4, 17, 20, 32}, {7, 9, 18, 40},
                             {2, 10, 125, 256}, {9, 5, 21, 40}, {1, 4, 64, 32}, {3, 3, 70, 64}, {6, 4, 17, 32}};
    for (const auto& st : stacks)
      if (check_run_stack(st[0], st[1], st[2], st[3])) return 1;
  }
  static_assert(Wino4LinGeom::LDS_BYTES + 16 <= 160 * 1024, "linear units: patch blocks + two U buffers must fit LDS");
  {
    const int stacks[][4] = {{3, 10, 38, 256}, {5, 20, 75, 128}, {2, 40, 149, 64}, {7, 1, 1, 32}, {4, 17, 9, 32},
                             {2, 10, 125, 256}, {9, 5, 3, 40}, {1, 4, 64, 32}, {3, 3, 70, 64}};
    for (const auto& st : stacks)
      if (check_lin_stack(st[0], st[1], st[2], st[3])) return 1;
  }
  static_assert(Wino4Geom::LDS_BYTES + 16 <= 160 * 1024, "patch blocks + two U buffers must fit the 160 KB of LDS");
  const int images[][3] = {{80, 998, 32}, {40, 499, 64}, {20, 250, 128}, {10, 125, 256}, {17, 9, 32}, {1, 1, 64},
                           {10, 38, 256}, {40, 149, 32}, {8, 128, 32}, {9, 129, 40}, {4, 64, 32}, {5, 65, 32}};
  for (const auto& im : images)
    if (check_image(im[0], im[1], im[2])) return 1;
  if (check_u_reads()) return 1;
  return check_order(7, 20, 250, 4) || check_order(3, 10, 125, 8) || check_order(1, 1, 1, 2) || check_order(5, 17, 9, 1) ||
         check_order(2, 80, 998, 1);
}
