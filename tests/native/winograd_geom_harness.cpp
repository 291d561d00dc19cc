// Host harness for csrc/emb_winograd_geom.h (the integer geometry of k_conv3x3_wino, compiled unchanged):
// replays, for every wave and lane, (1) the LDS-DMA of the input patch -- which global element, or a
// hardware zero, lands in which LDS row / slot -- and (2) the ds_read_b128 addresses of the input transform,
// and checks that every read returns exactly the patch element the Winograd tile needs (zeros in the halo
// outside the image), that no read touches an LDS location the DMA did not write, and that every read
// instruction is bank-conflict free (MI355X_MICROARCH.md: a wave's ds_read_b128 is served in 4 groups of 16
// lanes; 64 banks of 4 bytes).  Also: the tile order covers every (pixel tile, cout slice) exactly once and keeps
// the cout slices of one pixel tile on one XCD.  Exit code 0 = all good.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define __device__
#define __forceinline__ inline
#include "emb_winograd_geom.h"
using namespace pa;

struct Cell {
  int kind;      // 0 = never written, 1 = zero fill, 2 = data
  long gpix;     // linear pixel index (iy * W + ix) of the image
  int quad;      // channel quad 0..3 of the 16-channel stage
};

static const int GROUPS[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

template <int TR, int TCG>
static int check_tile(int H, int W, int CIN, int y0, int x0, int x0_last, int c0) {
  using G = WinoGeom<TR, TCG>;
  WinoTile q{0, 0, y0, x0, 1};
  const long img = (long)H * W * CIN;
  const long org = ((long)(y0 - 1) * W + (x0 - 1)) * CIN + c0;
  const unsigned num_records = (unsigned)((img - org) * 4);
  std::vector<Cell> lds((size_t)G::PINSTR * 16 * 4, Cell{0, 0, 0});
  const int keep = wino_patch_keep(q, x0_last);
  for (int slw = 0; slw < HW; ++slw)
    for (int lane = 0; lane < 64; ++lane) {
      int prel[G::NPP];
      wino_patch_lanes<TR, TCG>(prel, W, CIN, lane, slw, x0_last);
      for (int i = 0; i < G::NPP; ++i) {
        const int k = slw + HW * i;
        if (k >= G::PINSTR) break;
        const unsigned off = (unsigned)(prel[i] & keep);
        Cell& c = lds[(size_t)(16 * k + (lane >> 2)) * 4 + (lane & 3)];
        if (c.kind != 0) return printf("LDS location written twice\n"), 1;
        if (off >= num_records) {   // class bit survived, padding lane, or below the image
          c = Cell{1, 0, 0};
        } else {
          const long fl = (long)(off / 4) + org - c0;   // float index within the image, channel 0-based in the stage
          if (fl < 0 || (off % 16) != 0) return printf("bad offset\n"), 1;
          c = Cell{2, fl / CIN, (int)(fl % CIN) / 4};
          if ((fl % CIN) % 4 != 0 || (fl % CIN) >= WCB) return printf("channel quad out of the stage\n"), 1;
        }
      }
    }
  for (int slw = 0; slw < HW; ++slw) {
    const int wr = slw / TCG, wc = slw % TCG;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        const int K = (2 * i + (j & 1)) * G::PWH + (j >> 1);
        int slot16[64];
        for (int lane = 0; lane < 64; ++lane) {
          const int t = lane & 15, g = lane >> 4;
          int pbase[8];
          wino_patch_bases<TR, TCG>(pbase, t, g, wr, wc);
          const int addr = pbase[K & 7] + (K & ~7) * (WCB * 4);
          slot16[lane] = (addr / 16) % 16;
          if (addr % 16 != 0 || addr / 16 >= (int)lds.size()) return printf("read outside the patch image\n"), 1;
          const Cell& c = lds[addr / 16];
          const int py = 2 * wr + i, px = 2 * (16 * wc + t) + j;
          const int iy = y0 - 1 + py, ix = x0 - 1 + px;
          if (c.kind == 0) return printf("read of an LDS location the DMA never wrote (i=%d j=%d lane=%d)\n", i, j, lane), 1;
          const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
          if (inside) {
            if (c.kind != 2 || c.gpix != (long)iy * W + ix || c.quad != g)
              return printf("tile (%d,%d) wave %d lane %d (i=%d,j=%d): wrong element\n", y0, x0, slw, lane, i, j), 1;
          } else if (iy < 0 || ix < 0 || iy >= H || ix == W) {
            if (c.kind != 1) return printf("halo element (%d,%d) is not a hardware zero\n", iy, ix), 1;
          }   // ix > W: feeds only output tiles that are never stored
        }
        for (const auto& grp : GROUPS) {
          std::set<int> banks;
          for (int l : grp) banks.insert(slot16[l]);
          if (banks.size() != 16) return printf("bank conflict in the transform read (i=%d, j=%d)\n", i, j), 1;
        }
      }
  }
  return 0;
}

template <int TR, int TCG>
static int check_image(int H, int W, int CIN) {
  const int th = 2 * TR, tw = 32 * TCG;
  const int tiles_h = (H + th - 1) / th, tiles_w = (W + tw - 1) / tw;
  const int x0_last = (tiles_w - 1) * tw;
  for (int ty = 0; ty < tiles_h; ++ty)
    for (int tx = 0; tx < tiles_w; ++tx) {
      if (ty > 1 && ty < tiles_h - 2 && tx > 1 && tx < tiles_w - 2 && (ty * 7 + tx) % 5) continue;   // sample the interior
      for (int c0 = 0; c0 < CIN; c0 += CIN - WCB > 0 ? CIN - WCB : WCB)
        if (check_tile<TR, TCG>(H, W, CIN, ty * th, tx * tw, x0_last, c0)) {
          printf("  geometry <%d,%d> image %dx%dx%d tile (%d,%d) stage %d\n", TR, TCG, H, W, CIN, ty, tx, c0);
          return 1;
        }
    }
  return 0;
}

static int check_decode(int tiles_w, int tiles_h, int B, int n_tiles) {
  const int tiles_hw = tiles_w * tiles_h;
  const int num_pb = tiles_hw * B;
  const int total = ((num_pb + 7) / 8) * 8 * n_tiles;
  std::vector<int> seen((size_t)num_pb * n_tiles, 0);
  for (int q = 0; q < total; ++q) {
    const WinoTile t = wino_decode(q, tiles_w, tiles_hw, n_tiles, 8, 32, num_pb);
    const int pix = (t.y0 / 8) * tiles_w + t.x0 / 32, pb = t.b * tiles_hw + pix;
    if (t.n0 % W_BN || t.n0 / W_BN >= n_tiles || pb >= num_pb) return printf("decode out of range\n"), 1;
    if (!t.valid) continue;
    if ((pb & 7) != (q & 7)) return printf("tile %d: pixel tile %d is not on XCD %d\n", q, pb, q & 7), 1;
    seen[(size_t)pb * n_tiles + t.n0 / W_BN]++;
  }
  for (int v : seen)
    if (v != 1) return printf("a (pixel tile, cout slice) pair is handed out %d times\n", v), 1;
  return 0;
}

int main() {
  const int images[][3] = {{80, 998, 32}, {40, 499, 64}, {20, 250, 128}, {10, 125, 256}, {17, 9, 32}, {1, 1, 64},
                           {10, 38, 256}, {40, 149, 32}};
  for (const auto& im : images) {
    if (check_image<4, 1>(im[0], im[1], im[2])) return 1;
    if (check_image<2, 2>(im[0], im[1], im[2])) return 1;
    if (check_image<1, 4>(im[0], im[1], im[2])) return 1;
  }
  if (check_decode(32, 10, 7, 1) || check_decode(16, 5, 3, 2) || check_decode(4, 5, 5, 4) ||
      check_decode(1, 5, 9, 8) || check_decode(1, 1, 1, 2))
    return 1;
  return 0;
}
