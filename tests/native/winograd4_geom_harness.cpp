// Host harness for csrc/emb_winograd4_geom.h (the integer geometry of k_conv3x3_wino4, compiled unchanged):
// replays, for every wave and lane, (1) the LDS-DMA of the input patch -- which global element, or a hardware zero,
// lands in which 16-byte LDS slot -- and (2) the ds_read_b64 addresses of the input transform, and checks that
// every read returns exactly the patch element (pixel, channel pair) the Winograd tile needs, zeros wherever the
// pixel lies outside the image (F(4x4) mixes all six patch columns into every output of a tile, so EVERY column
// past the border must be a zero, not only the first), that no read touches an LDS location the DMA did not write,
// and that the 64 lanes of one read cover 512 contiguous bytes (bank-conflict free at two passes).  Same for the
// U-slab reads.  Exit code 0 = all good.
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

static int check_tile(int H, int W, int CIN, int y0, int x0, int x0_last, int c0) {
  using G = Wino4Geom;
  WinoTile q{0, 0, y0, x0, 1};
  const long img = (long)H * W * CIN;
  const long org = ((long)(y0 - 1) * W + (x0 - 1)) * CIN + c0;
  const unsigned num_records = (unsigned)((img - org) * 4);
  std::vector<Cell> lds((size_t)G::PINSTR * 64, Cell{0, 0, 0});   // 16-byte slots
  const int keep = wino_patch_keep(q, x0_last);
  for (int slw = 0; slw < 4; ++slw)
    for (int lane = 0; lane < 64; ++lane) {
      int prel[G::NPP];
      wino4_patch_lanes(prel, W, CIN, lane, slw, x0_last);
      for (int i = 0; i < G::NPP; ++i) {
        const int k = slw + 4 * i;
        if (k >= G::PINSTR) break;
        const unsigned off = (unsigned)(prel[i] & keep);
        Cell& c = lds[(size_t)64 * k + lane];
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
  for (int slw = 0; slw < 4; ++slw) {
    const int wr = slw >> 1, wc = slw & 1;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        std::set<int> bytes;
        int lo = 1 << 30, hi = 0;
        for (int lane = 0; lane < 64; ++lane) {
          const int t = lane & 15, g = lane >> 4;
          const int addr = wino4_patch_base(t, g, wr, wc) + wino4_patch_k(i, j);
          if (addr % 8 != 0 || addr / 16 >= (int)lds.size()) return printf("read outside the patch image\n"), 1;
          lo = addr < lo ? addr : lo;
          hi = addr > hi ? addr : hi;
          bytes.insert(addr);
          const Cell& c = lds[addr / 16];
          const int py = 4 * wr + i, px = 4 * (16 * wc + t) + j;
          const int iy = y0 - 1 + py, ix = x0 - 1 + px;
          if (c.kind == 0) return printf("read of an LDS location the DMA never wrote (i=%d j=%d lane=%d)\n", i, j, lane), 1;
          const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
          if (inside) {
            // the 8-byte read takes channels 2g, 2g+1: quad g >> 1, second half of the slot when g is odd
            if (c.kind != 2 || c.gpix != (long)iy * W + ix || c.quad != (g >> 1) || ((addr % 16) / 8) != (g & 1))
              return printf("tile (%d,%d) wave %d lane %d (i=%d,j=%d): wrong element\n", y0, x0, slw, lane, i, j), 1;
          } else {
            if (c.kind != 1) return printf("element (%d,%d) outside the image is not a hardware zero\n", iy, ix), 1;
          }
        }
        if (bytes.size() != 64 || hi - lo != 504) return printf("transform read (i=%d,j=%d) is not 512 contiguous bytes\n", i, j), 1;
      }
  }
  return 0;
}

static int check_image(int H, int W, int CIN) {
  using G = Wino4Geom;
  const int tiles_h = (H + G::TH - 1) / G::TH, tiles_w = (W + G::TW - 1) / G::TW;
  const int x0_last = (tiles_w - 1) * G::TW;
  for (int ty = 0; ty < tiles_h; ++ty)
    for (int tx = 0; tx < tiles_w; ++tx) {
      if (ty > 1 && ty < tiles_h - 2 && tx > 1 && tx < tiles_w - 2 && (ty * 7 + tx) % 5) continue;
      for (int c0 = 0; c0 < CIN; c0 += CIN - G::CB > 0 ? CIN - G::CB : G::CB)
        if (check_tile(H, W, CIN, ty * G::TH, tx * G::TW, x0_last, c0)) {
          printf("  image %dx%dx%d tile (%d,%d) stage %d\n", H, W, CIN, ty, tx, c0);
          return 1;
        }
    }
  return 0;
}

static int check_u_reads() {
  for (int xi = 0; xi < 36; ++xi)
    for (int cg = 0; cg < 2; ++cg) {
      std::set<int> seen;
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, g = lane >> 4;
        const int addr = wino4_u_base(m, g) + wino4_u_k(xi, cg);
        // slab image: row = 32 xi + n (n = 16 cg + m), 8 input channels of 4 bytes; the pair g at byte 8 g
        if (addr != ((32 * xi + 16 * cg + m) * 8 + 2 * g) * 4) return printf("U read address\n"), 1;
        if (addr + 8 > Wino4Geom::USLAB_BYTES) return printf("U read outside the slab\n"), 1;
        seen.insert(addr);
      }
      if (seen.size() != 64 || *seen.rbegin() - *seen.begin() != 504) return printf("U read not contiguous\n"), 1;
    }
  return 0;
}

int main() {
  static_assert(2 * Wino4Geom::BUF_BYTES + 16 <= 160 * 1024, "two stage buffers must fit the 160 KB of LDS");
  const int images[][3] = {{80, 998, 32}, {40, 499, 64}, {20, 250, 128}, {10, 125, 256}, {17, 9, 32}, {1, 1, 64},
                           {10, 38, 256}, {40, 149, 32}, {8, 128, 32}, {9, 129, 40}};
  for (const auto& im : images)
    if (check_image(im[0], im[1], im[2])) return 1;
  return check_u_reads();
}
