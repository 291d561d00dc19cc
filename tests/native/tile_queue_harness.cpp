// Host harness for csrc/tile_queue.h: the device code itself, compiled with std::atomic stand-ins for the
// HIP atomics, driven by real threads the way the persistent convolution kernels drive it (claim one tile
// ahead, resolve at the tile boundary, steal when the own XCD is dry, last workgroup resets the block).
// usage: harness <workgroups> <tiles> <launches> <upto: 0|1> <seed>  ->  exit code 0 iff every tile of every
// launch was handed out exactly once, no hole was handed out, and the counter block is zero after each launch.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>
#define __device__
#define __forceinline__ inline
static inline int atomicAdd(int* p, int v) {
  return reinterpret_cast<std::atomic<int>*>(p)->fetch_add(v, std::memory_order_relaxed);
}
static inline int atomicExch(int* p, int v) {
  return reinterpret_cast<std::atomic<int>*>(p)->exchange(v, std::memory_order_relaxed);
}
#include "tile_queue.h"

int main(int argc, char** argv) {
  const int G = atoi(argv[1]), total = atoi(argv[2]), launches = atoi(argv[3]), upto = atoi(argv[4]);
  const unsigned seed = (unsigned)atoi(argv[5]);
  alignas(64) static int ctr[16] = {0};
  const int per_xcd = upto ? (total + 7) / 8 : total / 8;
  const int span = per_xcd * 8;
  for (int l = 0; l < launches; ++l) {
    std::vector<std::atomic<int>> seen(span > 0 ? span : 1);
    for (auto& s : seen) s.store(0);
    std::vector<std::thread> wgs;
    for (int w = 0; w < G; ++w)
      wgs.emplace_back([&, w] {
        std::mt19937 rng(seed * 7919u + (unsigned)l * 131u + (unsigned)w);
        if (rng() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));  // placed late
        const pa::TileQueue tq{ctr, w & 7, per_xcd};
        auto resolve = [&](int r) { return upto ? pa::tq_resolve_upto(tq, r, total) : pa::tq_resolve(tq, r); };
        int q = resolve(pa::tq_claim_own(tq));
        while (q >= 0) {
          const int ahead = pa::tq_claim_own(tq);                 // next tile, claimed while this one "runs"
          seen[q].fetch_add(1);
          if (rng() % 8 == 0) std::this_thread::yield();
          q = resolve(ahead);
        }
        pa::tq_done(tq, G);
      });
    for (auto& t : wgs) t.join();
    const int valid = upto ? total : span;
    for (int q = 0; q < span; ++q) {
      const int n = seen[q].load();
      if (n != (q < valid ? 1 : 0)) {
        printf("launch %d: tile %d handed out %d times (valid %d)\n", l, q, n, q < valid);
        return 1;
      }
    }
    for (int i = 0; i < 16; ++i)
      if (ctr[i] != 0) {
        printf("launch %d: counter %d = %d after the launch\n", l, i, ctr[i]);
        return 2;
      }
  }
  return 0;
}
