// Dynamic tile claiming for persistent kernels (included by common.h; also compiled for the HOST by
// tests/test_tile_queue_cpu.py, with std::atomic stand-ins for atomicAdd / atomicExch, to check the protocol
// under real concurrency).  Needs: __device__, __forceinline__, atomicAdd(int*, int), atomicExch(int*, int).
#pragma once

namespace pa {

// ---------------------------------------------------------------------------------------------
// Dynamic tile claiming for the persistent convolution kernels.  A grid of resident workgroups that
// partitions its tiles STATICALLY (tile q, q + grid, ...) is at the mercy of anything else on the chip:
// when a foreign workgroup (the dendrogram merge of the previous file, on another stream) keeps a few of
// them from being placed, those start a whole round late and the launch takes 1.6x as long
// (tools/probes/interference_probe.py).  Claimed tiles make a late workgroup harmless: it just finds
// less left.  Workgroup w runs on XCD w % 8 and each XCD has its own L2, so there is one counter per
// XCD (tile q belongs to XCD q % 8 -- the tile orders keep the workgroups that share an input patch on
// one XCD); a workgroup whose own XCD has run dry steals from the others.
//   counters: 16 ints (tile_counters()): [0..7] next index per XCD, [8] workgroups that have finished; the
//   last one to finish zeroes them again, so a block is ready for its next launch without a memset.
// ---------------------------------------------------------------------------------------------

struct TileQueue {
  int* ctr;
  int xcd;      // blockIdx.x & 7
  int per_xcd;  // tiles per XCD (total / 8)
};
// the workgroup's own XCD: returns the raw counter value (issue early, resolve late: the atomic's
// round trip through the fabric is hidden behind the tile being processed)
__device__ __forceinline__ int tq_claim_own(const TileQueue& tq) { return atomicAdd(tq.ctr + tq.xcd, 1); }
// raw value -> tile index q (q % 8 = owning XCD), stealing from the other XCDs when the own one is done;
// -1 when nothing is left anywhere
__device__ __forceinline__ int tq_resolve(const TileQueue& tq, int r) {
  if (r < tq.per_xcd) return r * 8 + tq.xcd;
  for (int j = 1; j < 8; ++j) {
    const int y = (tq.xcd + j) & 7;
    const int s = atomicAdd(tq.ctr + y, 1);
    if (s < tq.per_xcd) return s * 8 + y;
  }
  return -1;
}
// same for an index space that is NOT a multiple of 8: indices >= total are holes at the end of some XCDs' ranges
__device__ __forceinline__ int tq_resolve_upto(const TileQueue& tq, int r, int total) {
  for (;;) {
    const int q = tq_resolve(tq, r);
    if (q < total) return q;   // a tile, or -1
    r = tq_claim_own(tq);
  }
}
// one thread per workgroup, after its last claim
__device__ __forceinline__ void tq_done(const TileQueue& tq, int num_workgroups) {
  if (atomicAdd(tq.ctr + 8, 1) == num_workgroups - 1) {
#pragma unroll
    for (int i = 0; i < 9; ++i) atomicExch(tq.ctr + i, 0);
  }
}

}  // namespace pa
