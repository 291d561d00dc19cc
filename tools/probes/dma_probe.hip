// Development probe (not part of the product library): sustained global -> LDS (LDS-DMA) rate of ONE CU's
// workgroup as a function of the request shape, the number of issuing waves and the source footprint.
// Built by tools/probes/build.sh into tools/_dbg/libprobes.so.
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// mode 0: linear 1-KB pieces; mode 1: 64-byte segments `stride` bytes apart (4 lanes each);
// mode 2: 128-byte segments `stride` bytes apart (8 lanes each); mode 3: 256-byte segments (16 lanes)
extern "C" __global__ __launch_bounds__(512, 2) void k_dma_probe(const float* src, long src_bytes, int mode,
                                                              int stride, int nwaves, int pieces, int iters,
                                                              long long* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every workgroup streams its own power-of-two region so that hits come from re-use over `iters`
  const long region = src_bytes / gridDim.x;
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src) + (long)blockIdx.x * region / 4, 0, (int)region, 0x00020000);
  const int lanes_per_seg = mode == 1 ? 4 : mode == 2 ? 8 : mode == 3 ? 16 : 64;
  const int seg = lane / lanes_per_seg, within = lane % lanes_per_seg;
  const int segs_per_piece = 64 / lanes_per_seg;
  long long t0 = 0;
  long cursor = 0;  // in pieces
  __syncthreads();
  if (wv < nwaves) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
      for (int p = 0; p < pieces; ++p) {
        const long piece = cursor + (long)wv * pieces + p;
        long off;
        if (mode == 0) off = piece * 1024 + lane * 16;
        else off = (piece * segs_per_piece + seg) * (long)stride + within * 16;
        off &= (region - 1) & ~15L;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + (wv * pieces + p) % 96 * 256), 16,
                                                 (int)off, 0, 0, 0);
      }
      cursor += (long)nwaves * pieces;
      // allow one iteration of pieces in flight
      if (pieces <= 4) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
      else if (pieces <= 8) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
      else __builtin_amdgcn_s_waitcnt(0x0F70 | 12);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
  }
}

extern "C" int dma_probe(const void* src, long src_bytes, int mode, int stride, int nwaves, int pieces, int iters,
                         void* out, int grid, void* stream) {
  hipFuncSetAttribute((const void*)k_dma_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL(k_dma_probe, dim3(grid), dim3(512), 96 * 1024, (hipStream_t)stream, (const float*)src,
                     src_bytes, mode, stride, nwaves, pieces, iters, (long long*)out);
  return (int)hipGetLastError();
}
