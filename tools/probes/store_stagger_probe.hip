// Development probe (round 6): does a scattered 16-byte-per-lane store cost its wave less when the OTHER waves of the
// CU are not storing at the same time?  (The F(4x4) epilogue: all four waves of a workgroup reach their 32 stores
// together -- the stage barrier keeps them in step -- and each store holds its wave 200-245 cycles, 27 B/clk per CU,
// profiles/r5_wino4_store_anatomy.txt.  If that is the CU's store path being shared by four, a lone storing wave
// should get through in a quarter of the time while the other SIMDs run MFMAs.)
// One workgroup of 4 waves per CU (one per SIMD).  Waves in `mask` issue NS back-to-back buffer_store_dwordx4 of 16
// segments of 64 B (`seg_stride` apart); the others run `mfma_iters` x 16 v_mfma_f32_16x16x4_f32.  Cycles per wave.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NS>
__global__ __launch_bounds__(256, 1) void k_store_stagger(float* buf, long wave_bytes, int mask, int seg_stride,
                                                          int row_stride, int mfma_iters, int pattern,
                                                          long long* out, float* sink) {
  extern __shared__ unsigned char smem_ss[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, g = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + w;
  char* base = (char*)buf + wave * wave_bytes;
  const int off = pattern == 0 ? t * seg_stride + 16 * g : (t >> 1) * 2 * seg_stride + 64 * (t & 1) + 16 * g;
  __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)wave_bytes, 0x00020000);
  f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
  int zero = 0;
  asm volatile("" : "+s"(zero));
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = (float)(lane & 7) * 0.25f, b = (float)(lane >> 3) * 0.125f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  long long t1;
  if ((mask >> w) & 1) {
    if (pattern == 3) {
      // k_stem's pair of stores: lane i writes 16 B at 32 i, then the other 16 B of its 32 (two half-filled sectors per
      // lane and instruction)
      const int off16 = lane * 32;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(v), "v"(off16 + 16 * (i & 1)), "s"(srd),
                     "s"(__builtin_amdgcn_readfirstlane((i >> 1) * row_stride + zero)) : "memory");
      }
    } else if (pattern == 2) {
      // the direct convolution's epilogue (csrc/emb_resnet.hip): ONE dword per lane, lanes 0-31 = 32 consecutive channels
      // of a pixel (128 B), lanes 32-63 the pixel four further on
      const int off4 = (lane & 31) * 4 + (lane >> 5) * 4 * seg_stride;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        asm volatile("buffer_store_dword %0, %1, %2, %3 offen" :: "v"(v[0]), "v"(off4), "s"(srd),
                     "s"(__builtin_amdgcn_readfirstlane(i * row_stride + zero)) : "memory");
      }
    } else {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(srd),
                   "s"(__builtin_amdgcn_readfirstlane(i * row_stride + zero)) : "memory");
    }
    }
    t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
  }
  const long long t2 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += acc[i];
  if (s[0] == 123.456f) sink[threadIdx.x] = s[1] + smem_ss[threadIdx.x];
  if (lane == 0) {
    out[wave * 2] = t1 - t0;
    out[wave * 2 + 1] = t2 - t0;
  }
}

extern "C" int store_stagger_probe(float* buf, long wave_bytes, int mask, int seg_stride, int row_stride, int mfma_iters,
                                   int pattern, long long* out, float* sink, int grid, void* stream) {
  const int lds = 100 * 1024;
  (void)hipFuncSetAttribute((const void*)k_store_stagger<32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k_store_stagger<32>, dim3(grid), dim3(256), lds, (hipStream_t)stream, buf, wave_bytes, mask, seg_stride,
                     row_stride, mfma_iters, pattern, out, sink);
  return (int)hipGetLastError();
}
