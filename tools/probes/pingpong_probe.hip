// Development probe (not part of the product library): what can run beside an f32 MFMA stream?
// One 512-thread workgroup per CU = two halves of 4 waves (one wave of each half per SIMD).  Half 0 issues
// `nmfma` v_mfma_f32_16x16x4_f32 per iteration (optionally with the B-fragment ds_read_b128 pattern of the
// Winograd kernel); half 1 does a selectable amount of VALU / LDS-read / LDS-DMA work per iteration.  No
// barriers between the halves: each half is timed on its own (cycles per iteration).
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int K>
__device__ __forceinline__ void interleave(f32x4 (&acc)[16], float (&z)[8], f32x4 a, f32x4 b0, int iters) {
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int xi = 0; xi < 128; ++xi) {
      acc[xi & 15] = MFMA16(a[xi & 3], b0[xi & 3], acc[xi & 15]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) z[k] += a[k & 3];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

extern "C" __global__ __launch_bounds__(512, 2) void k_pingpong_probe(const float* src, int iters, int mfma_on,
                                                                   int breads, int valu, int dsr, int dma,
                                                                   long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wv >> 2;
  for (int i = tid; i < 24 * 1024; i += 512) smem[i] = src[i & 4095];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t srd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src) + blockIdx.x * 16384, 0, 65536, 0x00020000);
  long long t0 = __builtin_amdgcn_s_memtime();
  if (h == 0) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a = *reinterpret_cast<const f32x4*>(smem + lane * 4);
    f32x4 b0 = *reinterpret_cast<const f32x4*>(smem + 1024 + lane * 4), b1 = b0;
    const float* ub = smem + 2048 + lane * 4;
    if (mfma_on >= 2) {
      // same-wave interleave: K = mfma_on - 2 independent VALU adds after every MFMA
      float z[8];
      for (int i = 0; i < 8; ++i) z[i] = a[i & 3] + i;
      switch (mfma_on - 2) {
        case 0: interleave<0>(acc, z, a, b0, iters); break;
        case 1: interleave<1>(acc, z, a, b0, iters); break;
        case 2: interleave<2>(acc, z, a, b0, iters); break;
        case 4: interleave<4>(acc, z, a, b0, iters); break;
        case 6: interleave<6>(acc, z, a, b0, iters); break;
        default: interleave<8>(acc, z, a, b0, iters); break;
      }
      for (int i = 0; i < 8; ++i) acc[0][0] += z[i];
    } else if (mfma_on)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
          if (breads) {
            b0 = *reinterpret_cast<const f32x4*>(ub + (xi * 512) % 8192);
            b1 = *reinterpret_cast<const f32x4*>(ub + (xi * 512 + 256) % 8192);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            acc[xi & 15] = MFMA16(a[ks], b0[ks], acc[xi & 15]);
            acc[(xi + 8) & 15] = MFMA16(a[ks], b1[ks], acc[(xi + 8) & 15]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    f32x4 s = acc[0];
    for (int i = 1; i < 16; ++i) s += acc[i];
    if (s[0] == 123.456f) sink[tid] = s[1] + s[2] + s[3];
  } else {
    f32x4 x = *reinterpret_cast<const f32x4*>(smem + lane * 4), y = x + 1.0f;
    f32x4 r = x;
    const int lw = wv & 3;
    for (int it = 0; it < iters; ++it) {
      for (int p = 0; p < dma; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(smem + 12288 + ((lw * 10 + p) % 40) * 256), 16,
                                                 lane * 16 + ((it * 7 + p) & 31) * 1024, 0, 0, 0);
      for (int k = 0; k < dsr; ++k) r += *reinterpret_cast<const f32x4*>(smem + 4096 + ((k * 256 + lane * 4) & 4095));
#pragma unroll 8
      for (int k = 0; k < valu; k += 8) {   // 8 independent v_add/v_sub per trip
        x += y; r -= x; y -= r; x += r;
      }
      if (dma) __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    if (r[0] + x[1] + y[2] == 123.456f) sink[tid] = r[1];
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
}

extern "C" int pingpong_probe(const void* src, int iters, int mfma_on, int breads, int valu, int dsr, int dma,
                              void* out, void* sink, int grid, void* stream) {
  (void)hipFuncSetAttribute((const void*)k_pingpong_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL(k_pingpong_probe, dim3(grid), dim3(512), 96 * 1024, (hipStream_t)stream, (const float*)src,
                     iters, mfma_on, breads, valu, dsr, dma, (long long*)out, (float*)sink);
  return (int)hipGetLastError();
}
