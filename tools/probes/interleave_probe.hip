// Development probe (not part of the product library): what does a global->LDS DMA instruction cost the
// issuing wave when it is (a) bunched in front of a 128-MFMA run (the Winograd stage structure) or (b) spread
// through the MFMA run (software-pipelined prefetch)?  256-thread workgroups, `per_cu` of them per CU (LDS size
// forces the count); each wave: `iters` x { D DMA pieces + 128 v_mfma_f32_16x16x4_f32 }.
// pattern 0: linear 1-KB piece (U slab); pattern s > 0: 16 segments of 64 B, s bytes apart (input patch rows of
// a map with s/4 channels).  `span` = bytes of source a workgroup cycles through (cache-resident or not).
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int D, bool SPREAD>
__device__ __forceinline__ void run(f32x4 (&acc)[16], f32x4 a, f32x4 b, int iters, const __amdgpu_buffer_rsrc_t srd,
                                    float* lds, int voff, int step, int span_mask, int wave_off) {
  int pos = wave_off;
  for (int it = 0; it < iters; ++it) {
    if (!SPREAD) {
#pragma unroll
      for (int p = 0; p < D; ++p) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(lds + 256 * p), 16, voff, pos, 0, 0);
        pos = (pos + step) & span_mask;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int xi = 0; xi < 128; ++xi) {
      if (SPREAD && D > 0 && xi % (128 / (D > 0 ? D : 1)) == 0 && xi / (128 / (D > 0 ? D : 1)) < D) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(lds + 256 * (xi / (128 / (D > 0 ? D : 1)))), 16, voff,
                                                 pos, 0, 0);
        pos = (pos + step) & span_mask;
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[xi & 15] = MFMA16(a[xi & 3], b[xi & 3], acc[xi & 15]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (SPREAD) __builtin_amdgcn_s_waitcnt(0x0F70);
  }
}

extern "C" __global__ __launch_bounds__(256, 2) void k_interleave_probe(const float* src, int iters, int d, int spread,
                                                                        int pattern, int span, long long* out,
                                                                        float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2048; i += 256) smem[i] = src[i];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src) + (size_t)(blockIdx.x % 64) * (span / 4), 0, span, 0x00020000);
  float* lds = smem + 2048 + wv * 16 * 256;                 // 16 KB of landing space per wave
  const int voff = pattern == 0 ? lane * 16 : (lane >> 2) * pattern + (lane & 3) * 16;
  const int step = pattern == 0 ? 1024 : (pattern >= 1024 ? 64 : 16 * pattern);
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 a = *reinterpret_cast<const f32x4*>(smem + lane * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(smem + 1024 + lane * 4);
  const int mask = span / 2 - 1, woff = wv * (span / 8);
  const long long t0 = __builtin_amdgcn_s_memtime();
#define CASE(DD)                                                                                  \
  case DD:                                                                                        \
    if (spread) run<DD, true>(acc, a, b, iters, srd, lds, voff, step, mask, woff);                \
    else run<DD, false>(acc, a, b, iters, srd, lds, voff, step, mask, woff);                      \
    break;
  switch (d) {
    CASE(0) CASE(4) CASE(8) CASE(16)
    default: break;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0];
  for (int i = 1; i < 16; ++i) s += acc[i];
  if (s[0] == 123.456f) sink[tid] = s[1] + s[2] + s[3];
  if (lane == 0) out[blockIdx.x * 4 + wv] = t1 - t0;
}

extern "C" int interleave_probe(const void* src, int iters, int d, int spread, int pattern, int span, int per_cu,
                                void* out, void* sink, int grid, void* stream) {
  const int lds = per_cu == 1 ? 120 * 1024 : 72 * 1024;
  (void)hipFuncSetAttribute((const void*)k_interleave_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k_interleave_probe, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const float*)src, iters, d,
                     spread, pattern, span, (long long*)out, (float*)sink);
  return (int)hipGetLastError();
}
