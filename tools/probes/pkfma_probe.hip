// How many cycles does a v_pk_fma_f32 cost a wave, by operand kind?  (development probe: the input transform of
// csrc/emb_winograd4.hip is 144 of them per stage, with scalar-register constants.)  One wave per SIMD (256 threads per
// workgroup, one workgroup per CU), a chain-free block of 32 independent instructions repeated `iters` times, timed
// with s_memtime.  mode 0: v_pk_fma_f32 v, v, v(const), v   1: v_pk_fma_f32 v, v, s[pair], v   2: v_pk_add_f32 v, v, v
// 3: two v_fma_f32 (unpacked) per value pair, scalar constant   4: v_pk_mul_f32 v, v, s[pair]
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_pkfma(const float* __restrict__ src, int iters, long long* __restrict__ out,
                                               float* __restrict__ sink) {
  f2 a[16], c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = f2{src[threadIdx.x + 64 * i], src[threadIdx.x + 64 * i + 1]};
    c[i] = f2{src[threadIdx.x + 7 * i + 3], src[threadIdx.x + 5 * i + 1]};
  }
  float ks = src[0];
  asm volatile("" : "+s"(ks));
  const f2 kc = {ks, ks};
  f2 kv = {src[threadIdx.x & 3], src[(threadIdx.x & 3) + 1]};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 0) a[i] = __builtin_elementwise_fma(a[i], kv, c[i]);
        else if (MODE == 1) a[i] = __builtin_elementwise_fma(a[i], kc, c[i]);
        else if (MODE == 2) a[i] = a[i] + c[i];
        else if (MODE == 3) {
          a[i].x = __builtin_fmaf(a[i].x, ks, c[i].x);
          a[i].y = __builtin_fmaf(a[i].y, ks, c[i].y);
        } else a[i] = a[i] * kc;
      }
    asm volatile("" ::: "memory");
  }
  const long long t1 = __builtin_readcyclecounter();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += a[i].x + a[i].y;
  if (acc == 12345.678f) sink[0] = acc;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

extern "C" int pkfma_probe(const float* src, int iters, int mode, long long* out, float* sink, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(k_pkfma<0>, dim3(grid), dim3(256), 0, st, src, iters, out, sink); break;
    case 1: hipLaunchKernelGGL(k_pkfma<1>, dim3(grid), dim3(256), 0, st, src, iters, out, sink); break;
    case 2: hipLaunchKernelGGL(k_pkfma<2>, dim3(grid), dim3(256), 0, st, src, iters, out, sink); break;
    case 3: hipLaunchKernelGGL(k_pkfma<3>, dim3(grid), dim3(256), 0, st, src, iters, out, sink); break;
    default: hipLaunchKernelGGL(k_pkfma<4>, dim3(grid), dim3(256), 0, st, src, iters, out, sink); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
