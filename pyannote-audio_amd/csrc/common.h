// Common helpers for the gfx950 (MI355X / CDNA4) kernels of the speaker-diarization hot path.
// Everything here is written for wave64 + MFMA (f32-input forms) only; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// D(16x16) += A(16x4) * B(4x16);  lane l supplies A[l&15][l>>4], B[l>>4][l&15];
// D: lane l holds column l&15, rows 4*(l>>4)+r, r = 0..3.
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// D(32x32) += A(32x2) * B(2x32);  lane l supplies A[l&31][l>>5], B[l>>5][l&31];
// D: lane l holds column l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// launchers shared between translation units of the library that are NOT part of the C ABI (include/pyannote_amd.h
// declares every exported symbol; tests/test_capi.py checks both directions)
#define PA_INTERNAL __attribute__((visibility("hidden")))

namespace pa {

void set_error(const char* fmt, ...);

// tile queue of the persistent convolution kernels: protocol in tile_queue.h, counter pool in pa_core.cpp
int* tile_counters();  // a zeroed 16-int block in device memory of the current device (rotating pool)
}  // namespace pa
#include "tile_queue.h"
namespace pa {

// HIP-event profiler scope (pa_core.cpp): brackets the launches issued while it is alive with two
// events on `stream`; `flops` / `bytes` are the ALGORITHMIC work of those launches.
struct ProfScope {
  ProfScope(const char* name, void* stream, double flops, double bytes);
  ~ProfScope();
  long idx_;
  void* stream_;
};

#define PA_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      pa::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return e__ == hipErrorOutOfMemory ? 2 : 1;                           \
    }                                                                      \
  } while (0)

#define PA_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      pa::set_error(__VA_ARGS__);    \
      return 3;                      \
    }                                \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blocks of NW waves; `red` is an LDS array of >= NW floats.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) s += red[i];
  return s;
}

__device__ __forceinline__ float leaky_relu(float x) { return x > 0.f ? x : 0.01f * x; }
// torch.nn.functional.gelu (approximate="none"): x * Phi(x) with the exact error function
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace pa
