// Development probe: ONE 1024-thread workgroup that holds a CU for `cycles` shader cycles, as the dendrogram
// merge does, with a selectable memory behaviour -- to find out what the merge does to a concurrently running
// persistent convolution kernel (tools/probes/interference_probe.py).
//   mode 0: spin on s_memtime only (residency alone)
//   mode 1: + strided 8-byte reads (one 128-B line each) over `n` doubles          (scattered reads)
//   mode 2: + strided 8-byte read-modify-writes                                    (scattered updates)
//   mode 3: + contiguous 8-byte reads                                              (streaming reads)
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(1024) void k_hog(long long cycles, int mode, double* buf, long long n,
                                                         long long stride, double* sink) {
  extern __shared__ double lds[];
  const long long t0 = __builtin_amdgcn_s_memtime();
  double acc = 0.0;
  long long pos = threadIdx.x;
  lds[threadIdx.x] = 0.0;
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
    if (mode == 1) {
      acc += buf[(pos * stride) % n];
    } else if (mode == 2) {
      const long long i = (pos * stride) % n;
      buf[i] = buf[i] * 0.999 + 1.0;
    } else if (mode == 3) {
      acc += buf[pos % n];
    }
    pos += 1024;
    __syncthreads();
  }
  if (acc == 123.456) sink[threadIdx.x] = acc;
}
extern "C" int hog_probe(long long cycles, int mode, void* buf, long long n, long long stride, void* sink, int lds_kb,
                         void* stream) {
  (void)hipFuncSetAttribute((const void*)k_hog, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
  hipLaunchKernelGGL(k_hog, dim3(1), dim3(1024), lds_kb * 1024, (hipStream_t)stream, cycles, mode, (double*)buf, n,
                     stride, (double*)sink);
  return (int)hipGetLastError();
}
