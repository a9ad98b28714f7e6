// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on the MSM's access pattern: every lane gathers one
// random 64-byte record with 4 x 16-byte loads from a buffer much larger than the caches, and every
// lane writes one 128-byte record with 8 x 16-byte stores. Known bytes: n*64 read, n*128 written.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void gather64_calib(const uint4* __restrict__ src, uint32_t nrec, uint4* __restrict__ dst, uint32_t n) {
  uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint32_t r = (uint32_t)(((uint64_t)i * 2654435761u + 12345u) % nrec);
  const uint4* p = src + 4 * (size_t)r;
  uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  uint4* o = dst + 8 * (size_t)i;
  o[0] = a; o[1] = b; o[2] = c; o[3] = d; o[4] = d; o[5] = c; o[6] = b; o[7] = a;
}
int main() {
  const uint32_t nrec = 1u << 26;   // 4 GiB of 64-byte records
  const uint32_t n = 1u << 24;      // 1 GiB read (16M x 64 B), 2 GiB written
  uint4 *src, *dst;
  CK(hipMalloc(&src, (size_t)nrec * 64)); CK(hipMalloc(&dst, (size_t)n * 128));
  CK(hipMemset(src, 1, (size_t)nrec * 64));
  for (int it = 0; it < 3; it++) hipLaunchKernelGGL(gather64_calib, dim3(n / 256), dim3(256), 0, 0, (const uint4*)src, nrec, dst, n);
  CK(hipDeviceSynchronize());
  printf("calib: expected read %llu bytes, written %llu bytes per launch\n", (unsigned long long)n * 64, (unsigned long long)n * 128);
  return 0;
}
