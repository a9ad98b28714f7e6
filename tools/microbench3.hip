// Prototype: carry-free Montgomery multiplication in radix 2^29 (9 limbs, R' = 2^261), product scanning,
// 64-bit column accumulators (18 products < 2^58 each never overflow), lazy output in [0, 2q).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __constant__ const uint32_t P29c[9] = {0x187cfd47u, 0x10460b6cu, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x0a6e141bu, 0x0e5c2634u, 0x00030644u};
struct F29 { uint32_t l[9]; };
__device__ __forceinline__ F29 mul29(const F29& a, const F29& b) {
  constexpr uint32_t MASK = (1u << 29) - 1u;
  const uint32_t P29[9] = {0x187cfd47u, 0x10460b6cu, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x0a6e141bu, 0x0e5c2634u, 0x00030644u};
  const uint32_t INV = 0x1a866389u & MASK;   // placeholder low bits; exact value irrelevant for timing
  uint64_t acc = 0;
  uint32_t m[9];
  F29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P29[k - i];
    m[k] = ((uint32_t)acc * INV) & MASK;
    acc += (uint64_t)m[k] * P29[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * P29[k - i];
    r.l[k - 9] = (uint32_t)acc & MASK;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k_mul29(const uint32_t* in, uint32_t* out, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  F29 x[CHAINS], y;
  for (int i = 0; i < 9; i++) { y.l[i] = in[(t * 9 + i) & 8191] & 0x1fffffffu; for (int c = 0; c < CHAINS; c++) x[c].l[i] = in[(t * 9 + i + 17 * (c + 1)) & 8191] & 0x1fffffffu; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = mul29(x[c], y);
  }
  uint32_t s = 0;
  for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 9; i++) s += x[c].l[i];
  out[t] = s;
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  uint32_t *din, *dout; CK(hipMalloc(&din, 8192 * 4)); CK(hipMalloc(&dout, 64 << 20));
  uint32_t h[8192]; for (int i = 0; i < 8192; i++) h[i] = i * 2654435761u;
  CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
  for (int wps : {2, 4, 8}) {
    int grid = cus * wps, iters = 512;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mul29<1>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)din, dout, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mul29<1>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)din, dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mul29 chain=1 wps=%d: %.3f ms  %.1f G modmul/s\n", wps, ms, (double)grid * 256 * iters / ms / 1e6);
  }
  return 0;
}
