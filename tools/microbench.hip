// Register-only microbenchmarks: v_mad_u64_u32 issue rate, Montgomery multiply rate, XYZZ mixed-add rate.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk-proof-of-assets_amd/csrc tools/microbench.hip -o tools/microbench
#include "bn254_ec.hip.h"
#include <stdio.h>
#include <vector>
using namespace zkpoa;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// 8 independent mad chains per thread
__global__ __launch_bounds__(256) void k_mad(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint64_t acc[8];
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
  }
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) s += acc[k];
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
// same with the addc (the mac96 pair)
__global__ __launch_bounds__(256) void k_mac96(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint64_t lo[4]; uint32_t hi[4];
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++) { lo[k] = k; hi[k] = 0; }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) { mac96(lo[k], hi[k], a, b); mac96(lo[k], hi[k], b, a); }
  }
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) s += lo[k] + hi[k];
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
__global__ __launch_bounds__(256) void k_mul32(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint32_t acc[8];
  uint32_t a = a0 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = b0 + k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[k]) : "v"(a));
  }
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) s += acc[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma64(double* out, double a0, int iters) {
  double acc[8];
  double a = a0 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(acc[k]) : "v"(a));
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) s += acc[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void k_modmul(const uint4* in, uint4* out, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Fq x[CHAINS];
  Fq y = load_field<Fq>(in + 2 * (t & 1023));
#pragma unroll
  for (int c = 0; c < CHAINS; c++) x[c] = load_field<Fq>(in + 2 * ((t + c + 1) & 1023));
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = x[c] * y;
  }
  Fq s = x[0];
#pragma unroll
  for (int c = 1; c < CHAINS; c++) s = s + x[c];
  store_field(out + 2 * t, s);
}

// r03: dedicated squaring (36 + 72 products) against a * a (64 + 72), and the Fq2 product (lazy dot2 form)
template <int SQR>
__global__ __launch_bounds__(256) void k_modsqr(const uint4* in, uint4* out, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Fq x = load_field<Fq>(in + 2 * (t & 1023)), y = load_field<Fq>(in + 2 * ((t + 1) & 1023));
  for (int i = 0; i < iters; i++) {
    x = SQR ? x.sqr() : x * x;
    y = SQR ? y.sqr() : y * y;
  }
  store_field(out + 2 * t, x + y);
}
__global__ __launch_bounds__(256) void k_fq2mul(const uint4* in, uint4* out, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Fq2 x = {load_field<Fq>(in + 2 * (t & 1023)), load_field<Fq>(in + 2 * ((t + 1) & 1023))};
  Fq2 y = {load_field<Fq>(in + 2 * ((t + 2) & 1023)), load_field<Fq>(in + 2 * ((t + 3) & 1023))};
  for (int i = 0; i < iters; i++) x = x * y;
  store_field(out + 2 * t, x.c0 + x.c1);
}
__global__ __launch_bounds__(256) void k_madd2(const uint4* in, uint4* out, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  auto ld = [&](int o) { return Fq2{load_field<Fq>(in + 2 * ((t + o) & 1023)), load_field<Fq>(in + 2 * ((t + o + 1) & 1023))}; };
  Affine<Fq2> p = {ld(0), ld(2)};
  XYZZ<Fq2> acc = {ld(4), ld(6), ld(8), ld(10)};
  for (int i = 0; i < iters; i++) xyzz_add_affine(acc, p, false);
  Fq2 s = acc.x + acc.y + acc.zz + acc.zzz;
  store_field(out + 2 * t, s.c0 + s.c1);
}

__global__ __launch_bounds__(256) void k_madd(const uint4* in, uint4* out, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Affine<Fq> p = {load_field<Fq>(in + 2 * (t & 1023)), load_field<Fq>(in + 2 * ((t + 7) & 1023))};
  XYZZ<Fq> acc = {load_field<Fq>(in + 2 * ((t + 1) & 1023)), load_field<Fq>(in + 2 * ((t + 2) & 1023)),
                  load_field<Fq>(in + 2 * ((t + 3) & 1023)), load_field<Fq>(in + 2 * ((t + 4) & 1023))};
  for (int i = 0; i < iters; i++) xyzz_add_affine(acc, p, false);
  store_field(out + 2 * t, acc.x + acc.y + acc.zz + acc.zzz);
}

template <class K, class... A>
float timeit(K kernel, int grid, A... args) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, args...);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, args...);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs=%d clock=%d MHz\n", prop.name, cus, prop.clockRate / 1000);
  uint32_t* d32; CK(hipMalloc(&d32, 64 << 20));
  std::vector<uint32_t> h(8 * 1024);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffffu : 0xffffffffu);
  uint4* din; CK(hipMalloc(&din, 32 * 1024)); CK(hipMemcpy(din, h.data(), 32 * 1024, hipMemcpyHostToDevice));
  uint4* dout = (uint4*)d32;
  for (int wps : {1, 2, 4, 8}) {   // waves per SIMD
    int grid = cus * wps;          // 256 threads = 4 waves = 1 per SIMD per block
    int iters = 4096;
    float ms = timeit(k_mad, grid, d32, 3u, 5u, iters);
    double inst = (double)grid * 4 * iters * 8;  // wave-instructions
    printf("v_mad_u64_u32   wps=%d: %.3f ms  %.2f cycles/wave-inst/SIMD @2.4GHz  (%.1f G lane-mads/s)\n", wps, ms,
           ms * 1e-3 * 2.4e9 / (inst / (cus * 4)), inst * 64 / ms / 1e6);
    ms = timeit(k_mac96, grid, d32, 3u, 5u, iters);
    inst = (double)grid * 4 * iters * 8;
    printf("mac96 (mad+addc) wps=%d: %.3f ms  %.2f cycles/pair/SIMD\n", wps, ms, ms * 1e-3 * 2.4e9 / (inst / (cus * 4)));
    ms = timeit(k_mul32, grid, d32, 3u, 5u, iters);
    printf("v_mul_lo_u32    wps=%d: %.3f ms  %.2f cycles/wave-inst/SIMD\n", wps, ms, ms * 1e-3 * 2.4e9 / (inst / (cus * 4)));
    ms = timeit(k_fma64, grid, (double*)d32, 1.5, iters);
    printf("v_fma_f64       wps=%d: %.3f ms  %.2f cycles/wave-inst/SIMD\n", wps, ms, ms * 1e-3 * 2.4e9 / (inst / (cus * 4)));
  }
  for (int wps : {1, 2, 4, 8}) {
    int grid = cus * wps, iters = 512;
    float ms = timeit(k_modmul<1>, grid, (const uint4*)din, dout, iters);
    printf("modmul chain=1 wps=%d: %.3f ms  %.1f G modmul/s\n", wps, ms, (double)grid * 256 * iters / ms / 1e6);
    ms = timeit(k_modmul<2>, grid, (const uint4*)din, dout, iters);
    printf("modmul chain=2 wps=%d: %.3f ms  %.1f G modmul/s\n", wps, ms, (double)grid * 256 * iters * 2 / ms / 1e6);
  }
  for (int wps : {1, 2, 3, 4}) {
    int grid = cus * wps, iters = 128;
    float ms = timeit(k_madd, grid, (const uint4*)din, dout, iters);
    printf("xyzz_add_affine wps=%d: %.3f ms  %.2f G adds/s\n", wps, ms, (double)grid * 256 * iters / ms / 1e6);
  }
  for (int wps : {2, 4, 8}) {
    int grid = cus * wps, iters = 512;
    float m0 = timeit(k_modsqr<0>, grid, (const uint4*)din, dout, iters);
    float m1 = timeit(k_modsqr<1>, grid, (const uint4*)din, dout, iters);
    printf("a*a vs sqr() wps=%d: %.3f / %.3f ms  %.1f / %.1f G per s\n", wps, m0, m1, (double)grid * 256 * iters * 2 / m0 / 1e6,
           (double)grid * 256 * iters * 2 / m1 / 1e6);
    float m2 = timeit(k_fq2mul, grid, (const uint4*)din, dout, iters);
    printf("Fq2 mul (dot2) wps=%d: %.3f ms  %.1f G Fq2 products/s\n", wps, m2, (double)grid * 256 * iters / m2 / 1e6);
  }
  for (int wps : {1, 2}) {
    int grid = cus * wps, iters = 64;
    float ms = timeit(k_madd2, grid, (const uint4*)din, dout, iters);
    printf("G2 xyzz_add_affine wps=%d: %.3f ms  %.2f G adds/s\n", wps, ms, (double)grid * 256 * iters / ms / 1e6);
  }
  return 0;
}
