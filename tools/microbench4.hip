// r04: does instruction-level parallelism inside ONE wave help the multiplier? Every v_mad_u64_u32 of the field code
// writes its carry to VCC, so even "independent" accumulator chains share a register -- this measures chains with
// carries in distinct SGPR pairs, the interleaved mad+addc pair (mac96x2) and an Fq2 product whose two coordinates
// advance in lockstep, against the forms in use, by waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk-proof-of-assets_amd/csrc tools/microbench4.hip -o tools/microbench4
#include "bn254_ec.hip.h"
#include <stdio.h>
#include <vector>
using namespace zkpoa;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// 4 chains, all carries to VCC (as tools/microbench.hip k_mad)
__global__ __launch_bounds__(256) void k_mad_vcc(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint64_t acc[4];
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++) acc[k] = k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int k = 0; k < 4; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
  }
  uint64_t s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
// 4 chains, each carry into its own SGPR pair
__global__ __launch_bounds__(256) void k_mad_sgpr(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint64_t acc[4];
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++) acc[k] = k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      uint64_t c0, c1, c2, c3;
      asm volatile("v_mad_u64_u32 %0, %4, %8, %9, %0\n\t"
                   "v_mad_u64_u32 %1, %5, %8, %9, %1\n\t"
                   "v_mad_u64_u32 %2, %6, %8, %9, %2\n\t"
                   "v_mad_u64_u32 %3, %7, %8, %9, %3"
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
                   : "v"(a), "v"(b));
    }
  }
  uint64_t s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// (mac96x2: csrc/bn254_field.hip.h)
__global__ __launch_bounds__(256) void k_mac96_seq(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {   // 2 chains, one after the other
  uint64_t lo[2] = {1, 2}; uint32_t hi[2] = {0, 0};
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) { mac96(lo[0], hi[0], a, b); mac96(lo[1], hi[1], b, a); }
  }
  uint64_t s = lo[0] + lo[1] + hi[0] + hi[1];
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
__global__ __launch_bounds__(256) void k_mac96_x2(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint64_t lo[2] = {1, 2}; uint32_t hi[2] = {0, 0};
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) mac96x2(lo[0], hi[0], a, b, lo[1], hi[1], b, a);
  }
  uint64_t s = lo[0] + lo[1] + hi[0] + hi[1];
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// the Fq2 product with its two coordinates one after the other (r03) against the lockstep form now in the header
ZK_DEV Fq2 fq2_mul_seq(const Fq2& a, const Fq2& b) {
  return {Fq::dot2(a.c0, b.c0, a.c1, b.c1.neg_2p()), Fq::dot2(a.c0, b.c1, a.c1, b.c0)};
}
template <int LOCK>
__global__ __launch_bounds__(256) void k_fq2mul(const uint4* in, uint4* out, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Fq2 x = {load_field<Fq>(in + 2 * (t & 1023)), load_field<Fq>(in + 2 * ((t + 1) & 1023))};
  Fq2 y = {load_field<Fq>(in + 2 * ((t + 2) & 1023)), load_field<Fq>(in + 2 * ((t + 3) & 1023))};
  for (int i = 0; i < iters; i++) x = LOCK ? x * y : fq2_mul_seq(x, y);
  store_field(out + 2 * t, x.c0 + x.c1);
}

// the mixed additions as the accumulation kernels run them: G1 at 3 waves per SIMD, G2 at 2
__global__ __launch_bounds__(256, 3) void k_madd_g1(const uint4* in, uint4* out, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  Affine<Fq> p = {load_field<Fq>(in + 2 * (t & 1023)), load_field<Fq>(in + 2 * ((t + 7) & 1023))};
  XYZZ<Fq> acc = {load_field<Fq>(in + 2 * ((t + 1) & 1023)), load_field<Fq>(in + 2 * ((t + 2) & 1023)),
                  load_field<Fq>(in + 2 * ((t + 3) & 1023)), load_field<Fq>(in + 2 * ((t + 4) & 1023))};
  for (int i = 0; i < iters; i++) xyzz_add_affine(acc, p, false);
  store_field(out + 2 * t, acc.x + acc.y + acc.zz + acc.zzz);
}
__global__ __launch_bounds__(256, 2) void k_madd_g2(const uint4* in, uint4* out, int iters) {
  int t = blockIdx.x * 256 + threadIdx.x;
  auto ld = [&](int o) { return Fq2{load_field<Fq>(in + 2 * ((t + o) & 1023)), load_field<Fq>(in + 2 * ((t + o + 1) & 1023))}; };
  Affine<Fq2> p = {ld(0), ld(2)};
  XYZZ<Fq2> acc = {ld(4), ld(6), ld(8), ld(10)};
  for (int i = 0; i < iters; i++) xyzz_add_affine(acc, p, false);
  Fq2 s = acc.x + acc.y + acc.zz + acc.zzz;
  store_field(out + 2 * t, s.c0 + s.c1);
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
  uint32_t* d32; CK(hipMalloc(&d32, 64 << 20));
  std::vector<uint32_t> h(8 * 1024);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffffu : 0xffffffffu);
  uint4* din; CK(hipMalloc(&din, 32 * 1024)); CK(hipMemcpy(din, h.data(), 32 * 1024, hipMemcpyHostToDevice));
  uint4* dout = (uint4*)d32;
  for (int wps : {1, 2, 3, 4, 8}) {
    int grid = cus * wps, iters = 4096;
    double inst = (double)grid * 4 * iters * 8;
    float m0 = timeit(k_mad_vcc, grid, d32, 3u, 5u, iters), m1 = timeit(k_mad_sgpr, grid, d32, 3u, 5u, iters);
    printf("v_mad_u64_u32, 4 chains  wps=%d: carries in VCC %.2f cycles/inst/SIMD, in 4 SGPR pairs %.2f\n", wps,
           m0 * 1e-3 * 2.4e9 / (inst / (cus * 4)), m1 * 1e-3 * 2.4e9 / (inst / (cus * 4)));
    float p0 = timeit(k_mac96_seq, grid, d32, 3u, 5u, iters), p1 = timeit(k_mac96_x2, grid, d32, 3u, 5u, iters);
    printf("mad+addc pair, 2 chains  wps=%d: one after the other %.2f cycles/pair/SIMD, interleaved (mac96x2) %.2f\n", wps,
           p0 * 1e-3 * 2.4e9 / (inst / (cus * 4)), p1 * 1e-3 * 2.4e9 / (inst / (cus * 4)));
  }
  for (int wps : {1, 2, 3, 4, 8}) {
    int grid = cus * wps, iters = 512;
    float f0 = timeit(k_fq2mul<0>, grid, (const uint4*)din, dout, iters), f1 = timeit(k_fq2mul<1>, grid, (const uint4*)din, dout, iters);
    printf("Fq2 product wps=%d: two dot2 in sequence %.1f G/s, coordinates in lockstep %.1f G/s\n", wps,
           (double)grid * 256 * iters / f0 / 1e6, (double)grid * 256 * iters / f1 / 1e6);
  }
  for (int wps : {1, 2, 3}) {
    int grid = cus * wps, iters = 128;
    float ms = timeit(k_madd_g1, grid, (const uint4*)din, dout, iters);
    printf("G1 xyzz_add_affine (launch bounds 3 waves/SIMD) wps=%d: %.2f G adds/s\n", wps, (double)grid * 256 * iters / ms / 1e6);
  }
  for (int wps : {1, 2}) {
    int grid = cus * wps, iters = 64;
    float ms = timeit(k_madd_g2, grid, (const uint4*)din, dout, iters);
    printf("G2 xyzz_add_affine (launch bounds 2 waves/SIMD) wps=%d: %.2f G adds/s\n", wps, (double)grid * 256 * iters / ms / 1e6);
  }
  return 0;
}
