// Issue-rate probes for carry-handling alternatives (all independent chains, 8 per thread).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define PROBE(NAME, DECL, BODY, FOLD)                                                     \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t a0, uint32_t b0, int iters) { \
    uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x; (void)a; (void)b;                 \
    DECL                                                                                  \
    for (int i = 0; i < iters; i++) {                                                     \
      _Pragma("unroll") for (int k = 0; k < 8; k++) { BODY }                              \
    }                                                                                     \
    uint32_t s = 0;                                                                       \
    _Pragma("unroll") for (int k = 0; k < 8; k++) { FOLD }                                \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                              \
  }

PROBE(k_add_u32, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));, s += x[k];)
PROBE(k_add3_u32, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));, s += x[k];)
PROBE(k_add_co, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x[k]) : "v"(a) : "vcc");, s += x[k];)
PROBE(k_addc_co, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : "vcc");, s += x[k];)
PROBE(k_addc_zero, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(x[k]) : : "vcc");, s += x[k];)
PROBE(k_lshl_add_u64, uint64_t x[8]; for (int k = 0; k < 8; k++) x[k] = k; uint64_t y = ((uint64_t)a << 32) | b;,
      asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[k]) : "v"(y));, s += (uint32_t)x[k] ^ (uint32_t)(x[k] >> 32);)
PROBE(k_alignbit, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(x[k]) : "v"(a));, s += x[k];)
PROBE(k_and, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[k]) : "v"(a));, s += x[k];)
PROBE(k_mul_hi, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k + 3;,
      asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));, s += x[k];)
PROBE(k_mad_u32_u24, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b));, s += x[k];)
PROBE(k_mad64_sgprcarry, uint64_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(x[k]) : "v"(a), "v"(b) : "s20", "s21");, s += (uint32_t)x[k] ^ (uint32_t)(x[k] >> 32);)
// mad + addc with SGPR carry (VOP3 addc)
PROBE(k_pair_sgpr, uint64_t x[8]; uint32_t h[8]; for (int k = 0; k < 8; k++) { x[k] = k; h[k] = 0; },
      asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_addc_co_u32 %1, s[20:21], 0, %1, s[20:21]" : "+v"(x[k]), "+v"(h[k]) : "v"(a), "v"(b) : "s20", "s21");,
      s += (uint32_t)x[k] ^ (uint32_t)(x[k] >> 32) ^ h[k];)
// mad (no carry needed) + plain 32-bit add of the high word into a separate sum
PROBE(k_mad_plus_add, uint64_t x[8]; uint32_t h[8]; for (int k = 0; k < 8; k++) { x[k] = k; h[k] = 0; },
      asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %1, %2" : "+v"(x[k]), "+v"(h[k]) : "v"(a), "v"(b) : "vcc");,
      s += (uint32_t)x[k] ^ (uint32_t)(x[k] >> 32) ^ h[k];)
// two mads back to back then two addc (does grouping help?)
PROBE(k_mad_mad, uint64_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0" : "+v"(x[k]) : "v"(a), "v"(b) : "vcc");,
      s += (uint32_t)x[k] ^ (uint32_t)(x[k] >> 32);)
PROBE(k_mad_i32_i24, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[k]) : "v"(a));, s += x[k];)
PROBE(k_mul_hi_u24, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k + 7;,
      asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x[k]) : "v"(a));, s += x[k];)
PROBE(k_fma_f32, float x[8]; for (int k = 0; k < 8; k++) x[k] = k; float fa = (float)a;,
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(fa));, s += (uint32_t)x[k];)
PROBE(k_cndmask, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a) : );, s += x[k];)
PROBE(k_mov, uint32_t x[8]; for (int k = 0; k < 8; k++) x[k] = k;,
      asm volatile("v_mov_b32 %0, %1" : "+v"(x[k]) : "v"(a));, s += x[k];)

template <class K>
void run(const char* name, K kernel, int cus, uint32_t* d, int per_body) {
  for (int wps : {2, 8}) {
    int grid = cus * wps, iters = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d, 3u, 5u, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d, 3u, 5u, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double bodies = (double)grid * 4 * iters * 8;
    printf("%-22s wps=%d: %.2f cycles per body (%d instr) per SIMD\n", name, wps, ms * 1e-3 * 2.4e9 / (bodies / (cus * 4)), per_body);
  }
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  uint32_t* d; CK(hipMalloc(&d, 64 << 20));
  run("v_mov_b32", k_mov, cus, d, 1);
  run("v_add_u32", k_add_u32, cus, d, 1);
  run("v_add3_u32", k_add3_u32, cus, d, 1);
  run("v_and_b32", k_and, cus, d, 1);
  run("v_alignbit_b32", k_alignbit, cus, d, 1);
  run("v_cndmask_b32", k_cndmask, cus, d, 1);
  run("v_fma_f32", k_fma_f32, cus, d, 1);
  run("v_add_co_u32", k_add_co, cus, d, 1);
  run("v_addc_co_u32", k_addc_co, cus, d, 1);
  run("v_addc_co_u32 (0)", k_addc_zero, cus, d, 1);
  run("v_lshl_add_u64", k_lshl_add_u64, cus, d, 1);
  run("v_mul_hi_u32", k_mul_hi, cus, d, 1);
  run("v_mul_u32_u24", k_mad_i32_i24, cus, d, 1);
  run("v_mul_hi_u32_u24", k_mul_hi_u24, cus, d, 1);
  run("v_mad_u32_u24", k_mad_u32_u24, cus, d, 1);
  run("mad64 sgpr carry", k_mad64_sgprcarry, cus, d, 1);
  run("mad64+addc sgpr", k_pair_sgpr, cus, d, 2);
  run("mad64+add_u32", k_mad_plus_add, cus, d, 2);
  run("mad64,mad64", k_mad_mad, cus, d, 2);
  return 0;
}
