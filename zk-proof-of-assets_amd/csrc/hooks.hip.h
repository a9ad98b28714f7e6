// Element-wise device hooks (parity tests) and synthetic base generation, generic over the group.
#pragma once
#include "msm.hip.h"
#include "zkpoa_internal.hpp"

namespace zkpoa {

// ---------------------------------------------------------------------------------------------
// element-wise test kernels
// ---------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) void field_op_kernel(int op, const void* a, const void* b, void* out, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  F x = load_field<F>(reinterpret_cast<const char*>(a) + 32 * i);
  F y = F::zero();
  if (b) y = load_field<F>(reinterpret_cast<const char*>(b) + 32 * i);
  F r;
  switch (op) {
    case 0: r = x * y; break;
    case 1: r = x + y; break;
    case 2: r = x - y; break;
    case 3: r = x.inv(); break;
    case 4: r = x.to_mont(); break;
    default: r = x.from_mont(); break;
  }
  store_field(reinterpret_cast<char*>(out) + 32 * i, r);
}

template <class F>
__global__ __launch_bounds__(256) void group_add_kernel(const void* a, const void* b, void* out_xyzz, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  Affine<F> pa = load_affine<F>(a, i), pb = load_affine<F>(b, i);
  XYZZ<F> acc = XYZZ<F>::from_affine(pa);
  xyzz_add_affine(acc, pb, false);
  store_xyzz(out_xyzz, i, acc);
}

// XYZZ -> affine on the device (one Fermat inversion per point; setup/test use only)
template <class F>
__global__ __launch_bounds__(256) void xyzz_to_affine_kernel(const void* in_xyzz, void* out_affine, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  XYZZ<F> p = load_xyzz<F>(in_xyzz, i);
  constexpr int FB = FieldBytes<F>::N;
  char* o = reinterpret_cast<char*>(out_affine) + i * (2 * FB);
  if (p.is_inf()) {
    store_field(o, F::zero());
    store_field(o + FB, F::zero());
    return;
  }
  F i3 = p.zzz.inv();
  F i2 = (p.zz * i3).sqr();
  store_field(o, p.x * i2);
  store_field(o + FB, p.y * i3);
}

// P_i = (a + i*b) * G for i in [i0 + t*CH, +CH): one double-and-add per thread, then CH-1 mixed
// additions of B = b*G. Output XYZZ into scratch (converted by xyzz_to_affine_kernel).
template <class F>
__global__ __launch_bounds__(256) void gen_bases_kernel(Affine<F> G, Affine<F> Bstep, Fr a_m, Fr b_m, uint64_t i0,
                                                        uint64_t n, uint32_t CH, void* out_xyzz) {
  uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  uint64_t first = t * CH;
  if (first >= n) return;
  // s = a + (i0 + first) * b  (Montgomery), then to standard form for the bit scan
  uint64_t idx = i0 + first;
  Fr im = Fr::zero();
  im.l[0] = (uint32_t)idx;
  im.l[1] = (uint32_t)(idx >> 32);
  im = im.to_mont();
  Fr s = (a_m + b_m * im).from_mont().canon();  // canonical: the bits are scanned below
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int limb = 7; limb >= 0; limb--) {
    uint32_t w = s.l[limb];
    for (int bit = 31; bit >= 0; bit--) {
      acc = xyzz_dbl(acc);
      if ((w >> bit) & 1u) xyzz_add_affine(acc, G, false);
    }
  }
  uint64_t last = first + CH < n ? first + CH : n;
  for (uint64_t k = first; k < last; k++) {
    store_xyzz(out_xyzz, k, acc);
    xyzz_add_affine(acc, Bstep, false);
  }
}


// group generator (affine, Montgomery); specialised in hooks_g1.hip / hooks_g2.hip
template <class HF>
Affine<HF> host_generator();

template <class F, class HF>
void gen_bases(zkpoa_context* ctx, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t i0, uint64_t n,
               void* d_out) {
  if (n == 0) return;
  Lane& lane = ctx->dev.lanes[0];
  Affine<HF> G = host_generator<HF>();
  uint64_t bk[4];
  memcpy(bk, b_le, 32);
  Affine<HF> Bs = h_to_affine(h_mul(XYZZ<HF>::from_affine(G), bk));
  Affine<F> dG, dB;
  static_assert(sizeof(Affine<F>) == sizeof(Affine<HF>), "layout");
  memcpy(&dG, &G, sizeof(dG));
  memcpy(&dB, &Bs, sizeof(dB));
  Fr am, bm;
  HFr ha = HFr::from_bytes(a_le).to_mont(), hb = HFr::from_bytes(b_le).to_mont();
  memcpy(&am, &ha, 32);
  memcpy(&bm, &hb, 32);
  const uint32_t CH = 32;
  // scratch: XYZZ for every point, processed in slabs to bound memory
  const uint64_t slab = 1ull << 22;
  DevBuf scratch((n < slab ? n : slab) * MsmSizes<F>::kXyzz);
  for (uint64_t off = 0; off < n; off += slab) {
    uint64_t cnt = n - off < slab ? n - off : slab;
    uint64_t threads = (cnt + CH - 1) / CH;
    hipLaunchKernelGGL((gen_bases_kernel<F>), dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, lane.stream, dG,
                       dB, am, bm, i0 + off, cnt, CH, scratch.p);
    hipLaunchKernelGGL((xyzz_to_affine_kernel<F>), dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, lane.stream,
                       (const void*)scratch.p, (void*)(reinterpret_cast<char*>(d_out) + off * MsmSizes<F>::kAffine),
                       cnt);
  }
  ZK_HIP(hipStreamSynchronize(lane.stream));
  ZK_HIP(hipGetLastError());
}


template <class F>
inline void group_add_run(zkpoa_context* ctx, const void* a, const void* b, void* out, uint64_t n) {
  constexpr size_t A = MsmSizes<F>::kAffine;
  DevBuf da(n * A), db(n * A), dx(n * MsmSizes<F>::kXyzz), dout(n * A);
  ZK_HIP(hipMemcpy(da.p, a, n * A, hipMemcpyHostToDevice));
  ZK_HIP(hipMemcpy(db.p, b, n * A, hipMemcpyHostToDevice));
  hipStream_t st = ctx->dev.lanes[0].stream;
  dim3 grid((uint32_t)((n + 255) / 256));
  hipLaunchKernelGGL((group_add_kernel<F>), grid, dim3(256), 0, st, (const void*)da.p, (const void*)db.p, dx.p, n);
  hipLaunchKernelGGL((xyzz_to_affine_kernel<F>), grid, dim3(256), 0, st, (const void*)dx.p, dout.p, n);
  ZK_HIP(hipStreamSynchronize(st));
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpy(out, dout.p, n * A, hipMemcpyDeviceToHost));
}


}  // namespace zkpoa
