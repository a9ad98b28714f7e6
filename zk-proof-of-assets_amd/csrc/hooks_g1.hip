// G1 / Fq / Fr instantiations of the element-wise hooks and the synthetic base generator.
#include "hooks.hip.h"

namespace zkpoa {
template <>
Affine<HFq> host_generator<HFq>() {
  return {HFq::from_u64(1), HFq::from_u64(2)};
}
void group_add_run_g1(zkpoa_context* ctx, const void* a, const void* b, void* out, uint64_t n) {
  group_add_run<Fq>(ctx, a, b, out, n);
}
void gen_bases_g1(zkpoa_context* ctx, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t i0, uint64_t n, void* d_out) {
  gen_bases<Fq, HFq>(ctx, a_le, b_le, i0, n, d_out);
}
}  // namespace zkpoa
using namespace zkpoa;

extern "C" int zkpoa_field_op(zkpoa_context* ctx, int field, int op, const void* a, const void* b, void* out,
                              uint64_t n) {
  ZK_API_BEGIN(ctx)
  if (field < 0 || field > 1 || op < 0 || op > 5) throw HipError("field_op: bad field/op");
  if (n == 0) return PROVER_OK;
  DevBuf da(n * 32), db(n * 32), dout(n * 32);
  ZK_HIP(hipMemcpy(da.p, a, n * 32, hipMemcpyHostToDevice));
  if (b) ZK_HIP(hipMemcpy(db.p, b, n * 32, hipMemcpyHostToDevice));
  hipStream_t st = ctx->dev.lanes[0].stream;
  dim3 grid((uint32_t)((n + 255) / 256));
  const void* bp = b ? db.p : nullptr;
  if (field == 0) hipLaunchKernelGGL((zkpoa::field_op_kernel<Fq>), grid, dim3(256), 0, st, op, (const void*)da.p, bp, dout.p, n);
  else hipLaunchKernelGGL((zkpoa::field_op_kernel<Fr>), grid, dim3(256), 0, st, op, (const void*)da.p, bp, dout.p, n);
  ZK_HIP(hipStreamSynchronize(st));
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpy(out, dout.p, n * 32, hipMemcpyDeviceToHost));
  ZK_API_END(ctx)
}
