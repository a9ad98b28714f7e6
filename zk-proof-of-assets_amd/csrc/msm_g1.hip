// G1 instantiation of the Pippenger MSM (zkey sections 5, 6, 8, 9: A, B1, C, H queries).
#include "msm_run.hip.h"

namespace zkpoa {
void msm_run_g1(zkpoa_context* ctx, int lane_id, const void* d_bases, const void* d_scalars, uint64_t n, uint8_t* out,
                float* ms2, const MsmTable* table) {
  msm_run<Fq, HFq>(ctx, lane_id, d_bases, d_scalars, n, out, ms2, table);
}
MsmTable* msm_table_build_g1(zkpoa_context* ctx, const void* d_bases, uint64_t n, int c) {
  MsmTable* t = new MsmTable();
  try {
    *t = msm_table_build<Fq>(ctx->dev.lanes[0].stream, d_bases, n, msm_table_c(n, c, false));
  } catch (...) {
    delete t;
    throw;
  }
  return t;
}
size_t msm_table_bytes_g1(uint64_t n, int c) { return msm_table_bytes<Fq>(n, msm_table_c(n, c, false)); }
uint32_t msm_table_width(uint64_t n, int c, bool g2) { return msm_table_c(n, c, g2); }
void msm_table_release(MsmTable* t) {
  if (!t) return;
  msm_table_free(*t);
  delete t;
}
const void* msm_table_data(const MsmTable* t) { return t ? t->d : nullptr; }
void msm_set_forced_k0(int k0) { msm_forced_k0() = k0; }
void msm_set_density_hint(const double* density) { msm_density_hint() = density; }
void msm_density(hipStream_t st, const void* d_scalars, uint64_t n, double out[32], void* d_scratch) {
  for (int c = 0; c < 32; c++) out[c] = (double)msm_windows(c < 4 ? 4u : (uint32_t)c);   // uniform default
  if (n == 0) return;
  unsigned long long h[kDensityHi - kDensityLo + 1];
  ZK_HIP(hipMemsetAsync(d_scratch, 0, 32 * 8, st));
  hipLaunchKernelGGL(msm_density_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, d_scalars, n,
                     (unsigned long long*)d_scratch);
  ZK_HIP(hipMemcpyAsync(h, d_scratch, sizeof(h), hipMemcpyDeviceToHost, st));
  ZK_HIP(hipStreamSynchronize(st));
  for (int c = kDensityLo; c <= kDensityHi; c++) out[c] = (double)h[c - kDensityLo] / (double)n;
}
size_t msm_workspace_g1(uint64_t n, int force_c, bool for_g2, int table_c, bool accum) {
  if (n == 0) return 0;
  const MsmPlan p = msm_make_plan((size_t)n, table_c > 0 ? table_c : force_c, for_g2, table_c > 0);
  return msm_sort_workspace_bytes(p) + (accum ? msm_accum_workspace_bytes<Fq>(p) : 0);
}
void msm_table_info(const MsmTable* t, uint64_t out[4]) {
  out[0] = t->n;
  out[1] = t->c;
  out[2] = t->W;
  out[3] = t->bytes;
}
MsmSorted* msm_sort_run(zkpoa_context* ctx, int lane_id, const void* d_scalars, uint64_t n, bool for_g2, int table_c) {
  // the sorting lane also accumulates a G1 array afterwards: reserve room for that in the same arena
  MsmSorted* sr = new MsmSorted();
  try {
    if (lane_id) ctx->dev.wait_lanes();
    *sr = msm_sort_phase(ctx->dev.lanes[lane_id], d_scalars, (size_t)n, table_c > 0 ? table_c : ctx->opt_msm_c,
                         &msm_accum_workspace_bytes<Fq>, true, for_g2, table_c > 0);
  } catch (...) {
    delete sr;
    throw;
  }
  return sr;
}
void msm_sorted_free(MsmSorted* sr) { delete sr; }
void msm_accum_g1(zkpoa_context* ctx, int lane_id, const MsmSorted* sr, bool own_arena, const void* d_bases,
                  uint8_t* out, float* ms2) {
  msm_accum_run<Fq, HFq>(ctx, lane_id, *sr, own_arena, d_bases, out, ms2);
}
}  // namespace zkpoa
