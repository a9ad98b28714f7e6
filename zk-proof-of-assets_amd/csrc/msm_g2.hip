// G2 instantiation of the Pippenger MSM (zkey section 7: B2 query).
#include "msm_run.hip.h"

namespace zkpoa {
void msm_run_g2(zkpoa_context* ctx, int lane_id, const void* d_bases, const void* d_scalars, uint64_t n, uint8_t* out,
                float* ms2, const MsmTable* table) {
  msm_run<Fq2, HFq2>(ctx, lane_id, d_bases, d_scalars, n, out, ms2, table);
}
MsmTable* msm_table_build_g2(zkpoa_context* ctx, const void* d_bases, uint64_t n, int c) {
  MsmTable* t = new MsmTable();
  try {
    *t = msm_table_build<Fq2>(ctx->dev.lanes[0].stream, d_bases, n, msm_table_c(n, c, true));
  } catch (...) {
    delete t;
    throw;
  }
  return t;
}
size_t msm_workspace_g2(uint64_t n, int force_c, int table_c, bool sort) {
  if (n == 0) return 0;
  const MsmPlan p = msm_make_plan((size_t)n, table_c > 0 ? table_c : force_c, true, table_c > 0);
  return (sort ? msm_sort_workspace_bytes(p) : 0) + msm_accum_workspace_bytes<Fq2>(p);
}
size_t msm_table_bytes_g2(uint64_t n, int c) { return msm_table_bytes<Fq2>(n, msm_table_c(n, c, true)); }
void msm_accum_g2(zkpoa_context* ctx, int lane_id, const MsmSorted* sr, bool own_arena, const void* d_bases,
                  uint8_t* out, float* ms2) {
  msm_accum_run<Fq2, HFq2>(ctx, lane_id, *sr, own_arena, d_bases, out, ms2);
}
}  // namespace zkpoa
