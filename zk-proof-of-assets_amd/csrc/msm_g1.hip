// G1 instantiation of the Pippenger MSM (zkey sections 5, 6, 8, 9: A, B1, C, H queries).
#include "msm_run.hip.h"

namespace zkpoa {
void msm_run_g1(zkpoa_context* ctx, int lane_id, const void* d_bases, const void* d_scalars, uint64_t n, uint8_t* out,
                float* ms2) {
  msm_run<Fq, HFq>(ctx, lane_id, d_bases, d_scalars, n, out, ms2);
}
MsmSorted* msm_sort_run(zkpoa_context* ctx, int lane_id, const void* d_scalars, uint64_t n, bool for_g2) {
  // the sorting lane also accumulates a G1 array afterwards: reserve room for that in the same arena
  MsmSorted* sr = new MsmSorted();
  try {
    if (lane_id) ctx->dev.wait_lanes();
    *sr = msm_sort_phase(ctx->dev.lanes[lane_id], d_scalars, (size_t)n, ctx->opt_msm_c, &msm_accum_workspace_bytes<Fq>,
                         true, for_g2);
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
