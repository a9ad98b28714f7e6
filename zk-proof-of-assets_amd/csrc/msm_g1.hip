// G1 instantiation of the Pippenger MSM (zkey sections 5, 6, 8, 9: A, B1, C, H queries).
#include "msm_run.hip.h"

namespace zkpoa {
void msm_run_g1(zkpoa_context* ctx, int lane_id, const void* d_bases, const void* d_scalars, uint64_t n, uint8_t* out,
                float* ms2) {
  msm_run<Fq, HFq>(ctx, lane_id, d_bases, d_scalars, n, out, ms2);
}
}  // namespace zkpoa
