// Host driver shared by msm_g1.hip / msm_g2.hip: device MSM + host window combination.
#pragma once
#include "msm.hip.h"

#include <chrono>
#include <thread>
#include "zkpoa_internal.hpp"

namespace zkpoa {
// ms2 (optional): [0] = device time of the whole MSM, [1] = its level-0 bucket-accumulation kernel.
// Entry positions are 32-bit (n x windows < 2^32), so an MSM over more than `opt_msm_max_points` points
// (default 2^27; the largest domain a zkey may have is 2^28) runs in chunks whose results are added on the host.
// `table` (optional): fixed-base table built from exactly these n bases -> merged-bucket form, d_bases unread.
template <class F, class HF>
void msm_run(zkpoa_context* ctx, int lane_id, const void* d_bases, const void* d_scalars, uint64_t n, uint8_t* out,
             float* ms2, const MsmTable* table = nullptr) {
  if (lane_id) ctx->dev.wait_lanes();
  ctx->dev.ensure_lane(lane_id);
  Lane& lane = ctx->dev.lanes[lane_id];
  std::vector<char>& wsums = lane.host_sums;   // room for the largest read-back (msm_accum_phase checks it)
  XYZZ<HF> total = XYZZ<HF>::inf();
  float tot_ms = 0, acc_sum = 0;
  uint64_t done = 0, adds = 0;
  uint32_t hold_waits = 0;
  do {
    // (read every time round: a failed reservation below lowers it)
    const uint64_t max_pts = ctx->msm_points_limit(lane_id);
    if (table && (n > max_pts || table->n != n)) table = nullptr;   // a chunked MSM cannot index a whole-array table
    const uint64_t cnt = n - done < max_pts ? n - done : max_pts;
    float acc_ms = 0;
    uint32_t entries = 0;
    ZK_HIP(hipEventRecord(ctx->ev_a[lane_id], lane.stream));
    MsmPlan p;
    try {
      p = msm_device<F>(lane, reinterpret_cast<const char*>(d_bases) + done * MsmSizes<F>::kAffine,
                        reinterpret_cast<const char*>(d_scalars) + done * 32, (size_t)cnt, wsums.data(),
                        ctx->opt_msm_c, &acc_ms, table, &entries);
    } catch (const OomError& e) {
      // HBM is full (a 2^27 key beside five whole-MSM workspaces; another process on the card): the workspace is
      // proportional to the points sorted at once, so go through them in halves -- the sum is the same
      if (!ctx->shrink_after_oom(lane_id, cnt)) {
        // nothing smaller to try. If the refusal came from the hold-back of a key that is still being loaded (an upper
        // bound), its last allocations are moments away: wait for them (bounded) before giving up
        if (ctx->key_hold_back.load() > 0 && hold_waits < 5000) {
          hold_waits++;
          std::this_thread::sleep_for(std::chrono::milliseconds(1));
          continue;
        }
        throw;
      }
      if (getenv("ZKPOA_VERBOSE"))
        fprintf(stderr, "zkpoa:   lane %d: %s for %llu points at once; continuing with at most %llu\n", lane_id, e.what(),
                (unsigned long long)cnt, (unsigned long long)ctx->msm_points_limit(lane_id));
      continue;
    }
    adds += entries;
    ZK_HIP(hipEventRecord(ctx->ev_b[lane_id], lane.stream));
    ZK_HIP(hipEventSynchronize(ctx->ev_b[lane_id]));
    float tot = 0;
    ZK_HIP(hipEventElapsedTime(&tot, ctx->ev_a[lane_id], ctx->ev_b[lane_id]));
    tot_ms += tot;
    acc_sum += acc_ms;
    XYZZ<HF> r = h_combine_windows<HF>(wsums.data(), p.Wb, p.c, p.logS);
    if (done == 0) total = r;
    else xyzz_add(total, r);
    done += cnt;
  } while (done < n);
  if (ms2) {
    ms2[0] = tot_ms;
    ms2[1] = acc_sum;
  }
  ctx->lane_adds[lane_id] = (double)adds;
  h_affine_to_bytes<HF>(h_to_affine(total), out);
}


// Phase B only, for a sort result that may live on another lane (prover: A, B1, B2 share the witness sort).
template <class F, class HF>
void msm_accum_run(zkpoa_context* ctx, int lane_id, const MsmSorted& sr, bool own_arena, const void* d_bases,
                   uint8_t* out, float* ms2) {
  if (lane_id) ctx->dev.wait_lanes();
  Lane& lane = ctx->dev.lanes[lane_id];
  std::vector<char>& wsums = lane.host_sums;
  float acc_ms = 0;
  ZK_HIP(hipEventRecord(ctx->ev_a[lane_id], lane.stream));
  msm_accum_phase<F>(lane, sr, d_bases, wsums.data(), own_arena, &acc_ms);
  ZK_HIP(hipEventRecord(ctx->ev_b[lane_id], lane.stream));
  ZK_HIP(hipEventSynchronize(ctx->ev_b[lane_id]));
  float tot = 0;
  ZK_HIP(hipEventElapsedTime(&tot, ctx->ev_a[lane_id], ctx->ev_b[lane_id]));
  if (ms2) {
    ms2[0] = tot;
    ms2[1] = acc_ms;
  }
  ctx->lane_adds[lane_id] = (double)sr.total_entries;
  XYZZ<HF> r = h_combine_windows<HF>(wsums.data(), sr.p.Wb, sr.p.c, sr.p.logS);
  h_affine_to_bytes<HF>(h_to_affine(r), out);
}

}  // namespace zkpoa
