// Internal C++ view of the opaque handles declared in include/zkpoa_prover.h.
#pragma once
#include "../../include/zkpoa_prover.h"
#include "device_ctx.hpp"
#include "fast_upload.hpp"
#include "host_curve.hpp"

#include <algorithm>
#include <atomic>
#include <string>
#include <vector>

namespace zkpoa {
struct NttEngine;
struct MsmTable;
}

struct zkpoa_poseidon_state;   // poseidon.hip: device copy of the Poseidon parameters

struct zkpoa_context {
  zkpoa::DeviceCtx dev;
  zkpoa_poseidon_state* poseidon = nullptr;
  zkpoa::NttEngine* ntt = nullptr;
  zkpoa::FastUploader uploader;   // pinned, multi-threaded host -> HBM path for zkey sections / witnesses
  std::string last_error;
  float ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float io_ms[2] = {0, 0};   // last prove from a host buffer: [0] witness -> HBM (host clock), [1] its bytes / 1e6
  float lane_ms[zkpoa::DeviceCtx::kLanes][2] = {};   // per-lane {whole MSM, accumulation kernel} of the last MSM
  double lane_adds[zkpoa::DeviceCtx::kLanes] = {};   // per-lane mixed additions of that kernel (non-zero digits)
  int opt_msm_c = 0;
  long opt_msm_max_points = 0;   // 0 = default (2^27): larger MSMs run in chunks
  // Set when a lane's workspace for a whole MSM did not fit in HBM (found out by a failed reservation: msm_run, or
  // ahead of a proof: prover.hip budget_lane_workspaces): from then on the MSMs of that lane take at most this many
  // points at a time (never below 2^16); 0 = no limit of this kind.
  std::atomic<uint64_t> oom_max_points[zkpoa::DeviceCtx::kLanes] = {};
  uint64_t msm_points_limit(int lane) const {
    uint64_t lim = opt_msm_max_points ? (uint64_t)opt_msm_max_points : (1ull << 27);
    const uint64_t o = oom_max_points[lane].load();
    return o && o < lim ? o : lim;
  }
  uint64_t msm_points_limit_min() const {
    uint64_t lim = msm_points_limit(0);
    for (int l = 1; l < zkpoa::DeviceCtx::kLanes; l++) lim = std::min(lim, msm_points_limit(l));
    return lim;
  }
  static uint64_t below(uint64_t points) {   // the largest power of two below `points`, at least 2^16
    uint64_t want = 1ull << 16;
    while (want * 2 < points) want *= 2;
    return want;
  }
  // called with the size that failed; false when there is nothing smaller to try
  bool shrink_after_oom(int lane, uint64_t failed_points) {
    if (failed_points <= (1ull << 16)) return false;
    const uint64_t want = below(failed_points);
    uint64_t cur = oom_max_points[lane].load();
    while ((cur == 0 || cur > want) && !oom_max_points[lane].compare_exchange_weak(cur, want)) {
    }
    return true;
  }
  // HBM the key of a staged one-shot prove has still to allocate (prover.hip load_prove_staged); the lanes' arenas read it
  std::atomic<int64_t> key_hold_back{0};
  int opt_prove_serial = 0;      // measurement: run the stages of a prove one at a time (solo device times)
  // split chain: the witness copy of zkpoa_split_stage1 is enqueued on lane 0; the other lanes' MSMs wait for it
  hipEvent_t ev_witness = nullptr;
  bool ev_witness_set = false;
  hipEvent_t ev_a[zkpoa::DeviceCtx::kLanes] = {};
  hipEvent_t ev_b[zkpoa::DeviceCtx::kLanes] = {};
};

struct zkpoa_msm_table {   // C-ABI handle of a fixed-base table (zkpoa_msm_table_build)
  zkpoa::MsmTable* t = nullptr;
  int group = 1;
};

namespace zkpoa {
void set_err(char* buf, unsigned long cap, const std::string& msg);

// hipFree waits for the whole device, copies in flight included. While a one-shot prove overlaps the key upload with
// compute, temporaries are therefore parked in the calling thread's sink and freed after the proof.
inline std::vector<void*>*& deferred_free_sink() {
  static thread_local std::vector<void*>* sink = nullptr;
  return sink;
}

struct DevBuf {  // RAII device allocation for the host-buffer entry points
  void* p = nullptr;
  explicit DevBuf(size_t bytes) { ZK_HIP(hipMalloc(&p, bytes ? bytes : 1)); }
  ~DevBuf() {
    if (!p) return;
    if (deferred_free_sink()) deferred_free_sink()->push_back(p);
    else (void)hipFree(p);
  }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};


// per-group entry points, each compiled in its own translation unit (msm_g1.hip, msm_g2.hip, ...)
struct MsmTable;   // msm.hip.h: fixed-base table 2^(c*j) * P_i of one base array
void msm_run_g1(zkpoa_context* ctx, int lane_id, const void* d_bases, const void* d_scalars, uint64_t n, uint8_t* out,
                float* ms2, const MsmTable* table = nullptr);
void msm_run_g2(zkpoa_context* ctx, int lane_id, const void* d_bases, const void* d_scalars, uint64_t n, uint8_t* out,
                float* ms2, const MsmTable* table = nullptr);
// tables: built on lane 0's stream (synchronised on return); c = 0 picks the width from the cost model
MsmTable* msm_table_build_g1(zkpoa_context* ctx, const void* d_bases, uint64_t n, int c);
MsmTable* msm_table_build_g2(zkpoa_context* ctx, const void* d_bases, uint64_t n, int c);
size_t msm_table_bytes_g1(uint64_t n, int c);
size_t msm_table_bytes_g2(uint64_t n, int c);
uint32_t msm_table_width(uint64_t n, int c, bool g2);
void msm_table_release(MsmTable* t);
const void* msm_table_data(const MsmTable* t);
void msm_table_info(const MsmTable* t, uint64_t out[4]);   // n, c, W, bytes
// bytes of lane workspace an MSM over n points reserves (msm.hip.h msm_sort/accum_workspace_bytes of its plan, which
// follows the calling thread's density hint). g1: bucket sort (for_g2: one that also feeds a G2 accumulation) and, with
// `accum`, the G1 accumulation in the same arena; g2: the G2 accumulation and, with `sort`, its own sort.
// table_c > 0: the fixed-base form with that window width.
size_t msm_workspace_g1(uint64_t n, int force_c, bool for_g2, int table_c, bool accum);
size_t msm_workspace_g2(uint64_t n, int force_c, int table_c, bool sort);
void msm_set_forced_k0(int k0);
// non-zero digits per scalar for every window width (index c, 4..25) of n scalars on the device; synchronises st
// (d_scratch: 256 B of device memory of the caller's -- no allocation here: a hipFree would wait for every lane)
void msm_density(hipStream_t st, const void* d_scalars, uint64_t n, double out[32], void* d_scratch);
void msm_set_density_hint(const double* density);   // for the calling thread; nullptr = uniform scalars
void group_add_run_g1(zkpoa_context* ctx, const void* a, const void* b, void* out, uint64_t n);
void group_add_run_g2(zkpoa_context* ctx, const void* a, const void* b, void* out, uint64_t n);
void gen_bases_g1(zkpoa_context* ctx, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t i0, uint64_t n, void* d_out);
void gen_bases_g2(zkpoa_context* ctx, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t i0, uint64_t n, void* d_out);
// shared-sort form (prover: the A, B1 and B2 queries use the same witness scalars): sort once on `lane`
// (stream synchronised on return), then accumulate each base array on its own lane.
struct MsmSorted;
// for_g2: the result also feeds a G2 accumulation (shorter pieces: a G2 addition has 3x the latency)
// table_c > 0: sort for the fixed-base (merged-bucket) form with that window width
MsmSorted* msm_sort_run(zkpoa_context* ctx, int lane_id, const void* d_scalars, uint64_t n,
                        bool for_g2 = false, int table_c = 0);  // delete with msm_sorted_free
void msm_sorted_free(MsmSorted* sr);
void msm_accum_g1(zkpoa_context* ctx, int lane_id, const MsmSorted* sr, bool own_arena, const void* d_bases,
                  uint8_t* out, float* ms2);
void msm_accum_g2(zkpoa_context* ctx, int lane_id, const MsmSorted* sr, bool own_arena, const void* d_bases,
                  uint8_t* out, float* ms2);
// ntt.hip
void ntt_prepare(zkpoa_context* ctx, hipStream_t st, uint32_t k);  // builds twiddle tables (hipMalloc) once per k
// batch > 1: that many vectors of 2^k elements, `stride` bytes apart, transformed together (one launch per pass)
void ntt_to_odd_coset(zkpoa_context* ctx, hipStream_t st, void* d_data, uint32_t k, uint32_t batch = 1, size_t stride = 0);
void ntt_natural(zkpoa_context* ctx, hipStream_t st, void* d_data, uint32_t k, bool inverse);
// unscaled, unpermuted halves of a transform: DIF natural -> bit-reversed, DIT bit-reversed -> natural
void ntt_dif(zkpoa_context* ctx, hipStream_t st, void* d_data, uint32_t k, bool inverse, uint32_t batch = 1, size_t stride = 0);
void ntt_dit(zkpoa_context* ctx, hipStream_t st, void* d_data, uint32_t k, bool inverse, uint32_t batch = 1, size_t stride = 0);
// H-scalar chain split over G ranks: the step between the two exchanges (ntt.hip.h, ntt_split_mid_kernel)
void ntt_split_mid(zkpoa_context* ctx, hipStream_t st, const void* in, void* out, uint32_t k, uint32_t G, uint32_t h,
                   uint32_t rank_stride);
void ntt_release(zkpoa_context* ctx);
void poseidon_release(zkpoa_context* ctx);
}  // namespace zkpoa

#define ZK_API_BEGIN(ctx)  \
  if (!(ctx)) return PROVER_ERROR; \
  try {                    \
    ZK_HIP(hipSetDevice((ctx)->dev.device));
#define ZK_API_END(ctx)                 \
  }                                     \
  catch (const std::exception& e) {     \
    (ctx)->last_error = e.what();       \
    return PROVER_ERROR;                \
  }                                     \
  return PROVER_OK;

