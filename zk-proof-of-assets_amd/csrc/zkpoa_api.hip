// C-ABI glue of libzkpoa_prover.so (see include/zkpoa_prover.h): context, options, timings,
// MSM entry points, host-only group helpers. Kernels live in the per-group translation units.
#include "zkpoa_internal.hpp"

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>

using namespace zkpoa;

namespace zkpoa {
void set_err(char* buf, unsigned long cap, const std::string& msg) {
  if (!buf || cap == 0) return;
  size_t k = msg.size() < cap - 1 ? msg.size() : cap - 1;
  memcpy(buf, msg.data(), k);
  buf[k] = 0;
}
}  // namespace zkpoa

// ---- context -----------------------------------------------------------------------------------
extern "C" int zkpoa_context_create(int device, zkpoa_context** out, char* error_msg, unsigned long error_msg_maxsize) {
  if (!out) return PROVER_ERROR;
  *out = nullptr;
  zkpoa_context* c = new zkpoa_context();
  try {
    // the uploader's pinned staging buffers (24 MiB, ~15 ms to pin) come up with the copy stream, off the main thread
    c->dev.after_copy_stream = [c, device](hipStream_t st) { c->uploader.prepare(device, st); };
    c->dev.after_lanes = [c, device] { c->uploader.add_streams(device); };
    c->dev.init(device);
    for (auto& l : c->dev.lanes) l.ws.hold_back = &c->key_hold_back;
    // cap of every MSM lane's workspace (as option lane_workspace_max_mb): a card shared with other work
    if (const char* e = getenv("ZKPOA_LANE_WORKSPACE_MAX_MB")) {
      if (*e) {
        char* end = nullptr;
        const long v = strtol(e, &end, 10);
        if (end == e || *end || v < 0 || v > (1l << 20))
          throw std::runtime_error(std::string("ZKPOA_LANE_WORKSPACE_MAX_MB='") + e + "' is not a number of MiB");
        for (auto& l : c->dev.lanes) l.ws.limit = (size_t)v << 20;
      }
    }
    for (int i = 0; i < DeviceCtx::kLanes; i++) {
      ZK_HIP(hipEventCreate(&c->ev_a[i]));
      ZK_HIP(hipEventCreate(&c->ev_b[i]));
    }
  } catch (const std::exception& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    delete c;
    return PROVER_ERROR;
  }
  *out = c;
  return PROVER_OK;
}

extern "C" void zkpoa_context_destroy(zkpoa_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->dev.device);
  (void)hipDeviceSynchronize();
  if (ctx->ev_witness) (void)hipEventDestroy(ctx->ev_witness);
  for (int i = 0; i < DeviceCtx::kLanes; i++) {
    if (ctx->ev_a[i]) (void)hipEventDestroy(ctx->ev_a[i]);
    if (ctx->ev_b[i]) (void)hipEventDestroy(ctx->ev_b[i]);
  }
  ntt_release(ctx);
  poseidon_release(ctx);
  ctx->dev.join_background();   // it may still be adding the uploader's streams
  ctx->uploader.release();
  ctx->dev.destroy();
  delete ctx;
}

extern "C" const char* zkpoa_last_error(const zkpoa_context* ctx) {
  return ctx ? ctx->last_error.c_str() : "null context";
}

extern "C" float zkpoa_last_ms(const zkpoa_context* ctx, int id) {
  if (!ctx || id < 0 || id >= 8) return -1.f;
  return ctx->ms[id];
}

extern "C" uint64_t zkpoa_msm_points_limit(const zkpoa_context* ctx) { return ctx ? ctx->msm_points_limit_min() : 0; }

extern "C" int zkpoa_set_option(zkpoa_context* ctx, const char* key, long value) {
  if (!ctx || !key) return PROVER_ERROR;
  if (!strcmp(key, "msm_c")) {
    ctx->opt_msm_c = (int)value;
    return PROVER_OK;
  }
  if (!strcmp(key, "msm_max_points")) {   // tests: force the chunked path at small sizes
    if (value < 0) return PROVER_ERROR;
    ctx->opt_msm_max_points = value;
    return PROVER_OK;
  }
  if (!strcmp(key, "msm_k0")) {           // experiments: level-0 piece length of the bucket accumulation (process-wide)
    msm_set_forced_k0((int)value);
    return PROVER_OK;
  }
  if (!strcmp(key, "prove_serial")) {     // measurement: no overlap between the stages of a prove
    ctx->opt_prove_serial = value != 0;
    return PROVER_OK;
  }
  if (!strcmp(key, "scan_poll_limit_log2")) {   // polls of one status word before a scan look-back gives up (default 24)
    if (value < 4 || value > 30) return PROVER_ERROR;
    for (auto& l : ctx->dev.lanes) l.scan_poll_limit = 1u << value;
    return PROVER_OK;
  }
  if (!strcmp(key, "lane_workspace_max_mb")) {  // cap of every MSM lane's workspace (0 = none): an MSM whose sort does
    if (value < 0) return PROVER_ERROR;         // not fit under it goes through its points in pieces (msm_run)
    for (auto& l : ctx->dev.lanes) l.ws.limit = (size_t)value << 20;
    for (auto& o : ctx->oom_max_points) o = 0;
    return PROVER_OK;
  }
  if (!strcmp(key, "scan_test_withhold")) {     // tests only: tile 0 of every scan withholds its prefix
    for (auto& l : ctx->dev.lanes) l.scan_test_withhold = value != 0;
    return PROVER_OK;
  }
  ctx->last_error = std::string("unknown option ") + key;
  return PROVER_ERROR;
}

// measurement hook (tools/upload_bench.py; not in the public header): `bytes` of the file at `path` from offset `off`
// into device memory through the context's uploader, exactly as a witness or a zkey section travels. *ms <- host time.
extern "C" int zkpoa_test_upload(zkpoa_context* ctx, const char* path, uint64_t off, uint64_t bytes, void* d_dst, float* ms) {
  if (!ctx || !path || !d_dst) return PROVER_ERROR;
  int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return PROVER_ERROR;
  int rc = PROVER_OK;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    ctx->dev.wait_lanes();
    const auto t0 = std::chrono::steady_clock::now();
    ctx->uploader.upload(d_dst, nullptr, bytes, ctx->dev.device, ctx->dev.lanes[0].stream, fd, off);
    if (ms) *ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  } catch (const std::exception& e) {
    ctx->last_error = e.what();
    rc = PROVER_ERROR;
  }
  close(fd);
  return rc;
}

// ---- MSM ---------------------------------------------------------------------------------------
static void check_n(uint64_t n) {
  if (n > (1ull << 28)) throw HipError("msm: n too large (max 2^28 points per call)");
}

extern "C" int zkpoa_msm_g1_device(zkpoa_context* ctx, const void* d_bases, const void* d_scalars, uint64_t n,
                                   uint8_t out[64]) {
  ZK_API_BEGIN(ctx)
  check_n(n);
  msm_run_g1(ctx, 0, d_bases, d_scalars, n, out, ctx->ms);
  ZK_API_END(ctx)
}
// Same MSM on an explicit lane (stream + workspace), so a caller can keep several MSMs in flight from
// several host threads: each lane is independent; calls on the SAME lane must not overlap.
extern "C" int zkpoa_msm_g1_device_lane(zkpoa_context* ctx, int lane, const void* d_bases, const void* d_scalars,
                                        uint64_t n, uint8_t out[64]) {
  if (!ctx || lane < 0 || lane >= DeviceCtx::kLanes) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    check_n(n);
    msm_run_g1(ctx, lane, d_bases, d_scalars, n, out, ctx->lane_ms[lane]);
  } catch (const std::exception& e) {
    return PROVER_ERROR;   // last_error is not touched: it is shared between lanes
  }
  return PROVER_OK;
}

extern "C" float zkpoa_last_ms_lane(const zkpoa_context* ctx, int lane, int id) {
  if (!ctx || lane < 0 || lane >= DeviceCtx::kLanes || id < 0 || id > 2) return -1.f;
  if (id == 2) return (float)(ctx->lane_adds[lane] * 1e-6);   // millions of mixed additions (float: 24-bit mantissa)
  return ctx->lane_ms[lane][id];
}

extern "C" int zkpoa_msm_g2_device(zkpoa_context* ctx, const void* d_bases, const void* d_scalars, uint64_t n,
                                   uint8_t out[128]) {
  ZK_API_BEGIN(ctx)
  check_n(n);
  msm_run_g2(ctx, 0, d_bases, d_scalars, n, out, ctx->ms);
  ZK_API_END(ctx)
}
extern "C" int zkpoa_msm_g1(zkpoa_context* ctx, const void* bases, const void* scalars, uint64_t n, uint8_t out[64]) {
  ZK_API_BEGIN(ctx)
  check_n(n);
  DevBuf db(n * 64), ds(n * 32);
  // pageable host buffers: the multi-threaded pinned-staging uploader (a plain hipMemcpy stages at ~4 GB/s)
  ctx->uploader.upload(db.p, bases, n * 64, ctx->dev.device, ctx->dev.lanes[0].stream);
  ctx->uploader.upload(ds.p, scalars, n * 32, ctx->dev.device, ctx->dev.lanes[0].stream);
  msm_run_g1(ctx, 0, db.p, ds.p, n, out, ctx->ms);
  ZK_API_END(ctx)
}
extern "C" int zkpoa_msm_g2(zkpoa_context* ctx, const void* bases, const void* scalars, uint64_t n, uint8_t out[128]) {
  ZK_API_BEGIN(ctx)
  check_n(n);
  DevBuf db(n * 128), ds(n * 32);
  // pageable host buffers: the multi-threaded pinned-staging uploader (a plain hipMemcpy stages at ~4 GB/s)
  ctx->uploader.upload(db.p, bases, n * 128, ctx->dev.device, ctx->dev.lanes[0].stream);
  ctx->uploader.upload(ds.p, scalars, n * 32, ctx->dev.device, ctx->dev.lanes[0].stream);
  msm_run_g2(ctx, 0, db.p, ds.p, n, out, ctx->ms);
  ZK_API_END(ctx)
}

// ---- fixed-base tables (resident bases: zkey sections never change) ----------------------------------------------
extern "C" int zkpoa_msm_table_build(zkpoa_context* ctx, int group, const void* d_bases, uint64_t n, int window_bits,
                                     zkpoa_msm_table** out) {
  if (!out) return PROVER_ERROR;
  *out = nullptr;
  ZK_API_BEGIN(ctx)
  check_n(n);
  if (group != 1 && group != 2) throw HipError("msm_table_build: group must be 1 or 2");
  zkpoa_msm_table* h = new zkpoa_msm_table();
  h->group = group;
  try {
    h->t = group == 1 ? msm_table_build_g1(ctx, d_bases, n, window_bits) : msm_table_build_g2(ctx, d_bases, n, window_bits);
  } catch (...) {
    delete h;
    throw;
  }
  *out = h;
  ZK_API_END(ctx)
}
extern "C" void zkpoa_msm_table_free(zkpoa_context* ctx, zkpoa_msm_table* table) {
  if (!table) return;
  if (ctx) {
    (void)hipSetDevice(ctx->dev.device);
    (void)hipDeviceSynchronize();
  }
  msm_table_release(table->t);
  delete table;
}
extern "C" int zkpoa_msm_table_info(const zkpoa_msm_table* table, uint64_t out[4]) {
  if (!table || !table->t || !out) return PROVER_ERROR;
  msm_table_info(table->t, out);
  return PROVER_OK;
}
extern "C" int zkpoa_msm_table_run_lane(zkpoa_context* ctx, int lane, const zkpoa_msm_table* table,
                                        const void* d_scalars, uint8_t* out) {
  if (!ctx || !table || !table->t || !out || lane < 0 || lane >= DeviceCtx::kLanes) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    uint64_t info[4];
    msm_table_info(table->t, info);
    if (table->group == 1) msm_run_g1(ctx, lane, nullptr, d_scalars, info[0], out, ctx->lane_ms[lane], table->t);
    else msm_run_g2(ctx, lane, nullptr, d_scalars, info[0], out, ctx->lane_ms[lane], table->t);
  } catch (const std::exception& e) {
    return PROVER_ERROR;   // last_error is shared between lanes: not touched
  }
  return PROVER_OK;
}

// ---- synthetic bases / element-wise hooks --------------------------------------------------------
extern "C" int zkpoa_gen_bases_g1_device(zkpoa_context* ctx, const uint8_t a_le[32], const uint8_t b_le[32],
                                         uint64_t i0, uint64_t n, void* d_out) {
  ZK_API_BEGIN(ctx)
  gen_bases_g1(ctx, a_le, b_le, i0, n, d_out);
  ZK_API_END(ctx)
}
extern "C" int zkpoa_gen_bases_g2_device(zkpoa_context* ctx, const uint8_t a_le[32], const uint8_t b_le[32],
                                         uint64_t i0, uint64_t n, void* d_out) {
  ZK_API_BEGIN(ctx)
  gen_bases_g2(ctx, a_le, b_le, i0, n, d_out);
  ZK_API_END(ctx)
}
extern "C" int zkpoa_group_add(zkpoa_context* ctx, int group, const void* a, const void* b, void* out, uint64_t n) {
  ZK_API_BEGIN(ctx)
  if (n == 0) return PROVER_OK;
  if (group == 1) group_add_run_g1(ctx, a, b, out, n);
  else if (group == 2) group_add_run_g2(ctx, a, b, out, n);
  else throw HipError("group_add: group must be 1 or 2");
  ZK_API_END(ctx)
}

// ---- host-only group helpers ---------------------------------------------------------------------
template <class HF>
static int group_sum(const void* points, uint64_t count, uint8_t* out) {
  constexpr size_t A = 2 * HostBytes<HF>::N;
  XYZZ<HF> acc = XYZZ<HF>::inf();
  for (uint64_t i = 0; i < count; i++) {
    Affine<HF> p = h_affine_from_bytes<HF>(reinterpret_cast<const char*>(points) + i * A);
    xyzz_add_affine(acc, p, false);
  }
  h_affine_to_bytes<HF>(h_to_affine(acc), out);
  return PROVER_OK;
}
extern "C" int zkpoa_g1_sum(const void* points, uint64_t count, uint8_t out[64]) {
  return group_sum<HFq>(points, count, out);
}
extern "C" int zkpoa_g2_sum(const void* points, uint64_t count, uint8_t out[128]) {
  return group_sum<HFq2>(points, count, out);
}
template <class HF>
static int group_mul(const uint8_t* point, const uint8_t* scalar_le, uint8_t* out) {
  uint64_t k[4];
  memcpy(k, scalar_le, 32);
  Affine<HF> p = h_affine_from_bytes<HF>(point);
  h_affine_to_bytes<HF>(h_to_affine(h_mul(XYZZ<HF>::from_affine(p), k)), out);
  return PROVER_OK;
}
extern "C" int zkpoa_g1_mul(const uint8_t point[64], const uint8_t scalar_le[32], uint8_t out[64]) {
  return group_mul<HFq>(point, scalar_le, out);
}
extern "C" int zkpoa_g2_mul(const uint8_t point[128], const uint8_t scalar_le[32], uint8_t out[128]) {
  return group_mul<HFq2>(point, scalar_le, out);
}
