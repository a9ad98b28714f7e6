// NTT translation unit: kernels from ntt.hip.h + the C-ABI entry points zkpoa_ntt / zkpoa_ntt_device.
#include "ntt.hip.h"
#include "zkpoa_internal.hpp"

namespace zkpoa {

static NttEngine* engine(zkpoa_context* ctx) {
  if (!ctx->ntt) ctx->ntt = new NttEngine();
  return ctx->ntt;
}

void ntt_to_odd_coset(zkpoa_context* ctx, hipStream_t st, void* d_data, uint32_t k, uint32_t batch, size_t stride) {
  engine(ctx)->to_odd_coset(st, d_data, k, batch, stride);
}
void ntt_natural(zkpoa_context* ctx, hipStream_t st, void* d_data, uint32_t k, bool inverse) {
  engine(ctx)->transform_natural(st, d_data, k, inverse);
}
void ntt_prepare(zkpoa_context* ctx, hipStream_t st, uint32_t k) {
  if (k == 0) return;
  NttEngine* e = engine(ctx);
  (void)e->tables(st, k, false);
  (void)e->tables(st, k, true);
  HFr inc = (k == 28) ? HFr::from_u64(25) : hfr_root_of_unity(k + 1);
  HFr ninv = HFr::from_u64(1ull << k).inv();
  (void)e->pow_tables(st, k, inc, ninv, k);
}
void ntt_dif(zkpoa_context* ctx, hipStream_t st, void* d_data, uint32_t k, bool inverse, uint32_t batch, size_t stride) {
  engine(ctx)->dif(st, d_data, k, inverse, batch, stride);
}
void ntt_dit(zkpoa_context* ctx, hipStream_t st, void* d_data, uint32_t k, bool inverse, uint32_t batch, size_t stride) {
  engine(ctx)->dit(st, d_data, k, inverse, batch, stride);
}
void ntt_split_mid(zkpoa_context* ctx, hipStream_t st, const void* in, void* out, uint32_t k, uint32_t G, uint32_t h,
                   uint32_t rank_stride) {
  engine(ctx)->split_mid(st, in, out, k, G, h, rank_stride);
}
void ntt_release(zkpoa_context* ctx) {
  if (ctx->ntt) {
    ctx->ntt->release();
    delete ctx->ntt;
    ctx->ntt = nullptr;
  }
}

}  // namespace zkpoa
using namespace zkpoa;

extern "C" int zkpoa_ntt_device(zkpoa_context* ctx, void* d_data, unsigned log_n, int inverse) {
  ZK_API_BEGIN(ctx)
  if (log_n > 28) throw HipError("ntt: log_n > 28 (two-adicity of Fr)");
  Lane& lane = ctx->dev.lanes[0];
  ntt_prepare(ctx, lane.stream, log_n);
  ZK_HIP(hipEventRecord(ctx->ev_a[0], lane.stream));
  ntt_natural(ctx, lane.stream, d_data, log_n, inverse != 0);
  ZK_HIP(hipEventRecord(ctx->ev_b[0], lane.stream));
  ZK_HIP(hipStreamSynchronize(lane.stream));
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipEventElapsedTime(&ctx->ms[2], ctx->ev_a[0], ctx->ev_b[0]));
  ZK_API_END(ctx)
}

extern "C" int zkpoa_ntt(zkpoa_context* ctx, void* data, unsigned log_n, int inverse) {
  ZK_API_BEGIN(ctx)
  if (log_n > 28) throw HipError("ntt: log_n > 28 (two-adicity of Fr)");
  size_t bytes = (size_t)32 << log_n;
  DevBuf d(bytes);
  ZK_HIP(hipMemcpy(d.p, data, bytes, hipMemcpyHostToDevice));
  int rc = zkpoa_ntt_device(ctx, d.p, log_n, inverse);
  if (rc != PROVER_OK) return rc;
  ZK_HIP(hipMemcpy(data, d.p, bytes, hipMemcpyDeviceToHost));
  ZK_API_END(ctx)
}
