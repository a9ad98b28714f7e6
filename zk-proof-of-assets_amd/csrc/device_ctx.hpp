// Device context shared by the MSM / NTT / ABC stages: one HIP device, a set of streams, and a
// grow-only workspace arena per stream so that no hipMalloc happens inside a timed step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <atomic>
#include <condition_variable>
#include <string.h>
#include <exception>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <string>
#include <vector>

namespace zkpoa {

struct HipError : std::runtime_error {
  explicit HipError(const std::string& s) : std::runtime_error(s) {}
};

// a lane's workspace did not fit: the MSM drivers answer by working through the points in smaller pieces (msm_run)
struct OomError : HipError {
  explicit OomError(const std::string& s) : HipError(s) {}
};

#define ZK_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      throw ::zkpoa::HipError(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " + \
                              __FILE__ + ":" + std::to_string(__LINE__));                         \
  } while (0)

// Grow-only bump arena. reset() rewinds; take() never frees, it reallocates when too small
// (only legal while nothing is in flight on the owning stream -> callers reserve() up front).
struct Arena {
  char* base = nullptr;
  size_t cap = 0;
  size_t off = 0;
  size_t limit = 0;   // 0 = whatever HBM holds; else reserve() refuses more than this (option lane_workspace_max_mb)
  // bytes of HBM a key that is still being loaded will ask for (a staged one-shot prove: its lanes start while the
  // sections are still arriving): a workspace may not take them
  const std::atomic<int64_t>* hold_back = nullptr;
  void reserve(size_t bytes) {
    if (bytes <= cap) return;
    if (limit && bytes > limit)
      throw OomError("workspace of " + std::to_string(bytes >> 20) + " MiB is over the lane limit of " +
                     std::to_string(limit >> 20) + " MiB");
    if (hold_back) {
      const int64_t keep = hold_back->load();
      size_t free_b = 0, total_b = 0;
      if (keep > 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b + (double)cap < (double)bytes + (double)keep)
        throw OomError("workspace of " + std::to_string(bytes >> 20) + " MiB would take HBM the key still being loaded needs (" +
                       std::to_string(keep >> 20) + " MiB to come, " + std::to_string(free_b >> 20) + " MiB free)");
    }
    if (base) ZK_HIP(hipFree(base));
    base = nullptr;
    cap = off = 0;   // (a failed allocation below must not leave a capacity without memory behind it)
    const hipError_t e = hipMalloc(&base, bytes);
    if (e == hipErrorOutOfMemory) {
      base = nullptr;
      (void)hipGetLastError();   // the error is reported by the exception, not left behind for the next launch check
      throw OomError("workspace of " + std::to_string(bytes >> 20) + " MiB: out of device memory");
    }
    ZK_HIP(e);
    cap = bytes;
  }
  void reset() { off = 0; }
  template <class T>
  T* take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    if (off + bytes > cap) throw HipError("arena overflow (reserve() too small)");
    T* p = reinterpret_cast<T*>(base + off);
    off += bytes;
    return p;
  }
  void release() {
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = off = 0;
  }
};

struct Lane {  // one stream + its workspace + a pinned host staging area for small read-backs
  hipStream_t stream = nullptr;
  Arena ws;
  void* pinned = nullptr;
  void* pinned_dev = nullptr;   // the same memory as kernels address it (read-backs are written by a kernel, not copied)
  size_t pinned_cap = 0;
  // host copy of an MSM's read-back (the per-bit totals of its bucket reduction), owned by the lane: a fresh
  // 256 KB buffer per call would be an mmap + munmap each time, and unmapping is slow in a process whose address
  // space the GPU driver watches (measured: 11 % of the six-in-flight 2^20 MSM rate)
  std::vector<char> host_sums;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // scratch of the single-launch scans (msm.hip.h scan_pair): status words + ticket counters, cleared only when
  // (re)allocated -- every call stamps its words with the next generation number
  void* scan_scratch = nullptr;
  uint32_t scan_tiles = 0, scan_gen = 0;
  // The look-back of scan_pair waits for words other workgroups publish: a wait that can never end (an ordering mistake
  // on the host side, a wiped scratch) must surface as an error, not as a hung GPU. After scan_poll_limit polls of one
  // word a wavefront gives up, raises the lane's fault word -- the last 64 bytes of the pinned staging area, which the
  // host looks at after every read-back -- and carries on with a zero prefix so that the grid still drains.
  // scan_test_withhold (tests only, zkpoa_set_option): tile 0 never publishes its prefix.
  uint32_t scan_poll_limit = 1u << 24, scan_test_withhold = 0;
  static constexpr size_t kFaultBytes = 64;
  volatile uint32_t* fault_word() const { return reinterpret_cast<volatile uint32_t*>(static_cast<char*>(pinned) + pinned_cap); }
  uint32_t* fault_word_dev() const { return reinterpret_cast<uint32_t*>(static_cast<char*>(pinned_dev) + pinned_cap); }
  // level 2 = highest, 1 = middle, 0 = lowest stream priority
  void init(int level = 0) {
    int lo = 0, hi = 0;  // numerically lower = higher priority
    ZK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    int prio = level >= 2 ? hi : (level == 1 ? (lo + hi) / 2 : lo);
    // creating the stream is the expensive part (10-50 ms: a hardware queue comes up with it)
    ZK_HIP(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio));
    pinned_cap = (1 << 20) - kFaultBytes;   // read-backs use [0, pinned_cap); the fault word sits behind them
    ZK_HIP(hipHostMalloc(&pinned, pinned_cap + kFaultBytes, hipHostMallocDefault));
    ZK_HIP(hipHostGetDevicePointer(&pinned_dev, pinned, 0));
    memset(static_cast<char*>(pinned) + pinned_cap, 0, kFaultBytes);
    host_sums.assign(pinned_cap + kFaultBytes, 0);
    ZK_HIP(hipEventCreate(&ev0));
    ZK_HIP(hipEventCreate(&ev1));
  }
  void destroy() {
    ws.release();
    if (scan_scratch) (void)hipFree(scan_scratch);
    scan_scratch = nullptr;
    scan_tiles = scan_gen = 0;
    if (pinned) (void)hipHostFree(pinned);
    if (stream) (void)hipStreamDestroy(stream);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    pinned = nullptr;
    stream = nullptr;
    ev0 = ev1 = nullptr;
  }
};

struct DeviceCtx {
  int device = 0;
  int num_cu = 256;
  static constexpr int kLanes = 12;      // lanes a caller may address (zkpoa_msm_g1_device_lane ...)
  static constexpr int kEagerLanes = 5;  // created with the context (the prover uses 0-4); the rest on first use
  Lane lanes[kLanes];
  std::mutex lazy_mutex_;
  bool ok = false;
  // last-run kernel timing (HIP events on the lane's stream), for bench.py's roofline object
  float last_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::thread bg_;
  std::mutex bg_mutex_;
  std::condition_variable bg_cv_;
  bool lanes_ready_ = false;     // lanes 1 .. kEagerLanes-1 exist (or bg_err_ says why not); extras may still be coming up
  std::exception_ptr bg_err_;
  // A stream of its own for host -> HBM copies that run WHILE lanes compute (one-shot proves overlap the zkey upload
  // with the MSMs). First thing the background thread creates; copy_stream_wait() blocks until it exists.
  hipStream_t copy_stream = nullptr;
  std::function<void(hipStream_t)> after_copy_stream;   // runs on the background thread once the copy stream exists
  std::function<void()> after_lanes;                    // runs on the background thread once every eager lane is up
  std::mutex copy_mutex_;
  std::condition_variable copy_cv_;
  bool copy_done_ = false;

  void init(int dev) {
    const bool verbose = getenv("ZKPOA_VERBOSE") != nullptr;
    auto now_ms = [] {
      struct timespec ts;
      clock_gettime(CLOCK_MONOTONIC, &ts);
      return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
    };
    double t0 = now_ms();
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
      throw HipError("zkpoa: no HIP device visible (the MSM/NTT path is HIP-only; there is no CPU fallback)");
    if (dev < 0 || dev >= count) throw HipError("zkpoa: device index out of range");
    device = dev;
    ZK_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    ZK_HIP(hipGetDeviceProperties(&prop, dev));
    num_cu = prop.multiProcessorCount;
    double t1 = now_ms();
    // lane 0 carries the prover's critical path (H-scalar chain -> H MSM): highest stream priority
    // A stream with its pinned buffer and events costs ~12 ms to create: lane 0 now, the others on a background
    // thread (they are first needed by the MSMs of a prove, after the key has been uploaded) -> wait_lanes().
    {
      const char* pe0 = getenv("ZKPOA_LANE_PRIO");
      lanes[0].init(pe0 && !strcmp(pe0, "none") ? 0 : 2);
    }
    bg_ = std::thread([this] {
      try {
        ZK_HIP(hipSetDevice(device));
        // highest priority: a copy queued at normal priority waits behind the lanes' kernels (fast_upload.hpp)
        int plo = 0, phi = 0;
        (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
        hipError_t ce = hipStreamCreateWithPriority(&copy_stream, hipStreamNonBlocking, phi);
        if (ce != hipSuccess) copy_stream = nullptr;
        if (copy_stream && after_copy_stream) {
          try {
            after_copy_stream(copy_stream);
          } catch (...) {   // only a head start: the first upload does it again
          }
        }
        {
          std::lock_guard<std::mutex> lk(copy_mutex_);
          copy_done_ = true;
        }
        copy_cv_.notify_all();
        // Stream priorities, two lanes per level (high, high, middle, middle, low, low). Measured on one box
        // (tools/ab_streams.sh): every lane at the same priority costs 8-10 % of the six-in-flight MSM rate and of a
        // 2^21 proof (the lanes then move in lock step and their accumulation kernels collide); which lanes get which
        // level matters little (within 3 %); ZKPOA_LANE_PRIO = flat | none | ladder2 selects the other patterns tried.
        const char* pe = getenv("ZKPOA_LANE_PRIO");
        const bool flat = pe && !strcmp(pe, "flat"), none = pe && !strcmp(pe, "none"), ladder2 = pe && !strcmp(pe, "ladder2");
        const bool top1 = pe && !strcmp(pe, "top1");   // the chain's lane alone at the top level
        static const int kLadder[6] = {2, 2, 1, 1, 0, 0}, kLadder2[6] = {2, 0, 1, 2, 0, 1}, kTop1[6] = {2, 1, 1, 0, 0, 0};
        // (created one after the other: side by side they take just as long -- the runtime serialises stream creation;
        // measured r04, "lanes up" 47-58 ms either way)
        for (int i = 1; i < kEagerLanes; i++)
          lanes[i].init(none ? 0 : ladder2 ? kLadder2[i] : top1 ? kTop1[i] : flat ? (i == 3 ? 1 : 0) : kLadder[i]);
      } catch (...) {
        bg_err_ = std::current_exception();
        {
          std::lock_guard<std::mutex> lk(copy_mutex_);
          copy_done_ = true;
        }
        copy_cv_.notify_all();
      }
      {
        std::lock_guard<std::mutex> lk(bg_mutex_);
        lanes_ready_ = true;
      }
      bg_cv_.notify_all();
      // extras nobody waits for (the uploader's own streams: ~10-50 ms each, and a one-shot layer-one proof is 190 ms)
      if (!bg_err_ && after_lanes) {
        try {
          after_lanes();
        } catch (...) {
        }
      }
    });
    if (verbose)
      fprintf(stderr, "zkpoa: HIP runtime up in %.1f ms, first lane in %.1f ms (the other %d in the background)\n",
              t1 - t0, now_ms() - t1, kEagerLanes - 1);
    ok = true;
  }
  // the copy stream (nullptr when it could not be created: callers then copy on lane 0's stream)
  hipStream_t copy_stream_wait() {
    std::unique_lock<std::mutex> lk(copy_mutex_);
    copy_cv_.wait(lk, [this] { return copy_done_; });
    return copy_stream;
  }
  // lanes kEagerLanes.. come up the first time somebody asks for them (a stream costs ~12 ms to create)
  void ensure_lane(int i) {
    if (i < kEagerLanes) return;
    std::lock_guard<std::mutex> lk(lazy_mutex_);
    if (!lanes[i].stream) {
      ZK_HIP(hipSetDevice(device));
      lanes[i].init(0);
    }
  }
  // every lane other than 0 may only be used after this (cheap once the background thread has been joined)
  void wait_lanes() {
    std::unique_lock<std::mutex> lk(bg_mutex_);
    bg_cv_.wait(lk, [this] { return lanes_ready_; });
    if (bg_err_) {
      std::exception_ptr e = bg_err_;
      bg_err_ = nullptr;
      std::rethrow_exception(e);
    }
  }
  // the background thread has ended (extras included): before anything it touches is torn down
  void join_background() {
    if (bg_.joinable()) bg_.join();
  }
  void destroy() {
    if (!ok) return;
    join_background();
    (void)hipSetDevice(device);
    for (auto& l : lanes) l.destroy();
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    copy_stream = nullptr;
    ok = false;
  }
};

}  // namespace zkpoa
