// Host -> HBM upload of large pageable buffers (an mmap'd .zkey: 1 GB for layer one, 21 GB for layer
// three). A plain hipMemcpy from pageable memory stages through one internal buffer
// (~4 GB/s measured); here T host threads each own a slice of the transfer, copy it chunk-wise into
// their own pinned double buffer and issue hipMemcpyAsync on their own stream, so page-cache reads,
// staging copies and PCIe DMA all overlap.
#pragma once
#include "device_ctx.hpp"

#include <ctype.h>
#include <pthread.h>
#include <sched.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

namespace zkpoa {

class FastUploader {
 public:
  static constexpr int kMaxThreads = 16;
  // 2 MiB per buffer, two per thread: large enough for full-rate DMA, and pinning the staging memory is a
  // start-up cost that a one-shot prover pays every time (12 x 8 MiB cost ~60 ms, 12 x 2 MiB ~15 ms)
  static constexpr size_t kChunk = 2u << 20;

  ~FastUploader() { release(); }

  // `stream`: an existing stream of the caller that is idle during the upload (the context's lane 0). Creating
  // a HIP stream costs 10-50 ms (it brings up a hardware queue), which a one-shot prover pays on every run, so
  // the uploader owns none by default: the copies of all threads interleave on the one stream at full PCIe rate.
  // fd >= 0: the bytes are read with pread(fd, file_off + ...) straight into the pinned buffers instead of being
  // copied out of `src` (a fresh mmap of a page-cached 1 GB file costs a page fault per 64 KiB on top of the copy:
  // ~10 GB/s; pread from the page cache runs at memcpy speed and skips the second copy); `src` may then be null.
  void upload(void* dst, const void* src, size_t bytes, int device, hipStream_t stream, int fd = -1, uint64_t file_off = 0) {
    if (bytes < (4u << 20)) {  // small: not worth the threads
      std::vector<char> tmp;
      if (fd >= 0) {
        tmp.resize(bytes);
        read_all(fd, tmp.data(), bytes, file_off);
        src = tmp.data();
      }
      ZK_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
      return;
    }
    ensure(device, stream);
    const int T = threads_;
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> errs(T);
    size_t per = ((bytes / T) + 4095) & ~size_t(4095);
    for (int t = 0; t < T; t++) {
      size_t lo = (size_t)t * per;
      if (lo >= bytes) break;
      size_t hi = lo + per < bytes ? lo + per : bytes;
      th.emplace_back([this, t, lo, hi, dst, src, device, stream, fd, file_off, &errs] {
        try {
          ZK_HIP(hipSetDevice(device));
          if (have_cpus_) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpus_), &cpus_);
          Slot& s = slots_[t];
          // streams: the caller's plus the uploader's own, as many as exist by now (they come up in the background);
          // one stream -- one SDMA queue -- moves ~36 GB/s whatever the thread count, four reach the link's ~53 GB/s
          const int ns = n_own_.load(std::memory_order_acquire);
          const int pick = t % (ns + 1);
          hipStream_t st = pick == ns ? stream : own_[pick];
          int b = 0;
          for (size_t off = lo; off < hi; off += kChunk, b ^= 1) {
            size_t len = hi - off < kChunk ? hi - off : kChunk;
            ZK_HIP(hipEventSynchronize(s.done[b]));  // the DMA that last used this buffer has finished
            if (fd >= 0) read_all(fd, static_cast<char*>(s.pinned[b]), len, file_off + off);
            else memcpy(s.pinned[b], reinterpret_cast<const char*>(src) + off, len);
            ZK_HIP(hipMemcpyAsync(reinterpret_cast<char*>(dst) + off, s.pinned[b], len, hipMemcpyHostToDevice, st));
            ZK_HIP(hipEventRecord(s.done[b], st));
          }
          for (int k = 0; k < 2; k++) ZK_HIP(hipEventSynchronize(s.done[k]));
        } catch (...) {
          errs[t] = std::current_exception();
        }
      });
    }
    for (auto& t : th) t.join();
    for (auto& e : errs)
      if (e) std::rethrow_exception(e);
  }

  // Streams of the uploader's own (ZKPOA_UPLOAD_STREAMS, default 3: four with the caller's). Called from the context's
  // background thread AFTER the lanes are up: a stream costs 10-50 ms to create, so a one-shot prover's first upload
  // (the witness) starts on the caller's stream alone and later sections find more streams as they appear.
  // Their priority is HIGH: on this runtime a host -> device copy is queued like a kernel, and at normal priority it
  // waits behind the lanes' kernels -- measured on an MI355X (tools/upload_bench.py, 8 GB from the page cache, four lanes
  // running 2^22-point MSMs): 3 GB/s on three normal-priority streams, 12 GB/s on one, 34 GB/s on one high-priority
  // stream (47 GB/s with the chip idle), 40 GB/s on three with GPU_MAX_HW_QUEUES=16 (which the `prover` executable sets
  // for its own process: with the default of 4 hardware queues the eleven streams of a context share queues, and a
  // copy then sits behind a kernel of the lane it shares with). Default: 1 own stream, 3 when GPU_MAX_HW_QUEUES >= 8.
  void add_streams(int device) {
    int want = 1;
    if (const char* q = getenv("GPU_MAX_HW_QUEUES"))
      if (atoi(q) >= 8) want = 3;
    if (const char* e = getenv("ZKPOA_UPLOAD_STREAMS")) want = atoi(e);
    if (want > kMaxOwnStreams) want = kMaxOwnStreams;
    for (int i = n_own_.load(); i < want; i++) {
      if (hipSetDevice(device) != hipSuccess) return;
      hipStream_t st = nullptr;
      int lo = 0, hi = 0;   // numerically lower = higher priority
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      const char* pe = getenv("ZKPOA_UPLOAD_STREAM_PRIO");
      const int prio = pe && !strcmp(pe, "normal") ? (lo + hi) / 2 : pe && !strcmp(pe, "low") ? lo : hi;
      if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess) {
        (void)hipGetLastError();
        return;
      }
      own_[i] = st;
      n_own_.store(i + 1, std::memory_order_release);
    }
  }

  // pin the staging buffers ahead of the first upload (a one-shot prover: on the context's background thread)
  void prepare(int device, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(ensure_mutex_);
    ensure_locked(device, stream);
  }

  void release() {
    for (auto& s : slots_) {
      for (int b = 0; b < 2; b++) {
        if (s.done[b]) (void)hipEventDestroy(s.done[b]);
        s.pinned[b] = nullptr;
        s.done[b] = nullptr;
      }
    }
    const int ns = n_own_.exchange(0);
    for (int i = 0; i < ns; i++) (void)hipStreamDestroy(own_[i]);
    if (block_) (void)hipHostFree(block_);
    block_ = nullptr;
    ready_ = false;
  }

 private:
  struct Slot {
    void* pinned[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
  };
  Slot slots_[kMaxThreads];
  void* block_ = nullptr;
  bool ready_ = false;
  int threads_ = 8;
  cpu_set_t cpus_;            // the CPUs of the GPU's NUMA node (staging copies stay on its socket)
  bool have_cpus_ = false;
  static constexpr int kMaxOwnStreams = 8;
  hipStream_t own_[kMaxOwnStreams] = {};
  std::atomic<int> n_own_{0};

  static void read_all(int fd, char* out, size_t len, uint64_t pos) {
    size_t got = 0;
    while (got < len) {
      ssize_t r = pread(fd, out + got, len - got, (off_t)(pos + got));
      if (r <= 0) throw HipError("upload: short read from the file");
      got += (size_t)r;
    }
  }

  // CPUs of the NUMA node the GPU hangs off (sysfs: the PCI device's numa_node, that node's cpulist); false when unknown
  static bool gpu_node_cpus(int device, cpu_set_t* out) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) return false;
    for (char* c = bus; *c; c++) *c = (char)tolower(*c);
    char path[256];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = fopen(path, "r");
    int node = -1;
    if (!f || fscanf(f, "%d", &node) != 1) node = -1;
    if (f) fclose(f);
    if (node < 0) return false;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return false;
    CPU_ZERO(out);
    int a = 0, b = 0, n = 0;
    char sep = 0;
    while (fscanf(f, "%d", &a) == 1) {
      b = a;
      int ch = fgetc(f);
      if (ch == '-') {
        if (fscanf(f, "%d", &b) != 1) break;
        ch = fgetc(f);
      }
      for (int c = a; c <= b && c < CPU_SETSIZE; c++) {
        CPU_SET(c, out);
        n++;
      }
      sep = (char)ch;
      if (sep != ',') break;
    }
    fclose(f);
    // only CPUs this process may use at all
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
      CPU_AND(out, out, &allowed);
      n = CPU_COUNT(out);
    }
    return n > 0;
  }

  std::mutex ensure_mutex_;
  void ensure(int device, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(ensure_mutex_);
    ensure_locked(device, stream);
  }
  void ensure_locked(int device, hipStream_t stream) {
    if (ready_) return;
    // Threads: each copies its slice through its own pair of pinned buffers. Measured on an MI355X box (r04,
    // tools/file_inclusive.py): see DESIGN.md section 7 for the rates; ZKPOA_UPLOAD_THREADS overrides (1..16).
    if (const char* e = getenv("ZKPOA_UPLOAD_THREADS")) {
      const int v = atoi(e);
      if (v >= 1 && v <= kMaxThreads) threads_ = v;
    }
    unsigned hw = std::thread::hardware_concurrency();
    if (hw && (int)hw < threads_) threads_ = (int)hw;
    ZK_HIP(hipSetDevice(device));
    // the staging threads (and the first touch of their pinned buffers) stay on the CPUs of the GPU's NUMA node: 51 vs
    // 47 GB/s on a two-socket box (ZKPOA_UPLOAD_AFFINITY=off: wherever the scheduler puts them)
    {
      const char* e = getenv("ZKPOA_UPLOAD_AFFINITY");
      if (!e || strcmp(e, "off") != 0) have_cpus_ = gpu_node_cpus(device, &cpus_);
    }
    cpu_set_t before;
    const bool moved = have_cpus_ && pthread_getaffinity_np(pthread_self(), sizeof(before), &before) == 0 &&
                       pthread_setaffinity_np(pthread_self(), sizeof(cpus_), &cpus_) == 0;
    ZK_HIP(hipHostMalloc(&block_, (size_t)threads_ * 2 * kChunk, hipHostMallocDefault));   // one pinning call
    if (moved) {
      memset(block_, 0, (size_t)threads_ * 2 * kChunk);   // first touch on the GPU's node
      (void)pthread_setaffinity_np(pthread_self(), sizeof(before), &before);
    }
    for (int t = 0; t < threads_; t++) {
      Slot& s = slots_[t];
      for (int b = 0; b < 2; b++) {
        s.pinned[b] = static_cast<char*>(block_) + ((size_t)t * 2 + b) * kChunk;
        ZK_HIP(hipEventCreate(&s.done[b]));
        ZK_HIP(hipEventRecord(s.done[b], stream));
      }
    }
    ready_ = true;
  }
};

}  // namespace zkpoa
