// One proof over several GPUs INSIDE the drop-in prover (BASELINE.json configs[3], [4]).
//
// The reference's boundary is one exec'd binary (scripts/g16_prove.sh:248-252) and, per batch, one such process
// (scripts/full_workflow.sh:552). So the multi-GPU path lives behind that same entry point, in C++, with no Python and
// no process group: ONE process, one zkpoa_context per device each driven by its own host thread ("rank"), the key
// sharded straight from the file's byte ranges (zkey_load_impl: block-cyclic sections 5-8, cyclic section 9, the
// coefficient rows c = rank mod G), the H-scalar chain split as in zkpoa_split_stage1/2/3 with its two all-to-all
// exchanges done as peer-to-peer writes over xGMI (one kernel per rank and exchange stores chunk h of its buffer straight
// into rank h's receive buffer: a single hop, all links busy at once, no ring; hipMemcpyPeerAsync where a pair has no
// direct path), and the five 64/128-byte partial results summed on the host.
// Included by prover.hip inside its anonymous namespace.
//
// Which devices (env, read once per process):
//   ZKPOA_DEVICES=0,1,2,3   exactly these HIP devices, one rank each (a device may be listed more than once: ranks
//                           then share it -- how the path is rehearsed on a one-GPU box);
//   ZKPOA_DEVICE=n          that one device (the single-GPU path, as before);
//   neither                 automatic: each GPU has a lock file /tmp/zkpoa-<uid>/gpu<N>.lock held for the life of the
//                           process that proves on it. A key whose domain is >= 2^ZKPOA_MULTI_MIN_POWER (default 24)
//                           takes every GPU that is free (rounded down to a power of two), a smaller key one free GPU
//                           -- so the reference's parallel batch jobs (full_workflow.sh:552) spread over the GPUs of the
//                           node instead of all landing on GPU 0, and a layer-three proof run alone uses all of them.
//                           When no GPU is free the process waits for GPU (pid mod count).
#pragma once

struct RankBarrier {   // host barrier of the rank threads (C++17: no std::barrier)
  std::mutex m;
  std::condition_variable cv;
  unsigned n, waiting = 0, generation = 0;
  explicit RankBarrier(unsigned count) : n(count) {}
  void arrive_and_wait() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned gen = generation;
    if (++waiting == n) {
      waiting = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != generation; });
    }
  }
  // `missing` participants will never arrive (their threads could not be started): the others must not wait for them
  void abandon(unsigned missing) {
    std::unique_lock<std::mutex> lk(m);
    n -= missing < n ? missing : n;
    if (n && waiting >= n) {
      waiting = 0;
      generation++;
      cv.notify_all();
    }
  }
};

struct DeviceSet {
  std::vector<int> ids;                // HIP device of every rank
  std::vector<zkpoa_context*> ctx;     // one context per rank (ranks that share a device have their own streams)
  std::vector<int> lock_fds;           // automatic selection: the GPUs' lock files, held until the process ends
  bool automatic = false;
  bool peer_ok = true;                 // every rank can address every other rank's memory (peer access enabled)
  DeviceSet() = default;
  DeviceSet(const DeviceSet&) = delete;
  DeviceSet& operator=(const DeviceSet&) = delete;
  // A set that never became the process's set (context creation or the peer-access loop failed) gives everything
  // back: a retry in the same process (the resident server's next request) opens the lock files anew, and a lock this
  // process still held through a leaked descriptor would make its own blocking flock wait for ever.
  ~DeviceSet() {
    for (auto* c : ctx)
      if (c) zkpoa_context_destroy(c);
    for (int fd : lock_fds)
      if (fd >= 0) close(fd);
  }
};

// env integer with a range, or `dflt` when unset / empty; a malformed or out-of-range value is an input error
long env_long(const char* name, long lo, long hi, long dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  errno = 0;
  const long v = strtol(e, &end, 10);
  if (errno || end == e || *end || v < lo || v > hi)
    throw ProverError(PROVER_ERROR, std::string(name) + "='" + e + "' is not an integer in [" + std::to_string(lo) + ", " +
                                    std::to_string(hi) + "]");
  return v;
}

std::mutex g_devset_mutex;
DeviceSet* g_devset = nullptr;

bool devices_ready() {
  std::lock_guard<std::mutex> lk(g_devset_mutex);
  return g_devset != nullptr;
}

std::string gpu_lock_dir() { return "/tmp/zkpoa-" + std::to_string((long)getuid()); }

// The lock directory lives in /tmp, so it must be a real directory of this user that nobody else can write to (not a
// symlink, not something another user created first); otherwise no lock is taken and the caller falls back to its
// pid-based choice. Lock files are opened without following symlinks and are never written to.
bool gpu_lock_dir_ok(const std::string& dir) {
  if (mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return false;
  struct stat sb;
  return lstat(dir.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode) && sb.st_uid == getuid() && (sb.st_mode & 0022) == 0;
}

int try_lock_gpu(int dev, bool block) {
  const std::string dir = gpu_lock_dir();
  if (!gpu_lock_dir_ok(dir)) return -1;
  const std::string path = dir + "/gpu" + std::to_string(dev) + ".lock";
  int fd = open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
  if (fd < 0) return -1;
  if (flock(fd, LOCK_EX | LOCK_NB) == 0) return fd;
  if (block) {
    // Wait for the holder, but not for ever: a wedged holder must not hang every later prover. After
    // ZKPOA_GPU_LOCK_WAIT_S seconds (default 1800: a queue of layer-three proofs is minutes, not hours) the wait is
    // reported and this process shares the GPU without the lock.
    const long wait_s = env_long("ZKPOA_GPU_LOCK_WAIT_S", 0, 86400, 1800);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      if (flock(fd, LOCK_EX | LOCK_NB) == 0) return fd;
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= (double)wait_s) break;
      usleep(20000);
    }
    fprintf(stderr, "zkpoa: GPU %d: %s is still held after %ld s (another prover of this user; `fuser %s` names it); "
                    "proceeding without the lock\n", dev, path.c_str(), wait_s, path.c_str());
  }
  close(fd);
  return -1;
}

// The automatic choice, free of HIP and of the file system so that it can be tested without GPUs
// (zkpoa_test_auto_pick_devices): a key with domain >= 2^min_power wants every GPU (at most 8), a smaller one wants one;
// GPUs are tried in order -- for a single-GPU proof starting at pid mod count, so that concurrent batch jobs start at
// different GPUs -- and taken when `lock(d, false)` succeeds; nothing free: wait for GPU pid mod count (`lock(d, true)`);
// the list is cut to a power of two (`release(keep)` gives the surplus locks back, last taken first).
template <class Lock, class Release>
std::vector<int> auto_pick_devices(int count, uint32_t power, uint32_t min_power, unsigned long pid, Lock lock, Release release) {
  std::vector<int> ids;
  const int want = power >= min_power ? (count > 8 ? 8 : count) : 1;
  const int first = want == 1 ? (int)(pid % (unsigned long)count) : 0;
  for (int k = 0; k < count && (int)ids.size() < want; k++) {
    const int d = (first + k) % count;
    if (lock(d, false)) ids.push_back(d);
  }
  if (ids.empty()) {
    const int d = (int)(pid % (unsigned long)count);
    (void)lock(d, true);
    ids.push_back(d);
  }
  size_t keep = 1;
  while (keep * 2 <= ids.size()) keep *= 2;   // the split chain wants 2, 4 or 8 ranks
  ids.resize(keep);
  release(keep);
  return ids;
}

// the device list for this process; `power` = log2 of the first key's domain (automatic selection only)
void select_devices(DeviceSet& ds, uint32_t power) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    throw HipError("zkpoa: no HIP device visible (the MSM/NTT path is HIP-only; there is no CPU fallback)");
  if (const char* e = getenv("ZKPOA_DEVICES")) {
    if (*e && strcmp(e, "auto") != 0) {
      for (const char* p = e; *p;) {
        char* end = nullptr;
        long v = strtol(p, &end, 10);
        if (end == p || v < 0 || v >= count)
          throw ProverError(PROVER_ERROR, std::string("ZKPOA_DEVICES: '") + e + "' is not a comma-separated list of device "
                                          "indices below " + std::to_string(count));
        ds.ids.push_back((int)v);
        p = end;
        if (*p == ',') p++;
        else if (*p) throw ProverError(PROVER_ERROR, std::string("ZKPOA_DEVICES: unexpected character in '") + e + "'");
      }
      if (ds.ids.empty() || ds.ids.size() > 8) throw ProverError(PROVER_ERROR, "ZKPOA_DEVICES: between 1 and 8 devices");
      return;
    }
  }
  if (const char* e = getenv("ZKPOA_DEVICE")) {
    if (*e) {
      ds.ids.push_back((int)env_long("ZKPOA_DEVICE", 0, count - 1, 0));
      return;
    }
  }
  if (count == 1) {
    ds.ids.push_back(0);
    return;
  }
  ds.automatic = true;
  const uint32_t min_power = (uint32_t)env_long("ZKPOA_MULTI_MIN_POWER", 0, 64, 24);
  std::vector<int> fds;
  ds.ids = auto_pick_devices(count, power, min_power, (unsigned long)getpid(), [&](int d, bool block) {
    int fd = try_lock_gpu(d, block);
    if (fd >= 0) fds.push_back(fd);
    return fd >= 0 || block;   // a blocking attempt that could not even open its lock file still proceeds on that GPU
  }, [&](size_t keep) {
    while (fds.size() > keep) {
      close(fds.back());
      fds.pop_back();
    }
  });
  ds.lock_fds = fds;
}

// Contexts of the process, created once (first prove decides the device list). Ranks come up in parallel: a context
// costs 60-200 ms (HIP runtime, first stream).
// On failure nullptr with err / *code set: PROVER_ERROR for an input error (a malformed ZKPOA_* variable),
// PROVER_ERROR_RUNTIME when the HIP runtime failed (no device, context creation, peer access) -- the caller reports it
// as such, so a resident server goes away and the fault is logged instead of being retried on every request.
DeviceSet* process_devices(uint32_t power, std::string& err, int* code) {
  std::lock_guard<std::mutex> lk(g_devset_mutex);
  if (g_devset) return g_devset;
  if (code) *code = PROVER_ERROR;
  std::unique_ptr<DeviceSet> ds(new DeviceSet());   // ~DeviceSet releases locks and contexts on every failure path
  try {
    auto t0 = std::chrono::steady_clock::now();
    select_devices(*ds, power);
    const size_t G = ds->ids.size();
    ds->ctx.assign(G, nullptr);
    std::vector<std::string> errs(G);
    std::vector<std::thread> th;
    for (size_t g = 0; g < G; g++)
      th.emplace_back([&, g] {
        char msg[512] = {0};
        if (zkpoa_context_create(ds->ids[g], &ds->ctx[g], msg, sizeof(msg)) != PROVER_OK) errs[g] = msg[0] ? msg : "context creation failed";
      });
    for (auto& t : th) t.join();
    for (size_t g = 0; g < G; g++)
      if (!errs[g].empty()) throw HipError(errs[g]);
    // peer access between distinct devices: the exchanges then go GPU to GPU over xGMI instead of through the host
    for (size_t g = 0; g < G; g++)
      for (size_t h = 0; h < G; h++) {
        if (ds->ids[g] == ds->ids[h]) continue;
        int can = 0;
        ZK_HIP(hipSetDevice(ds->ids[g]));
        if (hipDeviceCanAccessPeer(&can, ds->ids[g], ds->ids[h]) == hipSuccess && can) {
          hipError_t pe = hipDeviceEnablePeerAccess(ds->ids[h], 0);
          if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) ds->peer_ok = false;
          if (pe != hipSuccess) (void)hipGetLastError();   // already enabled (devices listed twice), or not: copies still work
        } else {
          ds->peer_ok = false;   // no direct path between this pair: the exchanges go through hipMemcpyPeerAsync
        }
      }
    if (req_getenv("ZKPOA_VERBOSE")) {
      std::string list;
      for (size_t g = 0; g < G; g++) list += (g ? "," : "") + std::to_string(ds->ids[g]);
      fprintf(stderr, "zkpoa: %zu rank(s) on HIP device(s) %s%s, contexts ready in %.1f ms\n", G, list.c_str(),
              ds->automatic ? " (picked by lock file)" : "",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
  } catch (const ProverError& e) {
    err = e.what();
    if (code) *code = e.code;
    return nullptr;
  } catch (const std::exception& e) {   // HipError, bad_alloc, a thread that could not be started
    err = e.what();
    if (code) *code = PROVER_ERROR_RUNTIME;
    return nullptr;
  }
  g_devset = ds.release();
  g_ctx = g_devset->ctx[0];
  return g_devset;
}

// ---- a key sharded over the ranks of the process ----------------------------------------------------------------------
struct MultiKey {
  std::vector<zkpoa_zkey*> shards;          // shard g lives on context g
  bool split = false;                       // H-scalar chain split (G in {2, 4, 8}, G^2 <= domain) or replicated
  std::vector<void*> xa, xb1, xb2;          // per rank: send buffer, receive buffers of the two exchanges
  std::vector<hipEvent_t> ev1, ev2;         // per rank: "my pushes of exchange 1 / 2 are enqueued up to here"
  std::vector<hipEvent_t> evw;              // per rank: "my slice of the witness is on its way to every peer"
  uint64_t xbytes = 0;                      // bytes of one exchange buffer: 3 * (domain / G) * 32
  uint64_t proofs_done = 0, table_bytes = 0;
  bool tables_tried = false;                // the shards' fixed-base tables have been built (or the attempt has been made)
  double load_ms = 0;
};

void multi_key_release(DeviceSet* ds, MultiKey* mk) {
  if (!mk) return;
  for (size_t g = 0; g < mk->shards.size(); g++) {
    (void)hipSetDevice(ds->ids[g]);
    (void)hipDeviceSynchronize();
    for (std::vector<void*>* v : {&mk->xa, &mk->xb1, &mk->xb2})
      if (g < v->size() && (*v)[g]) (void)hipFree((*v)[g]);
    if (g < mk->ev1.size() && mk->ev1[g]) (void)hipEventDestroy(mk->ev1[g]);
    if (g < mk->ev2.size() && mk->ev2[g]) (void)hipEventDestroy(mk->ev2[g]);
    if (g < mk->evw.size() && mk->evw[g]) (void)hipEventDestroy(mk->evw[g]);
    if (mk->shards[g]) {
      mk->shards[g]->release();
      delete mk->shards[g];
    }
  }
  delete mk;
}

// run fn(g) on one thread per rank; the first exception (by rank) is rethrown after all have joined
template <class Fn>
void for_each_rank(DeviceSet* ds, Fn fn) {
  const size_t G = ds->ids.size();
  std::vector<std::exception_ptr> errs(G);
  std::vector<std::thread> th;
  for (size_t g = 0; g < G; g++)
    th.emplace_back([&, g] {
      try {
        ZK_HIP(hipSetDevice(ds->ids[g]));
        fn(g);
      } catch (...) {
        errs[g] = std::current_exception();
      }
    });
  for (auto& t : th) t.join();
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
}

// Block size of the block-cyclic sections: 2^16 items, less for small keys so that every rank still gets at least
// eight blocks (and the test-size keys exercise the same path). ZKPOA_SHARD_BLOCK_LOG overrides; 0 = contiguous ranges.
uint32_t multi_block_log_default(uint64_t n_vars, size_t G) {
  uint32_t L = 16;
  while (L > 4 && (n_vars >> L) < 8 * (uint64_t)G) L--;
  return L;
}
uint32_t multi_block_log(uint64_t n_vars, size_t G) {
  if (const char* e = getenv("ZKPOA_SHARD_BLOCK_LOG"))
    if (*e) {
      const long v = env_long("ZKPOA_SHARD_BLOCK_LOG", 0, 24, 0);
      if (v != 0 && v < 4) throw ProverError(PROVER_ERROR, "ZKPOA_SHARD_BLOCK_LOG: 0 (contiguous ranges) or 4 .. 24");
      return (uint32_t)v;
    }
  return multi_block_log_default(n_vars, G);
}

MultiKey* multi_key_load(DeviceSet* ds, const uint8_t* buf, uint64_t size) {
  const size_t G = ds->ids.size();
  std::unique_ptr<MultiKey> mk(new MultiKey());
  mk->shards.assign(G, nullptr);
  auto t0 = std::chrono::steady_clock::now();
  try {
    ZkeySections zs;
    std::unique_ptr<zkpoa_zkey> hdr = zkey_parse(buf, size, zs);   // validates the container once, before G uploads start
    mk->split = (G == 2 || G == 4 || G == 8) && (uint64_t)hdr->domain >= (uint64_t)G * G;
    const uint32_t bc = multi_block_log(hdr->nVars, G);
    for_each_rank(ds, [&](size_t g) { mk->shards[g] = zkey_load_impl(ds->ctx[g], buf, size, g, G, mk->split, bc); });
    mk->evw.assign(G, nullptr);
    for (size_t g = 0; g < G; g++) {
      ZK_HIP(hipSetDevice(ds->ids[g]));
      ZK_HIP(hipEventCreateWithFlags(&mk->evw[g], hipEventDisableTiming));
    }
    if (mk->split) {
      mk->xbytes = (uint64_t)3 * (hdr->domain / G) * 32;
      mk->xa.assign(G, nullptr);
      mk->xb1.assign(G, nullptr);
      mk->xb2.assign(G, nullptr);
      mk->ev1.assign(G, nullptr);
      mk->ev2.assign(G, nullptr);
      for (size_t g = 0; g < G; g++) {
        ZK_HIP(hipSetDevice(ds->ids[g]));
        ZK_HIP(hipMalloc(&mk->xa[g], mk->xbytes));
        ZK_HIP(hipMalloc(&mk->xb1[g], mk->xbytes));
        ZK_HIP(hipMalloc(&mk->xb2[g], mk->xbytes));
        ZK_HIP(hipEventCreateWithFlags(&mk->ev1[g], hipEventDisableTiming));
        ZK_HIP(hipEventCreateWithFlags(&mk->ev2[g], hipEventDisableTiming));
      }
    }
  } catch (...) {
    multi_key_release(ds, mk.release());
    throw;
  }
  mk->load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return mk.release();
}

// One exchange of the split chain, rank g's half: push chunk h of `src` (what rank h needs from g) into slot g of rank
// h's receive buffer, all on g's lane-0 stream behind the stage that produced `src`; then mark the stream.
// ZKPOA_EXCHANGE=copy: hipMemcpyPeerAsync exchanges, whole witness per rank -- also switched on for the rest of the
// process when a proof made with the peer-store exchanges fails its self-check (multi_prove_to_json)
inline std::atomic<int>& multi_copies_state() {
  static std::atomic<int> v{[] {
    const char* e = getenv("ZKPOA_EXCHANGE");
    return e && !strcmp(e, "copy") ? 1 : 0;
  }()};
  return v;
}
inline bool multi_force_copies() { return multi_copies_state().load() != 0; }
// tests: ZKPOA_TEST_STALE_EXCHANGE=<n> -- in the n-th multi-rank proof of the process rank 0 does not push its first
// exchange (its peers keep what the previous proof left there): what a missing release / acquire would look like
inline long multi_test_stale_at() {
  static const long v = [] {
    const char* e = getenv("ZKPOA_TEST_STALE_EXCHANGE");
    return e && *e ? strtol(e, nullptr, 10) : 0L;
  }();
  return v;
}
inline std::atomic<long>& multi_proof_counter() {
  static std::atomic<long> v{0};
  return v;
}

void multi_push(DeviceSet* ds, MultiKey* mk, size_t g, const void* src, std::vector<void*>& dst, hipEvent_t done) {
  const size_t G = ds->ids.size();
  const uint64_t chunk = mk->xbytes / G;
  hipStream_t st = ds->ctx[g]->dev.lanes[0].stream;
  if (ds->peer_ok && !multi_force_copies() && chunk % 16 == 0) {
    if (g == 0 && &dst == &mk->xb1 && multi_test_stale_at() && multi_proof_counter().load() == multi_test_stale_at()) {
      ZK_HIP(hipEventRecord(done, st));   // (test hook: this push is withheld)
      return;
    }
    // one kernel writes all G chunks into the peers' receive buffers: all links busy at once (abc.hip.h)
    XchgDst d;
    for (size_t h = 0; h < 8; h++) d.p[h] = h < G ? reinterpret_cast<char*>(dst[h]) + g * chunk : nullptr;
    const uint64_t chunk16 = chunk / 16, total = chunk16 * G;
    hipLaunchKernelGGL(xchg_push_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const uint4*>(src), d, chunk16, (uint32_t)G, (uint32_t)g);
    ZK_HIP(hipGetLastError());
  } else {
    for (size_t k = 0; k < G; k++) {
      const size_t h = (g + k) % G;   // start with the own slot, then round the ring: no two ranks aim at one peer at once
      const char* s = reinterpret_cast<const char*>(src) + h * chunk;
      char* d = reinterpret_cast<char*>(dst[h]) + g * chunk;
      if (ds->ids[g] == ds->ids[h]) ZK_HIP(hipMemcpyAsync(d, s, chunk, hipMemcpyDeviceToDevice, st));
      else ZK_HIP(hipMemcpyPeerAsync(d, ds->ids[h], s, ds->ids[g], chunk, st));
    }
  }
  ZK_HIP(hipEventRecord(done, st));
}

// The proof's five partial sums over all ranks: parts_sum = A(64) B1(64) B2(128) C(64) H(64).
void multi_prove_partials(DeviceSet* ds, MultiKey* mk, const WtnsView& w, uint8_t parts_sum[384]) {
  const size_t G = ds->ids.size();
  std::vector<std::array<uint8_t, 384>> parts(G);
  RankBarrier bar((unsigned)G);
  std::atomic<bool> failed{false};
  std::vector<std::exception_ptr> errs(G);
  std::vector<std::thread> th;
  th.reserve(G);
  auto rank_main = [&](size_t g) {
      // Phases separated by barriers; a rank that fails keeps arriving at the barriers so that nobody waits for ever.
      auto phase = [&](const std::function<void()>& fn) {
        if (failed.load()) return;
        try {
          fn();
        } catch (...) {
          errs[g] = std::current_exception();
          failed.store(true);
        }
      };
      zkpoa_context* ctx = ds->ctx[g];
      zkpoa_zkey* zk = mk->shards[g];
      hipStream_t st = nullptr;
      // tests: ZKPOA_TEST_FAIL_RANK=<g>[:<phase>] makes rank g fail at the start of that phase (1 = before the first
      // barrier, 2 = between the exchanges, 3 = before its MSMs): the other ranks must come back, not wait for ever
      // (read once per process: no getenv on the proving path)
      auto test_fail = [&](int at) {
        static const std::pair<long, long> hook = [] {
          const char* e = getenv("ZKPOA_TEST_FAIL_RANK");
          if (!e || !*e) return std::pair<long, long>(-1, 0);
          const char* colon = strchr(e, ':');
          return std::pair<long, long>(strtol(e, nullptr, 10), colon ? strtol(colon + 1, nullptr, 10) : 1);
        }();
        if (hook.first == (long)g && hook.second == at)
          throw ProverError(PROVER_ERROR, "test: rank " + std::to_string(g) + " failed in phase " + std::to_string(at));
      };
      // The witness is needed whole on every rank. Each rank uploads 1 / G of it over its own PCIe link and stores that
      // slice into the other ranks' buffers over xGMI (61 M wires = 2 GB: 8 x 2 GB through the host would cost twice
      // the proof); ranks without a direct path to each other upload it whole.
      const bool sliced = ds->peer_ok && G > 1 && !multi_force_copies();
      phase([&] {
        ZK_HIP(hipSetDevice(ds->ids[g]));
        st = ctx->dev.lanes[0].stream;
        test_fail(1);
        if (sliced) {
          const uint64_t lo = (uint64_t)w.n * g / G, hi = (uint64_t)w.n * (g + 1) / G;
          char* mine = reinterpret_cast<char*>(zk->d_witness) + lo * 32;
          if (hi > lo) {
            w.upload(ctx, mine, lo, hi - lo, st);
            XchgDst d;
            for (size_t h = 0; h < 8; h++) d.p[h] = h < G ? reinterpret_cast<char*>(mk->shards[h]->d_witness) + lo * 32 : nullptr;
            const uint64_t n16 = (hi - lo) * 2, total = n16 * (G - 1);
            hipLaunchKernelGGL(xchg_bcast_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st,
                               reinterpret_cast<const uint4*>(mine), d, n16, (uint32_t)G, (uint32_t)g);
            ZK_HIP(hipGetLastError());
          }
          ZK_HIP(hipEventRecord(mk->evw[g], st));
        } else {
          w.upload(ctx, zk->d_witness, 0, w.n, st);   // replicated
        }
      });
      if (sliced) {
        bar.arrive_and_wait();   // every slice's event has been recorded
        phase([&] {
          for (size_t h = 0; h < G; h++) ZK_HIP(hipStreamWaitEvent(st, mk->evw[h], 0));
          // the witness MSMs run on other lanes: they wait for this point of lane 0 (prove_partials, ev_witness)
          if (!ctx->ev_witness) ZK_HIP(hipEventCreateWithFlags(&ctx->ev_witness, hipEventDisableTiming));
          ZK_HIP(hipEventRecord(ctx->ev_witness, st));
          ctx->ev_witness_set = true;
        });
      }
      phase([&] {
        zk->h_ready = false;
        if (mk->split) {
          split_stage1(ctx, zk, mk->xa[g]);
          multi_push(ds, mk, g, mk->xa[g], mk->xb1, mk->ev1[g]);
        }
      });
      if (mk->split) {
        bar.arrive_and_wait();   // every rank's event has been recorded: waiting on an unrecorded event is a no-op
        phase([&] {
          test_fail(2);
          for (size_t h = 0; h < G; h++) ZK_HIP(hipStreamWaitEvent(st, mk->ev1[h], 0));
          split_stage2(ctx, zk, mk->xb1[g], mk->xa[g]);   // xa[g] is free: this stream's own pushes precede this stage
          multi_push(ds, mk, g, mk->xa[g], mk->xb2, mk->ev2[g]);
        });
        bar.arrive_and_wait();
        phase([&] {
          for (size_t h = 0; h < G; h++) ZK_HIP(hipStreamWaitEvent(st, mk->ev2[h], 0));
          split_stage3(ctx, zk, mk->xb2[g]);
        });
      }
      // the witness MSMs start at once on their own lanes and overlap the chain still in flight on lane 0
      phase([&] {
        test_fail(3);
        prove_partials(ctx, zk, parts[g].data());
      });
      // a rank must not start the next proof's pushes into a peer that still reads this proof's buffers
      phase([&] { ZK_HIP(hipStreamSynchronize(st)); });
  };
  std::exception_ptr spawn_err;
  for (size_t g = 0; g < G; g++) {
    try {
      th.emplace_back(rank_main, g);
    } catch (...) {   // out of threads: the ranks that did start must not wait at the barriers for the ones that did not
      spawn_err = std::current_exception();
      failed.store(true);
      bar.abandon((unsigned)(G - g));
      break;
    }
  }
  for (auto& t : th) t.join();
  if (failed.load()) {
    // A failed proof skipped its last phase: peers' pushes and this proof's lane work may still be in flight, and a
    // rank may have announced a witness event nobody consumed. Drain every rank (best effort) and reset that state
    // before the error leaves, so that the next proof on this key does not depend on what this one left behind.
    for (size_t g = 0; g < G; g++) {
      (void)hipSetDevice(ds->ids[g]);
      (void)hipDeviceSynchronize();
      ds->ctx[g]->ev_witness_set = false;
      mk->shards[g]->h_ready = false;
    }
    (void)hipSetDevice(ds->ids[0]);
  }
  if (spawn_err) std::rethrow_exception(spawn_err);
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
  // "all-reduce of the partial sums": five tiny host-side group sums (G points each)
  const struct { size_t off, len; bool g2; } kParts[5] = {{0, 64, false}, {64, 64, false}, {128, 128, true}, {256, 64, false}, {320, 64, false}};
  for (const auto& p : kParts) {
    std::vector<uint8_t> col(G * p.len);
    for (size_t g = 0; g < G; g++) memcpy(&col[g * p.len], parts[g].data() + p.off, p.len);
    int rc = p.g2 ? zkpoa_g2_sum(col.data(), G, parts_sum + p.off) : zkpoa_g1_sum(col.data(), G, parts_sum + p.off);
    if (rc != PROVER_OK) throw ProverError(PROVER_ERROR, "multi-GPU prove: a partial result is not a curve point");
  }
}

// fixed-base tables of every shard (1 / G of the memory per GPU: on 8 GPUs all five sections of a 2^26 key fit)
void multi_precompute(DeviceSet* ds, MultiKey* mk) {
  std::vector<uint64_t> used(ds->ids.size(), 0);
  for_each_rank(ds, [&](size_t g) {
    try {
      used[g] = zkey_precompute(ds->ctx[g], mk->shards[g], 0);
    } catch (const HipError&) {   // out of HBM: the classic form keeps working
      mk->shards[g]->release_tables();
      (void)hipGetLastError();
    }
  });
  mk->table_bytes = 0;
  for (uint64_t u : used) mk->table_bytes += u;
}

// witness -> proof JSON on a loaded MultiKey (the multi-GPU twin of prove_to_json)
int multi_prove_to_json(DeviceSet* ds, MultiKey* mk, const WtnsSrc& wsrc, char* proof_buffer,
                        unsigned long* proof_size, char* public_buffer, unsigned long* public_size, char* error_msg,
                        unsigned long error_msg_maxsize, uint64_t zkey_size, bool cache_hit) {
  zkpoa_zkey* z0 = mk->shards[0];
  zkpoa_context* c0 = ds->ctx[0];
  WtnsView w = parse_wtns(wsrc);
  if (w.n != z0->nVars)
    throw ProverError(PROVER_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " + std::to_string(z0->nVars) +
                                                         ", witness: " + std::to_string(w.n));
  uint8_t rb[32], sb[32], parts[384], header[448], pts[256];
  const uint8_t *rp = nullptr, *sp = nullptr;
  env_blinding(rb, sb, rp, sp);
  auto t0 = std::chrono::steady_clock::now();
  multi_proof_counter()++;
  const bool kernel_exchange = ds->ids.size() > 1 && ds->peer_ok && !multi_force_copies();
  multi_prove_partials(ds, mk, w, parts);
  zkey_header_bytes(z0, header);
  prove_assemble(header, parts, rp, sp, pts);
  c0->ms[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::vector<uint8_t> pub_store;
  const uint8_t* pubs = w.publics(z0->nPublic, pub_store);
  // The peer-store exchanges rest on a memory-visibility rule that no run on a multi-GPU node has confirmed yet (DESIGN
  // section 6), so the pairing check that normally guards a key's first proof guards its first three here -- a stale
  // receive buffer can only show from the second proof on -- and a proof that fails it is not an error yet: it is
  // repeated with hipMemcpyPeerAsync exchanges (DMA, ordered by the runtime), which then stay on for this process.
  try {
    selfcheck(c0, z0, pts, pubs, kernel_exchange ? 3 : 1);
  } catch (const SelfCheckFailed&) {
    if (!kernel_exchange) throw;
    multi_copies_state().store(1);
    fprintf(stderr, "zkpoa: WARNING: a proof over %zu ranks failed its self-check with the peer-store exchanges; repeating it "
                    "with hipMemcpyPeerAsync exchanges, which stay on for the rest of this process (ZKPOA_EXCHANGE=copy makes "
                    "them the default; DESIGN.md section 6 names the suspects)\n", ds->ids.size());
    multi_prove_partials(ds, mk, w, parts);
    prove_assemble(header, parts, rp, sp, pts);
    c0->ms[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    selfcheck(c0, z0, pts, pubs, ~0ull);   // this one must verify
  }
  if (req_getenv("ZKPOA_VERBOSE"))
    fprintf(stderr, "zkpoa: one proof over %zu ranks: H-scalar chain %s, sections 5-8 %s, %.2f GB of fixed-base tables; "
                    "prove %.2f ms\n", ds->ids.size(), mk->split ? "split (2 peer-to-peer exchanges)" : "replicated",
            z0->bc_log ? "block-cyclic" : "contiguous ranges", mk->table_bytes / 1e9, c0->ms[5]);
  return emit_outputs(c0, z0, pts, pubs, proof_buffer, proof_size, public_buffer, public_size, error_msg,
                      error_msg_maxsize, mk->load_ms, zkey_size, cache_hit ? "cached," : "sharded load");
}
