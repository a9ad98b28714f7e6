// `prover` -- drop-in for the executable the reference execs at scripts/g16_prove.sh:248-252:
//     prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>
// (rapidsnark's argv; g16_prove.sh:195-199 insists the file is named exactly `prover`).
// Exit status 0 on success; non-zero with a message on stderr otherwise, so the reference's
// `set -eE` / ERR trap (scripts/lib/error_handling.sh:14-41) fires. Outputs are written to a
// temporary name and renamed, so a failed run never leaves a partial proof.json.
//
// Server mode (opt-in, ZKPOA_SERVER=1 or =/path/to.sock): the reference proves layer one and layer two once
// per batch with the SAME zkey (scripts/full_workflow.sh:497-552), and a one-shot process spends most of its
// life on HIP start-up and on uploading 1-21 GB of key. With ZKPOA_SERVER set, `prover` keeps the same argv and
// exit codes but hands the four paths to a resident prover process over a unix socket (started on first use,
// one per device, exits after ZKPOA_SERVER_IDLE_S seconds without work, default 600); that process keeps the
// keys in HBM (groth16_prover_zkey_file's cache, keyed by inode + size + mtime). `prover --stop-server` ends it.
// The socket lives in a 0700 directory of the calling user and the server checks the peer's uid.
#include "../../include/zkpoa_prover.h"
#include "worker_exit.hpp"

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <poll.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// The library is loaded on demand (dlopen): a client of the resident server never proves itself, and resolving
// libzkpoa_prover.so with the HIP runtime behind it costs ~11 ms of every call (measured r04: 28 ms per layer-one
// proof through the server, of which 15 in the server). Looked for next to this executable, then on the usual path.
typedef int (*prover_files_fn)(const char*, const char*, char*, unsigned long*, char*, unsigned long*, char*, unsigned long);
typedef int (*thread_options_fn)(const char*, const char*, const char*, int);
static thread_options_fn g_set_thread_options = nullptr;   // per-request r, s, JSON style, verbosity of a server thread
typedef int (*idle_work_fn)(void);
static std::atomic<idle_work_fn> g_idle_work{nullptr};     // one step of the library's background work (fixed-base tables)
static std::mutex g_load_mutex;
static prover_files_fn load_prover(std::string& message) {
  static prover_files_fn fn = nullptr;
  std::lock_guard<std::mutex> lk(g_load_mutex);
  if (fn) return fn;
  std::string tried;
  char exe[PATH_MAX];
  ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
  void* h = nullptr;
  if (n > 0) {
    exe[n] = 0;
    std::string dir(exe);
    dir = dir.substr(0, dir.rfind('/'));
    h = dlopen((dir + "/libzkpoa_prover.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) tried = dlerror();
  }
  if (!h) h = dlopen("libzkpoa_prover.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    message = std::string("Error: cannot load libzkpoa_prover.so (") + (tried.empty() ? dlerror() : tried.c_str()) +
              "); the prover has no CPU fallback";
    return nullptr;
  }
  fn = reinterpret_cast<prover_files_fn>(dlsym(h, "zkpoa_groth16_prover_files"));
  if (!fn) message = "Error: libzkpoa_prover.so does not export zkpoa_groth16_prover_files";
  g_set_thread_options = reinterpret_cast<thread_options_fn>(dlsym(h, "zkpoa_set_thread_options"));
  g_idle_work.store(reinterpret_cast<idle_work_fn>(dlsym(h, "zkpoa_idle_work")));
  return fn;
}

static bool write_atomic(const char* path, const char* text) {
  std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  size_t len = strlen(text);
  bool ok = fwrite(text, 1, len, f) == len;
  ok = (fclose(f) == 0) && ok;
  if (!ok || rename(tmp.c_str(), path) != 0) {
    unlink(tmp.c_str());
    return false;
  }
  return true;
}

static double now_ms() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

// ---- one proof in this process: (zkey path, wtns path) -> files; message = what goes to stderr ------------
// *runtime_failure (optional) <- the failure came from the HIP runtime (sticky fault, out of memory, lost device),
// not from the inputs: the process's GPU context cannot be trusted any more.
static int prove_files(const char* zkey, const char* wtns_path, const char* proof_path, const char* public_path,
                       std::string& message, bool* runtime_failure = nullptr) {
  prover_files_fn prover = load_prover(message);
  if (!prover) return EXIT_FAILURE;
  {   // an unreadable witness is reported before the GPU is touched, in the words the reference's tests look for
    int fd = open(wtns_path, O_RDONLY | O_CLOEXEC);
    struct stat sb;
    const bool ok = fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    if (fd >= 0) close(fd);
    if (!ok) {
      message = std::string("Error: cannot read witness file ") + wtns_path;
      return EXIT_FAILURE;
    }
  }
  unsigned long proof_size = 1 << 12, public_size = 1 << 16;
  std::vector<char> proof(proof_size), pub(public_size);
  char err[1024] = {0};
  int rc = prover(zkey, wtns_path, proof.data(), &proof_size, pub.data(), &public_size, err, sizeof(err));
  if (rc == PROVER_ERROR_SHORT_BUFFER) {
    proof.resize(proof_size);
    pub.resize(public_size);
    rc = prover(zkey, wtns_path, proof.data(), &proof_size, pub.data(), &public_size, err, sizeof(err));
  }
  if (rc != PROVER_OK) {
    message = std::string("Error: ") + err;
    // the library tells HIP-runtime failures (sticky fault, lost device, out of memory) from input errors by code
    if (runtime_failure) *runtime_failure = rc == PROVER_ERROR_RUNTIME;
    return EXIT_FAILURE;
  }
  if (!write_atomic(proof_path, proof.data()) || !write_atomic(public_path, pub.data())) {
    message = std::string("Error: cannot write ") + proof_path + " / " + public_path;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}

// ---- the fault log ---------------------------------------------------------------------------------------
// A GPU fault must never pass unnoticed because a fallback produced the proof anyway (r02: a memory access fault inside
// the resident server was only found by counting lines of its log). Whenever this process proves in-process BECAUSE the
// server failed -- it reported a HIP runtime failure, died, or did not answer -- one line goes to a persistent log
// /tmp/zkpoa-<uid>/faults.log (time, pid, what happened, the key) and its path to stderr. ZKPOA_STRICT=1 turns the
// fallback into a non-zero exit instead (the workflow's ERR trap fires and somebody looks at the machine).
static std::string abs_path(const char* p);
static std::string fault_log_path() {
  return "/tmp/zkpoa-" + std::to_string((long)getuid()) + "/faults.log";
}
static void log_fault(const char* what, const std::string& detail, const char* zkey) {
  const std::string path = fault_log_path();
  const std::string dir = path.substr(0, path.rfind('/'));
  (void)mkdir(dir.c_str(), 0700);
  {   // /tmp is shared: only a real directory of this user that others cannot write to, and never through a symlink
    struct stat sb;
    if (lstat(dir.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != getuid() || (sb.st_mode & 0022) != 0) {
      fprintf(stderr, "zkpoa: GPU-side failure (%s: %s); %s is not a private directory of this user, nothing logged\n", what,
              detail.c_str(), dir.c_str());
      return;
    }
  }
  char when[64] = {0};
  time_t now = time(nullptr);
  struct tm tmv;
  gmtime_r(&now, &tmv);
  strftime(when, sizeof(when), "%Y-%m-%dT%H:%M:%SZ", &tmv);
  std::string line = std::string(when) + " pid " + std::to_string((long)getpid()) + " " + what + ": " + detail +
                     " | zkey " + (zkey ? abs_path(zkey) : std::string("-")) + "\n";
  for (char& c : line)
    if ((c == '\n' || c == '\r') && &c != &line[line.size() - 1]) c = ' ';
  int fd = open(path.c_str(), O_CREAT | O_WRONLY | O_APPEND | O_CLOEXEC | O_NOFOLLOW, 0600);
  if (fd >= 0) {
    (void)!write(fd, line.data(), line.size());   // O_APPEND: one write per line, concurrent provers do not interleave
    close(fd);
  }
  fprintf(stderr, "zkpoa: GPU-side failure recorded in %s\n", path.c_str());
}
static bool strict_mode() {
  const char* e = getenv("ZKPOA_STRICT");
  return e && *e && strcmp(e, "0") != 0;
}
enum { kExitStrict = 5 };   // ZKPOA_STRICT=1: the server failed and the fallback was refused

// ---- server mode ---------------------------------------------------------------------------------------
// Wire format, both directions: u32 length + payload. Request payload = NUL-separated fields
//   "prove" zkey wtns proof public R S JSON VERBOSE   (absolute paths; option fields may be empty)
//   "stop"
// Reply payload = one status byte followed by the message for stderr: '0' + exit code, or 'R' = the server hit a
// GPU runtime failure and is going away: the client proves in its own process (fresh HIP context) instead.
static bool send_all(int fd, const void* buf, size_t len) {
  const char* p = static_cast<const char*>(buf);
  while (len) {
    ssize_t n = send(fd, p, len, MSG_NOSIGNAL);
    if (n <= 0) {
      if (n < 0 && errno == EINTR) continue;
      return false;
    }
    p += n;
    len -= (size_t)n;
  }
  return true;
}
static bool recv_all(int fd, void* buf, size_t len) {
  char* p = static_cast<char*>(buf);
  while (len) {
    ssize_t n = recv(fd, p, len, 0);
    if (n <= 0) {
      if (n < 0 && errno == EINTR) continue;
      return false;
    }
    p += n;
    len -= (size_t)n;
  }
  return true;
}
static bool send_msg(int fd, const std::string& m) {
  uint32_t len = (uint32_t)m.size();
  return send_all(fd, &len, 4) && send_all(fd, m.data(), m.size());
}
static bool recv_msg(int fd, std::string& m) {
  uint32_t len = 0;
  if (!recv_all(fd, &len, 4) || len > (1u << 20)) return false;
  m.resize(len);
  return len == 0 || recv_all(fd, &m[0], len);
}

static std::string server_socket_path() {
  const char* e = getenv("ZKPOA_SERVER");
  if (e && e[0] == '/') return e;
  const char* dev = getenv("ZKPOA_DEVICE");
  return "/tmp/zkpoa-" + std::to_string((long)getuid()) + "/prover-dev" + (dev && *dev ? dev : "0") + ".sock";
}

// the socket's directory must be ours alone (created 0700 if missing)
static bool secure_dir(const std::string& sock) {
  std::string dir = sock.substr(0, sock.rfind('/'));
  if (dir.empty()) return false;
  if (mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return false;
  struct stat sb;
  if (lstat(dir.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != getuid()) return false;
  return (sb.st_mode & 0077) == 0 || getenv("ZKPOA_SERVER")[0] == '/';   // an explicit path is the caller's choice
}

// pid of the server behind a connected socket, taken from the kernel (SO_PEERCRED: the credentials of the process that
// listens), or 0 when the peer is not a process of this user -- the only identity a client ever acts on
static long peer_server_pid(int fd) {
  struct ucred cred;
  socklen_t cl = sizeof(cred);
  if (getsockopt(fd, SOL_SOCKET, SO_PEERCRED, &cred, &cl) != 0 || cred.uid != getuid()) return 0;
  return (long)cred.pid;
}

static int connect_to(const std::string& sock) {
  int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
  if (fd < 0) return -1;
  struct sockaddr_un sa;
  memset(&sa, 0, sizeof(sa));
  sa.sun_family = AF_UNIX;
  if (sock.size() >= sizeof(sa.sun_path)) {
    close(fd);
    return -1;
  }
  strcpy(sa.sun_path, sock.c_str());
  if (connect(fd, reinterpret_cast<struct sockaddr*>(&sa), sizeof(sa)) != 0) {
    close(fd);
    return -1;
  }
  return fd;
}

// The resident prover: serves requests one at a time (the GPU is the shared resource) until idle or told to stop.
enum { kIdleJob = -2 };   // a queue entry that is not a connection: "run the library's idle work"
static int server_main(const std::string& sock) {
  std::string lockp = sock + ".lock";
  int lock = open(lockp.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
  if (lock < 0) return 0;
  {   // Another server owns this socket -- unless it is on its way out (it removes its socket first, see `leave` below,
      // and drops the lock a moment later): with no socket in place, wait for the lock a little instead of giving up.
    bool mine = false;
    for (int i = 0; i < 150 && !mine; i++) {
      mine = flock(lock, LOCK_EX | LOCK_NB) == 0;
      if (!mine) {
        struct stat sb;
        if (stat(sock.c_str(), &sb) == 0) return 0;   // a live server is listening
        usleep(20000);
      }
    }
    if (!mine) return 0;
  }
  // (the lock file carries no pid: a client identifies the server by the kernel's SO_PEERCRED of its own connection)
  unlink(sock.c_str());                                           // stale socket of a dead server
  int ls = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
  struct sockaddr_un sa;
  memset(&sa, 0, sizeof(sa));
  sa.sun_family = AF_UNIX;
  strcpy(sa.sun_path, sock.c_str());
  mode_t old = umask(0077);
  bool ok = ls >= 0 && bind(ls, reinterpret_cast<struct sockaddr*>(&sa), sizeof(sa)) == 0 && listen(ls, 64) == 0;
  umask(old);
  if (!ok) return 1;
  int idle_s = 600;
  if (const char* e = getenv("ZKPOA_SERVER_IDLE_S")) idle_s = atoi(e) > 0 ? atoi(e) : idle_s;
  // Requests are served by a small pool of threads (ZKPOA_SERVER_WORKERS, default 2): the library stages the witness
  // of one request while it proves another (two locks inside zkpoa_groth16_prover_files), so the batch jobs the
  // reference runs in parallel (scripts/full_workflow.sh:552) do not pay the witness upload one after the other.
  double idle_work_ms = 300;
  if (const char* e = getenv("ZKPOA_SERVER_IDLE_WORK_MS")) idle_work_ms = atof(e) >= 0 ? atof(e) : idle_work_ms;
  int workers = 2;
  if (const char* e = getenv("ZKPOA_SERVER_WORKERS")) workers = atoi(e) >= 1 && atoi(e) <= 8 ? atoi(e) : workers;
  static const bool test_crash = getenv("ZKPOA_SERVER_TEST_CRASH") != nullptr;   // tests: die with a request in hand
  std::mutex qm;
  std::condition_variable qcv;
  std::vector<int> queue;          // accepted connections waiting for a worker
  int busy = 0;
  std::atomic<bool> running{true};
  std::atomic<int> exit_code{0};
  int wake[2] = {-1, -1};          // a worker that decides to leave wakes the accept loop at once
  if (pipe2(wake, O_CLOEXEC | O_NONBLOCK) != 0) wake[0] = wake[1] = -1;
  // leaving: the socket disappears BEFORE the answer goes out, so that the caller's next request finds no socket and
  // starts a fresh server instead of queueing behind one that is gone; the main loop is told after the answer is out
  auto leave = [&] { unlink(sock.c_str()); };
  std::atomic<bool> idle_pending{false};   // a proof has been served since the library last said "nothing to do"
  std::atomic<double> last_activity{now_ms()};   // a request arrived or a worker finished one
  auto serve = [&](int c) {
    if (c == kIdleJob) {   // the server has been idle for a moment: the library's background work, one step at a time,
      idle_work_fn f = g_idle_work.load();   // for as long as nobody is waiting
      while (f && running.load()) {
        {
          std::lock_guard<std::mutex> lk(qm);
          if (!queue.empty()) return;
        }
        if (f() == 0) {
          idle_pending.store(false);
          return;
        }
      }
      return;
    }
    {   // a client that connects and then says nothing must not hold a worker
      struct timeval tv = {10, 0};
      if (const char* e = getenv("ZKPOA_SERVER_RCV_TIMEOUT_S")) tv.tv_sec = atoi(e) > 0 ? atoi(e) : 10;
      (void)setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
      (void)setsockopt(c, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
    }
    struct ucred cred;
    socklen_t cl = sizeof(cred);
    std::string req, reply;
    bool leaving = false;
    int leave_code = 0;
    if (getsockopt(c, SOL_SOCKET, SO_PEERCRED, &cred, &cl) != 0 || cred.uid != getuid()) {
      close(c);
      return;
    }
    if (recv_msg(c, req)) {
      std::vector<std::string> f;
      size_t pos = 0;
      while (pos <= req.size()) {
        size_t e = req.find('\0', pos);
        if (e == std::string::npos) e = req.size();
        f.push_back(req.substr(pos, e - pos));
        pos = e + 1;
      }
      if (!f.empty() && f[0] == "stop") {
        reply = "0";
        leaving = true;
        leave();
      } else if (f.size() >= 9 && f[0] == "prove") {
        const double t0 = now_ms();
        std::string message;
        if (test_crash) _exit(3);
        bool runtime_failure = false;
        // r, s, JSON style and verbosity of THIS request, for this thread only (the library's phase line of a verbose
        // request goes to <socket>.log); never through the process environment: other workers are proving
        (void)load_prover(message);
        if (g_set_thread_options)
          g_set_thread_options(f[5].empty() ? nullptr : f[5].c_str(), f[6].empty() ? nullptr : f[6].c_str(),
                               f[7].empty() ? nullptr : f[7].c_str(), f[8].empty() ? 0 : 1);
        int rc = prove_files(f[1].c_str(), f[2].c_str(), f[3].c_str(), f[4].c_str(), message, &runtime_failure);
        if (rc == EXIT_SUCCESS) idle_pending.store(true);
        if (rc == EXIT_SUCCESS && !f[8].empty())
          message = "zkpoa: prover server pid " + std::to_string((long)getpid()) + " served the proof in " +
                    std::to_string(now_ms() - t0) + " ms";
        reply = std::string(1, (char)('0' + rc)) + message;
        if (runtime_failure) {
          // This process's HIP context (and every key cached in it) is suspect: hand the request back to the client
          // and leave, so that the next call starts a fresh server instead of failing until the idle timeout.
          reply = "R" + message;
          leaving = true;
          leave_code = 4;
          leave();
        }
      } else {
        reply = "1Error: malformed request to the prover server";
      }
      send_msg(c, reply);
    }
    close(c);
    if (leaving) {
      if (leave_code) exit_code.store(leave_code);
      running.store(false);
      if (wake[1] >= 0) (void)!write(wake[1], "x", 1);
    }
  };
  std::vector<std::thread> pool;
  for (int w = 0; w < workers; w++)
    pool.emplace_back([&] {
      for (;;) {
        int c = -1;
        {
          std::unique_lock<std::mutex> lk(qm);
          qcv.wait(lk, [&] { return !queue.empty() || !running.load(); });
          if (queue.empty()) return;
          c = queue.front();
          queue.erase(queue.begin());
          busy++;
        }
        serve(c);
        {
          std::lock_guard<std::mutex> lk(qm);
          busy--;
          if (c != kIdleJob) last_activity.store(now_ms());
        }
        qcv.notify_all();
      }
    });
  while (running.load()) {
    struct pollfd pfd[2] = {{ls, POLLIN, 0}, {wake[0], POLLIN, 0}};
    int pr = poll(pfd, wake[0] >= 0 ? 2 : 1, 200);   // (short slices all the same: the idle clock)
    if (pr < 0 && errno != EINTR) break;
    if (!running.load()) break;
    if (pr > 0 && (pfd[0].revents & POLLIN)) {
      int c = accept4(ls, nullptr, nullptr, SOCK_CLOEXEC);
      if (c >= 0) {
        {
          std::lock_guard<std::mutex> lk(qm);
          queue.push_back(c);
        }
        qcv.notify_one();
      }
      last_activity.store(now_ms());
      continue;
    }
    bool idle;
    {
      std::lock_guard<std::mutex> lk(qm);
      idle = busy == 0 && queue.empty();
      if (!idle) last_activity.store(now_ms());
      // nothing to do for a moment: let the library use the GPU (fixed-base tables of the keys it keeps, one per step)
      if (idle && idle_pending.load() && now_ms() - last_activity.load() >= idle_work_ms) queue.push_back(kIdleJob);
    }
    if (idle && idle_pending.load()) qcv.notify_one();
    if (idle && now_ms() - last_activity.load() >= idle_s * 1e3) break;   // idle: give the HBM back
  }
  running.store(false);
  unlink(sock.c_str());
  close(ls);
  qcv.notify_all();
  if (exit_code.load()) {   // a failed HIP context: no teardown, and nobody waits for workers stuck in it
    close(lock);
    _exit(exit_code.load());
  }
  {   // connections still queued get no answer: their clients prove in-process
    std::lock_guard<std::mutex> lk(qm);
    for (int c : queue)
      if (c >= 0) close(c);
    queue.clear();
  }
  for (auto& t : pool) t.join();
  close(lock);
  return 0;
}

static std::string abs_path(const char* p) {
  if (p[0] == '/') return p;
  char cwd[PATH_MAX];
  if (!getcwd(cwd, sizeof(cwd))) return p;
  return std::string(cwd) + "/" + p;
}

// Client side. Returns the exit code, or -1 when no server could be reached (the caller proves in-process).
static int client_main(char** argv, const std::string& sock, bool stop) {
  // the directory is checked BEFORE anything in it is trusted: a socket somebody else planted in a directory they own
  // is never connected to, and no lock file of theirs is ever read
  if (!secure_dir(sock)) {
    if (stop) return EXIT_SUCCESS;
    fprintf(stderr, "zkpoa: server socket directory of %s is not private to this user; proving in-process\n", sock.c_str());
    return -1;
  }
  int fd = connect_to(sock);
  if (fd < 0 && stop) return EXIT_SUCCESS;   // nothing to stop
  if (fd < 0) {
    // start the server: a detached child of this (GPU-free) process
    pid_t pid = fork();
    if (pid == 0) {
      setsid();
      pid_t p2 = fork();
      if (p2 != 0) _exit(0);
      int devnull = open("/dev/null", O_RDWR);
      std::string logp = sock + ".log";
      int log = open(logp.c_str(), O_CREAT | O_WRONLY | O_APPEND, 0600);
      dup2(devnull, 0);
      dup2(log >= 0 ? log : devnull, 1);
      dup2(log >= 0 ? log : devnull, 2);
      (void)chdir("/");
      _exit(server_main(sock));
    }
    if (pid > 0) (void)waitpid(pid, nullptr, 0);   // the intermediate child exits at once
    for (int i = 0; i < 200 && fd < 0; i++) {   // the server listens before it touches the GPU: normally a few ms
      usleep(i < 20 ? 5000 : 50000);
      fd = connect_to(sock);
    }
    if (fd < 0) return -1;
  }
  const long spid = peer_server_pid(fd);   // the process that listens on THIS connection, same uid -- or nobody
  if (spid == 0) {
    close(fd);
    if (stop) return EXIT_SUCCESS;
    fprintf(stderr, "zkpoa: the process behind %s is not a prover server of this user; proving in-process\n", sock.c_str());
    return -1;
  }
  std::string req;
  if (stop) {
    req = "stop";
  } else {
    auto env = [](const char* n) { const char* e = getenv(n); return std::string(e ? e : ""); };
    const std::string fields[] = {"prove", abs_path(argv[1]), abs_path(argv[2]), abs_path(argv[3]), abs_path(argv[4]),
                                  env("ZKPOA_R"), env("ZKPOA_S"), env("ZKPOA_JSON"), env("ZKPOA_VERBOSE")};
    for (size_t i = 0; i < 9; i++) {
      if (i) req.push_back('\0');
      req += fields[i];
    }
  }
  std::string reply;
  {   // a server stuck on the GPU must not hang the workflow for ever: generous, since a layer-three proof with a cold
      // 21 GB key takes tens of seconds (ZKPOA_SERVER_TIMEOUT_S)
    int to = 1800;
    if (const char* e = getenv("ZKPOA_SERVER_TIMEOUT_S")) to = atoi(e) > 0 ? atoi(e) : to;
    struct timeval tv = {to, 0};
    (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  }
  errno = 0;
  if (!send_msg(fd, req) || !recv_msg(fd, reply) || reply.empty()) {
    // no reply arrived (the server died, or was exiting when we connected): nothing has been written, so this
    // process can simply do the work itself
    const bool timed_out = errno == EAGAIN || errno == EWOULDBLOCK;
    close(fd);
    if (stop) return EXIT_SUCCESS;
    // The server is left alone by default. The single-threaded server serves one client at a time, so a client that
    // timed out may simply have been queued behind other long proofs -- killing the server then would take down a
    // healthy process in the middle of somebody else's proof; and a server that closed the connection is either gone or
    // running its loop, not wedged. Only with ZKPOA_SERVER_KILL_WEDGED=1, only after a timeout, only the pid the kernel
    // reported for this very connection (SO_PEERCRED, same uid) and only while the server's lock is still held (the
    // process that took it is still the one that listens) is the server ended -- it may hold tens of GB of cached keys
    // in HBM. Never a re-exec: this process proves itself, a later call starts a fresh server.
    bool killed = false, alive = kill((pid_t)spid, 0) == 0;
    const char* kw = getenv("ZKPOA_SERVER_KILL_WEDGED");
    if (timed_out && alive && kw && !strcmp(kw, "1") && spid > 1 && spid != (long)getpid()) {
      int lf = open((sock + ".lock").c_str(), O_RDWR | O_CLOEXEC | O_NOFOLLOW);
      bool held = false;
      if (lf >= 0) {
        held = flock(lf, LOCK_EX | LOCK_NB) != 0;   // we could take it: no server owns the socket any more
        if (!held) flock(lf, LOCK_UN);
        close(lf);
      }
      if (held) {
        killed = kill((pid_t)spid, SIGKILL) == 0;
        for (int i = 0; killed && i < 100 && kill((pid_t)spid, 0) == 0; i++) usleep(20000);
      }
    }
    log_fault(timed_out ? "prover server did not answer in time" : "prover server went away without answering",
              std::string("server pid ") + std::to_string(spid) + (killed ? " (killed)" : alive ? " (left running)" : " (gone)"),
              argv[1]);
    if (strict_mode()) {
      fprintf(stderr, "zkpoa: the prover server went away without answering; ZKPOA_STRICT is set: not proving in-process\n");
      return kExitStrict;
    }
    fprintf(stderr, "zkpoa: the prover server went away without answering; proving in-process\n");
    return -1;
  }
  close(fd);
  if (reply[0] == 'R') {
    log_fault("prover server reported a GPU runtime failure", reply.substr(1), argv[1]);
    if (strict_mode()) {
      fprintf(stderr, "zkpoa: the prover server reported a GPU runtime failure (%s); ZKPOA_STRICT is set: not proving "
                      "in-process\n", reply.c_str() + 1);
      return kExitStrict;
    }
    fprintf(stderr, "zkpoa: the prover server reported a GPU runtime failure (%s) and is restarting; proving in-process\n",
            reply.c_str() + 1);
    return -1;
  }
  if (reply.size() > 1) fprintf(stderr, "%s\n", reply.c_str() + 1);
  return reply[0] - '0';
}

int main(int argc, char** argv) {
  const double t_start = now_ms();
  // This process's HIP runtime (loaded later, on demand; the resident server is a child of this process and inherits
  // it): a context has eleven streams -- six lanes, the copy stream, the uploader's own, the null stream -- and with the
  // default of four hardware queues they share queues, so an upload that should overlap the compute sits behind a
  // lane's kernels. Never overrides the caller's own setting.
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  const char* srv = getenv("ZKPOA_SERVER");
  const bool use_server = srv && *srv && strcmp(srv, "0") != 0;
  if (argc == 2 && strcmp(argv[1], "--stop-server") == 0) {
    if (!use_server) setenv("ZKPOA_SERVER", "1", 1);
    return client_main(argv, server_socket_path(), true);
  }
  if (argc != 5) {
    fprintf(stderr, "Invalid number of parameters\nUsage: prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>\n");
    return EXIT_FAILURE;
  }
  if (use_server) {
    int rc = client_main(argv, server_socket_path(), false);
    if (rc >= 0) {
      if (getenv("ZKPOA_VERBOSE")) fprintf(stderr, "zkpoa: prover process total %.1f ms\n", now_ms() - t_start);
      return rc;
    }
    // no server reachable: prove in this process (still on the GPU)
  }
  // A key of 2 GB or more: the proof is made by a worker process and this one leaves as soon as both outputs are in
  // place (csrc/worker_exit.hpp).
  struct stat zs;
  zkpoa::WorkerExit we = zkpoa::WorkerExit::start(stat(argv[1], &zs) == 0 && zs.st_size >= (2ll << 30), "prover");
  std::string message;
  bool runtime_failure = false;
  int rc = prove_files(argv[1], argv[2], argv[3], argv[4], message, &runtime_failure);
  if (rc != EXIT_SUCCESS) {
    fprintf(stderr, "%s\n", message.c_str());
    if (runtime_failure) log_fault("in-process prove hit a HIP runtime failure", message, argv[1]);
    if (we.is_worker()) we.leave(rc);
    return rc;
  }
  if (getenv("ZKPOA_VERBOSE"))   // (CLOCK_MONOTONIC stamps: a caller can see what it waited for before main and after it)
    fprintf(stderr, "zkpoa: prover process total %.1f ms (main entered at %.1f, leaving at %.1f)%s\n", now_ms() - t_start, t_start,
            now_ms(), we.is_worker() ? "; worker process, the caller is released now" : "");
  // Both outputs are complete and renamed into place: leave without running the HIP runtime's teardown
  // (freeing tens of GB of device memory and its queues costs ~0.25 s that a one-shot prover never gets back).
  we.leave(EXIT_SUCCESS);
}
