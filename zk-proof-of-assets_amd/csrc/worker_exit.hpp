// One-shot commands that hold tens of GB (on the GPU: `prover` with a 13-116 GB key; on the host: `zkpoa-setup`) do their
// work in a WORKER process, and the command itself leaves the moment the worker reports that the outputs are renamed
// into place. Measured (tools/file_inclusive.py `after_main_ms`): after `_exit` -- no runtime teardown -- the kernel
// still needs 130-165 ms to dismantle a process holding a layer-two or layer-three key on the GPU (2 ms at layer one),
// seconds for ~100 GB of host arrays, and a caller that waits for the exit waits for that too.
//   * forked before anything touches the GPU (the commands call this first); never when a GPU runtime is already loaded
//     into the process (a preloaded profiler's): RTLD_NOLOAD only asks;
//   * the worker dies with the command (PR_SET_PDEATHSIG), reports its exit status over a pipe, then closes the caller's
//     stdin / stdout / stderr so that a pipeline sees the end of the command, not of the worker;
//   * a worker that ends without a report (a crash) is waited for and its status becomes the command's;
//   * the worker keeps its GPU lock file until it is gone, so the next command on that GPU starts after it.
// ZKPOA_DETACH_EXIT=0: one process. =always: a worker whatever `wanted` says (tests).
#pragma once
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

namespace zkpoa {

struct WorkerExit {
  int report_fd = -1;   // >= 0: this IS the worker
  bool is_worker() const { return report_fd >= 0; }

  // Returns in the worker (is_worker()) or, when no worker was made, in the one process there is. In the command's own
  // process of a pair it does not return: it exits with the worker's status.
  static WorkerExit start(bool wanted, const char* who) {
    WorkerExit w;
    const char* det = getenv("ZKPOA_DETACH_EXIT");
    if (det && strcmp(det, "0") == 0) return w;
    if (!wanted && !(det && strcmp(det, "always") == 0)) return w;
    for (const char* lib : {"libhsa-runtime64.so.1", "libhsa-runtime64.so", "libamdhip64.so.7", "libamdhip64.so.6", "libamdhip64.so"})
      if (void* h = dlopen(lib, RTLD_NOLOAD | RTLD_LAZY)) {
        dlclose(h);
        if (gpu_runtime_in_use()) return w;
      }
    int pfd[2];
    if (pipe2(pfd, O_CLOEXEC) != 0) return w;
    fflush(stdout);
    fflush(stderr);
    const pid_t self = getpid();
    const pid_t worker = fork();
    if (worker < 0) {
      close(pfd[0]);
      close(pfd[1]);
      return w;
    }
    if (worker > 0) {
      close(pfd[1]);
      unsigned char code = 0;
      ssize_t got;
      do got = read(pfd[0], &code, 1);
      while (got < 0 && errno == EINTR);
      if (got == 1) _exit(code);   // outputs complete, or the failure is on stderr: the worker's exit is not ours to wait for
      int st = 0;                  // the worker ended without a report
      while (waitpid(worker, &st, 0) < 0 && errno == EINTR) {
      }
      if (WIFEXITED(st) && WEXITSTATUS(st) != 0) _exit(WEXITSTATUS(st));
      fprintf(stderr, "%s: the worker process ended abnormally (%s %d)\n", who, WIFSIGNALED(st) ? "signal" : "status",
              WIFSIGNALED(st) ? WTERMSIG(st) : WEXITSTATUS(st));
      _exit(EXIT_FAILURE);
    }
    close(pfd[0]);
    w.report_fd = pfd[1];
    prctl(PR_SET_PDEATHSIG, SIGKILL);            // killing the command kills its work, as in one process
    if (getppid() != self) _exit(EXIT_FAILURE);   // (the parent went away before the line above took effect)
    return w;
  }

  // worker: hand the status over, stop holding the caller's pipes open, and leave without any teardown.
  // one process: flush and leave the same way (the outputs are complete: nothing a destructor does is still needed).
  [[noreturn]] void leave(int code) {
    fflush(stdout);
    fflush(stderr);
    if (report_fd >= 0) {
      const unsigned char b = (unsigned char)code;
      (void)!write(report_fd, &b, 1);
      close(report_fd);
      close(STDIN_FILENO);
      close(STDOUT_FILENO);
      close(STDERR_FILENO);
    }
    _exit(code);
  }

 private:
  // A command that links the library has the runtime LOADED from its first instruction; what rules a fork out is a
  // runtime that is already running (its threads do not survive a fork). The HSA runtime starts its threads at hsa_init;
  // a process with one thread has not initialised it.
  static bool gpu_runtime_in_use() {
    FILE* f = fopen("/proc/self/status", "r");
    if (!f) return true;
    char line[256];
    long threads = -1;
    while (fgets(line, sizeof line, f))
      if (strncmp(line, "Threads:", 8) == 0) threads = strtol(line + 8, nullptr, 10);
    fclose(f);
    return threads != 1;
  }
};

}  // namespace zkpoa
