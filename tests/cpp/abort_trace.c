// abort_trace.c -- TEST INFRASTRUCTURE (loaded by tests/conftest.py through ctypes; not part of libringhip.so).
// A SIGABRT in the GPU suite comes without a word when a runtime library calls abort() directly: this handler writes the native call stack
// of the aborting thread to stderr (module + offset per frame: resolve with llvm-symbolizer against the same image), then hands over to the
// handler that was installed before (Python's faulthandler), which prints the Python stack and lets the process die with SIGABRT as before.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction prev_abrt;
static int out_fd = 2;          /* where the report goes: the REAL stderr (pytest's capture has fd 2 pointing at a temporary file) */

static void on_abort(int sig, siginfo_t* info, void* ctx) {
  static const char head[] = "\n==== SIGABRT: native call stack of the aborting thread (tests/cpp/abort_trace.c) ====\n";
  void* frames[96];
  int n;
  (void)!write(out_fd, head, sizeof head - 1);
  n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, out_fd);
  {
    static const char tail[] = "==== end of native call stack ====\n";
    (void)!write(out_fd, tail, sizeof tail - 1);
  }
  if (out_fd != 2) {          /* what the dying test wrote to its CAPTURED stderr (glibc's and the HIP runtime's last words land there): the tail of that file */
    static char buf[8192];
    off_t end = lseek(2, 0, SEEK_CUR);
    if (end > 0) {
      size_t want = end > (off_t)sizeof buf ? sizeof buf : (size_t)end;
      ssize_t got = pread(2, buf, want, end - (off_t)want);
      if (got > 0) {
        static const char h2[] = "==== captured stderr of the running test (tail) ====\n";
        static const char t2[] = "\n==== end of captured stderr ====\n";
        (void)!write(out_fd, h2, sizeof h2 - 1);
        (void)!write(out_fd, buf, (size_t)got);
        (void)!write(out_fd, t2, sizeof t2 - 1);
      }
    }
  }
  if (prev_abrt.sa_flags & SA_SIGINFO) {
    if (prev_abrt.sa_sigaction) { prev_abrt.sa_sigaction(sig, info, ctx); return; }
  } else if (prev_abrt.sa_handler != SIG_DFL && prev_abrt.sa_handler != SIG_IGN) {
    prev_abrt.sa_handler(sig);
    return;
  }
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
}

int abort_trace_install_fd(int fd);
int abort_trace_install(void) { return abort_trace_install_fd(2); }
int abort_trace_install_fd(int fd) {
  out_fd = fd;
  struct sigaction sa;
  void* warm[4];
  (void)backtrace(warm, 4);              // loads libgcc's unwinder now, not inside the handler
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_abort;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  return sigaction(SIGABRT, &sa, &prev_abrt);
}
