// micro_hostpath.hip -- what the host-pointer seam is made of (tools/bench_host_path.py measures the whole call): cost of
// hipPointerGetAttributes, of one hipMemcpyAsync + synchronise by size and direction (page-locked memory), of N back-to-back copies on
// one stream, and of H2D / D2H running on two streams at once.  Build: hipcc --offload-arch=gfx950 -O2 tools/micro_hostpath.hip -o tools/micro_hostpath
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t MAXB = (size_t)16 << 20;
  char *hp, *hq, *d, *d2; hipHostMalloc((void**)&hp, MAXB); hipHostMalloc((void**)&hq, MAXB); hipMalloc((void**)&d, MAXB); hipMalloc((void**)&d2, MAXB);
  char* pg = (char*)malloc(MAXB);
  for (size_t i = 0; i < MAXB; i += 4096) { hp[i] = 1; hq[i] = 1; pg[i] = 1; }
  hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipPointerAttribute_t a;
  const int R = 200;
  double t0 = now(); for (int i = 0; i < R; ++i) (void)hipPointerGetAttributes(&a, hp + (i % 16) * 65536); double t1 = now();
  printf("hipPointerGetAttributes pinned   %.2f us\n", (t1 - t0) / R);
  t0 = now(); for (int i = 0; i < R; ++i) { (void)hipPointerGetAttributes(&a, pg + (i % 16) * 65536); (void)hipGetLastError(); } t1 = now();
  printf("hipPointerGetAttributes pageable %.2f us\n", (t1 - t0) / R);
  for (size_t b : {(size_t)32 << 10, (size_t)128 << 10, (size_t)512 << 10, (size_t)2 << 20, (size_t)8 << 20}) {
    for (int dir = 0; dir < 2; ++dir) {
      for (int w = 0; w < 3; ++w) { if (dir) hipMemcpyAsync(hp, d, b, hipMemcpyDeviceToHost, s0); else hipMemcpyAsync(d, hp, b, hipMemcpyHostToDevice, s0); hipStreamSynchronize(s0); }
      t0 = now();
      for (int i = 0; i < 50; ++i) { if (dir) hipMemcpyAsync(hp, d, b, hipMemcpyDeviceToHost, s0); else hipMemcpyAsync(d, hp, b, hipMemcpyHostToDevice, s0); hipStreamSynchronize(s0); }
      t1 = now();
      printf("%s %6zu KiB copy+sync           %8.1f us  %.1f GB/s\n", dir ? "D2H" : "H2D", b >> 10, (t1 - t0) / 50, b / ((t1 - t0) / 50) / 1e3);
    }
  }
  for (size_t b : {(size_t)32 << 10, (size_t)512 << 10}) {
    t0 = now();
    for (int r = 0; r < 20; ++r) { for (int i = 0; i < 16; ++i) hipMemcpyAsync(d + i * b, hp + i * b, b, hipMemcpyHostToDevice, s0); hipStreamSynchronize(s0); }
    t1 = now();
    printf("16 x %zu KiB H2D back to back, one sync: %8.1f us (%.1f per copy)\n", b >> 10, (t1 - t0) / 20, (t1 - t0) / 20 / 16);
  }
  for (size_t b : {(size_t)2 << 20, (size_t)8 << 20}) {
    t0 = now();
    for (int r = 0; r < 20; ++r) { hipMemcpyAsync(d, hp, b, hipMemcpyHostToDevice, s0); hipMemcpyAsync(hq, d2, b, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s0); hipStreamSynchronize(s1); }
    t1 = now();
    printf("H2D + D2H of %zu KiB on two streams at once: %8.1f us  (%.1f GB/s each way)\n", b >> 10, (t1 - t0) / 20, b / ((t1 - t0) / 20) / 1e3);
  }
  // host memcpy rate (staging of pageable limbs)
  t0 = now(); for (int r = 0; r < 10; ++r) memcpy(hp, pg, (size_t)8 << 20); t1 = now();
  printf("memcpy pageable -> pinned 8 MiB: %.1f us (%.1f GB/s)\n", (t1 - t0) / 10, (double)((size_t)8 << 20) / ((t1 - t0) / 10) / 1e3);
  // pageable hipMemcpyAsync (runtime staging)
  t0 = now(); for (int r = 0; r < 10; ++r) { hipMemcpyAsync(d, pg, (size_t)512 << 10, hipMemcpyHostToDevice, s0); hipStreamSynchronize(s0); } t1 = now();
  printf("H2D 512 KiB from PAGEABLE memory + sync: %.1f us\n", (t1 - t0) / 10);
  return 0;
}
