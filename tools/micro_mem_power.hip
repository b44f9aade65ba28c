// Energy per byte of streaming traffic on MI355X by access width: c[i] = a[i] + b[i] over three 2 GiB buffers, held for some seconds while
// rocm-smi is sampled from outside (tools/exp_mem_power.sh).  mode 0: 8 bytes per lane per access (global_load_dwordx2, what the NTT
// kernels use); mode 1: 16 bytes per lane (dwordx4, what the element-wise kernels use); mode 2: read-only sum of 8-byte accesses;
// mode 3: read-only sum of 16-byte accesses.
//   hipcc --offload-arch=gfx950 -O2 tools/micro_mem_power.hip -o tools/micro_mem_power.bin && tools/micro_mem_power.bin <mode> <seconds>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;

__global__ void __launch_bounds__(256) add8(const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ c) {
  const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;          // 16 accesses of 8 B per lane, each wave-contiguous (512 B)
  u64 x[16], y[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { x[k] = a[base + k * 256]; y[k] = b[base + k * 256]; }
#pragma unroll
  for (int k = 0; k < 16; ++k) c[base + k * 256] = x[k] + y[k];
}
__global__ void __launch_bounds__(256) add16(const ulonglong2* __restrict__ a, const ulonglong2* __restrict__ b, ulonglong2* __restrict__ c) {
  const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;          // 8 accesses of 16 B per lane (1 KiB per wave instruction)
  ulonglong2 x[8], y[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { x[k] = a[base + k * 256]; y[k] = b[base + k * 256]; }
#pragma unroll
  for (int k = 0; k < 8; ++k) { ulonglong2 r; r.x = x[k].x + y[k].x; r.y = x[k].y + y[k].y; c[base + k * 256] = r; }
}
__global__ void __launch_bounds__(256) sum8(const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ c) {
  const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += a[base + k * 256] ^ b[base + k * 256];
  if (s == 0x1234567) c[threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) sum16(const ulonglong2* __restrict__ a, const ulonglong2* __restrict__ b, u64* __restrict__ c) {
  const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { ulonglong2 x = a[base + k * 256], y = b[base + k * 256]; s += (x.x ^ y.x) + (x.y ^ y.y); }
  if (s == 0x1234567) c[threadIdx.x] = s;
}
// L2-resident reads: every block re-reads its own 32 KiB (x2 buffers) `reps` times; the grid's footprint is 2 x 16 MiB = 4 MiB per XCD
__global__ void __launch_bounds__(256) sum16_rep(const ulonglong2* a, const ulonglong2* b, u64* c, int reps) {
  const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
  u64 s = 0;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ulonglong2 x = a[base + k * 256], y = b[base + k * 256];
      s += (x.x ^ y.x) + (x.y ^ y.y) + r;
    }
    asm volatile("" ::: "memory");                                        // re-read every trip (the 128 KiB a CU's two blocks cycle through do not fit its L1)
  }
  if (s == 0x1234567) c[threadIdx.x] = s;
}
// L2-resident WRITES: every block re-writes its own 32 KiB `reps` times (footprint 16 MiB = 2 MiB per XCD): are they absorbed by L2 or written through?
__global__ void __launch_bounds__(256) store16_rep(ulonglong2* c, int reps, u64 seed) {
  const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { ulonglong2 v; v.x = seed + r + k; v.y = seed ^ (u64)(r * 8 + k); c[base + k * 256] = v; }
    asm volatile("" ::: "memory");
  }
}
__global__ void fill(u64* p, size_t n, u64 seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (i + seed) * 0x9E3779B97F4A7C15ull;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 4.0;
  const size_t words = (size_t)1 << 28;                                 // 2 GiB per buffer
  u64 *a, *b, *c;
  CK(hipMalloc(&a, words * 8)); CK(hipMalloc(&b, words * 8)); CK(hipMalloc(&c, words * 8));
  fill<<<4096, 256>>>(a, words, 1); fill<<<4096, 256>>>(b, words, 2); fill<<<4096, 256>>>(c, words, 3);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)(words / 4096);
  const auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  CK(hipEventRecord(e0));
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 8; ++i, ++n) {
      if (mode == 0) add8<<<grid, 256>>>(a, b, c);
      else if (mode == 1) add16<<<grid, 256>>>((const ulonglong2*)a, (const ulonglong2*)b, (ulonglong2*)c);
      else if (mode == 2) sum8<<<grid, 256>>>(a, b, c);
      else if (mode == 3) sum16<<<grid, 256>>>((const ulonglong2*)a, (const ulonglong2*)b, c);
      else if (mode == 4) sum16_rep<<<512, 256>>>((const ulonglong2*)a, (const ulonglong2*)b, c, 256);       // 512 blocks x 32 KiB x 2 = 32 MiB: L2-resident (4 MiB per XCD)
      else if (mode == 5) sum16<<<4096, 256>>>((const ulonglong2*)a, (const ulonglong2*)b, c);              // 4096 blocks x 32 KiB x 2 = 256 MiB re-read every launch
      else store16_rep<<<512, 256>>>((ulonglong2*)c, 256, (u64)n);                                           // 512 blocks x 32 KiB = 16 MiB, re-written 256 times per launch
    }
    CK(hipDeviceSynchronize());
  }
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = (double)words * 8 * (mode < 2 ? 3 : 2) * n;
  if (mode == 4) bytes = 512.0 * 32768 * 2 * 256 * n;
  if (mode == 5) bytes = 4096.0 * 32768 * 2 * n;
  if (mode == 6) bytes = 512.0 * 32768 * 256 * n;            // 128 MiB per buffer pair half... 2 x 128 MiB re-read every launch: Infinity-Cache-resident
  const char* names[] = {"add, 8 B per lane", "add, 16 B per lane", "read-only, 8 B per lane", "read-only, 16 B per lane",
                         "read-only, 16 B per lane, L2-resident (32 MiB footprint, 256 re-reads per launch)",
                         "read-only, 16 B per lane, 256 MiB footprint re-read every launch (Infinity Cache)",
                         "write-only, 16 B per lane, 16 MiB footprint re-written 256 times per launch (L2-resident?)"};
  printf("%s: %.0f GB/s over %.1f s\n", names[mode % 7], bytes / (ms * 1e-3) / 1e9, ms * 1e-3);
  return 0;
}
