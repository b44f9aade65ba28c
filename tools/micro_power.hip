// Sustained (power-capped) VALU issue rates on MI355X: tools/micro_valu.hip measures bursts of a few milliseconds, during which the chip
// holds its 2.4 GHz peak clock; the headline kernel runs for seconds at the 1400 W package cap with the shader clock near 1.7 GHz
// (tools/exp_clocks.sh).  This program runs each instruction stream for ~1.2 s with full-entropy operands and reports the rate of the
// last 0.8 s next to the rate of one short burst after an idle gap: sustained / burst = the clock the power cap leaves that stream.
//   hipcc --offload-arch=gfx950 -O2 tools/micro_power.hip -o /tmp/micro_power && /tmp/micro_power [waves_per_simd]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define ITERS 2048

// 8 independent 32-bit chains, 16 instructions per loop trip
#define K32(NAME, INS)                                                                                                                    \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                                                             \
    uint32_t a0 = (threadIdx.x + 1) * 2654435761u ^ seed, a1 = a0 * 2246822519u + 1, a2 = a1 * 3266489917u + 3, a3 = a2 * 668265263u + 5, \
             a4 = a3 * 374761393u + 7, a5 = a4 * 2654435761u + 9, a6 = a5 * 2246822519u + 11, a7 = a6 * 3266489917u + 13;                \
    uint32_t b = (seed * 2654435761u) | 0x80000001u;                                                                                      \
    for (int it = 0; it < ITERS; ++it)                                                                                                    \
      asm volatile(INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")                                        \
                   INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")                                        \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                  \
  }
#define I_ADD(r) "v_add_u32 " r ", " r ", %8\n"
#define I_XOR(r) "v_xor_b32 " r ", " r ", %8\n"
#define I_MULLO(r) "v_mul_lo_u32 " r ", " r ", %8\n"
#define I_MULHI(r) "v_mul_hi_u32 " r ", " r ", %8\n"
#define I_MUL24(r) "v_mul_u32_u24 " r ", " r ", %8\n"
#define I_MAD24(r) "v_mad_u32_u24 " r ", " r ", %8, " r "\n"
#define I_BFI(r) "v_bfi_b32 " r ", %8, " r ", " r "\n"
K32(k_add, I_ADD)
K32(k_xor, I_XOR)
K32(k_mullo, I_MULLO)
K32(k_mulhi, I_MULHI)
K32(k_mul24, I_MUL24)
K32(k_mad24, I_MAD24)
K32(k_bfi, I_BFI)

// 8 independent 64-bit chains (even-aligned pairs), 16 instructions per loop trip
#define K64(NAME, INS)                                                                                                                    \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                                                             \
    uint64_t a0 = ((uint64_t)(threadIdx.x + 1) * 0x9E3779B97F4A7C15ull) ^ seed, a1 = a0 * 0xBF58476D1CE4E5B9ull + 1, a2 = a1 * 0x94D049BB133111EBull + 3, \
             a3 = a2 * 0x9E3779B97F4A7C15ull + 5, a4 = a3 * 0xBF58476D1CE4E5B9ull + 7, a5 = a4 * 0x94D049BB133111EBull + 9,              \
             a6 = a5 * 0x9E3779B97F4A7C15ull + 11, a7 = a6 * 0xBF58476D1CE4E5B9ull + 13;                                                 \
    uint32_t b = (seed * 2654435761u) | 0x80000001u;                                                                                      \
    uint64_t c = ((uint64_t)b << 32) | (b * 3u);                                                                                          \
    for (int it = 0; it < ITERS; ++it)                                                                                                    \
      asm volatile(INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")                                        \
                   INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")                                        \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");            \
    uint64_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);                                                      \
  }
#define I_MAD64(r) "v_mad_u64_u32 " r ", vcc, %8, %8, " r "\n"
#define I_LSHLADD(r) "v_lshl_add_u64 " r ", " r ", 0, %9\n"
#define I_FMA64(r) "v_fma_f64 " r ", " r ", %9, %9\n"
K64(k_mad64, I_MAD64)
K64(k_lshladd, I_LSHLADD)

// v_mad_u64_u32 whose multiplicand changes every instruction (the accumulator's own low word): full multiplier activity.
// Fixed registers v[100:115] (the operand syntax has no sub-register modifier), whole loop inside one asm block.
__global__ void __launch_bounds__(256) k_mad64d(uint32_t* out, uint32_t seed) {
  uint32_t x = (threadIdx.x + 1) * 2654435761u ^ seed, b = (seed * 2654435761u) | 0x80000001u, r;
  asm volatile(
      "v_mov_b32 v100, %2\nv_mul_lo_u32 v101, %2, %1\nv_mul_lo_u32 v102, v101, %1\nv_mul_lo_u32 v103, v102, %1\n"
      "v_mul_lo_u32 v104, v103, %1\nv_mul_lo_u32 v105, v104, %1\nv_mul_lo_u32 v106, v105, %1\nv_mul_lo_u32 v107, v106, %1\n"
      "v_mul_lo_u32 v108, v107, %1\nv_mul_lo_u32 v109, v108, %1\nv_mul_lo_u32 v110, v109, %1\nv_mul_lo_u32 v111, v110, %1\n"
      "v_mul_lo_u32 v112, v111, %1\nv_mul_lo_u32 v113, v112, %1\nv_mul_lo_u32 v114, v113, %1\nv_mul_lo_u32 v115, v114, %1\n"
      "s_movk_i32 s20, 2048\n"
      "1:\n"
      "v_mad_u64_u32 v[100:101], vcc, v100, %1, v[100:101]\nv_mad_u64_u32 v[102:103], vcc, v102, %1, v[102:103]\n"
      "v_mad_u64_u32 v[104:105], vcc, v104, %1, v[104:105]\nv_mad_u64_u32 v[106:107], vcc, v106, %1, v[106:107]\n"
      "v_mad_u64_u32 v[108:109], vcc, v108, %1, v[108:109]\nv_mad_u64_u32 v[110:111], vcc, v110, %1, v[110:111]\n"
      "v_mad_u64_u32 v[112:113], vcc, v112, %1, v[112:113]\nv_mad_u64_u32 v[114:115], vcc, v114, %1, v[114:115]\n"
      "v_mad_u64_u32 v[100:101], vcc, v100, %1, v[100:101]\nv_mad_u64_u32 v[102:103], vcc, v102, %1, v[102:103]\n"
      "v_mad_u64_u32 v[104:105], vcc, v104, %1, v[104:105]\nv_mad_u64_u32 v[106:107], vcc, v106, %1, v[106:107]\n"
      "v_mad_u64_u32 v[108:109], vcc, v108, %1, v[108:109]\nv_mad_u64_u32 v[110:111], vcc, v110, %1, v[110:111]\n"
      "v_mad_u64_u32 v[112:113], vcc, v112, %1, v[112:113]\nv_mad_u64_u32 v[114:115], vcc, v114, %1, v[114:115]\n"
      "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n"
      "v_xor_b32 v100, v100, v102\nv_xor_b32 v100, v100, v104\nv_xor_b32 v100, v100, v106\nv_xor_b32 v100, v100, v108\n"
      "v_xor_b32 v100, v100, v110\nv_xor_b32 v100, v100, v112\nv_xor_b32 v100, v100, v114\nv_xor_b32 %0, v100, v101\n"
      : "=v"(r) : "v"(b), "v"(x)
      : "vcc", "scc", "s20", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static_assert(ITERS == 2048, "k_mad64d hard-codes its trip count");

struct Case { const char* name; void (*fn)(uint32_t*, uint32_t); };

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 8;              // waves per SIMD = blocks of 256 threads per CU
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int grid = cus * wps;
  uint32_t* out; CK(hipMalloc(&out, (size_t)grid * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const Case cases[] = {{"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_bfi_b32", k_bfi}, {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24},
                        {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_lshl_add_u64", k_lshladd},
                        {"v_mad_u64_u32 (constant multiplicands)", k_mad64}, {"v_mad_u64_u32 (changing multiplicand)", k_mad64d}};
  printf("device %s, %d CUs, %d waves/SIMD; G wave-instructions/s chip-wide: burst (one launch after 0.7 s idle) | sustained (last 0.8 s of 1.2 s) | ratio\n",
         prop.name, cus, wps);
  const double winstr = (double)grid * 4 /*waves per block*/ * ITERS * 16;
  if (argc > 3) {                                              // micro_power <wps> <case> <seconds>: one stream held for a power sample from outside
    const Case& c = cases[atoi(argv[2]) % (int)(sizeof(cases) / sizeof(cases[0]))];
    const double secs = atof(argv[3]);
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    CK(hipEventRecord(e0));
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      for (int i = 0; i < 8; ++i, ++n) c.fn<<<grid, 256>>>(out, (uint32_t)n);
      CK(hipDeviceSynchronize());
    }
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s held %.1f s at %d waves/SIMD: %.1f G wave-instructions/s\n", c.name, ms * 1e-3, wps, winstr * n / (ms * 1e-3) / 1e9);
    return 0;
  }
  for (const Case& c : cases) {
    std::this_thread::sleep_for(std::chrono::milliseconds(700));
    c.fn<<<grid, 256>>>(out, 1u);                              // warm the code path (tiny next to the idle gap? no: redo the gap)
    CK(hipDeviceSynchronize());
    std::this_thread::sleep_for(std::chrono::milliseconds(700));
    CK(hipEventRecord(e0)); c.fn<<<grid, 256>>>(out, 2u); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms_burst; CK(hipEventElapsedTime(&ms_burst, e0, e1));
    // sustained: ~0.4 s untimed, then ~0.8 s timed
    const int n_pre = (int)(400.0f / ms_burst) + 1, n_timed = (int)(800.0f / ms_burst) + 1;
    for (int i = 0; i < n_pre; ++i) c.fn<<<grid, 256>>>(out, 3u + i);
    CK(hipEventRecord(e0));
    for (int i = 0; i < n_timed; ++i) c.fn<<<grid, 256>>>(out, 100u + i);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms_sus; CK(hipEventElapsedTime(&ms_sus, e0, e1));
    const double burst = winstr / (ms_burst * 1e-3) / 1e9, sus = winstr * n_timed / (ms_sus * 1e-3) / 1e9;
    printf("%-42s | %8.1f | %8.1f | %.3f\n", c.name, burst, sus, sus / burst);
    fflush(stdout);
  }
  CK(hipFree(out));
  return 0;
}
