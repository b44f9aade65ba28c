// automorphism.hip -- Galois automorphisms X -> X^gen on (poly, limb, coefficient) blocks.
//
// Replaces ring/automorphism.go: AutomorphismNTTIndex (:12-35), AutomorphismNTT / ...WithIndex (:39-81),
// AutomorphismNTTWithIndexThenAddLazy (:86-117) and the coefficient-domain Automorphism for standard rings (:121-176).
// Pure index gathers/scatters: 16*N bytes per limb.  The NTT-domain index is computed in the kernel
// (two bit reversals and one multiply) instead of being read from a table.
#include <hip/hip_runtime.h>
#include "engine_internal.hpp"

RH_DEV u32 brevn(u32 x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// out[row][j] (=|+=) in[row][index(j)], index(j) = bitrev(((gen*(2*bitrev(j)+1) mod 2N) - 1)/2)    (:26-33)
__global__ void __launch_bounds__(256)
automorphism_ntt_kernel(const u64* in, u64* out, int logN, u32 gen, int add_lazy) {
  const u32 N = 1u << logN, mask = 2 * N - 1;
  const size_t base = (size_t)blockIdx.x << logN;
  for (u32 j = blockIdx.y * blockDim.x + threadIdx.x; j < N; j += gridDim.y * blockDim.x) {
    const u32 t1 = 2 * brevn(j, logN) + 1;
    const u32 t2 = (((gen * t1) & mask) - 1) >> 1;
    const u64 v = in[base + brevn(t2, logN)];
    out[base + j] = add_lazy ? out[base + j] + v : v;
  }
}
// coefficient domain, standard ring: out[(i*gen) mod N] = +-in[i], sign from bit logN of i*gen   (:162-175)
__global__ void __launch_bounds__(256)
automorphism_coeff_kernel(const u64* in, u64* out, int logN, u64 gen, const LimbConsts* __restrict__ consts, int L) {
  const u32 N = 1u << logN;
  const u64 q = consts[blockIdx.x % (u32)L].q;
  const size_t base = (size_t)blockIdx.x << logN;
  for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < N; i += gridDim.y * blockDim.x) {
    const u64 raw = (u64)i * gen;
    const u32 index = (u32)(raw & (N - 1));
    const u64 tmp = (raw >> logN) & 1;
    const u64 v = in[base + i];
    out[base + index] = v * (tmp ^ 1) | (q - v) * tmp;
  }
}

static int common(rh_ring* r, int level, const void* in, const void* out, int npoly) {
  if (!r || !in || !out) return rh_fail(RH_ERR_ARG, "automorphism: null argument");
  if (r->kind != RH_RING_STANDARD) return rh_fail(RH_ERR_UNSUPPORTED, "automorphism: power-of-two rings only (ring/automorphism.go:14-20)");
  if (in == out) return rh_fail(RH_ERR_ARG, "automorphism: the result cannot be in place (ring/automorphism.go:38)");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "automorphism: level %d out of range [0,%d)", level, r->L);
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "automorphism: npoly < 0");
  (void)hipSetDevice(r->device);
  (void)hipGetLastError();
  return 0;
}
static int done(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "%s launch failed: %s", what, hipGetErrorString(e));
  return RH_OK;
}

extern "C" int rh_ring_automorphism_ntt(rh_ring* r, int level, const uint64_t* in, uint64_t gen, uint64_t* out, int npoly, int add_lazy) {
  if (int rc = common(r, level, in, out, npoly)) return rc;
  const unsigned rows = (unsigned)npoly * (level + 1);
  if (!rows) return RH_OK;
  unsigned chunks = ((unsigned)r->N + 1023) / 1024; if (chunks > 64) chunks = 64;
  automorphism_ntt_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(in, out, r->logN, (u32)(gen & (2 * (u64)r->N - 1)), add_lazy ? 1 : 0);
  return done("automorphism_ntt_kernel");
}
extern "C" int rh_ring_automorphism(rh_ring* r, int level, const uint64_t* in, uint64_t gen, uint64_t* out, int npoly) {
  if (int rc = common(r, level, in, out, npoly)) return rc;
  const unsigned rows = (unsigned)npoly * (level + 1);
  if (!rows) return RH_OK;
  unsigned chunks = ((unsigned)r->N + 1023) / 1024; if (chunks > 64) chunks = 64;
  automorphism_coeff_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(in, out, r->logN, gen, r->d_consts, level + 1);
  return done("automorphism_coeff_kernel");
}
