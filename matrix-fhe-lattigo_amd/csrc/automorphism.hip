// automorphism.hip -- Galois automorphisms X -> X^gen on (poly, limb, coefficient) blocks.
//
// Replaces ring/automorphism.go: AutomorphismNTTIndex (:12-35), AutomorphismNTT / ...WithIndex (:39-81),
// AutomorphismNTTWithIndexThenAddLazy (:86-117) and the coefficient-domain Automorphism, both branches: standard rings
// (:158-175) and conjugate-invariant rings Z[X+X^-1]/(X^2N+1) (:131-156).
// Pure index gathers/scatters: 16*N bytes per limb.  The NTT-domain index is computed in the kernel
// (two bit reversals and one multiply) instead of being read from a table.
#include <hip/hip_runtime.h>
#include "engine_internal.hpp"

RH_DEV u32 brevn(u32 x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// out[row][j] (=|+=) in[row][index(j)], index(j) = bitrev(((gen*(2*bitrev(j)+1) mod NthRoot) - 1)/2), bit reversals over
// lg = log2(NthRoot) - 1 bits (:26-33): NthRoot = 2N (standard, lg = logN) or 4N (conjugate invariant, lg = logN + 1)
__global__ void __launch_bounds__(256)
automorphism_ntt_kernel(const u64* in, u64* out, int logN, int lg, u32 gen, int add_lazy) {
  const u32 N = 1u << logN, mask = (2u << lg) - 1;
  const size_t base = (size_t)blockIdx.x << logN;
  for (u32 j = blockIdx.y * blockDim.x + threadIdx.x; j < N; j += gridDim.y * blockDim.x) {
    const u32 t1 = 2 * brevn(j, lg) + 1;
    const u32 t2 = (((gen * t1) & mask) - 1) >> 1;
    const u64 v = in[base + brevn(t2, lg)];
    out[base + j] = add_lazy ? out[base + j] + v : v;
  }
}
// coefficient domain, conjugate-invariant ring (:131-156): for i in [0, 2N): index = i*gen mod 2N, sign from bit log2(2N) of
// i*gen; only index < N is written, from coefficient i (i < N) or 2N - i with the sign flipped (i >= N).  i -> index is a
// bijection of [0, 2N), so every output coefficient is written exactly once: a scatter with one thread per i.
__global__ void __launch_bounds__(256)
automorphism_coeff_ci_kernel(const u64* in, u64* out, int logN, u64 gen, const LimbConsts* __restrict__ consts, int L) {
  const u32 N = 1u << logN;
  const u64 q = consts[blockIdx.x % (u32)L].q;
  const size_t base = (size_t)blockIdx.x << logN;
  for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < 2 * N; i += gridDim.y * blockDim.x) {
    const u64 raw = (u64)i * gen;
    const u32 index = (u32)(raw & (2 * N - 1));
    u64 tmp = (raw >> (logN + 1)) & 1;
    if (index >= N) continue;
    u32 idx = i;
    if (idx >= N) { idx = 2 * N - idx; tmp ^= 1; }
    const u64 v = in[base + idx];
    out[base + index] = v * (tmp ^ 1) | (q - v) * tmp;
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
  if (r->kind != RH_RING_STANDARD && r->kind != RH_RING_CI) return rh_fail(RH_ERR_UNSUPPORTED, "automorphism: power-of-two rings only (ring/automorphism.go:14-20)");
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
  const int lg = r->kind == RH_RING_CI ? r->logN + 1 : r->logN;          // log2(NthRoot) - 1 with NthRoot = 4N / 2N (ring/ring.go:178-183)
  const u64 nthroot = (u64)2 << lg;
  if ((gen & 1) == 0) return rh_fail(RH_ERR_ARG, "automorphism: the Galois element must be odd");
  // conjugate-invariant ring: the slots are the exponents = 1 mod 4 of the 4N-th root; gen = 3 mod 4 maps them onto exponents the
  // ring does not hold and the reference's table look-up runs past the N coefficients (index out of range panic)
  if (r->kind == RH_RING_CI && (gen & 3) != 1) return rh_fail(RH_ERR_ARG, "automorphism: on a conjugate-invariant ring the Galois element must be 1 mod 4");
  automorphism_ntt_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(in, out, r->logN, lg, (u32)(gen & (nthroot - 1)), add_lazy ? 1 : 0);
  return done("automorphism_ntt_kernel");
}
extern "C" int rh_ring_automorphism(rh_ring* r, int level, const uint64_t* in, uint64_t gen, uint64_t* out, int npoly) {
  if (int rc = common(r, level, in, out, npoly)) return rc;
  const unsigned rows = (unsigned)npoly * (level + 1);
  if (!rows) return RH_OK;
  unsigned chunks = ((unsigned)r->N + 1023) / 1024; if (chunks > 64) chunks = 64;
  if ((gen & 1) == 0) return rh_fail(RH_ERR_ARG, "automorphism: the Galois element must be odd");
  if (r->kind == RH_RING_CI) automorphism_coeff_ci_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(in, out, r->logN, gen, r->d_consts, level + 1);
  else automorphism_coeff_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(in, out, r->logN, gen, r->d_consts, level + 1);
  return done("automorphism_coeff_kernel");
}

// ---- ring.Shift and ring.MultByMonomial (ring/operations.go:278-282, 306-363): index maps over every limb of a block, out of place ----
// (rows = npoly * limbs on gridDim.x like every other kernel of the library: gridDim.y is capped at 65535)
__global__ void __launch_bounds__(256)
shift_kernel(const u64* in, u64* out, unsigned N, unsigned k) {              // p2[j] = p1[(j + k) mod N] (utils.RotateSliceAllocFree: left rotation)
  const size_t base = (size_t)blockIdx.x * N;
  for (unsigned j = blockIdx.y * blockDim.x + threadIdx.x; j < N; j += gridDim.y * blockDim.x) {
    unsigned src = j + k; if (src >= N) src -= N;
    out[base + j] = in[base + src];
  }
}
__global__ void __launch_bounds__(256)
monomial_kernel(const u64* in, u64* out, unsigned N, unsigned shift2n, const LimbConsts* __restrict__ consts, int L) {
  // shift2n = (k + 2N) mod 2N, not 0.  tmp = shift2n < N ? p1 : q - p1 (q - 0 = q, as the reference writes it); s = shift2n mod N;
  // p2[j] = q - tmp[N - s + j] for j < s, tmp[j - s] otherwise (:324-361)
  const u64 q = consts[blockIdx.x % (unsigned)L].q;
  const size_t base = (size_t)blockIdx.x * N;
  const bool neg = shift2n >= N;
  const unsigned s = shift2n >= N ? shift2n - N : shift2n;
  for (unsigned j = blockIdx.y * blockDim.x + threadIdx.x; j < N; j += gridDim.y * blockDim.x) {
    const unsigned src = j < s ? N - s + j : j - s;
    u64 v = in[base + src];
    if (neg) v = q - v;
    out[base + j] = j < s ? q - v : v;
  }
}
static int index_map_common(rh_ring* r, int level, const void* in, const void* out, int npoly, const char* who) {
  if (!r || !in || !out) return rh_fail(RH_ERR_ARG, "%s: null argument", who);
  if (in == out) return rh_fail(RH_ERR_ARG, "%s: the device form is out of place", who);
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "%s: level %d out of range [0,%d)", who, level, r->L);
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "%s: npoly < 0", who);
  (void)hipSetDevice(r->device);
  (void)hipGetLastError();
  return 0;
}
extern "C" int rh_ring_shift(rh_ring* r, int level, const uint64_t* in, uint64_t* out, int k, int npoly) {
  if (int rc = index_map_common(r, level, in, out, npoly, "shift")) return rc;
  const unsigned rows = (unsigned)npoly * (unsigned)(level + 1);
  if (!rows) return RH_OK;
  const int N = r->N;
  int kk = k % N; if (kk < 0) kk += N;
  unsigned chunks = ((unsigned)N + 1023) / 1024; if (chunks > 64) chunks = 64;
  shift_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(in, out, (unsigned)N, (unsigned)kk);
  return done("shift_kernel");
}
extern "C" int rh_ring_mult_by_monomial(rh_ring* r, int level, const uint64_t* in, uint64_t* out, int k, int npoly) {
  if (int rc = index_map_common(r, level, in, out, npoly, "mult_by_monomial")) return rc;
  const unsigned rows = (unsigned)npoly * (unsigned)(level + 1);
  if (!rows) return RH_OK;
  const long N = r->N;
  long sh = ((long)k % (2 * N) + 2 * N) % (2 * N);                         // (k + 2N) % 2N for any int k
  if (sh == 0) {
    if (hipMemcpyAsync(out, in, (size_t)rows * N * 8, hipMemcpyDeviceToDevice, rh_stream(r)) != hipSuccess) return rh_fail(RH_ERR_DEVICE, "mult_by_monomial: copy failed");
    return RH_OK;
  }
  unsigned chunks = ((unsigned)N + 1023) / 1024; if (chunks > 64) chunks = 64;
  monomial_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(in, out, (unsigned)N, (unsigned)sh, r->d_consts, level + 1);
  return done("monomial_kernel");
}

// ---- AutomorphismNTTWithIndex / ...ThenAddLazy (:50-117): the caller's lookup table (N words on the device), any permutation
__global__ void __launch_bounds__(256)
automorphism_index_kernel(const u64* in, u64* out, const u64* __restrict__ index, unsigned N, int add_lazy) {
  const size_t base = (size_t)blockIdx.x * N;
  for (unsigned j = blockIdx.y * blockDim.x + threadIdx.x; j < N; j += gridDim.y * blockDim.x) {
    const u64 v = in[base + index[j]];
    out[base + j] = add_lazy ? out[base + j] + v : v;
  }
}
extern "C" int rh_ring_automorphism_ntt_index(rh_ring* r, int level, const uint64_t* in, const uint64_t* index, uint64_t* out, int npoly,
                                              int add_lazy) {
  if (!index) return rh_fail(RH_ERR_ARG, "automorphism (with index): null index table");
  if (int rc = index_map_common(r, level, in, out, npoly, "automorphism (with index)")) return rc;
  const unsigned rows = (unsigned)npoly * (unsigned)(level + 1);
  if (!rows) return RH_OK;
  unsigned chunks = ((unsigned)r->N + 1023) / 1024; if (chunks > 64) chunks = 64;
  automorphism_index_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(in, out, index, (unsigned)r->N, add_lazy);
  return done("automorphism_index_kernel");
}

// ---- standard <-> conjugate-invariant bridges (ring/conjugate_invariant.go) ----------------------------------------------------------
// unfold (:8-26): std[j] = ci[j], std[n + k] = ci[n - 1 - k]; one thread per word of the 2n-word output row
__global__ void __launch_bounds__(256)
ci_unfold_kernel(const u64* ci, u64* std_, unsigned n) {
  const size_t bi = (size_t)blockIdx.x * n, bo = (size_t)blockIdx.x * 2 * n;
  for (unsigned j = blockIdx.y * blockDim.x + threadIdx.x; j < 2 * n; j += gridDim.y * blockDim.x)
    std_[bo + j] = ci[bi + (j < n ? j : 2 * n - 1 - j)];
}
// fold (:31-49): AutomorphismNTTWithIndex over the first n outputs, then SubRing.Add with the first n words of the standard poly
__global__ void __launch_bounds__(256)
ci_fold_std_kernel(const u64* std_, const u64* __restrict__ index, u64* ci, unsigned n, const LimbConsts* __restrict__ consts, int L) {
  const u64 q = consts[blockIdx.x % (unsigned)L].q;
  const size_t bi = (size_t)blockIdx.x * 2 * n, bo = (size_t)blockIdx.x * n;
  for (unsigned j = blockIdx.y * blockDim.x + threadIdx.x; j < n; j += gridDim.y * blockDim.x) {
    const u64 v = std_[bi + index[j]] + std_[bi + j];
    ci[bo + j] = v >= q ? v - q : v;                            // CRed (addvec, ring/vec_ops.go:7-29)
  }
}
// pad (:52-80): the reference copies the n words and then runs its loop IN PLACE over them, so the second half of the loop reads what the
// first half wrote; the closed form of that is what each thread writes (ringhip.h).  Words n..2n-1 of the output row are not touched.
__global__ void __launch_bounds__(256)
ci_pad_kernel(const u64* std_, u64* ci, unsigned n, int is_ntt, const LimbConsts* __restrict__ consts, int L) {
  const u64 q = consts[blockIdx.x % (unsigned)L].q;
  const size_t bi = (size_t)blockIdx.x * n, bo = (size_t)blockIdx.x * 2 * n;
  const unsigned h = n / 2;
  for (unsigned k = blockIdx.y * blockDim.x + threadIdx.x; k < n; k += gridDim.y * blockDim.x) {
    u64 v;
    if (is_ntt) v = std_[bi + (k < h ? k : n - 1 - k)];
    else if (k == 0) v = 0;
    else if (k < h) v = std_[bi + k];
    else if (k == h) v = q - std_[bi + h];
    else v = q - std_[bi + (n - k)];
    ci[bo + k] = v;
  }
}
static unsigned bridge_chunks(unsigned words) { unsigned c = (words + 1023) / 1024; return c > 64 ? 64 : (c ? c : 1); }
extern "C" int rh_ring_unfold_ci_to_standard(rh_ring* r, int level, const uint64_t* ci, uint64_t* std_, int npoly) {
  if (int rc = index_map_common(r, level, ci, std_, npoly, "unfold_ci_to_standard")) return rc;
  if (r->kind != RH_RING_STANDARD || r->N < 2) return rh_fail(RH_ERR_ARG, "unfold_ci_to_standard: the receiver is the standard ring of degree 2n");
  const unsigned rows = (unsigned)npoly * (unsigned)(level + 1), n = (unsigned)r->N / 2;
  if (!rows) return RH_OK;
  ci_unfold_kernel<<<dim3(rows, bridge_chunks(2 * n)), 256, 0, rh_stream(r)>>>(ci, std_, n);
  return done("ci_unfold_kernel");
}
extern "C" int rh_ring_fold_standard_to_ci(rh_ring* r, int level, const uint64_t* std_, const uint64_t* index, uint64_t* ci, int npoly) {
  if (!index) return rh_fail(RH_ERR_ARG, "fold_standard_to_ci: null index table");
  if (int rc = index_map_common(r, level, std_, ci, npoly, "fold_standard_to_ci")) return rc;
  const unsigned rows = (unsigned)npoly * (unsigned)(level + 1), n = (unsigned)r->N;
  if (!rows) return RH_OK;
  ci_fold_std_kernel<<<dim3(rows, bridge_chunks(n)), 256, 0, rh_stream(r)>>>(std_, index, ci, n, r->d_consts, level + 1);
  return done("ci_fold_std_kernel");
}
extern "C" int rh_ring_pad_default_to_ci(rh_ring* r, int level, const uint64_t* std_, int is_ntt, uint64_t* ci, int npoly) {
  if (int rc = index_map_common(r, level, std_, ci, npoly, "pad_default_to_ci")) return rc;
  if (r->N < 2) return rh_fail(RH_ERR_ARG, "pad_default_to_ci: degree < 2");
  const unsigned rows = (unsigned)npoly * (unsigned)(level + 1), n = (unsigned)r->N;
  if (!rows) return RH_OK;
  ci_pad_kernel<<<dim3(rows, bridge_chunks(n)), 256, 0, rh_stream(r)>>>(std_, ci, n, is_ntt ? 1 : 0, r->d_consts, level + 1);
  return done("ci_pad_kernel");
}
