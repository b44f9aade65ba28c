// rescale.hip -- RNS rescale: division of a polynomial by its last modulus (ring/scaling.go), SURVEY 8(f) rank 1.
//
// Replaces Ring.DivFloorByLastModulus(:21-28), DivRoundByLastModulus(:112-126), their Many forms (:56-88, :160-192)
// and the NTT-domain forms DivFloorByLastModulusManyNTT(:32-52), DivRoundByLastModulusNTT(:92-108),
// DivRoundByLastModulusManyNTT(:130-156).  Every output is a full MRed, i.e. canonical, so only the arithmetic
// meaning has to match:  floor: (x_i - x_L) * q_L^-1 mod q_i ;  round: (x_i + h - t) * q_L^-1 with h = (q_L-1)/2,
// t = (x_L + h) mod q_L.  The coefficient-domain kernels also reproduce the reference's in-place side effects on p0.
// One thread per coefficient, limb loop inside: 8*(2*level+1)*N bytes per poly and step -- HBM-bound.
#include <hip/hip_runtime.h>
#include <vector>
#include "engine_internal.hpp"
#include "hostmath.hpp"


// mode 0 = floor, 1 = round.  p0: rows0 limbs per poly (limbs 0..level), p1: rows1 limbs per poly (limbs 0..level-1).
__global__ void __launch_bounds__(256)
rescale_step_kernel(int mode, u64* p0, int rows0, u64* p1, int rows1, int level, int N, u64 qL, const RescaleLimb* __restrict__ T) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int poly = blockIdx.y;
  if (k >= N) return;
  u64* src = p0 + (size_t)poly * rows0 * N + k;
  u64* dst = p1 + (size_t)poly * rows1 * N + k;
  u64 t = src[(size_t)level * N];
  if (mode == 1) {
    t = cred(t + ((qL - 1) >> 1), qL);                 // AddScalar on the last limb (:120), in place like the reference
    src[(size_t)level * N] = t;
  }
  for (int i = 0; i < level; ++i) {
    const RescaleLimb l = T[i];
    const u64 x = src[(size_t)i * N];
    u64 r;
    if (mode == 1) {
      const u64 u = l.s + 2 * l.q - x;                 // AddScalarLazyThenNegTwoModulusLazy (:123), in place
      src[(size_t)i * N] = u;
      r = mred(t + u, l.c, l.q, l.qinv);               // AddLazyThenMulScalarMontgomery (:124)
    } else {
      r = mred(2 * l.q - x + t, l.c, l.q, l.qinv);     // SubThenMulScalarMontgomeryTwoModulus (:26)
    }
    dst[(size_t)i * N] = r;
  }
}

__global__ void __launch_bounds__(256)
gather_limb_kernel(const u64* p0, int rows0, int limb, u64* tmp, int N) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < N) tmp[(size_t)blockIdx.y * N + k] = p0[((size_t)blockIdx.y * rows0 + limb) * N + k];
}
// tmp: coefficient-domain last limb (canonical).  buff[i] = (t [+ h, recentred]) mod q_i for i < level
__global__ void __launch_bounds__(256)
rescale_expand_kernel(int mode, const u64* tmp, u64* buff, int rowsb, int level, int N, u64 qL, const RescaleLimb* __restrict__ T) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int poly = blockIdx.y;
  if (k >= N) return;
  u64 t = tmp[(size_t)poly * N + k];
  if (mode == 1) t = cred(t + ((qL - 1) >> 1), qL);
  for (int i = 0; i < level; ++i) {
    const RescaleLimb l = T[i];
    const u64 v = mode == 1 ? t + l.s : t;             // AddScalarLazy (:104); reduced here so the NTT sees < q
    buff[((size_t)poly * rowsb + i) * N + k] = bred_add(v, l.q, l.bred0);
  }
}
// p1[i] = MRed(2q - p0[i] + buff[i], c_i)
__global__ void __launch_bounds__(256)
rescale_finish_kernel(const u64* buff, int rowsb, const u64* p0, int rows0, u64* p1, int rows1, int level, int N, const RescaleLimb* __restrict__ T) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int poly = blockIdx.y;
  if (k >= N) return;
  for (int i = 0; i < level; ++i) {
    const RescaleLimb l = T[i];
    const u64 x = buff[((size_t)poly * rowsb + i) * N + k], y = p0[((size_t)poly * rows0 + i) * N + k];
    p1[((size_t)poly * rows1 + i) * N + k] = mred(2 * l.q - y + x, l.c, l.q, l.qinv);
  }
}

static int rescale_table(rh_ring* r, int level, RescaleLimb** d) {
  if ((int)r->rescale_tables.size() < r->L) r->rescale_tables.assign(r->L, nullptr);
  if (r->rescale_tables[level]) { *d = (RescaleLimb*)r->rescale_tables[level]; return 0; }
  const u64 qL = r->moduli[level];
  std::vector<RescaleLimb> h(level);
  for (int i = 0; i < level; ++i) {
    const u64 q = r->moduli[i];
    h[i].q = q; h[i].qinv = r->mred[i]; h[i].bred0 = r->bred[2 * i];
    h[i].c = rh::mform(q - rh::invmod_prime(qL % q, q), q);            // rewRescaleConstants, ring/ring.go:363-380
    h[i].s = q - (((qL - 1) >> 1) % q);
  }
  RescaleLimb* p = nullptr;
  if (hipMalloc((void**)&p, (level ? level : 1) * sizeof(RescaleLimb)) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc failed");
  if (level && hipMemcpy(p, h.data(), level * sizeof(RescaleLimb), hipMemcpyHostToDevice) != hipSuccess) return rh_fail(RH_ERR_DEVICE, "hipMemcpy failed");
  r->rescale_tables[level] = p;
  *d = p;
  return 0;
}
void rh_rescale_teardown(rh_ring* r) { for (void* p : r->rescale_tables) if (p) (void)hipFree(p); r->rescale_tables.clear(); }

static int ensure_scratch(rh_ring* r, int which, size_t words) {
  if (r->rs_words[which] >= words) return 0;
  if (r->d_rs[which]) (void)hipFree(r->d_rs[which]);
  r->d_rs[which] = nullptr; r->rs_words[which] = 0;
  if (hipMalloc((void**)&r->d_rs[which], words * 8) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc(rescale scratch) failed");
  r->rs_words[which] = words;
  return 0;
}

static int check_args(rh_ring* r, int level, int nb, const void* p0, const void* p1, int npoly, int p1_rows) {
  if (!r || !p0 || !p1) return rh_fail(RH_ERR_ARG, "rescale: null argument");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "rescale: level %d out of range [0,%d)", level, r->L);
  if (nb < 0 || nb > level) return rh_fail(RH_ERR_ARG, "rescale: nbRescales %d exceeds level %d", nb, level);
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "rescale: npoly < 0");
  if (p1_rows < level + 1 - nb) return rh_fail(RH_ERR_ARG, "rescale: output block has %d limbs, needs %d", p1_rows, level + 1 - nb);
  (void)hipSetDevice(r->device);
  (void)hipGetLastError();
  return 0;
}
// npoly blocks of `limbs` leading limbs from a (poly, src_rows) block to a (poly, dst_rows) block: one strided copy
static int copy_leading_limbs(rh_ring* r, u64* dst, int dst_rows, const u64* src, int src_rows, int limbs, int npoly) {
  const size_t N = (size_t)r->N;
  if (limbs <= 0 || npoly <= 0) return RH_OK;
  if (hipMemcpy2DAsync(dst, (size_t)dst_rows * N * 8, src, (size_t)src_rows * N * 8, (size_t)limbs * N * 8, (size_t)npoly, hipMemcpyDeviceToDevice,
                       rh_stream(r)) != hipSuccess) return rh_fail(RH_ERR_DEVICE, "rescale: strided copy failed");
  return RH_OK;
}
static int launched(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "%s launch failed: %s", what, hipGetErrorString(e));
  return RH_OK;
}

// coefficient domain, nb sequential divisions.  p0 has level+1 limbs per poly and is modified like the reference's
// buffers are; p1 has p1_rows limbs per poly.
extern "C" int rh_ring_div_by_last_modulus_many(rh_ring* r, int round, int level, int nb, uint64_t* p0, uint64_t* p1, int p1_rows, int npoly) {
  if (int rc = check_args(r, level, nb, p0, p1, npoly, p1_rows)) return rc;
  if (npoly == 0) return RH_OK;
  std::lock_guard<std::recursive_mutex> lk(r->mu);          // lazily built per-level tables and shared scratch
  const int N = r->N;
  dim3 grid((N + 255) / 256, npoly);
  if (nb == 0) return copy_leading_limbs(r, p1, p1_rows, p0, level + 1, level + 1, npoly);
  // steps 0..nb-2 run in place on p0 (as the reference does on buff), the last one writes p1
  for (int j = 0; j < nb; ++j) {
    const int lv = level - j;
    RescaleLimb* T; if (int rc = rescale_table(r, lv, &T)) return rc;
    const bool last = j == nb - 1;
    rescale_step_kernel<<<grid, 256, 0, rh_stream(r)>>>(round ? 1 : 0, p0, level + 1, last ? p1 : p0, last ? p1_rows : level + 1, lv, N, r->moduli[lv], T);
  }
  return launched("rescale_step_kernel");
}

// NTT domain (DivFloorByLastModulusManyNTT / DivRoundByLastModulusManyNTT).  p0 is not modified.
extern "C" int rh_ring_div_by_last_modulus_many_ntt(rh_ring* r, int round, int level, int nb, const uint64_t* p0, uint64_t* p1, int p1_rows, int npoly) {
  if (int rc = check_args(r, level, nb, p0, p1, npoly, p1_rows)) return rc;
  if (npoly == 0) return RH_OK;
  std::lock_guard<std::recursive_mutex> lk(r->mu);          // lazily built per-level tables and shared scratch
  const int N = r->N;
  dim3 grid((N + 255) / 256, npoly);
  if (nb == 0) return copy_leading_limbs(r, p1, p1_rows, p0, level + 1, level + 1, npoly);
  if (nb == 1) {
    // INTT of the last limb only, re-expansion under every remaining modulus, NTT, fused subtract-multiply
    if (int rc = ensure_scratch(r, 0, (size_t)npoly * N)) return rc;
    if (int rc = ensure_scratch(r, 1, (size_t)npoly * level * N)) return rc;
    RescaleLimb* T; if (int rc = rescale_table(r, level, &T)) return rc;
    u64* tmp = r->d_rs[0]; u64* buff = r->d_rs[1];
    if (r->kind != RH_RING_STANDARD) {
      // 3N and conjugate-invariant rings (schemes/matrix_ckks/evaluator.go:235 rescales on the 3N ring): the same steps with the ring's own
      // transform -- INTT of the last limb, re-expansion under every remaining modulus, NTT, subtract-multiply (ring/scaling.go:97-124)
      gather_limb_kernel<<<grid, 256, 0, rh_stream(r)>>>(p0, level + 1, level, tmp, N);
      if (int rc = rh_ring_ntt_any(r, tmp, tmp, npoly, 1, level, true)) return rc;
      rescale_expand_kernel<<<grid, 256, 0, rh_stream(r)>>>(round ? 1 : 0, tmp, buff, level, level, N, r->moduli[level], T);
      if (level > 0) if (int rc = rh_ring_ntt_any(r, buff, buff, npoly, level, 0, false)) return rc;
      rescale_finish_kernel<<<grid, 256, 0, rh_stream(r)>>>(buff, level, p0, level + 1, p1, p1_rows, level, N, T);
      return launched("rescale (NTT domain)");
    }
    if (rh_can_intt_limb_strided(r)) {               // the inverse tile stages read the last limb where it lies
      if (int rc = rh_std_intt_limb_strided(r, p0, level + 1, level, tmp, npoly)) return rc;
    } else {
      gather_limb_kernel<<<grid, 256, 0, rh_stream(r)>>>(p0, level + 1, level, tmp, N);
      if (int rc = rh_std_ntt_launch(r, tmp, tmp, npoly, 1, level, true, true, 0)) return rc;
    }
    if (level > 0 && rh_can_fuse_submul(r)) {          // p1 = MRed(2q - p0 + NTT(buff), c_i) in the tile kernel's epilogue (:120-124)
      std::vector<u64> sc(level);
      for (int i = 0; i < level; ++i) sc[i] = rh::mform(r->moduli[i] - rh::invmod_prime(r->moduli[level] % r->moduli[i], r->moduli[i]), r->moduli[i]);
      const bool fuse_expand = r->logN > 12 && r->logN <= 17;       // the re-expansion feeds the column stages directly
      if (fuse_expand) { if (int rc = rh_std_ntt_expand_cols_launch(r, tmp, buff, npoly, level, T, round ? 1 : 0, r->moduli[level])) return rc; }
      else rescale_expand_kernel<<<grid, 256, 0, rh_stream(r)>>>(round ? 1 : 0, tmp, buff, level, level, N, r->moduli[level], T);
      return rh_std_ntt_submul_launch(r, buff, npoly, level, 0, p0, level + 1, p1, p1_rows, sc.data(), fuse_expand);
    }
    rescale_expand_kernel<<<grid, 256, 0, rh_stream(r)>>>(round ? 1 : 0, tmp, buff, level, level, N, r->moduli[level], T);
    if (level > 0) if (int rc = rh_std_ntt_launch(r, buff, buff, npoly, level, 0, false, false, 0)) return rc;
    rescale_finish_kernel<<<grid, 256, 0, rh_stream(r)>>>(buff, level, p0, level + 1, p1, p1_rows, level, N, T);
    return launched("rescale (NTT domain)");
  }
  // nb > 1: INTT everything, divide nb times in the coefficient domain, NTT what is left (:44-51, :142-150)
  if (int rc = ensure_scratch(r, 1, (size_t)npoly * (level + 1) * N)) return rc;
  u64* buff = r->d_rs[1];
  if (int rc = rh_ring_ntt_any(r, p0, buff, npoly, level + 1, 0, true)) return rc;
  for (int j = 0; j < nb; ++j) {
    const int lv = level - j;
    RescaleLimb* T; if (int rc = rescale_table(r, lv, &T)) return rc;
    rescale_step_kernel<<<grid, 256, 0, rh_stream(r)>>>(round ? 1 : 0, buff, level + 1, buff, level + 1, lv, N, r->moduli[lv], T);
  }
  const int out_limbs = level + 1 - nb;
  // forward transform of limbs 0..level-nb of each poly: rows are strided by level+1 in buff -> compact, transform, scatter
  if (int rc = ensure_scratch(r, 0, (size_t)npoly * out_limbs * N)) return rc;
  u64* cmp = r->d_rs[0];
  if (int rc = copy_leading_limbs(r, cmp, out_limbs, buff, level + 1, out_limbs, npoly)) return rc;
  if (int rc = rh_ring_ntt_any(r, cmp, cmp, npoly, out_limbs, 0, false)) return rc;
  if (int rc = copy_leading_limbs(r, p1, p1_rows, cmp, out_limbs, out_limbs, npoly)) return rc;
  return launched("rescale many (NTT domain)");
}

// the same with the NTT-domain layout of a 3N ring's data as a per-call argument (see rh_ring_ntt_layout)
extern "C" int rh_ring_div_by_last_modulus_many_ntt_layout(rh_ring* r, int round, int level, int nb, const uint64_t* p0, uint64_t* p1, int p1_rows, int npoly,
                                                           int block_order) {
  if (!r) return rh_fail(RH_ERR_ARG, "rescale: null ring");
  if (block_order && !rh_ring_ntt3n_block_order_supported(r)) return rh_fail(RH_ERR_UNSUPPORTED, "block order needs a 3N ring with N = 3 * 2^k, k >= 13");
  RhLayoutScope ls(r->kind == RH_RING_3N ? (block_order ? 1 : 0) : -1);
  return rh_ring_div_by_last_modulus_many_ntt(r, round, level, nb, p0, p1, p1_rows, npoly);
}

// rh_ring_reserve: the rescale scratch for batches of up to npoly polys and the per-level constant tables, so
// that no later call allocates (hipMalloc / hipFree synchronise the device and cannot be captured in a HIP graph)
int rh_rescale_reserve(rh_ring* r, int npoly) {
  if (int rc = ensure_scratch(r, 0, (size_t)npoly * r->L * r->N)) return rc;
  if (int rc = ensure_scratch(r, 1, (size_t)npoly * r->L * r->N)) return rc;
  for (int lv = 1; lv < r->L; ++lv) { RescaleLimb* T; if (int rc = rescale_table(r, lv, &T)) return rc; }
  return RH_OK;
}
