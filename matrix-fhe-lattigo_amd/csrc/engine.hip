// engine.hip -- C-ABI implementation (include/ringhip.h): ring construction, twiddle-table layout, kernel launches.
//
// Host logic here replaces, for device-resident data, the per-limb dispatch loops of the reference
// (Ring.NTT ring/ntt.go:127-152, SubRing ops ring/subring_ops.go, Ring ops ring/operations.go) with one batched
// launch over (poly, limb) -- limbs and polys are independent, so the batch IS the parallelism.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <vector>
#include <new>
#include "../../include/ringhip.h"
#include "hostmath.hpp"
#include "ntt_kernels.hip.hpp"
#include "vec_kernels.hip.hpp"
#include "ntt_kernels_asm.hip.hpp"
#include "ntt3n_kernels_asm.hip.hpp"
#include "engine_internal.hpp"

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
int rh_fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
  return code;
}
extern "C" const char* rh_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ per-call context
static thread_local hipStream_t tl_stream = nullptr;
static thread_local bool tl_has_stream = false;
static thread_local u64* tl_ws = nullptr;
static thread_local size_t tl_ws_words = 0;
static thread_local int tl_layout3n = -1;          // -1: the handle's tuning value; 0 / 1: the layout the running entry point was told
int rh_layout3n(const rh_ring* r) { return tl_layout3n >= 0 ? tl_layout3n : r->block_order3n; }
RhLayoutScope::RhLayoutScope(int layout) : prev(tl_layout3n) { tl_layout3n = layout; }
RhLayoutScope::~RhLayoutScope() { tl_layout3n = prev; }
hipStream_t rh_stream(const rh_ring* r) { return tl_has_stream ? tl_stream : r->stream; }
u64* rh_ws_override(size_t words) { return (tl_ws && tl_ws_words >= words) ? tl_ws : nullptr; }
RhCallScope::RhCallScope(hipStream_t st, u64* ws, size_t ws_words) : prev_st(tl_stream), prev_has(tl_has_stream), prev_ws(tl_ws), prev_words(tl_ws_words) {
  tl_stream = st; tl_has_stream = true;
  if (ws) { tl_ws = ws; tl_ws_words = ws_words; }
}
RhCallScope::~RhCallScope() { tl_stream = prev_st; tl_has_stream = prev_has; tl_ws = prev_ws; tl_ws_words = prev_words; }
extern "C" int rh_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

// ------------------------------------------------------------------------------------------------ table layout
// natural order -> the order ntt_fwd_tile / ntt_inv_tile read (ntt_kernels.hip.hpp): per 4096-tile T
//   [slot]                 round A, slot = (2^u - 1) + g          <- nat[2^(S1+u)   + T*2^u     + g]
//   [16 + slot*16 + hi4]   round B                                <- nat[2^(S1+4+u) + T*2^(4+u) + (hi4<<u) + g]
//   [256 + slot*256 + tid] round C                                <- nat[2^(S1+8+u) + T*2^(8+u) + (tid<<u) + g]
template <class T>
static void build_kernel_order(const T* nat, T* out, int logN) {
  const int S1 = logN - LT;
  const size_t ntiles = (size_t)1 << S1;
  for (size_t t = 0; t < ntiles; ++t) {
    T* o = out + (t << LT);
    memset(o, 0, sizeof(T) * TILE);
    for (int u = 0; u < 4; ++u)
      for (int g = 0; g < (1 << u); ++g) {
        const int slot = (1 << u) - 1 + g;
        o[slot] = nat[((size_t)1 << (S1 + u)) + (t << u) + g];
        for (int h = 0; h < 16; ++h) o[16 + slot * 16 + h] = nat[((size_t)1 << (S1 + 4 + u)) + (t << (4 + u)) + ((size_t)h << u) + g];
        for (int i = 0; i < 256; ++i) o[256 + slot * 256 + i] = nat[((size_t)1 << (S1 + 8 + u)) + (t << (8 + u)) + ((size_t)i << u) + g];
      }
  }
}

template <class T>
static int upload(T** dptr, const std::vector<T>& h) {
  if (hipMalloc((void**)dptr, h.size() * sizeof(T)) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc(%zu) failed", h.size() * sizeof(T));
  if (hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return rh_fail(RH_ERR_DEVICE, "hipMemcpy H2D failed");
  return 0;
}

// uploads natural-order Shoup tables (+ optional Montgomery forward table) and, for N >= 4096, their kernel-order copies
int rh_std_upload_tables(rh_ring* r, const std::vector<tw2>& fs, const std::vector<tw2>& is, const std::vector<u64>* mont,
                         const std::vector<tw2>& lastw) {
  const int L = r->L, N = r->N, logN = r->logN;
  int rc = 0;
  if (!rc) rc = upload(&r->d_tw_fwd, fs);
  if (!rc) rc = upload(&r->d_tw_inv, is);
  if (!rc && mont) rc = upload(&r->d_tw_fwd_mont, *mont);
  if (!rc) rc = upload(&r->d_lastw, lastw);
  if (!rc && logN >= LT) {
    std::vector<tw2> kf((size_t)L * N), ki((size_t)L * N);
    for (int i = 0; i < L; ++i) {
      build_kernel_order(fs.data() + (size_t)i * N, kf.data() + (size_t)i * N, logN);
      build_kernel_order(is.data() + (size_t)i * N, ki.data() + (size_t)i * N, logN);
    }
    if (!rc) rc = upload(&r->d_twk_fwd, kf);
    if (!rc) rc = upload(&r->d_twk_inv, ki);
    if (!rc && mont) {
      std::vector<u64> km((size_t)L * N);
      for (int i = 0; i < L; ++i) build_kernel_order(mont->data() + (size_t)i * N, km.data() + (size_t)i * N, logN);
      rc = upload(&r->d_twk_fwd_mont, km);
    }
  }
  return rc;
}
int rh_upload_consts(rh_ring* r, const std::vector<LimbConsts>& hc) { return upload(&r->d_consts, hc); }

// ------------------------------------------------------------------------------------------------ ring construction
static int validate_degree(int kind, int N) {
  if (kind == RH_RING_STANDARD || kind == RH_RING_CI) {
    // MinimumRingDegreeForLoopUnrolledOperations = 8 (ring/ring.go:21-23, :318)
    if (N < 8 || (N & (N - 1)) != 0 || N > (1 << 17)) return rh_fail(RH_ERR_ARG, "invalid ring degree: must be a power of 2 greater than 8 (and at most 2^17), got N=%d", N);
    return 0;
  }
  if (kind == RH_RING_3N) {
    int m = N; int a = 0, b = 0;
    while (m % 2 == 0) { m /= 2; ++a; }
    while (m % 3 == 0) { m /= 3; ++b; }
    if (m != 1 || a < 1 || b < 1) return rh_fail(RH_ERR_ARG, "invalid 3N ring degree: N=%d must be 2^a*3^b with a,b>=1", N);
    return 0;
  }
  return rh_fail(RH_ERR_ARG, "unknown ring kind %d", kind);
}

static bool one_pass_ok(rh_ring* r);
extern "C" int rh_ring_create(rh_ring** out, int device, int kind, int N, int L, const uint64_t* moduli, const uint64_t* mred,
                              const uint64_t* bred, const uint64_t* ninv, const uint64_t* roots_fwd, const uint64_t* roots_bwd,
                              const uint64_t* omega3n) {
  if (!out || !moduli || !mred || !bred || L < 1 || L > RH_MAX_LIMBS) return rh_fail(RH_ERR_ARG, "rh_ring_create: bad arguments (L=%d)", L);
  if (int e = validate_degree(kind, N)) return e;
  for (int i = 0; i < L; ++i) {
    if (moduli[i] >= ((u64)1 << 61) || (moduli[i] & 1) == 0) return rh_fail(RH_ERR_MODULUS, "modulus %d (%llu) must be odd and < 2^61", i, (unsigned long long)moduli[i]);
    for (int j = 0; j < i; ++j) if (moduli[i] == moduli[j]) return rh_fail(RH_ERR_MODULUS, "invalid moduli: duplicate modulus %llu", (unsigned long long)moduli[i]);
  }
  if (hipSetDevice(device) != hipSuccess) return rh_fail(RH_ERR_DEVICE, "hipSetDevice(%d) failed", device);
  rh_ring* r = new (std::nothrow) rh_ring();
  if (!r) return rh_fail(RH_ERR_NOMEM, "out of host memory");
  r->device = device; r->kind = kind; r->N = N; r->L = L;
  r->moduli.assign(moduli, moduli + L); r->mred.assign(mred, mred + L); r->bred.assign(bred, bred + 2 * L);
  int rc = 0;
  std::vector<LimbConsts> hc(L);
  for (int i = 0; i < L; ++i) {
    LimbConsts& c = hc[i];
    c.q = moduli[i]; c.qinv = mred[i]; c.bred0 = bred[2 * i]; c.bred1 = bred[2 * i + 1]; c.nq = (u64)0 - moduli[i];
    c.ninv_mont = 0; c.ninv_w = 0; c.ninv_wp = 0;
  }
  if (kind == RH_RING_STANDARD || kind == RH_RING_CI) {
    if (!ninv || !roots_fwd || !roots_bwd) { delete r; return rh_fail(RH_ERR_ARG, "rh_ring_create: ring needs ninv and root tables"); }
    int logN = 0; while ((1 << logN) < N) ++logN;
    r->logN = logN;
    // table length handed over: NthRoot/2 = N (standard, 2N-th root) or 2N (conjugate invariant, 4N-th root)
    const size_t TN = kind == RH_RING_CI ? (size_t)2 * N : (size_t)N;
    r->ninv.assign(ninv, ninv + L);
    r->roots_fwd.assign(roots_fwd, roots_fwd + (size_t)L * TN);
    r->roots_bwd.assign(roots_bwd, roots_bwd + (size_t)L * TN);
    std::vector<tw2> fs((size_t)L * N), is((size_t)L * N), lastw(L);
    std::vector<u64> mont((size_t)L * N);
    std::vector<CiFold> fold(L);
    for (int i = 0; i < L; ++i) {
      const u64 q = moduli[i];
      LimbConsts& c = hc[i];
      c.ninv_mont = ninv[i];
      c.ninv_w = rh::imform(ninv[i], q); c.ninv_wp = rh::shoup_quotient(c.ninv_w, q);
      // one modular inverse of 2^64 per limb, then a multiply per root
      const u64 rinv = rh::imform(1, q);
      const u64* rf = roots_fwd + (size_t)i * TN; const u64* rb = roots_bwd + (size_t)i * TN;
      for (int j = 0; j < N; ++j) {
        // standard: stage with m blocks uses roots[m+i] (ring/ntt.go:240-255).  Conjugate invariant: the same stage
        // uses roots[2m+i] of the 4N-th-root table (:768-781, :1119-1147), i.e. entry j = m+i reads 2m+i = j + 2^floor(log2 j)
        size_t src = (size_t)j;
        if (kind == RH_RING_CI && j > 0) { int hb = 31 - __builtin_clz((unsigned)j); src = (size_t)j + ((size_t)1 << hb); }
        u64 wf = rh::mulmod(rf[src] % q, rinv, q), wb = rh::mulmod(rb[src] % q, rinv, q);
        fs[(size_t)i * N + j] = tw2{wf, rh::shoup_quotient(wf, q)};
        is[(size_t)i * N + j] = tw2{wb, rh::shoup_quotient(wb, q)};
        mont[(size_t)i * N + j] = rf[src];
      }
      u64 lw = rh::mulmod(is[(size_t)i * N + 1].w, c.ninv_w, q);
      lastw[i] = tw2{lw, rh::shoup_quotient(lw, q)};
      if (kind == RH_RING_CI) {
        const u64 ff = rh::mulmod(rf[1] % q, rinv, q), fb = rh::mulmod(rb[1] % q, rinv, q);
        fold[i] = CiFold{tw2{ff, rh::shoup_quotient(ff, q)}, tw2{fb, rh::shoup_quotient(fb, q)}};
      }
    }
    if (!rc) rc = rh_std_upload_tables(r, fs, is, &mont, lastw);
    if (!rc && kind == RH_RING_STANDARD) {          // N^-1 * 2^64 = NInv as handed over (Montgomery form): constants of rh_ring_intt_mul
      std::vector<LimbConsts> hr(hc); std::vector<tw2> lr(L);
      for (int i = 0; i < L; ++i) {
        const u64 q = moduli[i], nr = ninv[i] % q;
        hr[i].ninv_w = nr; hr[i].ninv_wp = rh::shoup_quotient(nr, q);
        const u64 lw = rh::mulmod(is[(size_t)i * N + 1].w, nr, q);
        lr[i] = tw2{lw, rh::shoup_quotient(lw, q)};
      }
      rc = upload(&r->d_consts_r, hr);
      if (!rc) rc = upload(&r->d_lastw_r, lr);
    }
    if (!rc && N < 16) {
      std::vector<u64> bm((size_t)L * N);
      for (int i = 0; i < L; ++i) for (int j = 0; j < N; ++j) {
        size_t src = (size_t)j;
        if (kind == RH_RING_CI && j > 0) { int hb = 31 - __builtin_clz((unsigned)j); src = (size_t)j + ((size_t)1 << hb); }
        bm[(size_t)i * N + j] = roots_bwd[(size_t)i * TN + src];
      }
      rc = upload(&r->d_tw_inv_mont, bm);
    }
    if (!rc && kind == RH_RING_CI) rc = upload(&r->d_cifold, fold);
  } else {
    if (!omega3n) { delete r; return rh_fail(RH_ERR_ARG, "rh_ring_create: 3N ring needs omega3n"); }
    r->omega3n.assign(omega3n, omega3n + L);
    rc = rh_ring3n_setup(r, hc);
  }
  if (!rc) rc = upload(&r->d_consts, hc);
  if (rc) { rh_ring_destroy(r); return rc; }
  r->hconsts = hc;
  (void)one_pass_ok(r);          // N = 2^13 / 2^14: raise the one-pass kernels' dynamic-LDS limit now, while one thread owns the handle
  *out = r;
  return RH_OK;
}

extern "C" int rh_ring_create_auto(rh_ring** out, int device, int kind, int N, int L, const uint64_t* moduli, const uint64_t* omega3n) {
  if (!out || !moduli || L < 1 || L > RH_MAX_LIMBS) return rh_fail(RH_ERR_ARG, "rh_ring_create_auto: bad arguments");
  if (int e = validate_degree(kind, N)) return e;
  std::vector<u64> mred(L), bred(2 * L), ninv(L), rf, rb, om;
  const bool pow2 = kind == RH_RING_STANDARD || kind == RH_RING_CI;
  const u64 nthroot = kind == RH_RING_STANDARD ? (u64)2 * N : (kind == RH_RING_CI ? (u64)4 * N : (u64)3 * N);
  const size_t TN = (size_t)(nthroot >> 1);
  if (pow2) { rf.resize((size_t)L * TN); rb.resize((size_t)L * TN); }
  else om.resize(L);
  for (int i = 0; i < L; ++i) {
    const u64 q = moduli[i];
    if (!rh::is_prime(q)) return rh_fail(RH_ERR_MODULUS, "invalid modulus: %llu is not prime)", (unsigned long long)q);
    if (q % nthroot != 1) return rh_fail(RH_ERR_MODULUS, "invalid modulus: %llu != 1 mod NthRoot)", (unsigned long long)q);
    mred[i] = rh::gen_mred_constant(q);
    rh::gen_bred_constant(q, &bred[2 * i]);
    if (pow2) {
      if (rh::gen_ntt_tables(q, nthroot, &rf[(size_t)i * TN], &rb[(size_t)i * TN], &ninv[i])) return rh_fail(RH_ERR_MODULUS, "table generation failed for modulus %llu", (unsigned long long)q);
    } else {
      ninv[i] = rh::mform(rh::invmod_prime((u64)N % q, q), q);
      om[i] = omega3n ? omega3n[i] : rh::powmod(rh::primitive_root(q), (q - 1) / nthroot, q);
    }
  }
  return rh_ring_create(out, device, kind, N, L, moduli, mred.data(), bred.data(), ninv.data(),
                        pow2 ? rf.data() : nullptr, pow2 ? rb.data() : nullptr, kind == RH_RING_3N ? om.data() : nullptr);
}

void rh_poly_slots_teardown(rh_ring* r);
extern "C" void rh_ring_destroy(rh_ring* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  void* ptrs[] = {r->d_cifold, r->d_consts, r->d_tw_fwd, r->d_tw_inv, r->d_tw_fwd_mont, r->d_twk_fwd, r->d_twk_inv, r->d_twk_fwd_mont, r->d_lastw, r->d_tw_inv_mont, r->d_consts_r, r->d_lastw_r};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (RhHostSlot* sl : r->all_slots) {
    if (sl->buf) (void)hipFree(sl->buf);
    if (sl->stream) (void)hipStreamDestroy(sl->stream);
    delete sl;
  }
  rh_poly_slots_teardown(r);
  rh_rescale_teardown(r);
  for (int i = 0; i < 2; ++i) if (r->d_rs[i]) (void)hipFree(r->d_rs[i]);
  if (r->d_rows) (void)hipFree(r->d_rows);
  rh_ring3n_teardown(r);
  delete r;
}
extern "C" int rh_ring_n(const rh_ring* r) { return r ? r->N : 0; }
extern "C" int rh_ring_limbs(const rh_ring* r) { return r ? r->L : 0; }
extern "C" int rh_ring_get_constants(const rh_ring* r, uint64_t* moduli, uint64_t* mred, uint64_t* bred, uint64_t* ninv,
                                     uint64_t* roots_fwd, uint64_t* roots_bwd, uint64_t* omega3n) {
  if (!r) return rh_fail(RH_ERR_ARG, "null ring");
  if (moduli) memcpy(moduli, r->moduli.data(), r->moduli.size() * 8);
  if (mred) memcpy(mred, r->mred.data(), r->mred.size() * 8);
  if (bred) memcpy(bred, r->bred.data(), r->bred.size() * 8);
  if (ninv && !r->ninv.empty()) memcpy(ninv, r->ninv.data(), r->ninv.size() * 8);
  if (roots_fwd && !r->roots_fwd.empty()) memcpy(roots_fwd, r->roots_fwd.data(), r->roots_fwd.size() * 8);
  if (roots_bwd && !r->roots_bwd.empty()) memcpy(roots_bwd, r->roots_bwd.data(), r->roots_bwd.size() * 8);
  if (omega3n && !r->omega3n.empty()) memcpy(omega3n, r->omega3n.data(), r->omega3n.size() * 8);
  return RH_OK;
}
extern "C" int rh_ring_set_stream(rh_ring* r, void* s) { if (!r) return rh_fail(RH_ERR_ARG, "null ring"); r->stream = (hipStream_t)s; return RH_OK; }
extern "C" int rh_ring_sync(rh_ring* r) {
  if (!r) return rh_fail(RH_ERR_ARG, "null ring");
  hipError_t e = hipStreamSynchronize(rh_stream(r));
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "hipStreamSynchronize: %s", hipGetErrorString(e));
  return RH_OK;
}

// ------------------------------------------------------------------------------------------------ device memory
extern "C" int rh_dev_alloc(rh_ring* r, size_t words, uint64_t** dptr) {
  if (!r || !dptr) return rh_fail(RH_ERR_ARG, "rh_dev_alloc: null argument");
  (void)hipSetDevice(r->device);
  if (hipMalloc((void**)dptr, words * 8) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc(%zu words) failed", words);
  return RH_OK;
}
extern "C" int rh_dev_free(rh_ring* /*r: unused, may already be destroyed*/, uint64_t* dptr) { if (dptr) (void)hipFree(dptr); return RH_OK; }
extern "C" int rh_dev_upload(rh_ring* r, uint64_t* dst, const uint64_t* src, size_t words) {
  if (!r || !dst || !src) return rh_fail(RH_ERR_ARG, "rh_dev_upload: null argument");
  hipError_t e = hipMemcpyAsync(dst, src, words * 8, hipMemcpyHostToDevice, rh_stream(r));
  if (e == hipSuccess) e = hipStreamSynchronize(rh_stream(r));
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "upload: %s", hipGetErrorString(e));
  return RH_OK;
}
// Poly.CopyLvl on device blocks (ring/poly.go): limbs 0..level of every poly of src (src_rows limbs per poly) into dst (dst_rows limbs per
// poly), asynchronous on the ring's stream
extern "C" int rh_ring_copy_rows(rh_ring* r, uint64_t* dst, int dst_rows, const uint64_t* src, int src_rows, int npoly, int level) {
  if (!r || !dst || !src) return rh_fail(RH_ERR_ARG, "rh_ring_copy_rows: null argument");
  if (level < 0 || level >= r->L || dst_rows < level + 1 || src_rows < level + 1 || npoly < 0) return rh_fail(RH_ERR_ARG, "rh_ring_copy_rows: bad level / rows / npoly");
  if (npoly == 0) return RH_OK;
  (void)hipSetDevice(r->device);
  const size_t N = (size_t)r->N;
  hipError_t e = hipMemcpy2DAsync(dst, (size_t)dst_rows * N * 8, src, (size_t)src_rows * N * 8, (size_t)(level + 1) * N * 8, (size_t)npoly,
                                  hipMemcpyDeviceToDevice, rh_stream(r));
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "rh_ring_copy_rows: %s", hipGetErrorString(e));
  return RH_OK;
}
extern "C" int rh_dev_download(rh_ring* r, uint64_t* dst, const uint64_t* src, size_t words) {
  if (!r || !dst || !src) return rh_fail(RH_ERR_ARG, "rh_dev_download: null argument");
  hipError_t e = hipMemcpyAsync(dst, src, words * 8, hipMemcpyDeviceToHost, rh_stream(r));
  if (e == hipSuccess) e = hipStreamSynchronize(rh_stream(r));
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "download: %s", hipGetErrorString(e));
  return RH_OK;
}

// ------------------------------------------------------------------------------------------------ NTT launches
static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rh_fail(RH_ERR_DEVICE, "%s launch failed: %s", what, hipGetErrorString(e));
  return RH_OK;
}

template <class P>
static void launch_fwd_cols(int S1, dim3 grid, hipStream_t st, const u64* in, u64* out, const typename P::tw_t* tw,
                            const LimbConsts* c, int L, int logN, int Ls = 0, int Lso = 0) {
  switch (S1) {
    case 1: ntt_fwd_cols<P, 1><<<grid, 256, 0, st>>>(in, out, tw, c, L, logN, Ls, Lso); break;
    case 2: ntt_fwd_cols<P, 2><<<grid, 256, 0, st>>>(in, out, tw, c, L, logN, Ls, Lso); break;
    case 3: ntt_fwd_cols<P, 3><<<grid, 256, 0, st>>>(in, out, tw, c, L, logN, Ls, Lso); break;
    case 4: ntt_fwd_cols<P, 4><<<grid, 256, 0, st>>>(in, out, tw, c, L, logN, Ls, Lso); break;
    case 5: ntt_fwd_cols<P, 5><<<grid, 256, 0, st>>>(in, out, tw, c, L, logN, Ls, Lso); break;
  }
}
static void launch_inv_cols(int S1, dim3 grid, hipStream_t st, u64* data, const tw2* tw, const tw2* lastw,
                            const LimbConsts* c, int L, int logN, int scale, int Ls = 0) {
  switch (S1) {
    case 1: ntt_inv_cols<1><<<grid, 256, 0, st>>>(data, tw, lastw, c, L, logN, scale, Ls); break;
    case 2: ntt_inv_cols<2><<<grid, 256, 0, st>>>(data, tw, lastw, c, L, logN, scale, Ls); break;
    case 3: ntt_inv_cols<3><<<grid, 256, 0, st>>>(data, tw, lastw, c, L, logN, scale, Ls); break;
    case 4: ntt_inv_cols<4><<<grid, 256, 0, st>>>(data, tw, lastw, c, L, logN, scale, Ls); break;
    case 5: ntt_inv_cols<5><<<grid, 256, 0, st>>>(data, tw, lastw, c, L, logN, scale, Ls); break;
  }
}

// One-pass launches (N = 2^13, 2^14; ntt_kernels_asm.hip.hpp): 68 / 136 KiB of dynamic LDS per workgroup, beyond the 64 KiB default limit.
static bool one_pass_ok(rh_ring* r) {
  const int S1 = r->logN - LT;
  if (!(r->one_pass && r->asm_tile && (S1 == 1 || S1 == 2))) return false;
  if (!r->one_pass_ready) {
    const int b1 = 2 * LDS_WORDS * 8, b2 = 4 * LDS_WORDS * 8;
    hipError_t e = hipSuccess;
    auto raise = [&](const void* f, int bytes) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    raise((const void*)ntt_fwd_onepass_asm<1, false>, b1); raise((const void*)ntt_fwd_onepass_asm<1, true>, b1);
    raise((const void*)ntt_fwd_onepass_asm<2, false>, b2); raise((const void*)ntt_fwd_onepass_asm<2, true>, b2);
    raise((const void*)ntt_inv_onepass_asm<1, false>, b1); raise((const void*)ntt_inv_onepass_asm<1, true>, b1);
    raise((const void*)ntt_inv_onepass_asm<2, false>, b2); raise((const void*)ntt_inv_onepass_asm<2, true>, b2);
    if (e != hipSuccess) { (void)hipGetLastError(); r->one_pass = false; return false; }     // (a device without that much LDS: the two-pass launches)
    r->one_pass_ready = true;
  }
  return true;
}

// limb0: first limb of the table set to use (host-pointer single-limb path); rows = npoly * Lrows.
// phase: 0 = whole transform, 1 = column kernel only, 2 = tile kernel only (profiling aid, rh_ring_ntt_phase).
static int std_ntt_launch_span(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, bool inverse, bool lazy, int phase,
                               int Ls = 0, int Lso = 0) {   // Ls / Lso: rows per poly of the input / output block when > Lrows (0: Lrows / Ls): AtLevel views
  (void)hipGetLastError();                       // drop any stale error of an unrelated earlier call
  if (Ls && !Lso) Lso = Ls;                      // one stride given: both sides
  if (Lso && !Ls) Ls = Lrows;
  const int logN = r->logN, N = r->N;
  const size_t toff = (size_t)limb0 * N;
  const LimbConsts* c = r->d_consts + limb0;
  hipStream_t st = rh_stream(r);
  const unsigned rows = (unsigned)npoly * (unsigned)Lrows;
  if (rows == 0) return RH_OK;
  if (logN < LT) {
    if (!inverse) {
      if (lazy) ntt_fwd_small<MontPolicy><<<rows, 256, 0, st>>>(in, out, r->d_tw_fwd_mont + toff, c, Lrows, logN, 0, Ls, Lso);
      else      ntt_fwd_small<ShoupPolicy><<<rows, 256, 0, st>>>(in, out, r->d_tw_fwd + toff, c, Lrows, logN, 1, Ls, Lso);
    } else if (lazy && logN < 4 && r->d_tw_inv_mont && r->kind == RH_RING_STANDARD) {
      if (Ls || Lso) return rh_fail(RH_ERR_UNSUPPORTED, "strided non-canonical BackwardLazy of N = 8");          // (ntt_rows compacts this one shape)
      ntt_inv_small_lazy_mont<<<rows, 64, 0, st>>>(in, out, r->d_tw_inv_mont + toff, c, Lrows, logN);   // N = 8: BackwardLazy is not canonical
    } else {
      ntt_inv_small<<<rows, 256, 0, st>>>(in, out, r->d_tw_inv + toff, c, Lrows, logN, r->inv_scale ? 1 : 0, Ls, Lso);
    }
    return check_launch("ntt_small");
  }
  const int S1 = logN - LT;
  const unsigned tiles = rows << S1;
  const bool nt = r->nt_streams && (size_t)rows * (size_t)r->N * 8 >= ((size_t)512 << 20);   // non-temporal data streams beyond the Infinity Cache (see rh_streams_beyond_cache)
  if (phase == 0 && !(lazy && !inverse) && one_pass_ok(r)) {          // (the forward lazy form keeps the reference's representatives: Montgomery bodies, two passes)
    const size_t lds = ((size_t)LDS_WORDS * 8) << S1;
    const dim3 wg(256u << S1);
    const int sc = r->inv_scale ? 1 : 0;
    const unsigned grid = rows;
    if (!inverse) {
      if (S1 == 1 && nt) ntt_fwd_onepass_asm<1, true><<<grid, wg, lds, st>>>(in, out, r->d_tw_fwd + toff, r->d_twk_fwd + toff, c, Lrows, Ls, Lso);
      else if (S1 == 1) ntt_fwd_onepass_asm<1, false><<<grid, wg, lds, st>>>(in, out, r->d_tw_fwd + toff, r->d_twk_fwd + toff, c, Lrows, Ls, Lso);
      else if (nt) ntt_fwd_onepass_asm<2, true><<<grid, wg, lds, st>>>(in, out, r->d_tw_fwd + toff, r->d_twk_fwd + toff, c, Lrows, Ls, Lso);
      else ntt_fwd_onepass_asm<2, false><<<grid, wg, lds, st>>>(in, out, r->d_tw_fwd + toff, r->d_twk_fwd + toff, c, Lrows, Ls, Lso);
    } else {
      if (S1 == 1 && nt) ntt_inv_onepass_asm<1, true><<<grid, wg, lds, st>>>(in, out, r->d_twk_inv + toff, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Ls, Lso, sc);
      else if (S1 == 1) ntt_inv_onepass_asm<1, false><<<grid, wg, lds, st>>>(in, out, r->d_twk_inv + toff, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Ls, Lso, sc);
      else if (nt) ntt_inv_onepass_asm<2, true><<<grid, wg, lds, st>>>(in, out, r->d_twk_inv + toff, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Ls, Lso, sc);
      else ntt_inv_onepass_asm<2, false><<<grid, wg, lds, st>>>(in, out, r->d_twk_inv + toff, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Ls, Lso, sc);
    }
    return check_launch("ntt (one pass)");
  }
  if (!inverse) {
    const u64* src = in;
    if (S1 > 0) {
      dim3 g1(rows * 16);
      if (phase != 2) {
        if (lazy) launch_fwd_cols<MontPolicy>(S1, g1, st, in, out, r->d_tw_fwd_mont + toff, c, Lrows, logN, Ls, Lso);
        else if (S1 == 4 && r->asm_cols && r->asm_tile && nt) ntt_fwd_cols_asm<4, true><<<g1, 256, 0, st>>>(in, out, r->d_tw_fwd + toff, c, Lrows, Ls, Lso);
        else if (S1 == 3 && r->asm_cols && r->asm_tile && nt) ntt_fwd_cols_asm<3, true><<<g1, 256, 0, st>>>(in, out, r->d_tw_fwd + toff, c, Lrows, Ls, Lso);
        else if (S1 == 2 && r->asm_cols && r->asm_tile && nt) ntt_fwd_cols_asm<2, true><<<g1, 256, 0, st>>>(in, out, r->d_tw_fwd + toff, c, Lrows, Ls, Lso);
        else if (S1 == 4 && r->asm_cols && r->asm_tile) ntt_fwd_cols_asm<4><<<g1, 256, 0, st>>>(in, out, r->d_tw_fwd + toff, c, Lrows, Ls, Lso);
        else if (S1 == 3 && r->asm_cols && r->asm_tile) ntt_fwd_cols_asm<3><<<g1, 256, 0, st>>>(in, out, r->d_tw_fwd + toff, c, Lrows, Ls, Lso);
        else if (S1 == 2 && r->asm_cols && r->asm_tile) ntt_fwd_cols_asm<2><<<g1, 256, 0, st>>>(in, out, r->d_tw_fwd + toff, c, Lrows, Ls, Lso);
      else      launch_fwd_cols<ShoupPolicy>(S1, g1, st, in, out, r->d_tw_fwd + toff, c, Lrows, logN, Ls, Lso);
      }
      if (phase != 2) src = out;                   // phase 2 (tile stages only): the caller's `in` holds the column stages' output
    }
    if (phase != 1) {
      // the tile stages read what the column stages wrote (the OUTPUT block, its stride on both sides); without column stages (N = 4096) they go in -> out
      const int lso = Lso ? Lso : (Ls ? Ls : Lrows);
      const int ls = (src == out) ? lso : (Ls ? Ls : Lrows);
      if (lazy) ntt_fwd_tile<MontPolicy><<<tiles, 256, 0, st>>>(src, out, r->d_twk_fwd_mont + toff, c, Lrows, logN, 0, npoly, ls, lso);
      else if (r->asm_tile && nt) ntt_fwd_tile_asm<true><<<tiles, 256, 0, st>>>(src, out, r->d_twk_fwd + toff, c, Lrows, logN, npoly, ls, lso);
      else if (r->asm_tile) ntt_fwd_tile_asm<false><<<tiles, 256, 0, st>>>(src, out, r->d_twk_fwd + toff, c, Lrows, logN, npoly, ls, lso);
      else      ntt_fwd_tile<ShoupPolicy><<<tiles, 256, 0, st>>>(src, out, r->d_twk_fwd + toff, c, Lrows, logN, 1, npoly, ls, lso);
    }
  } else {
    if (phase != 1) {
      // the hand-scheduled body leaves values < 4q unscaled: right whenever column stages follow, and for N = 4096
      // sub-rings of the 3N transform (inv_scale = false), which scale in their own last layer; N = 4096 WITH scaling: the same body, then N^-1 out of LDS
      if (r->asm_tile && S1 == 0 && r->inv_scale && r->one_pass && phase == 0) {
        if (nt) ntt_inv_onepass_asm<0, true><<<rows, 256, (size_t)LDS_WORDS * 8, st>>>(in, out, r->d_twk_inv + toff, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Ls, Lso, 1);
        else ntt_inv_onepass_asm<0, false><<<rows, 256, (size_t)LDS_WORDS * 8, st>>>(in, out, r->d_twk_inv + toff, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Ls, Lso, 1);
        return check_launch("ntt (N = 4096, inverse)");
      }
      if (r->asm_tile && (S1 > 0 || !r->inv_scale) && nt) ntt_inv_tile_asm<true><<<tiles, 256, 0, st>>>(in, out, r->d_twk_inv + toff, c, Lrows, logN, npoly, Ls, Lso);
      else if (r->asm_tile && (S1 > 0 || !r->inv_scale)) ntt_inv_tile_asm<false><<<tiles, 256, 0, st>>>(in, out, r->d_twk_inv + toff, c, Lrows, logN, npoly, Ls, Lso);
      else ntt_inv_tile<<<tiles, 256, 0, st>>>(in, out, r->d_twk_inv + toff, c, Lrows, logN, (S1 == 0 && r->inv_scale) ? 1 : 0, npoly, Ls, Lso);
    }
    const bool acols = phase != 2 && r->asm_cols && r->asm_tile && r->inv_scale;
    if (S1 == 4 && acols && nt) ntt_inv_cols_asm<4, true><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Lso);
    else if (S1 == 3 && acols && nt) ntt_inv_cols_asm<3, true><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Lso);
    else if (S1 == 2 && acols && nt) ntt_inv_cols_asm<2, true><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Lso);
    else if (S1 == 4 && acols) ntt_inv_cols_asm<4><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Lso);
    else if (S1 == 3 && acols) ntt_inv_cols_asm<3><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Lso);
    else if (S1 == 2 && acols) ntt_inv_cols_asm<2><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, Lso);
    else if (S1 > 0 && phase != 2) launch_inv_cols(S1, dim3(rows * 16), st, out, r->d_tw_inv + toff, r->d_lastw + limb0, c, Lrows, logN, r->inv_scale ? 1 : 0, Lso);
  }
  return check_launch("ntt");
}

template <int S1>
static void launch_fused(rh_ring* r, const u64* in1, u64* out1, unsigned n1, u64* data2, unsigned n2, int npoly2,
                         size_t toff, const LimbConsts* c, int Lrows) {
  const unsigned grid = n1 > n2 ? n1 : n2;
  hipStream_t st = rh_stream(r);
  if (r->asm_tile && S1 >= 2 && S1 <= 4 && r->asm_cols)
    if (r->nt_streams) ntt_fwd_fused_asm<S1, true, true><<<grid, 256, 0, st>>>(in1, out1, n1, data2, n2, npoly2, r->d_tw_fwd + toff, r->d_twk_fwd + toff, c, Lrows, r->logN);
    else ntt_fwd_fused_asm<S1, true, false><<<grid, 256, 0, st>>>(in1, out1, n1, data2, n2, npoly2, r->d_tw_fwd + toff, r->d_twk_fwd + toff, c, Lrows, r->logN);
  else if (r->asm_tile)
    ntt_fwd_fused_asm<S1, false><<<grid, 256, 0, st>>>(in1, out1, n1, data2, n2, npoly2, r->d_tw_fwd + toff, r->d_twk_fwd + toff, c, Lrows, r->logN);
  else
    ntt_fwd_fused<ShoupPolicy, S1><<<grid, 256, 0, st>>>(in1, out1, n1, data2, n2, npoly2, r->d_tw_fwd + toff, r->d_twk_fwd + toff,
                                                      c, Lrows, r->logN, 1);
}

// Forward canonical transform of a large batch: software pipeline over spans of `chunk` polys in ONE stream; launch j
// runs the column stages of span j fused with the tile stages of span j-1 (ntt_fwd_fused).  The spans may come from several
// blocks (rh_ring_ntt_many): the pipeline runs through the block boundaries, so only the first launch of the first block and
// the last launch of the last block are not fused.
struct NttSeg { const u64* in; u64* out; int npoly; };
static int std_ntt_fwd_pipelined_segs(rh_ring* r, const NttSeg* segs, int nseg, int Lrows, int limb0, int chunk) {
  (void)hipGetLastError();
  const int N = r->N, S1 = r->logN - LT;
  const size_t toff = (size_t)limb0 * N, stride = (size_t)Lrows * N;
  const LimbConsts* c = r->d_consts + limb0;
  std::vector<NttSeg> spans;
  for (int k = 0; k < nseg; ++k)
    for (int p = 0; p < segs[k].npoly; p += chunk)
      spans.push_back(NttSeg{segs[k].in + (size_t)p * stride, segs[k].out + (size_t)p * stride, segs[k].npoly - p < chunk ? segs[k].npoly - p : chunk});
  const int nspans = (int)spans.size();
  for (int j = 0; j <= nspans; ++j) {
    const int n1p = j < nspans ? spans[j].npoly : 0, n2p = j >= 1 ? spans[j - 1].npoly : 0;
    const unsigned n1 = (unsigned)n1p * Lrows * 16, n2 = ((unsigned)n2p * Lrows) << S1;
    const u64* i1 = j < nspans ? spans[j].in : nullptr; u64* o1 = j < nspans ? spans[j].out : nullptr; u64* d2 = j >= 1 ? spans[j - 1].out : nullptr;
    switch (S1) {
      case 1: launch_fused<1>(r, i1, o1, n1, d2, n2, n2p, toff, c, Lrows); break;
      case 2: launch_fused<2>(r, i1, o1, n1, d2, n2, n2p, toff, c, Lrows); break;
      case 3: launch_fused<3>(r, i1, o1, n1, d2, n2, n2p, toff, c, Lrows); break;
      case 4: launch_fused<4>(r, i1, o1, n1, d2, n2, n2p, toff, c, Lrows); break;
      case 5: launch_fused<5>(r, i1, o1, n1, d2, n2, n2p, toff, c, Lrows); break;
    }
  }
  return check_launch("ntt_fwd_fused");
}
static int std_ntt_fwd_pipelined(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, int chunk) {
  const NttSeg seg{in, out, npoly};
  return std_ntt_fwd_pipelined_segs(r, &seg, 1, Lrows, limb0, chunk);
}

template <int S1>
static void launch_inv_fused(rh_ring* r, const u64* in1, const u64* in1b, u64* out1, unsigned n1, int npoly1, u64* data2, unsigned n2,
                             size_t toff, int limb0, const LimbConsts* c, int Lrows) {
  const unsigned grid = n1 > n2 ? n1 : n2;
  hipStream_t st = rh_stream(r);
  const bool acols = S1 >= 2 && S1 <= 4 && r->asm_cols;
  if (in1b) {                 // rh_ring_intt_mul: product on load, 2^64-scaled N^-1 constants (c = d_consts_r + limb0)
    const tw2* lw = r->d_lastw_r + limb0;
    if (acols) ntt_inv_fused_asm<S1, true, true><<<grid, 256, 0, st>>>(in1, in1b, out1, n1, npoly1, data2, n2, r->d_twk_inv + toff, r->d_tw_inv + toff, lw, c, Lrows, r->logN);
    else ntt_inv_fused_asm<S1, false, true><<<grid, 256, 0, st>>>(in1, in1b, out1, n1, npoly1, data2, n2, r->d_twk_inv + toff, r->d_tw_inv + toff, lw, c, Lrows, r->logN);
    return;
  }
  const tw2* lw = r->d_lastw + limb0;
  if (acols && !r->nt_streams) ntt_inv_fused_asm<S1, true, false, false><<<grid, 256, 0, st>>>(in1, nullptr, out1, n1, npoly1, data2, n2, r->d_twk_inv + toff, r->d_tw_inv + toff, lw, c, Lrows, r->logN);
  else if (acols) ntt_inv_fused_asm<S1, true, false><<<grid, 256, 0, st>>>(in1, nullptr, out1, n1, npoly1, data2, n2, r->d_twk_inv + toff, r->d_tw_inv + toff, lw, c, Lrows, r->logN);
  else ntt_inv_fused_asm<S1, false, false><<<grid, 256, 0, st>>>(in1, nullptr, out1, n1, npoly1, data2, n2, r->d_twk_inv + toff, r->d_tw_inv + toff, lw, c, Lrows, r->logN);
}
// Inverse transform of a large batch: launch j = tile stages of span j fused with column stages (+ N^-1) of span j-1.
static int std_ntt_inv_pipelined(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, int chunk, const u64* in_b = nullptr) {
  (void)hipGetLastError();
  const int N = r->N, S1 = r->logN - LT;
  const size_t toff = (size_t)limb0 * N, stride = (size_t)Lrows * N;
  const LimbConsts* c = (in_b ? r->d_consts_r : r->d_consts) + limb0;
  const int nspans = (npoly + chunk - 1) / chunk;
  for (int j = 0; j <= nspans; ++j) {
    const int p1 = j * chunk, n1p = j < nspans ? ((npoly - p1 < chunk) ? npoly - p1 : chunk) : 0;
    const int p2 = (j - 1) * chunk, n2p = j >= 1 ? ((npoly - p2 < chunk) ? npoly - p2 : chunk) : 0;
    const unsigned n1 = ((unsigned)n1p * Lrows) << S1, n2 = (unsigned)n2p * Lrows * 16;
    const u64* i1 = in + (size_t)p1 * stride; u64* o1 = out + (size_t)p1 * stride; u64* d2 = out + (size_t)p2 * stride;
    const u64* ib = in_b ? in_b + (size_t)p1 * stride : nullptr;
    switch (S1) {
      case 1: launch_inv_fused<1>(r, i1, ib, o1, n1, n1p, d2, n2, toff, limb0, c, Lrows); break;
      case 2: launch_inv_fused<2>(r, i1, ib, o1, n1, n1p, d2, n2, toff, limb0, c, Lrows); break;
      case 3: launch_inv_fused<3>(r, i1, ib, o1, n1, n1p, d2, n2, toff, limb0, c, Lrows); break;
      case 4: launch_inv_fused<4>(r, i1, ib, o1, n1, n1p, d2, n2, toff, limb0, c, Lrows); break;
      case 5: launch_inv_fused<5>(r, i1, ib, o1, n1, n1p, d2, n2, toff, limb0, c, Lrows); break;
    }
  }
  return check_launch("ntt_inv_fused_asm");
}

int rh_std_ntt_launch(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, bool inverse, bool lazy, int phase) {
  int chunk = r->chunk_polys;
  if (chunk < 0) {                                  // auto: pipeline batches of more than ~2048 limb rows in spans of ~2048 rows
    const int c = r->auto_span_rows / (Lrows > 0 ? Lrows : 1) > 0 ? r->auto_span_rows / Lrows : 1;   // 128 polys at 16 limbs (measured optimum: 64..128)
    chunk = npoly > c ? c : 0;
  }
  const bool two_pass = r->logN > LT && !one_pass_ok(r);
  if (chunk > 0 && two_pass && phase == 0 && !inverse && !lazy && npoly > chunk)
    return std_ntt_fwd_pipelined(r, in, out, npoly, Lrows, limb0, chunk);
  if (chunk > 0 && two_pass && phase == 0 && inverse && r->asm_tile && r->inv_scale && npoly > chunk)
    return std_ntt_inv_pipelined(r, in, out, npoly, Lrows, limb0, chunk);
  return std_ntt_launch_span(r, in, out, npoly, Lrows, limb0, inverse, lazy, phase);
}

// Forward canonical transform, in place, of limbs [limb0, limb0 + Lrows) of each poly of a block that has Ls >= Lrows
// rows per poly (`data` points at row limb0 of poly 0): lets a caller skip rows it will overwrite anyway (the digit's own
// limbs in DecomposeSingleNTT).  N >= 4096 standard rings; the pipelined launches are not involved.
int rh_std_ntt_fwd_strided(rh_ring* r, u64* data, int npoly, int Lrows, int limb0, int Ls) {
  if (r->logN < LT || Ls < Lrows) return rh_fail(RH_ERR_ARG, "strided transform needs N >= 4096 and a row stride >= the row count");
  if (Lrows <= 0 || npoly <= 0) return RH_OK;
  return std_ntt_launch_span(r, data, data, npoly, Lrows, limb0, false, false, 0, Ls);
}

// Forward canonical transform, in place, of the non-digit limbs of every digit block of a hybrid decomposition
// (DecomposeSingleNTT, core/rlwe/evaluator_gadget_product.go:455-478, for all digits at once): block j holds npoly polys
// of LQ rows and its limbs [j*LP, min((j+1)*LP, LQ)) are left alone.  One software-pipelined stream of launches instead of
// two strided transforms (before / after the digit's limbs) per digit.  Needs the hand-scheduled bodies (N = 2^14 .. 2^16).
// lazy_out: the outputs stay < 8q, not canonical (for a consumer that reduces anyway, never for a caller-visible block).
bool rh_can_ntt_digits(const rh_ring* r) {
  const int S1 = r->logN - LT;
  return r->kind == RH_RING_STANDARD && r->asm_tile && r->asm_cols && r->digit_pipeline && S1 >= 2 && S1 <= 4;
}
// Small batches: every block of ring r -- and, when r2 is given, every (gap-free) block of a second ring of the same degree: the P blocks of a key switch
// beside its Q blocks -- in ONE launch of the column stages and one of the tile stages (blockIdx.y = block).  At most 8 blocks per ring.
int rh_std_ntt_fwd_blocks_small(rh_ring* r, u64* data, size_t block_stride, int nblocks, int Ls, const int* gap0, const int* gap_len,
                                rh_ring* r2, u64* data2, size_t block_stride2, int nblocks2, int Ls2, int npoly, bool lazy_out) {
  if (!rh_can_ntt_digits(r) || (r2 && (!rh_can_ntt_digits(r2) || r2->logN != r->logN)) || nblocks > 8 || nblocks2 > 8)
    return rh_fail(RH_ERR_UNSUPPORTED, "small-batch block transform: hand-scheduled bodies, one ring degree, at most 8 blocks per ring");
  if (npoly <= 0 || nblocks <= 0) return RH_OK;
  (void)hipGetLastError();
  const int S1 = r->logN - LT;
  hipStream_t st = rh_stream(r);
  unsigned n1 = 0, n2 = 0;
  auto fill = [&](BlockSet& bs, rh_ring* R, u64* d, size_t stride, int nb, int rows, const int* g0, const int* gl, bool cols) {
    memset(&bs, 0, sizeof bs);
    if (!R) return;
    bs.data = d; bs.stride = stride; bs.tw = cols ? R->d_tw_fwd : R->d_twk_fwd; bs.consts = R->d_consts; bs.nblocks = nb; bs.g.Ls = rows;
    for (int j = 0; j < nb; ++j) {
      const int gl_j = gl ? gl[j] : 0;
      bs.g.L[j] = rows - gl_j; bs.g.gap0[j] = (u32)(g0 ? g0[j] : 0); bs.g.gap_len[j] = (u32)gl_j;
      const unsigned a = (unsigned)npoly * bs.g.L[j] * 16, b = ((unsigned)npoly * bs.g.L[j]) << S1;
      if (a > n1) n1 = a;
      if (b > n2) n2 = b;
    }
  };
  BlockSet ca, cb, ta, tb;
  fill(ca, r, data, block_stride, nblocks, Ls, gap0, gap_len, true); fill(cb, r2, data2, block_stride2, nblocks2, Ls2, nullptr, nullptr, true);
  fill(ta, r, data, block_stride, nblocks, Ls, gap0, gap_len, false); fill(tb, r2, data2, block_stride2, nblocks2, Ls2, nullptr, nullptr, false);
  if (!n1) return RH_OK;
  const unsigned ny = (unsigned)(nblocks + (r2 ? nblocks2 : 0));
#define RH_BLK(S) do { ntt_fwd_cols_blocks_asm<S><<<dim3(n1, ny), 256, 0, st>>>(ca, cb, npoly);                                  \
                       if (lazy_out) ntt_fwd_tile_blocks_asm<S, true><<<dim3(n2, ny), 256, 0, st>>>(ta, tb, npoly);                \
                       else ntt_fwd_tile_blocks_asm<S, false><<<dim3(n2, ny), 256, 0, st>>>(ta, tb, npoly); } while (0)
  switch (S1) { case 2: RH_BLK(2); break; case 3: RH_BLK(3); break; case 4: RH_BLK(4); break; }
#undef RH_BLK
  return check_launch("ntt_fwd_blocks (small batch)");
}
int rh_std_ntt_fwd_blocks(rh_ring* r, u64* data, size_t block_stride, int npoly, int nblocks, int Ls, const int* gap0, const int* gap_len,
                          bool lazy_out, int small) {       // small: 1 / 0 = every block in one launch pair / the pipelined stream; -1 = by the ring's ks_small_rows                  // block j: npoly polys of Ls rows, rows [gap0[j], gap0[j] + gap_len[j]) left alone
  if (!rh_can_ntt_digits(r)) return rh_fail(RH_ERR_UNSUPPORTED, "digit-block transform needs the hand-scheduled bodies (2^14 <= N <= 2^16)");
  if (npoly <= 0 || nblocks <= 0) return RH_OK;
  (void)hipGetLastError();
  const int S1 = r->logN - LT;
  hipStream_t st = rh_stream(r);
  auto rows = [&](int j) {                          // block j: its transformed rows
    GapRows g; g.Ls = Ls; g.gap0 = (u32)gap0[j]; g.gap_len = (u32)gap_len[j]; g.L = Ls - gap_len[j];
    return g;
  };
  if (small < 0) small = r->ks_small_rows > 0 && (long)npoly * Ls <= r->ks_small_rows;
  if (small && nblocks <= 8) return rh_std_ntt_fwd_blocks_small(r, data, block_stride, nblocks, Ls, gap0, gap_len, nullptr, nullptr, 0, 0, 0, npoly, lazy_out);
  for (int j = 0; j <= nblocks; ++j) {
    GapRows g1 = j < nblocks ? rows(j) : GapRows{1, 1, 0, 0}, g2 = j >= 1 ? rows(j - 1) : GapRows{1, 1, 0, 0};
    const unsigned n1 = j < nblocks ? (unsigned)npoly * g1.L * 16 : 0, n2 = j >= 1 ? ((unsigned)npoly * g2.L) << S1 : 0;
    const unsigned grid = n1 > n2 ? n1 : n2;
    if (!grid) continue;
    if (!g1.L) g1.L = 1;
    if (!g2.L) g2.L = 1;
    u64* d1 = data + (size_t)(j < nblocks ? j : 0) * block_stride; u64* d2 = data + (size_t)(j >= 1 ? j - 1 : 0) * block_stride;
    // non-temporal data streams once a block is too large to be re-read from the Infinity Cache by the next launch's tile stages
#define RH_GAP2(S, Z) do { if (nt) ntt_fwd_fused_gap_asm<S, Z, true><<<grid, 256, 0, st>>>(d1, n1, g1, d2, n2, npoly, g2, r->d_tw_fwd, r->d_twk_fwd, r->d_consts); \
                           else ntt_fwd_fused_gap_asm<S, Z, false><<<grid, 256, 0, st>>>(d1, n1, g1, d2, n2, npoly, g2, r->d_tw_fwd, r->d_twk_fwd, r->d_consts); } while (0)
    const bool nt = r->nt_streams && (size_t)npoly * (size_t)Ls * (size_t)r->N * 8 >= ((size_t)256 << 20);
    switch (S1) {
      case 2: if (lazy_out) RH_GAP2(2, true); else RH_GAP2(2, false); break;
      case 3: if (lazy_out) RH_GAP2(3, true); else RH_GAP2(3, false); break;
      case 4: if (lazy_out) RH_GAP2(4, true); else RH_GAP2(4, false); break;
    }
#undef RH_GAP2
  }
  return check_launch("ntt_fwd_fused_gap_asm");
}
int rh_std_ntt_fwd_digits(rh_ring* r, u64* data, size_t digit_stride, int npoly, int beta, int LQ, int LP, bool lazy_out, int small) {
  if (beta <= 0) return RH_OK;
  std::vector<int> g0(beta), gl(beta);
  for (int j = 0; j < beta; ++j) {                  // digit j skips its own limbs [j LP, min((j+1) LP, LQ))
    int n = LQ - j * LP; if (n > LP) n = LP; if (n < 0) n = 0;
    g0[j] = j * LP; gl[j] = n;
  }
  return rh_std_ntt_fwd_blocks(r, data, digit_stride, npoly, beta, LQ, g0.data(), gl.data(), lazy_out, small);
}

// Inverse canonical transform of ONE limb of every poly of a block with in_rows limbs per poly into a dense block of npoly rows
// (the last limb of a rescale step): the tile kernel reads the limb where it lies -- no gather copy.  Hand-scheduled bodies only.
bool rh_can_intt_limb_strided(const rh_ring* r) {
  const int S1 = r->logN - LT;
  return r->kind == RH_RING_STANDARD && r->asm_tile && r->asm_cols && r->inv_scale && S1 >= 2 && S1 <= 4;
}
int rh_std_intt_limb_strided(rh_ring* r, const u64* in, int in_rows, int limb, u64* out, int npoly) {
  if (!rh_can_intt_limb_strided(r)) return rh_fail(RH_ERR_UNSUPPORTED, "strided single-limb inverse transform needs the hand-scheduled bodies");
  if (npoly <= 0) return RH_OK;
  (void)hipGetLastError();
  const int S1 = r->logN - LT;
  const size_t toff = (size_t)limb * r->N;
  hipStream_t st = rh_stream(r);
  ntt_inv_tile_asm<<<(unsigned)npoly << S1, 256, 0, st>>>(in + toff, out, r->d_twk_inv + toff, r->d_consts + limb, 1, r->logN, npoly, in_rows, 0);
  const dim3 g((unsigned)npoly * 16);
  if (S1 == 4) ntt_inv_cols_asm<4><<<g, 256, 0, st>>>(out, r->d_tw_inv + toff, r->d_lastw + limb, r->d_consts + limb, 1, 0);
  else if (S1 == 3) ntt_inv_cols_asm<3><<<g, 256, 0, st>>>(out, r->d_tw_inv + toff, r->d_lastw + limb, r->d_consts + limb, 1, 0);
  else ntt_inv_cols_asm<2><<<g, 256, 0, st>>>(out, r->d_tw_inv + toff, r->d_lastw + limb, r->d_consts + limb, 1, 0);
  return check_launch("strided single-limb inverse transform");
}

// Inverse canonical transform of limbs 0..Lrows-1 of every poly of a block with in_rows limbs per poly into a block with out_rows limbs per
// poly (ring.AtLevel views of larger polys): one batched launch pair.  Hand-scheduled bodies only (N = 2^14 .. 2^16).
int rh_std_intt_rows(rh_ring* r, const u64* in, int in_rows, u64* out, int out_rows, int npoly, int Lrows) {
  if (!rh_can_intt_limb_strided(r)) return rh_fail(RH_ERR_UNSUPPORTED, "strided inverse transform needs the hand-scheduled bodies");
  if (npoly <= 0 || Lrows <= 0) return RH_OK;
  (void)hipGetLastError();
  const int S1 = r->logN - LT;
  hipStream_t st = rh_stream(r);
  const unsigned rows = (unsigned)npoly * (unsigned)Lrows;
  ntt_inv_tile_asm<<<rows << S1, 256, 0, st>>>(in, out, r->d_twk_inv, r->d_consts, Lrows, r->logN, npoly, in_rows, out_rows);
  const dim3 g(rows * 16);
  if (S1 == 4) ntt_inv_cols_asm<4><<<g, 256, 0, st>>>(out, r->d_tw_inv, r->d_lastw, r->d_consts, Lrows, out_rows);
  else if (S1 == 3) ntt_inv_cols_asm<3><<<g, 256, 0, st>>>(out, r->d_tw_inv, r->d_lastw, r->d_consts, Lrows, out_rows);
  else ntt_inv_cols_asm<2><<<g, 256, 0, st>>>(out, r->d_tw_inv, r->d_lastw, r->d_consts, Lrows, out_rows);
  return check_launch("strided inverse transform");
}

// ---- 3N transform (ntt3n.hip), b = 1: the hand-scheduled layer kernels live in this translation unit with the tile bodies they fuse with
void rh_3n_launch_layer(bool inverse, int S1, unsigned nblocks, hipStream_t st, const u64* in, u64* out, const N3Layer& a, bool nt_streams) {
  // a unit moves 6 * 2^S1 coefficients per thread: nblocks * 256 * 6 * 2^S1 * 8 bytes per direction; non-temporal streams beyond 512 MiB
  const bool nt = nt_streams && (size_t)nblocks * 256 * 6 * ((size_t)8 << S1) >= ((size_t)512 << 20);   // tuning nt_streams = 0: default policy everywhere
#define RH_3NL2(S, I) do { if (nt) ntt3n_layer_asm<S, I, true><<<nblocks, 256, 0, st>>>(in, out, a); else ntt3n_layer_asm<S, I, false><<<nblocks, 256, 0, st>>>(in, out, a); } while (0)
#define RH_3NL(S) do { if (inverse) RH_3NL2(S, true); else RH_3NL2(S, false); } while (0)
  if (S1 == 1) RH_3NL(1); else if (S1 == 2) RH_3NL(2); else RH_3NL(3);
#undef RH_3NL
#undef RH_3NL2
}
bool rh_can_fuse_submul(const rh_ring* r) { return r->kind == RH_RING_STANDARD && r->logN >= LT && r->fuse_submul; }
// Forward canonical transform of `buf` (in place up to its tile stages) fused with out = MRed(2q - y + NTT(buf), s_limb):
// column stages as usual, then ntt_fwd_tile_submul.  y / out: (poly, limb) blocks with y_rows / out_rows limbs per poly.
// Cache policy of a launch's data streams: non-temporal once the rows it moves exceed twice the 256 MiB Infinity Cache (the generated
// bodies exist in both forms, tools/gen_tile_asm.py; smaller working sets are re-read from the caches and run 3-6 % slower with nt)
static bool rh_streams_beyond_cache(const rh_ring* r, unsigned rows) { return r->nt_streams && (size_t)rows * (size_t)r->N * 8 >= ((size_t)512 << 20); }

// rescale: column stages of limbs 0..Lrows-1 fed by the re-expansion of the coefficient-domain last limb `tmp` (N >= 8192)
int rh_std_ntt_expand_cols_launch(rh_ring* r, const u64* tmp, u64* buf, int npoly, int Lrows, const void* table_dev, int mode, u64 qL) {
  const int S1 = r->logN - LT;
  if (S1 < 1 || S1 > 5) return rh_fail(RH_ERR_UNSUPPORTED, "expand + column stages need 8192 <= N <= 2^17");
  const unsigned rows = (unsigned)npoly * (unsigned)Lrows;
  if (rows == 0) return RH_OK;
  (void)hipGetLastError();
  const RescaleLimb* T = (const RescaleLimb*)table_dev;
  const dim3 g(rows * 16);
  bool lazy_ok = r->asm_tile && r->asm_cols && S1 >= 2 && S1 <= 4;      // hand-scheduled body: needs qL + q <= 8q for every limb
  for (int i = 0; i < Lrows && lazy_ok; ++i) lazy_ok = qL / 7 <= r->moduli[i] && qL - 1 < 7 * r->moduli[i];
  if (lazy_ok) {
    const bool nt = rh_streams_beyond_cache(r, rows);          // non-temporal data streams for working sets far beyond the Infinity Cache
    switch (S1 * 2 + (nt ? 1 : 0)) {
      case 4: ntt_fwd_cols_expand_asm<2, false><<<g, 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, mode, qL); break;
      case 5: ntt_fwd_cols_expand_asm<2, true><<<g, 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, mode, qL); break;
      case 6: ntt_fwd_cols_expand_asm<3, false><<<g, 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, mode, qL); break;
      case 7: ntt_fwd_cols_expand_asm<3, true><<<g, 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, mode, qL); break;
      case 8: ntt_fwd_cols_expand_asm<4, false><<<g, 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, mode, qL); break;
      case 9: ntt_fwd_cols_expand_asm<4, true><<<g, 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, mode, qL); break;
    }
    return check_launch("ntt_fwd_cols_expand_asm");
  }
  switch (S1) {
    case 1: ntt_fwd_cols_expand<1><<<g, 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, r->logN, mode, qL); break;
    case 2: ntt_fwd_cols_expand<2><<<g, 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, r->logN, mode, qL); break;
    case 3: ntt_fwd_cols_expand<3><<<g, 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, r->logN, mode, qL); break;
    case 4: ntt_fwd_cols_expand<4><<<g, 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, r->logN, mode, qL); break;
    case 5: ntt_fwd_cols_expand<5><<<dim3(rows * 16), 256, 0, rh_stream(r)>>>(tmp, buf, r->d_tw_fwd, r->d_consts, T, Lrows, r->logN, mode, qL); break;
  }
  return check_launch("ntt_fwd_cols_expand");
}

// Both components of a ModDown in ONE launch: buf holds 2 * npoly polys (component 0 then component 1, column stages done), component c goes with
// (y_c, out_c, z_c); z0 and z1 are both given or both null.  Hand-scheduled bodies only (the caller checks rh_can_fuse_submul and asm_tile).
int rh_std_ntt_submul_launch_pair(rh_ring* r, u64* buf, int npoly, int Lrows, const u64* y0, const u64* y1, int y_rows, u64* out0, u64* out1, int out_rows,
                                  const u64* scalars_host, const u64* z0, const u64* z1, int z_rows) {
  if (!rh_can_fuse_submul(r) || !r->asm_tile || (z0 == nullptr) != (z1 == nullptr)) return rh_fail(RH_ERR_UNSUPPORTED, "paired subtract-multiply: hand-scheduled bodies, both addends or none");
  const unsigned rows = 2u * (unsigned)npoly * (unsigned)Lrows;
  if (rows == 0) return RH_OK;
  const int S1 = r->logN - LT;
  (void)hipGetLastError();
  LimbShoup sh; memset(&sh, 0, sizeof(sh));
  for (int i = 0; i < Lrows; ++i) {
    const u64 q = r->moduli[i];
    sh.w[i] = rh::imform(scalars_host[i] % q, q); sh.wp[i] = rh::shoup_quotient(sh.w[i], q);
  }
  const bool nt = rh_streams_beyond_cache(r, rows / 2);       // (the policy of one component's launch, as before)
  hipStream_t st = rh_stream(r);
#define RH_SMP(ADD, NTF) ntt_fwd_tile_submul_asm<ADD, NTF><<<rows << S1, 256, 0, st>>>(buf, r->d_twk_fwd, r->d_consts, Lrows, r->logN, 2 * npoly, y0, y_rows, out0, out_rows, sh, z0, z_rows, npoly, y1, out1, z1)
  if (z0 && nt) RH_SMP(true, true); else if (z0) RH_SMP(true, false); else if (nt) RH_SMP(false, true); else RH_SMP(false, false);
#undef RH_SMP
  return check_launch("ntt_fwd_tile_submul_asm (pair)");
}
int rh_std_ntt_submul_launch(rh_ring* r, u64* buf, int npoly, int Lrows, int limb0, const u64* y, int y_rows, u64* out, int out_rows,
                             const u64* scalars_host, bool cols_done, const u64* z, int z_rows) {
  static_assert(RH_MAX_LIMBS_K == RH_MAX_LIMBS, "limb bound mismatch");
  if (!rh_can_fuse_submul(r)) return rh_fail(RH_ERR_UNSUPPORTED, "fused subtract-multiply needs a standard ring with N >= 4096");
  const unsigned rows = (unsigned)npoly * (unsigned)Lrows;
  if (rows == 0) return RH_OK;
  const int S1 = r->logN - LT;
  if (S1 > 0 && !cols_done) if (int rc = rh_std_ntt_launch(r, buf, buf, npoly, Lrows, limb0, false, false, 1)) return rc;   // column stages only
  (void)hipGetLastError();
  const size_t toff = (size_t)limb0 * r->N;
  if (r->asm_tile) {                                   // MRed by the limb's scalar = Shoup multiply by s * 2^-64 mod q
    LimbShoup sh; memset(&sh, 0, sizeof(sh));
    for (int i = 0; i < Lrows; ++i) {
      const u64 q = r->moduli[limb0 + i];
      sh.w[i] = rh::imform(scalars_host[i] % q, q); sh.wp[i] = rh::shoup_quotient(sh.w[i], q);
    }
    const bool nt = rh_streams_beyond_cache(r, rows);
    if (z && nt) ntt_fwd_tile_submul_asm<true, true><<<rows << S1, 256, 0, rh_stream(r)>>>(buf, r->d_twk_fwd + toff, r->d_consts + limb0, Lrows, r->logN, npoly, y, y_rows,
                                                                                               out, out_rows, sh, z, z_rows);
    else if (z) ntt_fwd_tile_submul_asm<true, false><<<rows << S1, 256, 0, rh_stream(r)>>>(buf, r->d_twk_fwd + toff, r->d_consts + limb0, Lrows, r->logN, npoly, y, y_rows,
                                                                                               out, out_rows, sh, z, z_rows);
    else if (nt) ntt_fwd_tile_submul_asm<false, true><<<rows << S1, 256, 0, rh_stream(r)>>>(buf, r->d_twk_fwd + toff, r->d_consts + limb0, Lrows, r->logN, npoly, y, y_rows,
                                                                                                out, out_rows, sh, nullptr, 0);
    else ntt_fwd_tile_submul_asm<false, false><<<rows << S1, 256, 0, rh_stream(r)>>>(buf, r->d_twk_fwd + toff, r->d_consts + limb0, Lrows, r->logN, npoly, y, y_rows,
                                                                                         out, out_rows, sh, nullptr, 0);
    return check_launch("ntt_fwd_tile_submul_asm");
  }
  LimbScalars sc; memset(&sc, 0, sizeof(sc)); memcpy(sc.s, scalars_host, (size_t)Lrows * 8);
  ntt_fwd_tile_submul<<<rows << S1, 256, 0, rh_stream(r)>>>(buf, r->d_twk_fwd + toff, r->d_consts + limb0, Lrows, r->logN, npoly, y, y_rows,
                                                         out, out_rows, sc, z, z_rows);
  return check_launch("ntt_fwd_tile_submul");
}

// ---- conjugate-invariant ring Z[X+X^-1]/(X^2N+1) (ring/ntt.go:716-1311): the negacyclic kernels on the re-indexed
// 4N-th-root table, plus the fold with F = roots[1] before (forward :761-768) / after (inverse :1149-1156) them.
__global__ void __launch_bounds__(256)
ci_fold_kernel(const u64* in, u64* out, int logN, const CiFold* __restrict__ fold, const LimbConsts* __restrict__ consts, int L, int inverse) {
  const u32 N = 1u << logN;
  const u32 limb = blockIdx.x % (u32)L;
  const LimbConsts c = consts[limb];
  const tw2 F = inverse ? fold[limb].b : fold[limb].f;
  const u64 q4 = 4 * c.q;
  const size_t base = (size_t)blockIdx.x << logN;
  for (u32 jx = blockIdx.y * blockDim.x + threadIdx.x; jx <= (N >> 1); jx += gridDim.y * blockDim.x) {
    if (jx == 0) {
      const u64 x = in[base];
      out[base] = inverse ? cred(2 * csub(csub(x, 2 * c.q), c.q), c.q) : x;          // p2[0] = CRed(p2[0] << 1) (:1157) / p1[0] (:770)
    } else if (jx == (N >> 1)) {
      const u64 x = csub(in[base + jx], q4);
      const u64 v = x + q4 - shoup_mul(x, F.w, F.wp, c.nq);
      out[base + jx] = inverse ? canon8(v, c.q) : v;
    } else {
      const u32 jy = N - jx;
      const u64 a = csub(in[base + jx], q4), b = csub(in[base + jy], q4);
      const u64 va = a + q4 - shoup_mul(b, F.w, F.wp, c.nq), vb = b + q4 - shoup_mul(a, F.w, F.wp, c.nq);
      out[base + jx] = inverse ? canon8(va, c.q) : va;
      out[base + jy] = inverse ? canon8(vb, c.q) : vb;
    }
  }
}
static int ci_ntt_launch(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, bool inverse) {
  const unsigned rows = (unsigned)npoly * Lrows;
  if (!rows) return RH_OK;
  unsigned chunks = ((unsigned)r->N / 2 + 256) / 256; if (chunks > 64) chunks = 64;
  (void)hipGetLastError();
  const int S1 = r->logN - LT;
  if (r->asm_tile && r->asm_cols && r->fuse_ci && S1 >= 2 && S1 <= 4) {
    // the fold rides in the column stages (a thread owns the column pair (c, 4096 - c) the fold couples; column 0 apart): no pass of its own
    hipStream_t st = rh_stream(r);
    const size_t toff = (size_t)limb0 * r->N;
    const CiFoldTw* cf = reinterpret_cast<const CiFoldTw*>(r->d_cifold + limb0);
    const LimbConsts* c = r->d_consts + limb0;
    // large batches: the software pipeline of the standard ring's fused launches (spans of ~2048 rows): forward launch j = fold + column stages of
    // span j with the tile stages of span j-1; inverse launch j = tile stages of span j with the column stages + fold of span j-1 (round 3)
    int chunk = r->chunk_polys;
    if (chunk < 0) { const int c0 = r->auto_span_rows / Lrows > 0 ? r->auto_span_rows / Lrows : 1; chunk = npoly > c0 ? c0 : 0; }
    if (chunk > 0 && npoly > chunk && r->inv_scale) {
      const int nspans = (npoly + chunk - 1) / chunk;
      const size_t stride = (size_t)Lrows * r->N;
      const bool ntp = r->nt_streams;
      auto span = [&](int j, int* n) { const int p0 = j * chunk; *n = (j < 0 || j >= nspans) ? 0 : (npoly - p0 < chunk ? npoly - p0 : chunk); return (size_t)(j < 0 ? 0 : p0) * stride; };
      for (int j = 0; j <= nspans; ++j) {
        int p1, p2;
        const size_t o1 = span(j, &p1), o2 = span(j - 1, &p2);
        const unsigned rows1 = (unsigned)p1 * Lrows, rows2 = (unsigned)p2 * Lrows;
        if (!inverse) {
          const unsigned n1 = rows1 * 8, n2 = rows2 << S1, grid = n1 > n2 ? n1 : n2;
#define RH_CIF(S) do { if (rows1) ci_col0_kernel<S, false><<<(rows1 + 63) / 64, 64, 0, st>>>(in + o1, out + o1, r->d_tw_fwd + toff, nullptr, cf, c, Lrows, rows1);                 \
                       if (ntp) ntt_ci_fwd_fused_asm<S, true><<<grid, 256, 0, st>>>(in + o1, out + o1, n1, out + o2, n2, p2, r->d_tw_fwd + toff, r->d_twk_fwd + toff, cf, c, Lrows);   \
                       else ntt_ci_fwd_fused_asm<S, false><<<grid, 256, 0, st>>>(in + o1, out + o1, n1, out + o2, n2, p2, r->d_tw_fwd + toff, r->d_twk_fwd + toff, cf, c, Lrows); } while (0)
          if (S1 == 4) RH_CIF(4); else if (S1 == 3) RH_CIF(3); else RH_CIF(2);
#undef RH_CIF
        } else {
          const unsigned n1 = rows1 << S1, n2 = rows2 * 8, grid = n1 > n2 ? n1 : n2;
#define RH_CII(S) do { if (ntp) ntt_ci_inv_fused_asm<S, true><<<grid, 256, 0, st>>>(in + o1, out + o1, n1, p1, out + o2, n2, r->d_twk_inv + toff, r->d_tw_inv + toff, r->d_lastw + limb0, cf, c, Lrows);  \
                       else ntt_ci_inv_fused_asm<S, false><<<grid, 256, 0, st>>>(in + o1, out + o1, n1, p1, out + o2, n2, r->d_twk_inv + toff, r->d_tw_inv + toff, r->d_lastw + limb0, cf, c, Lrows); \
                       if (rows2) ci_col0_kernel<S, true><<<(rows2 + 63) / 64, 64, 0, st>>>(out + o2, out + o2, r->d_tw_inv + toff, r->d_lastw + limb0, cf, c, Lrows, rows2); } while (0)
          if (S1 == 4) RH_CII(4); else if (S1 == 3) RH_CII(3); else RH_CII(2);
#undef RH_CII
        }
      }
      return check_launch("conjugate-invariant transform (pipelined)");
    }
    const unsigned g8 = rows * 8, g0 = (rows + 63) / 64;
#define RH_CI(S, INV, SRC, TW, LW) do { ntt_cols_ci_asm<S, INV><<<g8, 256, 0, st>>>(SRC, out, TW, LW, cf, c, Lrows);  \
                                        ci_col0_kernel<S, INV><<<g0, 64, 0, st>>>(SRC, out, TW, LW, cf, c, Lrows, rows); } while (0)
    if (!inverse) {
      if (S1 == 4) RH_CI(4, false, in, r->d_tw_fwd + toff, nullptr); else if (S1 == 3) RH_CI(3, false, in, r->d_tw_fwd + toff, nullptr);
      else RH_CI(2, false, in, r->d_tw_fwd + toff, nullptr);
      if (int rc = check_launch("ntt_cols_ci_asm")) return rc;
      return rh_std_ntt_launch(r, out, out, npoly, Lrows, limb0, false, false, 2);          // tile stages
    }
    if (int rc = rh_std_ntt_launch(r, in, out, npoly, Lrows, limb0, true, false, 2)) return rc;   // tile stages
    if (S1 == 4) RH_CI(4, true, out, r->d_tw_inv + toff, r->d_lastw + limb0); else if (S1 == 3) RH_CI(3, true, out, r->d_tw_inv + toff, r->d_lastw + limb0);
    else RH_CI(2, true, out, r->d_tw_inv + toff, r->d_lastw + limb0);
#undef RH_CI
    return check_launch("ntt_cols_ci_asm");
  }
  if (!inverse) {
    ci_fold_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(in, out, r->logN, r->d_cifold + limb0, r->d_consts + limb0, Lrows, 0);
    return rh_std_ntt_launch(r, out, out, npoly, Lrows, limb0, false, false, 0);
  }
  if (int rc = rh_std_ntt_launch(r, in, out, npoly, Lrows, limb0, true, false, 0)) return rc;
  ci_fold_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(out, out, r->logN, r->d_cifold + limb0, r->d_consts + limb0, Lrows, 1);
  return check_launch("ci_fold_kernel");
}

// canonical transform of limbs [limb0, limb0 + Lrows) of a dense block, for a ring of any type (rescale.hip)
int rh_ring_ntt_any(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, bool inverse) {
  if (r->kind == RH_RING_3N) {
    std::lock_guard<std::recursive_mutex> lk(r->mu);
    return rh_ring3n_ntt_launch(r, in, out, npoly, Lrows, limb0, inverse, rh_layout3n(r) != 0);
  }
  if (r->kind == RH_RING_CI) return ci_ntt_launch(r, in, out, npoly, Lrows, limb0, inverse);
  return rh_std_ntt_launch(r, in, out, npoly, Lrows, limb0, inverse, false, 0);
}

static int ntt_batch(rh_ring* r, const uint64_t* in, uint64_t* out, int npoly, int level, bool inverse, bool lazy, int phase = 0) {
  if (!r || !in || !out) return rh_fail(RH_ERR_ARG, "ntt: null argument");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "ntt: level %d out of range [0,%d)", level, r->L);
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "ntt: npoly < 0");
  (void)hipSetDevice(r->device);
  if (r->kind == RH_RING_3N) {
    std::lock_guard<std::recursive_mutex> lk(r->mu);          // the 3N workspace is shared and grows lazily
    return rh_ring3n_ntt_launch(r, in, out, npoly, level + 1, 0, inverse, rh_layout3n(r) != 0);
  }
  if (r->kind == RH_RING_CI) return ci_ntt_launch(r, in, out, npoly, level + 1, 0, inverse);
  return rh_std_ntt_launch(r, in, out, npoly, level + 1, 0, inverse, lazy, phase);
}
extern "C" int rh_ring_ntt(rh_ring* r, const uint64_t* in, uint64_t* out, int npoly, int level, int lazy) { return ntt_batch(r, in, out, npoly, level, false, lazy != 0); }
extern "C" int rh_ring_intt(rh_ring* r, const uint64_t* in, uint64_t* out, int npoly, int level, int lazy) { return ntt_batch(r, in, out, npoly, level, true, lazy != 0); }
// Ring.NTT on several blocks in one call (device-API extension; e.g. both operands of a product): for two-pass standard rings the
// software pipeline of the fused launches runs through all the blocks; otherwise the blocks are transformed one after the other.
extern "C" int rh_ring_ntt_many(rh_ring* r, const uint64_t* const* in, uint64_t* const* out, const int* npoly, int nblocks, int level) {
  if (!r || !in || !out || !npoly || nblocks < 0) return rh_fail(RH_ERR_ARG, "ntt_many: bad argument");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "ntt_many: level %d out of range [0,%d)", level, r->L);
  long total = 0;
  for (int k = 0; k < nblocks; ++k) {
    if (!in[k] || !out[k] || npoly[k] < 0) return rh_fail(RH_ERR_ARG, "ntt_many: block %d: null pointer or npoly < 0", k);
    total += npoly[k];
  }
  (void)hipSetDevice(r->device);
  const int Lrows = level + 1;
  int chunk = r->chunk_polys;
  if (chunk < 0) chunk = r->auto_span_rows / Lrows > 0 ? r->auto_span_rows / Lrows : 1;
  if (r->kind == RH_RING_STANDARD && r->logN > LT && !one_pass_ok(r) && chunk > 0 && total > chunk) {
    std::vector<NttSeg> segs;
    for (int k = 0; k < nblocks; ++k) if (npoly[k] > 0) segs.push_back(NttSeg{in[k], out[k], npoly[k]});
    return std_ntt_fwd_pipelined_segs(r, segs.data(), (int)segs.size(), Lrows, 0, chunk);
  }
  for (int k = 0; k < nblocks; ++k)
    if (int rc = ntt_batch(r, in[k], out[k], npoly[k], level, false, false)) return rc;
  return RH_OK;
}
// ---- views at a lower level of blocks allocated with MORE limbs per poly (ring.AtLevel(level) on max-level polys and buffers,
// ring/ring.go:192-213: the idiomatic use inside the reference's evaluators).  Forward transforms with one row stride on both sides
// (N >= 4096) and inverse transforms through the hand-scheduled bodies (N = 2^14 .. 2^16) are batched (row strides inside the
// kernels); the remaining shapes run poly by poly (correct, not the throughput path).
static int ntt_rows(rh_ring* r, const uint64_t* in, int in_rows, uint64_t* out, int out_rows, int npoly, int level, bool inverse, bool lazy) {
  if (!r) return rh_fail(RH_ERR_ARG, "ntt: null argument");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "ntt: level %d out of range [0,%d)", level, r->L);
  if (in_rows < level + 1 || out_rows < level + 1) return rh_fail(RH_ERR_ARG, "ntt: blocks with %d / %d limbs per poly used at level %d", in_rows, out_rows, level);
  if (in_rows == level + 1 && out_rows == level + 1) return ntt_batch(r, in, out, npoly, level, inverse, lazy);
  if (!in || !out) return rh_fail(RH_ERR_ARG, "ntt: null argument");
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "ntt: npoly < 0");
  (void)hipSetDevice(r->device);
  // standard rings: the row strides of the two blocks go into the kernels (every N, forward and inverse, exact-lazy forward included); the one
  // exception is the non-canonical BackwardLazy of N = 8, which compacts like the other ring types
  if (r->kind == RH_RING_STANDARD && !(inverse && lazy && r->logN < 4)) {
    ++r->stats_rows_direct;
    return std_ntt_launch_span(r, in, out, npoly, level + 1, 0, inverse, lazy, 0, in_rows, out_rows);
  }
  // conjugate-invariant and 3N rings: still ONE batched transform -- the leading limbs are compacted with a strided copy on the way in and / or
  // expanded on the way out (round 3; these shapes ran poly by poly before)
  if (npoly == 0) return RH_OK;
  ++r->stats_rows_compacted;
  const size_t N = (size_t)r->N, Lr = (size_t)(level + 1);
  hipStream_t st = rh_stream(r);
  auto copy2d = [&](u64* dst, size_t drows, const u64* src, size_t srows) {
    return hipMemcpy2DAsync(dst, drows * N * 8, src, srows * N * 8, Lr * N * 8, (size_t)npoly, hipMemcpyDeviceToDevice, st) == hipSuccess
               ? RH_OK : rh_fail(RH_ERR_DEVICE, "ntt: strided copy of the leading limbs failed");
  };
  if ((size_t)out_rows == Lr) {                       // dense destination: gather into it, transform in place
    if (int rc = copy2d(out, Lr, in, (size_t)in_rows)) return rc;
    return ntt_batch(r, out, out, npoly, level, inverse, lazy);
  }
  std::lock_guard<std::recursive_mutex> lk(r->mu);   // the dense scratch is shared and grows lazily; its use is stream-ordered
  const size_t words = (size_t)npoly * Lr * N;
  if (r->rows_words < words) {
    if (r->d_rows) (void)hipFree(r->d_rows);
    r->d_rows = nullptr; r->rows_words = 0;
    if (hipMalloc((void**)&r->d_rows, words * 8) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc(AtLevel scratch, %zu words) failed", words);
    r->rows_words = words;
  }
  const u64* src = in;
  if ((size_t)in_rows != Lr) { if (int rc = copy2d(r->d_rows, Lr, in, (size_t)in_rows)) return rc; src = r->d_rows; }
  if (int rc = ntt_batch(r, src, r->d_rows, npoly, level, inverse, lazy)) return rc;
  return copy2d(out, (size_t)out_rows, r->d_rows, Lr);
}
extern "C" int rh_ring_stats(const rh_ring* r, const char* key, long* value) {
  if (!r || !key || !value) return rh_fail(RH_ERR_ARG, "rh_ring_stats: null argument");
  if (!strcmp(key, "rows_direct")) { *value = r->stats_rows_direct; return RH_OK; }
  if (!strcmp(key, "rows_compacted")) { *value = r->stats_rows_compacted; return RH_OK; }
  if (!strcmp(key, "rows_poly_by_poly")) { *value = 0; return RH_OK; }   // no entry point loops over polys any more (kept so a regression has a name)
  return rh_fail(RH_ERR_ARG, "rh_ring_stats: unknown key %s", key);
}
extern "C" int rh_ring_ntt_rows(rh_ring* r, const uint64_t* in, int in_rows, uint64_t* out, int out_rows, int npoly, int level, int lazy) {
  return ntt_rows(r, in, in_rows, out, out_rows, npoly, level, false, lazy != 0);
}
extern "C" int rh_ring_intt_rows(rh_ring* r, const uint64_t* in, int in_rows, uint64_t* out, int out_rows, int npoly, int level, int lazy) {
  return ntt_rows(r, in, in_rows, out, out_rows, npoly, level, true, lazy != 0);
}
// Explicit-layout forms for 3N rings (hosts that TAG their device polys with the NTT-domain layout instead of switching the whole handle with
// the tuning key): block_order 1 = the NTT side of this call is in block order, 0 = the reference's ascending-totative order.  The layout is a
// per-call argument (thread-local inside the call), so concurrent callers of one handle may use different layouts.  Other ring kinds: ignored.
extern "C" int rh_ring_ntt3n_block_order_supported(const rh_ring* r) {
  return r && r->kind == RH_RING_3N && rh_ring3n_block_order_ok(r) ? 1 : 0;
}
extern "C" int rh_ring_ntt_layout(rh_ring* r, const uint64_t* in, int in_rows, uint64_t* out, int out_rows, int npoly, int level, int inverse, int block_order) {
  if (!r) return rh_fail(RH_ERR_ARG, "ntt: null argument");
  if (block_order && !rh_ring_ntt3n_block_order_supported(r)) return rh_fail(RH_ERR_UNSUPPORTED, "block order needs a 3N ring with N = 3 * 2^k, k >= 13");
  RhLayoutScope ls(r->kind == RH_RING_3N ? (block_order ? 1 : 0) : -1);
  return ntt_rows(r, in, in_rows, out, out_rows, npoly, level, inverse != 0, false);
}
extern "C" int rh_ring_ntt_phase(rh_ring* r, const uint64_t* in, uint64_t* out, int npoly, int level, int inverse, int phase) {
  if (phase < 0 || phase > 2) return rh_fail(RH_ERR_ARG, "phase must be 0, 1 or 2");
  return ntt_batch(r, in, out, npoly, level, inverse != 0, false, phase);
}

// INTT(a . b) for NTT-domain blocks a, b (canonical or lazy < 2q): the values ring.MForm(a) -> ring.MulCoeffsMontgomery(., b) ->
// ring.INTT produce (schemes/ckks/evaluator.go:821-834 + :INTT; BASELINE config 3), with the product formed on load by the inverse
// tile kernel (MRedLazy) and the factor 2^64 restored by the N^-1 constants of the last inverse stage.  Outputs are canonical
// residues of a*b*N^-1-transform, hence bit-identical to the three-call sequence.  out may alias a or b.
static int std_intt_mul_launch(rh_ring* r, const u64* a, const u64* b, u64* out, int npoly, int Lrows) {
  const int logN = r->logN;
  const unsigned rows = (unsigned)npoly * (unsigned)Lrows;
  if (rows == 0) return RH_OK;
  hipStream_t st = rh_stream(r);
  if (logN < LT) {                                   // small rings: the three calls as they are (not a throughput path)
    if (int rc = rh_vec_launch(r, RH_OP_MFORM, a, nullptr, out, npoly, Lrows, 0, nullptr, nullptr)) return rc;
    if (int rc = rh_vec_launch(r, RH_OP_MUL_MONT, out, b, out, npoly, Lrows, 0, nullptr, nullptr)) return rc;
    return rh_std_ntt_launch(r, out, out, npoly, Lrows, 0, true, false, 0);
  }
  (void)hipGetLastError();
  const int S1 = logN - LT;
  if (S1 > 0 && r->asm_tile) {
    // large batches: the software pipeline of std_ntt_inv_pipelined (tile stages of span j with the column stages of span j-1).
    // The output may alias an input: a span's tiles read (a, b) of that span only, before its column stages write them.
    int chunk = r->chunk_polys;
    if (chunk < 0) { const int c = r->auto_span_rows / Lrows > 0 ? r->auto_span_rows / Lrows : 1; chunk = npoly > c ? c : 0; }
    if (chunk > 0 && npoly > chunk) return std_ntt_inv_pipelined(r, a, out, npoly, Lrows, 0, chunk, b);
    ntt_inv_tile_mul_asm<<<rows << S1, 256, 0, st>>>(a, b, out, r->d_twk_inv, r->d_consts_r, Lrows, logN, npoly);
  } else
  ntt_inv_tile_mul<<<rows << S1, 256, 0, st>>>(a, b, out, r->d_twk_inv, r->d_consts_r, Lrows, logN, S1 == 0 ? 1 : 0, npoly);
  if (S1 >= 2 && S1 <= 4 && r->asm_cols && r->asm_tile) {
    if (S1 == 4) ntt_inv_cols_asm<4><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv, r->d_lastw_r, r->d_consts_r, Lrows, 0);
    else if (S1 == 3) ntt_inv_cols_asm<3><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv, r->d_lastw_r, r->d_consts_r, Lrows, 0);
    else ntt_inv_cols_asm<2><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv, r->d_lastw_r, r->d_consts_r, Lrows, 0);
  } else if (S1 > 0) launch_inv_cols(S1, dim3(rows * 16), st, out, r->d_tw_inv, r->d_lastw_r, r->d_consts_r, Lrows, logN, 1);
  return check_launch("intt_mul");
}
extern "C" int rh_ring_intt_mul(rh_ring* r, const uint64_t* a, const uint64_t* b, uint64_t* out, int npoly, int level) {
  if (!r || !a || !b || !out) return rh_fail(RH_ERR_ARG, "intt_mul: null argument");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "intt_mul: level %d out of range [0,%d)", level, r->L);
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "intt_mul: npoly < 0");
  (void)hipSetDevice(r->device);
  if (r->kind != RH_RING_STANDARD) {                 // 3N / conjugate-invariant rings: the three calls as they are, through the ring's own transform
    if (int rc = rh_vec_launch(r, RH_OP_MFORM, a, nullptr, out, npoly, level + 1, 0, nullptr, nullptr)) return rc;
    if (int rc = rh_vec_launch(r, RH_OP_MUL_MONT, out, b, out, npoly, level + 1, 0, nullptr, nullptr)) return rc;
    return rh_ring_ntt_any(r, out, out, npoly, level + 1, 0, true);
  }
  return std_intt_mul_launch(r, a, b, out, npoly, level + 1);
}

// c = INTT(NTT(a) . NTT(b)) with a and b given in the COEFFICIENT domain (BASELINE config 3: ring.NTT(a), ring.NTT(b), ring.MForm,
// ring.MulCoeffsMontgomery, ring.INTT -- schemes/ckks/evaluator.go:821-834 around a fresh product): forward column stages of a and of b in
// place, then ONE kernel for the forward tile stages of both, their product and the inverse tile stages (ntt_polymul_tile_asm), then the inverse
// column stages.  Same canonical values as the five ring calls; a and b are CONSUMED (they hold their column-stage intermediates afterwards,
// not NTT(a) / NTT(b)); out may alias either.  Standard rings, 2^13 <= N <= 2^17, hand-scheduled tile bodies; other shapes: RH_ERR_UNSUPPORTED
// (the caller falls back to rh_ring_ntt_many + rh_ring_intt_mul).
extern "C" int rh_ring_polymul(rh_ring* r, uint64_t* a, uint64_t* b, uint64_t* out, int npoly, int level) {
  if (!r || !a || !b || !out) return rh_fail(RH_ERR_ARG, "polymul: null argument");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "polymul: level %d out of range [0,%d)", level, r->L);
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "polymul: npoly < 0");
  if (a == b) return rh_fail(RH_ERR_ARG, "polymul: a and b must be different blocks (both are transformed in place)");
  const int S1 = r->logN - LT;
  if (r->kind != RH_RING_STANDARD || S1 < 1 || S1 > 5 || !r->asm_tile) return rh_fail(RH_ERR_UNSUPPORTED, "polymul: standard rings with 2^13 <= N <= 2^17 and the hand-scheduled tile bodies");
  const int Lrows = level + 1;
  const unsigned rows = (unsigned)npoly * (unsigned)Lrows;
  if (rows == 0) return RH_OK;
  (void)hipSetDevice(r->device);
  hipStream_t st = rh_stream(r);
  int chunk = r->chunk_polys;
  if (chunk < 0) { const int c = r->auto_span_rows / Lrows > 0 ? r->auto_span_rows / Lrows : 1; chunk = npoly > c ? c : 0; }
  if (chunk > 0 && npoly > chunk) {
    // large batches: ONE stream of launches, launch j = column stages of span j + tile middle of span j-1 + inverse column stages of span j-2
    (void)hipGetLastError();
    const int nspans = (npoly + chunk - 1) / chunk;
    const size_t stride = (size_t)Lrows * r->N;
    const bool nt = r->nt_streams;                         // a pipelined batch is always far beyond the Infinity Cache (three blocks of >= 2048 rows)
    auto span = [&](int j, int* n) { const int p0 = j * chunk; *n = (j < 0 || j >= nspans) ? 0 : (npoly - p0 < chunk ? npoly - p0 : chunk); return (size_t)(j < 0 ? 0 : p0) * stride; };
    for (int j = 0; j < nspans + 2; ++j) {
      int p1, p2, p3;
      const size_t o1 = span(j, &p1), o2 = span(j - 1, &p2), o3 = span(j - 2, &p3);
      const unsigned n1 = (unsigned)p1 * Lrows * 16, n2 = ((unsigned)p2 * Lrows) << S1, n3 = (unsigned)p3 * Lrows * 16;
      const unsigned grid = n1 > n2 ? (n1 > n3 ? n1 : n3) : (n2 > n3 ? n2 : n3);
      if (!grid) continue;
#define RH_PM_FUSED(S) do { if (nt) ntt_polymul_fused_asm<S, true><<<grid, 256, 0, st>>>(a + o1, b + o1, n1, a + o2, b + o2, out + o2, n2, p2, out + o3, n3, r->d_tw_fwd, r->d_twk_fwd, \
                                                                                          r->d_twk_inv, r->d_tw_inv, r->d_lastw_r, r->d_consts_r, Lrows, r->logN);                          \
                            else ntt_polymul_fused_asm<S, false><<<grid, 256, 0, st>>>(a + o1, b + o1, n1, a + o2, b + o2, out + o2, n2, p2, out + o3, n3, r->d_tw_fwd, r->d_twk_fwd,      \
                                                                                       r->d_twk_inv, r->d_tw_inv, r->d_lastw_r, r->d_consts_r, Lrows, r->logN); } while (0)
      switch (S1) { case 1: RH_PM_FUSED(1); break; case 2: RH_PM_FUSED(2); break; case 3: RH_PM_FUSED(3); break; case 4: RH_PM_FUSED(4); break; default: RH_PM_FUSED(5); break; }
#undef RH_PM_FUSED
    }
    return check_launch("polymul (pipelined)");
  }
  if (int rc = std_ntt_launch_span(r, a, a, npoly, Lrows, 0, false, false, 1)) return rc;       // column stages only (phase 1), in place
  if (int rc = std_ntt_launch_span(r, b, b, npoly, Lrows, 0, false, false, 1)) return rc;
  (void)hipGetLastError();
  if (rh_streams_beyond_cache(r, rows)) ntt_polymul_tile_asm<true><<<rows << S1, 256, 0, st>>>(a, b, out, r->d_twk_fwd, r->d_twk_inv, r->d_consts_r, Lrows, r->logN, npoly);
  else ntt_polymul_tile_asm<false><<<rows << S1, 256, 0, st>>>(a, b, out, r->d_twk_fwd, r->d_twk_inv, r->d_consts_r, Lrows, r->logN, npoly);
  const bool nt = rh_streams_beyond_cache(r, rows);
  if (S1 >= 2 && S1 <= 4 && r->asm_cols) {
#define RH_PM_COLS(S) do { if (nt) ntt_inv_cols_asm<S, true><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv, r->d_lastw_r, r->d_consts_r, Lrows, 0); \
                           else ntt_inv_cols_asm<S, false><<<dim3(rows * 16), 256, 0, st>>>(out, r->d_tw_inv, r->d_lastw_r, r->d_consts_r, Lrows, 0); } while (0)
    if (S1 == 4) RH_PM_COLS(4); else if (S1 == 3) RH_PM_COLS(3); else RH_PM_COLS(2);
#undef RH_PM_COLS
  } else launch_inv_cols(S1, dim3(rows * 16), st, out, r->d_tw_inv, r->d_lastw_r, r->d_consts_r, Lrows, r->logN, 1);
  return check_launch("polymul");
}

extern "C" int rh_ring_ntt3n_reorder(rh_ring* r, const uint64_t* in, uint64_t* out, int npoly, int level, int to_reference) {
  if (!r || !in || !out) return rh_fail(RH_ERR_ARG, "ntt3n_reorder: null argument");
  if (r->kind != RH_RING_3N) return rh_fail(RH_ERR_ARG, "ntt3n_reorder: not a 3N ring");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "ntt3n_reorder: level %d out of range [0,%d)", level, r->L);
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "ntt3n_reorder: npoly < 0");
  (void)hipSetDevice(r->device);
  return rh_ring3n_reorder_launch(r, in, out, npoly, level + 1, to_reference != 0);
}

// Pre-sizes every lazily grown scratch of the ring for batches of up to npoly polys (all limbs): afterwards no entry point of
// the ring allocates.  Optional: without it scratch grows on first use.
extern "C" int rh_ring_reserve(rh_ring* r, int npoly) {
  if (!r || npoly < 0) return rh_fail(RH_ERR_ARG, "rh_ring_reserve: bad argument");
  (void)hipSetDevice(r->device);
  std::lock_guard<std::recursive_mutex> lk(r->mu);
  if (r->kind == RH_RING_3N) if (int rc = rh_ring3n_reserve(r, npoly)) return rc;
  const size_t words = (size_t)npoly * r->L * r->N;            // the rows-per-poly transforms' dense scratch
  if (r->rows_words < words) {
    if (r->d_rows) (void)hipFree(r->d_rows);
    r->d_rows = nullptr; r->rows_words = 0;
    if (hipMalloc((void**)&r->d_rows, (words ? words : 1) * 8) != hipSuccess) return rh_fail(RH_ERR_NOMEM, "hipMalloc(AtLevel scratch) failed");
    r->rows_words = words;
  }
  return rh_rescale_reserve(r, npoly);
}

extern "C" int rh_ring_set_tuning(rh_ring* r, const char* key, long value) {
  if (!r || !key) return rh_fail(RH_ERR_ARG, "set_tuning: null argument");
  if (!strcmp(key, "chunk_polys")) { r->chunk_polys = (int)value; return RH_OK; }
  if (!strcmp(key, "asm_tile")) { r->asm_tile = value != 0; return RH_OK; }
  if (!strcmp(key, "auto_span_rows")) { if (value < 1) return rh_fail(RH_ERR_ARG, "auto_span_rows must be >= 1"); r->auto_span_rows = (int)value; return RH_OK; }
  if (!strcmp(key, "fuse_submul")) { r->fuse_submul = (int)value; return RH_OK; }
  if (!strcmp(key, "digit_pipeline")) { r->digit_pipeline = (int)value; return RH_OK; }
  if (!strcmp(key, "fuse_ci")) { r->fuse_ci = (int)value; return RH_OK; }
  if (!strcmp(key, "perm_inv_shape")) { r->perm_inv_shape = (int)value; return RH_OK; }
  if (!strcmp(key, "perm_fwd_shape")) { r->perm_fwd_shape = (int)value; return RH_OK; }
  if (!strcmp(key, "fuse3n")) { r->fuse3n = (int)value; return RH_OK; }
  if (!strcmp(key, "ntt3n_block_order")) {
    if (r->kind != RH_RING_3N) return rh_fail(RH_ERR_ARG, "ntt3n_block_order: not a 3N ring");
    r->block_order3n = value != 0;
    return RH_OK;
  }
  if (!strcmp(key, "asm_cols")) { r->asm_cols = (int)value; return RH_OK; }
  if (!strcmp(key, "ks_small_rows")) { if (value < 0) return rh_fail(RH_ERR_ARG, "ks_small_rows must be >= 0"); r->ks_small_rows = (int)value; return RH_OK; }
  if (!strcmp(key, "pair_submul")) { r->pair_submul = value != 0; return RH_OK; }
  if (!strcmp(key, "one_pass")) { r->one_pass = value != 0; return RH_OK; }
  if (!strcmp(key, "nt_streams")) { r->nt_streams = value != 0; if (r->kind == RH_RING_3N) rh_ring3n_set_nt_streams(r, value != 0); return RH_OK; }     // 0: default cache policy everywhere (A/B runs: bench.py --tune nt_streams=0)
  return rh_fail(RH_ERR_ARG, "set_tuning: unknown key %s", key);
}

// ---- one limb, host pointers: the NumberTheoreticTransformer interface -----------------------------------------------
// Every call takes a (stream, scratch) slot of its own from the ring's pool, so any number of OS threads may call
// Forward / Backward on one handle at once (ring/ring.go:192-194; goroutines migrate between threads).  A slot is created
// the first time a caller finds the pool empty; steady state allocates nothing.
static int slot_acquire(rh_ring* r, RhHostSlot** out) {
  {
    std::lock_guard<std::mutex> lk(r->slot_mu);
    if (!r->free_slots.empty()) { *out = r->free_slots.back(); r->free_slots.pop_back(); return RH_OK; }
  }
  RhHostSlot* sl = new (std::nothrow) RhHostSlot();
  if (!sl) return rh_fail(RH_ERR_NOMEM, "out of host memory");
  sl->words = (size_t)r->N * (r->kind == RH_RING_3N ? 2 : 1);                    // the limb + the 3N transform's workspace for one row
  if (hipStreamCreateWithFlags(&sl->stream, hipStreamNonBlocking) != hipSuccess) { delete sl; return rh_fail(RH_ERR_DEVICE, "hipStreamCreate failed"); }
  if (hipMalloc((void**)&sl->buf, sl->words * 8) != hipSuccess) { (void)hipStreamDestroy(sl->stream); delete sl; return rh_fail(RH_ERR_NOMEM, "hipMalloc(host-limb scratch) failed"); }
  std::lock_guard<std::mutex> lk(r->slot_mu);
  r->all_slots.push_back(sl);
  *out = sl;
  return RH_OK;
}
static void slot_release(rh_ring* r, RhHostSlot* sl) {
  std::lock_guard<std::mutex> lk(r->slot_mu);
  r->free_slots.push_back(sl);
}
static int ntt_host_limb(rh_ring* r, int limb, const uint64_t* p1, uint64_t* p2, bool inverse, bool lazy) {
  if (!r) return rh_fail(RH_ERR_ARG, "null ring");
  if (!p1 || !p2) return rh_fail(RH_ERR_ARG, "cannot NTT: nil slice (len(p1), len(p2) must be >= N=%d)", r->N);
  if (limb < 0 || limb >= r->L) return rh_fail(RH_ERR_ARG, "limb %d out of range [0,%d)", limb, r->L);
  (void)hipSetDevice(r->device);
  RhHostSlot* sl;
  if (int rc = slot_acquire(r, &sl)) return rc;
  int rc = RH_OK;
  {
    RhCallScope scope(sl->stream, r->kind == RH_RING_3N ? sl->buf + r->N : nullptr, (size_t)r->N);
    const size_t bytes = (size_t)r->N * 8;
    hipError_t e = hipMemcpyAsync(sl->buf, p1, bytes, hipMemcpyHostToDevice, sl->stream);
    if (e != hipSuccess) rc = rh_fail(RH_ERR_DEVICE, "H2D: %s", hipGetErrorString(e));
    if (!rc) rc = (r->kind == RH_RING_3N) ? rh_ring3n_ntt_launch(r, sl->buf, sl->buf, 1, 1, limb, inverse)
                : (r->kind == RH_RING_CI) ? ci_ntt_launch(r, sl->buf, sl->buf, 1, 1, limb, inverse)
                                          : rh_std_ntt_launch(r, sl->buf, sl->buf, 1, 1, limb, inverse, lazy, 0);
    if (!rc) {
      e = hipMemcpyAsync(p2, sl->buf, bytes, hipMemcpyDeviceToHost, sl->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(sl->stream);
      if (e != hipSuccess) rc = rh_fail(RH_ERR_DEVICE, "D2H: %s", hipGetErrorString(e));
    } else (void)hipStreamSynchronize(sl->stream);
  }
  slot_release(r, sl);
  return rc;
}
extern "C" int rh_ntt_forward(rh_ring* r, int limb, const uint64_t* p1, uint64_t* p2) { return ntt_host_limb(r, limb, p1, p2, false, false); }
extern "C" int rh_ntt_forward_lazy(rh_ring* r, int limb, const uint64_t* p1, uint64_t* p2) { return ntt_host_limb(r, limb, p1, p2, false, true); }
extern "C" int rh_ntt_backward(rh_ring* r, int limb, const uint64_t* p1, uint64_t* p2) { return ntt_host_limb(r, limb, p1, p2, true, false); }
extern "C" int rh_ntt_backward_lazy(rh_ring* r, int limb, const uint64_t* p1, uint64_t* p2) { return ntt_host_limb(r, limb, p1, p2, true, true); }

// ---- a whole Poly, host pointers: Ring.NTT / NTTLazy / INTT / INTTLazy (ring/ntt.go:127-152) as ONE call -------------------------------
// The reference loops `for i, s := range r.SubRings[:r.level+1] { s.NTT(p1.Coeffs[i], p2.Coeffs[i]) }` over Poly.Coeffs [][]uint64
// (ring/poly.go:13-24); through rh_ntt_forward that is level+1 synchronous H2D -> kernel -> D2H -> synchronise round trips.  Here the
// level+1 limb pointers arrive together: the limbs are cut into up to four groups that alternate between two streams, so group g's
// upload runs under group g-1's transform and download (PCIe is full duplex), every group is ONE batched launch pair over its limbs, and
// the host synchronises once.  Page-locked limbs (rh_host_alloc / rh_host_register) are DMA'd where they lie; pageable limbs are staged
// through the slot's page-locked buffers (the CPU copy of group g overlaps the DMA of group g-1).
struct RhPolySlot {
  hipStream_t st[2] = {nullptr, nullptr};
  hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr};
  u64* dbuf = nullptr; size_t dwords = 0;                 // device: the limbs (+ the 3N transform's workspace)
  u64* hin = nullptr; u64* hout = nullptr; size_t hwords = 0;   // page-locked staging, allocated when a pageable limb is first seen
};
static void poly_slot_free(RhPolySlot* sl) {
  if (sl->dbuf) (void)hipFree(sl->dbuf);
  if (sl->hin) (void)hipHostFree(sl->hin);
  if (sl->hout) (void)hipHostFree(sl->hout);
  for (hipStream_t s : sl->st) if (s) (void)hipStreamDestroy(s);
  for (hipEvent_t e : sl->done) if (e) (void)hipEventDestroy(e);
  delete sl;
}
void rh_poly_slots_teardown(rh_ring* r) { for (RhPolySlot* sl : r->all_poly_slots) poly_slot_free(sl); r->all_poly_slots.clear(); r->free_poly_slots.clear(); }
static int poly_slot_acquire(rh_ring* r, RhPolySlot** out) {
  {
    std::lock_guard<std::mutex> lk(r->slot_mu);
    if (!r->free_poly_slots.empty()) { *out = r->free_poly_slots.back(); r->free_poly_slots.pop_back(); return RH_OK; }
  }
  RhPolySlot* sl = new (std::nothrow) RhPolySlot();
  if (!sl) return rh_fail(RH_ERR_NOMEM, "out of host memory");
  sl->dwords = (size_t)r->N * r->L * (r->kind == RH_RING_3N ? 2 : 1);
  bool ok = hipMalloc((void**)&sl->dbuf, sl->dwords * 8) == hipSuccess;
  for (int k = 0; k < 2 && ok; ++k) ok = hipStreamCreateWithFlags(&sl->st[k], hipStreamNonBlocking) == hipSuccess;
  for (int k = 0; k < 4 && ok; ++k) ok = hipEventCreateWithFlags(&sl->done[k], hipEventDisableTiming) == hipSuccess;
  if (!ok) { poly_slot_free(sl); return rh_fail(RH_ERR_NOMEM, "whole-poly host path: device buffer / stream / event creation failed"); }
  std::lock_guard<std::mutex> lk(r->slot_mu);
  r->all_poly_slots.push_back(sl);
  *out = sl;
  return RH_OK;
}
static void poly_slot_release(rh_ring* r, RhPolySlot* sl) { std::lock_guard<std::mutex> lk(r->slot_mu); r->free_poly_slots.push_back(sl); }
// 1: page-locked host memory (DMA'd where it lies), 0: ordinary pageable memory (staged), -1: device memory (a caller's mistake: this is the HOST entry)
static int host_ptr_kind(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; }   // an ordinary (pageable) pointer is "invalid value" to the runtime
  if (a.type == hipMemoryTypeHost) return 1;
  return (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeArray) ? -1 : 0;
}
static int ntt_host_poly(rh_ring* r, int level, const uint64_t* const* p1, uint64_t* const* p2, bool inverse, bool lazy) {
  if (!r) return rh_fail(RH_ERR_ARG, "null ring");
  if (!p1 || !p2) return rh_fail(RH_ERR_ARG, "cannot NTT: nil Poly.Coeffs");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "ntt: level %d out of range [0,%d)", level, r->L);
  const int Lr = level + 1, N = r->N;
  for (int i = 0; i < Lr; ++i) if (!p1[i] || !p2[i]) return rh_fail(RH_ERR_ARG, "cannot NTT: nil slice at limb %d (len(p1[i]), len(p2[i]) must be >= N=%d)", i, N);
  (void)hipSetDevice(r->device);
  RhPolySlot* sl;
  if (int rc = poly_slot_acquire(r, &sl)) return rc;
  const size_t bytes = (size_t)N * 8;
  bool pin_in[RH_MAX_LIMBS], pin_out[RH_MAX_LIMBS], staged = false;
  int rc = RH_OK;
  for (int i = 0; i < Lr; ++i) {
    const int ki = host_ptr_kind(p1[i]), ko = host_ptr_kind(p2[i]);
    if (ki < 0 || ko < 0) { rc = rh_fail(RH_ERR_ARG, "cannot NTT: limb %d is a device pointer; device-resident polys go through rh_ring_ntt / rh_ring_intt", i); break; }
    pin_in[i] = ki == 1; pin_out[i] = ko == 1; staged |= !pin_in[i] || !pin_out[i];
  }
  if (rc) { poly_slot_release(r, sl); return rc; }
  if (staged && !sl->hin) {
    sl->hwords = (size_t)N * r->L;
    if (hipHostMalloc((void**)&sl->hin, sl->hwords * 8, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void**)&sl->hout, sl->hwords * 8, hipHostMallocDefault) != hipSuccess) {
      if (sl->hin) { (void)hipHostFree(sl->hin); sl->hin = nullptr; }
      rc = rh_fail(RH_ERR_NOMEM, "hipHostMalloc(page-locked staging) failed");
    }
  }
  // groups: enough bytes per group to amortise a launch pair (>= 256 KiB), at most 4
  int G = (int)(((size_t)Lr * bytes) / ((size_t)256 << 10)); if (G > 4) G = 4; if (G > Lr) G = Lr; if (G < 1) G = 1;
  // the host side of every limb: the caller's slice when it is page-locked, else its place in the staging blocks.  Limbs that are adjacent in
  // host memory move as ONE copy (a copy costs ~10 us to issue and start whatever its size): always the case for staged limbs, and for a
  // Poly whose limbs were carved from one page-locked allocation
  const u64* hsrc[RH_MAX_LIMBS]; u64* hdst[RH_MAX_LIMBS];
  for (int i = 0; i < Lr; ++i) { hsrc[i] = pin_in[i] ? p1[i] : sl->hin + (size_t)i * N; hdst[i] = pin_out[i] ? p2[i] : sl->hout + (size_t)i * N; }
  for (int g = 0; g < G && !rc; ++g) {
    const int g0 = (int)((long)Lr * g / G), g1 = (int)((long)Lr * (g + 1) / G);
    hipStream_t st = sl->st[g & 1];
    hipError_t e = hipSuccess;
    for (int i = g0; i < g1; ++i) if (!pin_in[i]) memcpy(sl->hin + (size_t)i * N, p1[i], bytes);
    for (int i = g0, j; i < g1 && e == hipSuccess; i = j) {
      for (j = i + 1; j < g1 && hsrc[j] == hsrc[j - 1] + N; ++j) {}
      e = hipMemcpyAsync(sl->dbuf + (size_t)i * N, hsrc[i], (size_t)(j - i) * bytes, hipMemcpyHostToDevice, st);
    }
    if (e != hipSuccess) { rc = rh_fail(RH_ERR_DEVICE, "H2D: %s", hipGetErrorString(e)); break; }
    {
      u64* d = sl->dbuf + (size_t)g0 * N;
      RhCallScope scope(st, r->kind == RH_RING_3N ? sl->dbuf + (size_t)(r->L + g0) * N : nullptr, (size_t)(g1 - g0) * N);
      rc = (r->kind == RH_RING_3N) ? rh_ring3n_ntt_launch(r, d, d, 1, g1 - g0, g0, inverse)
         : (r->kind == RH_RING_CI) ? ci_ntt_launch(r, d, d, 1, g1 - g0, g0, inverse)
                                   : rh_std_ntt_launch(r, d, d, 1, g1 - g0, g0, inverse, lazy, 0);
    }
    for (int i = g0, j; i < g1 && !rc; i = j) {
      for (j = i + 1; j < g1 && hdst[j] == hdst[j - 1] + N; ++j) {}
      e = hipMemcpyAsync(hdst[i], sl->dbuf + (size_t)i * N, (size_t)(j - i) * bytes, hipMemcpyDeviceToHost, st);
      if (e != hipSuccess) rc = rh_fail(RH_ERR_DEVICE, "D2H: %s", hipGetErrorString(e));
    }
    if (!rc && hipEventRecord(sl->done[g], st) != hipSuccess) rc = rh_fail(RH_ERR_DEVICE, "hipEventRecord failed");
  }
  if (rc) { (void)hipStreamSynchronize(sl->st[0]); (void)hipStreamSynchronize(sl->st[1]); poly_slot_release(r, sl); return rc; }
  for (int g = 0; g < G; ++g) {                         // staged outputs leave as soon as their group has landed
    const int g0 = (int)((long)Lr * g / G), g1 = (int)((long)Lr * (g + 1) / G);
    hipError_t e = hipEventSynchronize(sl->done[g]);
    if (e != hipSuccess) { rc = rh_fail(RH_ERR_DEVICE, "whole-poly host path: %s", hipGetErrorString(e)); (void)hipStreamSynchronize(sl->st[0]); (void)hipStreamSynchronize(sl->st[1]); break; }
    for (int i = g0; i < g1; ++i) if (!pin_out[i]) memcpy(p2[i], sl->hout + (size_t)i * N, bytes);
  }
  poly_slot_release(r, sl);
  return rc;
}
extern "C" int rh_ntt_poly_forward(rh_ring* r, int level, const uint64_t* const* p1, uint64_t* const* p2, int lazy) { return ntt_host_poly(r, level, p1, p2, false, lazy != 0); }
extern "C" int rh_ntt_poly_backward(rh_ring* r, int level, const uint64_t* const* p1, uint64_t* const* p2, int lazy) { return ntt_host_poly(r, level, p1, p2, true, lazy != 0); }
// page-locked host memory for Poly.Coeffs backing arrays: limbs that live here are DMA'd directly by rh_ntt_* / rh_ntt_poly_*
extern "C" int rh_host_alloc(size_t words, uint64_t** hptr) {
  if (!hptr) return rh_fail(RH_ERR_ARG, "rh_host_alloc: null argument");
  if (hipHostMalloc((void**)hptr, (words ? words : 1) * 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return rh_fail(RH_ERR_NOMEM, "hipHostMalloc(%zu words) failed", words); }
  return RH_OK;
}
extern "C" int rh_host_free(uint64_t* hptr) { if (hptr) (void)hipHostFree(hptr); return RH_OK; }
extern "C" int rh_host_register(uint64_t* hptr, size_t words) {
  if (!hptr || !words) return rh_fail(RH_ERR_ARG, "rh_host_register: null argument");
  // whole pages only: the runtime pins and GPU-maps every page the range touches, and a page shared with unrelated heap objects would be dragged along
  if (((uintptr_t)hptr & 4095u) || ((words * 8) & 4095u)) return rh_fail(RH_ERR_ARG, "rh_host_register: register whole 4 KiB pages (pointer %p, %zu bytes)", (void*)hptr, words * 8);
  hipError_t e = hipHostRegister(hptr, words * 8, hipHostRegisterDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); return rh_fail(RH_ERR_DEVICE, "hipHostRegister: %s", hipGetErrorString(e)); }
  return RH_OK;
}
extern "C" int rh_host_unregister(uint64_t* hptr) {
  if (!hptr) return RH_OK;
  hipError_t e = hipHostUnregister(hptr);
  if (e != hipSuccess) { (void)hipGetLastError(); return rh_fail(RH_ERR_DEVICE, "hipHostUnregister: %s", hipGetErrorString(e)); }
  return RH_OK;
}

// ------------------------------------------------------------------------------------------------ element-wise
struct ScalarPack { u64 s[RH_MAX_LIMBS]; };

struct RowStrides { int r1, r2, r3; unsigned pair0, pair1; int nt; };   // nt: non-temporal loads / stores (operands beyond the Infinity Cache; a broadcast row stays cached)   // limbs per poly of the three operand blocks (>= L: ring.AtLevel views; r2 < 0: p2 is ONE
                                                                 // row used for every (poly, limb)); the range of coefficient pairs processed
template <int OP>
__global__ void __launch_bounds__(256)
vec_op_packed(const u64* p1, const u64* p2, u64* p3, unsigned n, ScalarPack s0, ScalarPack s1,
              const LimbConsts* __restrict__ consts, int L, RowStrides rs) {
  const u32 row = blockIdx.x;
  const u32 limb = row % (u32)L, poly = row / (u32)L;
  const LimbConsts c = consts[limb];
  const u64 a0 = s0.s[limb], a1 = s1.s[limb];
  const size_t o1 = ((size_t)poly * rs.r1 + limb) * n, o2 = rs.r2 < 0 ? 0 : ((size_t)poly * rs.r2 + limb) * n, o3 = ((size_t)poly * rs.r3 + limb) * n;
  for (unsigned i = rs.pair0 + blockIdx.y * blockDim.x + threadIdx.x; i < rs.pair1; i += gridDim.y * blockDim.x) {
    const size_t e = 2 * (size_t)i;
    typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
    auto ld = [&](const u64* p, bool nt) {
      if (!nt) return *reinterpret_cast<const ulonglong2*>(p);
      const u64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(p));
      return make_ulonglong2(v.x, v.y);
    };
    ulonglong2 x = make_ulonglong2(0, 0), y = x, z = x;
    if (op_reads_x(OP)) x = ld(p1 + o1 + e, rs.nt);
    if (op_reads_y(OP)) y = ld(p2 + o2 + e, rs.nt && rs.r2 >= 0);
    if (op_reads_z(OP)) z = ld(p3 + o3 + e, rs.nt);
    ulonglong2 w;
    w.x = vec_apply<OP>(x.x, y.x, z.x, a0, a1, c);
    w.y = vec_apply<OP>(x.y, y.y, z.y, a0, a1, c);
    if (rs.nt) { u64x2_t v; v.x = w.x; v.y = w.y; __builtin_nontemporal_store(v, reinterpret_cast<u64x2_t*>(p3 + o3 + e)); }
    else *reinterpret_cast<ulonglong2*>(p3 + o3 + e) = w;
  }
}

typedef void (*vec_fn)(const u64*, const u64*, u64*, unsigned, ScalarPack, ScalarPack, const LimbConsts*, int, RowStrides);
template <int... I>
static const vec_fn* vec_table(std::integer_sequence<int, I...>) {
  static const vec_fn t[] = {vec_op_packed<I>...};
  return t;
}

int rh_vec_launch(rh_ring* r, int opcode, const u64* p1, const u64* p2, u64* p3, int npoly, int Lrows, int limb0,
                  const u64* s0, const u64* s1, int rows1, int rows2, int rows3, int half) {   // rows_k: limbs per poly of operand k (0: Lrows; rows2 < 0: one broadcast row); half: 0 all coefficients, 1 / 2 the first / second N/2
  static const vec_fn* table = vec_table(std::make_integer_sequence<int, RH_OP_COUNT>());
  (void)hipGetLastError();
  ScalarPack a, b;
  memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b));
  if (s0) memcpy(a.s, s0, (size_t)Lrows * 8);
  if (s1) memcpy(b.s, s1, (size_t)Lrows * 8);
  const unsigned rows = (unsigned)npoly * (unsigned)Lrows;
  if (rows == 0) return RH_OK;
  const unsigned n = (unsigned)r->N;
  unsigned chunks = (n / 2 + 256 * 4 - 1) / (256 * 4);
  if (chunks < 1) chunks = 1;
  if (chunks > 64) chunks = 64;
  const unsigned npairs = n >> 1;
  const RowStrides rs{rows1 ? rows1 : Lrows, rows2 ? rows2 : Lrows, rows3 ? rows3 : Lrows, half == 2 ? npairs / 2 : 0u, half == 1 ? npairs / 2 : npairs,
                      (r->nt_streams && (size_t)rows * (size_t)r->N * 8 >= ((size_t)512 << 20)) ? 1 : 0};
  hipLaunchKernelGGL(table[opcode], dim3(rows, chunks), dim3(256), 0, rh_stream(r), p1, p2, p3, n, a, b, r->d_consts + limb0, Lrows, rs);
  return check_launch("vec_op");
}

// p2 = ONE row of N words used for every (poly, limb): MulByVectorMontgomery(ThenAddLazy) (ring/operations.go:366-377) with the
// MUL_MONT / MUL_MONT_THEN_ADD_LAZY opcodes (any two-operand opcode works)
extern "C" int rh_ring_vec_op_bcast(rh_ring* r, int opcode, const uint64_t* p1, int rows1, const uint64_t* vector, uint64_t* p3, int rows3,
                                    int npoly, int level) {
  if (!r || !p1 || !vector || !p3) return rh_fail(RH_ERR_ARG, "vec_op_bcast: null argument");
  if (opcode < 0 || opcode >= RH_OP_COUNT || !op_reads_y(opcode)) return rh_fail(RH_ERR_ARG, "vec_op_bcast: opcode %d has no second operand", opcode);
  if (level < 0 || level >= r->L || rows1 < level + 1 || rows3 < level + 1 || npoly < 0) return rh_fail(RH_ERR_ARG, "vec_op_bcast: bad level / rows / npoly");
  (void)hipSetDevice(r->device);
  return rh_vec_launch(r, opcode, p1, vector, p3, npoly, level + 1, 0, nullptr, nullptr, rows1, -1, rows3);
}
// one-operand scalar opcodes with one RNS scalar for coefficients [0, N/2) and another for [N/2, N): Add / Sub / MulDoubleRNSScalar(ThenAdd)
// (ring/operations.go:167-184, 250-266); s_lo / s_hi: level+1 words each on the host, in the form the opcode takes them
extern "C" int rh_ring_vec_op_halves(rh_ring* r, int opcode, const uint64_t* p1, uint64_t* p3, int npoly, int level, const uint64_t* s_lo,
                                     const uint64_t* s_hi) {
  if (!r || !p1 || !p3 || !s_lo || !s_hi) return rh_fail(RH_ERR_ARG, "vec_op_halves: null argument");
  if (opcode < 0 || opcode >= RH_OP_COUNT || op_reads_y(opcode)) return rh_fail(RH_ERR_ARG, "vec_op_halves: opcode %d is not a one-operand scalar opcode", opcode);
  if (level < 0 || level >= r->L || npoly < 0) return rh_fail(RH_ERR_ARG, "vec_op_halves: bad level / npoly");
  if (r->N % 4) return rh_fail(RH_ERR_ARG, "vec_op_halves: N must be a multiple of 4");
  (void)hipSetDevice(r->device);
  if (int rc = rh_vec_launch(r, opcode, p1, nullptr, p3, npoly, level + 1, 0, s_lo, nullptr, 0, 0, 0, 1)) return rc;
  return rh_vec_launch(r, opcode, p1, nullptr, p3, npoly, level + 1, 0, s_hi, nullptr, 0, 0, 0, 2);
}

// Degree-1 x degree-1 tensoring of ckks mulRelin (schemes/ckks/evaluator.go:821-834) in one pass: the six ring calls
//   c00 = MForm(a0); c01 = MForm(a1); c0 = MulCoeffsMontgomery(c00, b0); c2 = MulCoeffsMontgomery(c01, b1);
//   c1 = MulCoeffsMontgomery(c00, b1); c1 = MulCoeffsMontgomeryThenAdd(c01, b0, c1)
// with the same formulas element by element (7 operands of traffic instead of 23).  Outputs may alias inputs element-wise.
__global__ void __launch_bounds__(256)
tensor_degree1_kernel(const u64* a0, const u64* a1, const u64* b0, const u64* b1, u64* c0, u64* c1, u64* c2, unsigned n,
                      const LimbConsts* __restrict__ consts, int L, int mform_first, int nt) {   // mform_first 0: no MForm (matrix_ckks.Evaluator.Mul, evaluator.go:166-173); nt: non-temporal streams
  typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
  auto ld = [&](const u64* p) {
    if (!nt) return *reinterpret_cast<const ulonglong2*>(p);
    const u64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(p));
    return make_ulonglong2(v.x, v.y);
  };
  auto st = [&](u64* p, const ulonglong2& w) {
    if (nt) { u64x2_t v; v.x = w.x; v.y = w.y; __builtin_nontemporal_store(v, reinterpret_cast<u64x2_t*>(p)); }
    else *reinterpret_cast<ulonglong2*>(p) = w;
  };
  const u32 row = blockIdx.x, limb = row % (u32)L;
  const LimbConsts c = consts[limb];
  const size_t ro = (size_t)row * n;
  for (unsigned i = blockIdx.y * blockDim.x + threadIdx.x; i < (n >> 1); i += gridDim.y * blockDim.x) {
    const size_t o = ro + 2 * (size_t)i;
    const ulonglong2 x0 = ld(a0 + o), x1 = ld(a1 + o);
    const ulonglong2 y0 = ld(b0 + o), y1 = ld(b1 + o);
    ulonglong2 r0, r1, r2;
    {
      const u64 m0 = mform_first ? mform(x0.x, c.q, c.bred0, c.bred1) : x0.x, m1 = mform_first ? mform(x1.x, c.q, c.bred0, c.bred1) : x1.x;
      r0.x = mred(m0, y0.x, c.q, c.qinv); r2.x = mred(m1, y1.x, c.q, c.qinv);
      r1.x = cred(mred(m0, y1.x, c.q, c.qinv) + mred(m1, y0.x, c.q, c.qinv), c.q);
    }
    {
      const u64 m0 = mform_first ? mform(x0.y, c.q, c.bred0, c.bred1) : x0.y, m1 = mform_first ? mform(x1.y, c.q, c.bred0, c.bred1) : x1.y;
      r0.y = mred(m0, y0.y, c.q, c.qinv); r2.y = mred(m1, y1.y, c.q, c.qinv);
      r1.y = cred(mred(m0, y1.y, c.q, c.qinv) + mred(m1, y0.y, c.q, c.qinv), c.q);
    }
    st(c0 + o, r0); st(c1 + o, r1); st(c2 + o, r2);
  }
}
extern "C" int rh_ring_tensor_degree1(rh_ring* r, const uint64_t* a0, const uint64_t* a1, const uint64_t* b0, const uint64_t* b1,
                                      uint64_t* c0, uint64_t* c1, uint64_t* c2, int npoly, int level, int mform_first) {
  if (!r || !a0 || !a1 || !b0 || !b1 || !c0 || !c1 || !c2) return rh_fail(RH_ERR_ARG, "tensor_degree1: null argument");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "tensor_degree1: level %d out of range [0,%d)", level, r->L);
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "tensor_degree1: npoly < 0");
  const unsigned rows = (unsigned)npoly * (unsigned)(level + 1), n = (unsigned)r->N;
  if (rows == 0) return RH_OK;
  (void)hipSetDevice(r->device);
  (void)hipGetLastError();
  unsigned chunks = (n / 2 + 1023) / 1024; if (chunks < 1) chunks = 1; if (chunks > 64) chunks = 64;
  tensor_degree1_kernel<<<dim3(rows, chunks), 256, 0, rh_stream(r)>>>(a0, a1, b0, b1, c0, c1, c2, n, r->d_consts, level + 1, mform_first,
                                                                      rh_streams_beyond_cache(r, rows) ? 1 : 0);
  return check_launch("tensor_degree1");
}

extern "C" int rh_ring_vec_op(rh_ring* r, int opcode, const uint64_t* p1, const uint64_t* p2, uint64_t* p3, int npoly, int level,
                              const uint64_t* s0, const uint64_t* s1) {
  if (!r || !p3) return rh_fail(RH_ERR_ARG, "vec_op: null argument");
  if (opcode < 0 || opcode >= RH_OP_COUNT) return rh_fail(RH_ERR_ARG, "vec_op: unknown opcode %d", opcode);
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "vec_op: level %d out of range [0,%d)", level, r->L);
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "vec_op: npoly < 0");
  if (op_reads_x(opcode) && !p1) return rh_fail(RH_ERR_ARG, "vec_op %d: p1 is null", opcode);
  if (op_reads_y(opcode) && !p2) return rh_fail(RH_ERR_ARG, "vec_op %d: p2 is null", opcode);
  (void)hipSetDevice(r->device);
  return rh_vec_launch(r, opcode, p1, p2, p3, npoly, level + 1, 0, s0, s1);
}
extern "C" int rh_ring_vec_op_rows(rh_ring* r, int opcode, const uint64_t* p1, int rows1, const uint64_t* p2, int rows2, uint64_t* p3, int rows3,
                                   int npoly, int level, const uint64_t* s0, const uint64_t* s1) {
  if (!r) return rh_fail(RH_ERR_ARG, "vec_op: null argument");
  if (level < 0 || level >= r->L) return rh_fail(RH_ERR_ARG, "vec_op: level %d out of range [0,%d)", level, r->L);
  const int need = level + 1;
  if ((p1 && rows1 < need) || (p2 && rows2 < need) || rows3 < need) return rh_fail(RH_ERR_ARG, "vec_op: a block has fewer than level+1 = %d limbs per poly", need);
  if ((!p1 || rows1 == need) && (!p2 || rows2 == need) && rows3 == need) return rh_ring_vec_op(r, opcode, p1, p2, p3, npoly, level, s0, s1);
  if (!p3) return rh_fail(RH_ERR_ARG, "vec_op: null argument");
  if (opcode < 0 || opcode >= RH_OP_COUNT) return rh_fail(RH_ERR_ARG, "vec_op: unknown opcode %d", opcode);
  if (npoly < 0) return rh_fail(RH_ERR_ARG, "vec_op: npoly < 0");
  if (op_reads_x(opcode) && !p1) return rh_fail(RH_ERR_ARG, "vec_op %d: p1 is null", opcode);
  if (op_reads_y(opcode) && !p2) return rh_fail(RH_ERR_ARG, "vec_op %d: p2 is null", opcode);
  (void)hipSetDevice(r->device);
  return rh_vec_launch(r, opcode, p1, p2, p3, npoly, need, 0, s0, s1, p1 ? rows1 : need, p2 ? rows2 : need, rows3);   // one launch, per-operand row strides
}
