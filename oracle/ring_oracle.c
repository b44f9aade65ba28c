/*
 * ring_oracle.c -- CPU restatement of the reference `ring` hot path.  TEST INFRASTRUCTURE ONLY (see ring_oracle.h).
 *
 * Plain C (gcc, unsigned __int128 for 64x64->128 products).  Each function cites the reference file:line whose
 * algorithm it restates.  Nothing here is used by the product path.
 */
#include "ring_oracle.h"
#include "../include/ringhip_ops.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

static inline u64 mulhi(u64 a, u64 b) { return (u64)(((u128)a * b) >> 64); }

/* ------------------------------------------------------------------ scalar primitives (ring/modular_reduction.go) */

u64 orc_mform_lazy(u64 a, u64 q, const u64 bred[2]) { /* :40-45 */
  u64 mhi = mulhi(a, bred[1]);
  return (u64)0 - (a * bred[0] + mhi) * q;
}
u64 orc_mform(u64 a, u64 q, const u64 bred[2]) { /* :11-35 */
  u64 r = orc_mform_lazy(a, q, bred);
  return r >= q ? r - q : r;
}
u64 orc_imform_lazy(u64 a, u64 q, u64 qinv) { /* :61-65 */
  return q - mulhi(a * qinv, q);
}
u64 orc_imform(u64 a, u64 q, u64 qinv) { /* :49-56 */
  u64 r = orc_imform_lazy(a, q, qinv);
  return r >= q ? r - q : r;
}
u64 orc_gen_mred_constant(u64 q) { /* :68-75 : q^(2^63-1) = q^-1 mod 2^64 */
  u64 r = 1;
  for (int i = 0; i < 63; i++) { r *= q; q *= q; }
  return r;
}
void orc_gen_bred_constant(u64 q, u64 out[2]) { /* :99-107 : floor(2^128/q) split hi,lo */
  /* 2^128 / q by two-step long division */
  u128 top = ((u128)1 << 64);            /* 2^64 */
  u64 hi = (u64)(top / q);               /* floor(2^64/q) */
  u128 rem = top % q;                    /* < q */
  u128 num = rem << 64;                  /* rem*2^64 < q*2^64 fits */
  u64 lo = (u64)(num / q);
  out[0] = hi; out[1] = lo;
}
u64 orc_mred_lazy(u64 x, u64 y, u64 q, u64 qinv) { /* :90-95 */
  u128 p = (u128)x * y;
  u64 ahi = (u64)(p >> 64), alo = (u64)p;
  u64 H = mulhi(alo * qinv, q);
  return ahi - H + q;
}
u64 orc_mred(u64 x, u64 y, u64 q, u64 qinv) { /* :78-86 */
  u64 r = orc_mred_lazy(x, y, q, qinv);
  return r >= q ? r - q : r;
}
u64 orc_bred_add_lazy(u64 a, u64 q, const u64 bred[2]) { /* :121-124 */
  return a - mulhi(a, bred[0]) * q;
}
u64 orc_bred_add(u64 a, u64 q, const u64 bred[2]) { /* :110-117 */
  u64 r = orc_bred_add_lazy(a, q, bred);
  return r >= q ? r - q : r;
}
u64 orc_bred_lazy(u64 x, u64 y, u64 q, const u64 bred[2]) { /* :166-197 */
  u128 m = (u128)x * y;
  u64 mhi = (u64)(m >> 64), mlo = (u64)m;
  u64 r = mhi * bred[0];
  u128 t = (u128)mlo * bred[0];
  u64 hhi = (u64)(t >> 64), hlo = (u64)t;
  r += hhi;
  u64 lhi = mulhi(mlo, bred[1]);
  u64 s0 = hlo + lhi;
  r += (s0 < hlo);
  t = (u128)mhi * bred[1];
  hhi = (u64)(t >> 64); hlo = (u64)t;
  r += hhi;
  u64 s1 = hlo + s0;
  r += (s1 < hlo);
  return mlo - r * q;
}
u64 orc_bred(u64 x, u64 y, u64 q, const u64 bred[2]) { /* :127-162 */
  u64 r = orc_bred_lazy(x, y, q, bred);
  return r >= q ? r - q : r;
}
u64 orc_cred(u64 a, u64 q) { return a >= q ? a - q : a; } /* :200-205 */

u64 orc_modexp(u64 x, u64 e, u64 p) { /* ring/utils.go:30-40 (BRed-based square and multiply) */
  u64 brc[2]; orc_gen_bred_constant(p, brc);
  u64 result = 1;
  for (u64 i = e; i > 0; i >>= 1) {
    if (i & 1) result = orc_bred(result, x, p, brc);
    x = orc_bred(x, x, p, brc);
  }
  return result;
}

/* ------------------------------------------------------------------ number theory for table generation */

static u64 mulmod(u64 a, u64 b, u64 m) { return (u64)(((u128)a * b) % m); }
static u64 powmod(u64 b, u64 e, u64 m) {
  u64 r = 1; b %= m;
  while (e) { if (e & 1) r = mulmod(r, b, m); b = mulmod(b, b, m); e >>= 1; }
  return r;
}
int orc_is_prime(u64 n) { /* deterministic Miller-Rabin for 64-bit */
  if (n < 2) return 0;
  static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (size_t i = 0; i < 12; i++) { if (n % small[i] == 0) return n == small[i]; }
  u64 d = n - 1; int s = 0;
  while ((d & 1) == 0) { d >>= 1; s++; }
  for (size_t i = 0; i < 12; i++) {
    u64 x = powmod(small[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int r = 1; r < s; r++) { x = mulmod(x, x, n); if (x == n - 1) { comp = 0; break; } }
    if (comp) return 0;
  }
  return 1;
}
static u64 gcd64(u64 a, u64 b) { while (b) { u64 t = a % b; a = b; b = t; } return a; }
static u64 pollard_rho(u64 n) {
  if ((n & 1) == 0) return 2;
  for (u64 c = 1;; c++) {
    u64 x = 2, y = 2, d = 1;
    while (d == 1) {
      x = (mulmod(x, x, n) + c) % n;
      y = (mulmod(y, y, n) + c) % n; y = (mulmod(y, y, n) + c) % n;
      d = gcd64(x > y ? x - y : y - x, n);
    }
    if (d != n) return d;
  }
}
static void factor_rec(u64 n, u64* out, int* cnt) {
  if (n == 1) return;
  if (orc_is_prime(n)) { for (int i = 0; i < *cnt; i++) if (out[i] == n) return; out[(*cnt)++] = n; return; }
  u64 d = pollard_rho(n);
  factor_rec(d, out, cnt); factor_rec(n / d, out, cnt);
}
u64 orc_primitive_root(u64 q) { /* ring/subring.go:218-251: g = 2; loop { g++; test all prime factors of q-1 } */
  u64 f[64]; int nf = 0;
  factor_rec(q - 1, f, &nf);
  u64 g = 2;
  for (;;) {
    g++;
    int ok = 1;
    for (int i = 0; i < nf; i++) if (orc_modexp(g, (q - 1) / f[i], q) == 1) { ok = 0; break; }
    if (ok) return g;
  }
}

static u64 bitrev64(u64 x, int bits) { u64 r = 0; for (int i = 0; i < bits; i++) { r = (r << 1) | ((x >> i) & 1); } return r; }

int orc_gen_ntt_tables(u64 q, u64 nthroot, u64* rf, u64* rb, u64* ninv, u64* prim) { /* ring/subring.go:129-214 */
  if (!orc_is_prime(q)) return -1;
  if (q % nthroot != 1) return -2;
  u64 brc[2]; orc_gen_bred_constant(q, brc);
  u64 qinv = orc_gen_mred_constant(q);
  u64 g = orc_primitive_root(q);
  if (prim) *prim = g;
  *ninv = orc_mform(orc_modexp(nthroot >> 1, q - 2, q), q, brc);                       /* :177-190 */
  u64 psi = orc_mform(orc_modexp(g, (q - 1) / nthroot, q), q, brc);                    /* :193 */
  u64 psiinv = orc_mform(orc_modexp(g, q - ((q - 1) / nthroot) - 1, q), q, brc);        /* :194 */
  u64 half = nthroot >> 1;
  rf[0] = orc_mform(1, q, brc); rb[0] = rf[0];
  if ((half & (half - 1)) == 0) {                                                       /* :201-205 */
    int lg = 0; while (((u64)1 << lg) < half) lg++;
    for (u64 j = 1; j < half; j++) {
      u64 prev = bitrev64(j - 1, lg), cur = bitrev64(j, lg);
      rf[cur] = orc_mred(rf[prev], psi, q, qinv);
      rb[cur] = orc_mred(rb[prev], psiinv, q, qinv);
    }
  } else {                                                                              /* :206-211 */
    for (u64 j = 1; j < half; j++) {
      rf[j] = orc_mred(rf[j - 1], psi, q, qinv);
      rb[j] = orc_mred(rb[j - 1], psiinv, q, qinv);
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ negacyclic NTT (ring/ntt.go) */

/* butterfly :155-161 with the 4q conditional subtraction selectable (the unrolled code skips it on some stages) */
static inline void fwd_bfly(u64* X, u64* Y, u64 U, u64 V, u64 psi, u64 q, u64 qinv, int reduce) {
  if (reduce && U >= 4 * q) U -= 4 * q;
  V = orc_mred_lazy(V, psi, q, qinv);
  *X = U + V; *Y = U + 2 * q - V;
}
/* invbutterfly :164-171 */
static inline void inv_bfly(u64* X, u64* Y, u64 U, u64 V, u64 psi, u64 q, u64 qinv) {
  u64 x = U + V;
  if (x >= 2 * q) x -= 2 * q;
  *X = x; *Y = orc_mred_lazy(U + 4 * q - V, psi, q, qinv);
}

void orc_ntt_core_lazy(const u64* p1, u64* p2, int N, u64 q, u64 qinv, const u64* roots) {
  /* Stage m (m = 1,2,4,..,N/2), t = N/(2m): block i spans [2it, 2it+2t) with twiddle roots[m+i]  (:240-255).
   * Reduce schedule: N < 16 -> every stage reduces (:223-257).  N >= 16 -> the first stage (m=1) never reduces
   * (:271-310); a later stage reduces iff bit-length(m) is odd (:315-318), except the last stage (t == 1) which
   * always reduces (:500-517). */
  int t = N >> 1;
  for (int m = 1; m < N; m <<= 1, t >>= 1) {
    int reduce;
    if (N < 16) reduce = 1;
    else if (m == 1) reduce = 0;
    else if (t == 1) reduce = 1;
    else { int bl = 0; for (int mm = m; mm; mm >>= 1) bl++; reduce = bl & 1; }
    const u64* src = (m == 1) ? p1 : p2;
    for (int i = 0; i < m; i++) {
      u64 F = roots[m + i];
      int j1 = 2 * i * t;
      for (int j = j1; j < j1 + t; j++) fwd_bfly(&p2[j], &p2[j + t], src[j], src[j + t], F, q, qinv, reduce);
    }
  }
}

void orc_intt_core_lazy(const u64* p1, u64* p2, int N, u64 q, u64 qinv, const u64* roots) {
  /* GS: t = 1,2,4,...; stage with h = m/2 uses roots[h+i] (:568-606); identical for the unrolled form (:608-714). */
  int t = 1;
  for (int m = N; m > 1; m >>= 1, t <<= 1) {
    int h = m >> 1;
    const u64* src = (m == N) ? p1 : p2;
    for (int i = 0; i < h; i++) {
      u64 F = roots[h + i];
      int j1 = 2 * i * t;
      for (int j = j1; j < j1 + t; j++) inv_bfly(&p2[j], &p2[j + t], src[j], src[j + t], F, q, qinv);
    }
  }
}

void orc_ntt_standard(const u64* p1, u64* p2, int N, u64 q, u64 qinv, const u64 bred[2], const u64* roots) {
  orc_ntt_core_lazy(p1, p2, N, q, qinv, roots);
  for (int i = 0; i < N; i++) p2[i] = orc_bred_add(p2[i], q, bred);      /* reducevec, ring/vec_ops.go:125-145 */
}
void orc_ntt_standard_lazy(const u64* p1, u64* p2, int N, u64 q, u64 qinv, const u64* roots) {
  orc_ntt_core_lazy(p1, p2, N, q, qinv, roots);
}
void orc_intt_standard(const u64* p1, u64* p2, int N, u64 ninv, u64 q, u64 qinv, const u64* roots) {
  orc_intt_core_lazy(p1, p2, N, q, qinv, roots);
  for (int i = 0; i < N; i++) p2[i] = orc_mred(p2[i], ninv, q, qinv);    /* :187-193 (both branches are MRed) */
}
void orc_intt_standard_lazy(const u64* p1, u64* p2, int N, u64 ninv, u64 q, u64 qinv, const u64* roots) {
  orc_intt_core_lazy(p1, p2, N, q, qinv, roots);
  if (N < 16) { for (int i = 0; i < N; i++) p2[i] = orc_mred_lazy(p2[i], ninv, q, qinv); }   /* :199-202 */
  else        { for (int i = 0; i < N; i++) p2[i] = orc_mred(p2[i], ninv, q, qinv); }        /* :203-205 */
}

/* Conjugate-invariant NTT in Z[X+X^-1]/(X^2N+1): ring/ntt.go:716-1311.  roots: the 4N-th-root tables (2N entries).
 * Restates the non-unrolled forms nttConjugateInvariantLazy (:751-783) / inttConjugateInvariantLazy (:1113-1157) and
 * the canonical wrappers NTTConjugateInvariant (:717-720) / INTTConjugateInvariant (:728-731).  The unrolled forms
 * differ only in which stages apply the lazy 4q subtraction, which does not change the canonical result. */
void orc_ntt_ci(const u64* p1, u64* p2, int N, u64 q, u64 qinv, const u64 bred[2], const u64* roots) {
  u64 twoQ = 2 * q;
  u64 F = roots[1];
  u64* in = (u64*)malloc((size_t)N * 8); memcpy(in, p1, (size_t)N * 8);
  for (int jx = 1, jy = N - 1; jx < (N >> 1); jx++, jy--) {
    p2[jx] = in[jx] + twoQ - orc_mred_lazy(in[jy], F, q, qinv);
    p2[jy] = in[jy] + twoQ - orc_mred_lazy(in[jx], F, q, qinv);
  }
  p2[N >> 1] = in[N >> 1] + twoQ - orc_mred_lazy(in[N >> 1], F, q, qinv);
  p2[0] = in[0];
  free(in);
  int t = N;
  for (int m = 2; m < 2 * N; m <<= 1) {
    t >>= 1;
    int h = m >> 1;
    for (int i = 0; i < h; i++) {
      u64 W = roots[m + i];
      int j1 = 2 * i * t;
      for (int j = j1; j < j1 + t; j++) fwd_bfly(&p2[j], &p2[j + t], p2[j], p2[j + t], W, q, qinv, 1);
    }
  }
  for (int i = 0; i < N; i++) p2[i] = orc_bred_add(p2[i], q, bred);
}
void orc_intt_ci(const u64* p1, u64* p2, int N, u64 ninv, u64 q, u64 qinv, const u64* roots) {
  u64 twoQ = 2 * q;
  int t = 1;
  const u64* src = p1;
  for (int m = N; m > 1; m >>= 1, t <<= 1) {           /* first stage uses roots[N+i] (h = N/2), then roots[m+i], h = m/2 */
    int h = m >> 1;
    for (int i = 0; i < h; i++) {
      u64 W = roots[m + i];
      int j1 = 2 * i * t;
      for (int j = j1; j < j1 + t; j++) inv_bfly(&p2[j], &p2[j + t], src[j], src[j + t], W, q, qinv);
    }
    src = p2;
  }
  u64 F = roots[1];
  for (int jx = 1, jy = N - 1; jx < (N >> 1); jx++, jy--) {
    u64 a = p2[jx], b = p2[jy];
    p2[jx] = a + twoQ - orc_mred_lazy(b, F, q, qinv);
    p2[jy] = b + twoQ - orc_mred_lazy(a, F, q, qinv);
  }
  p2[N >> 1] = p2[N >> 1] + twoQ - orc_mred_lazy(p2[N >> 1], F, q, qinv);
  p2[0] = orc_cred(p2[0] << 1, q);
  for (int i = 0; i < N; i++) p2[i] = orc_mred(p2[i], ninv, q, qinv);
}

/* ------------------------------------------------------------------ element-wise kernels (ring/vec_ops.go) */

int orc_vec_op(int op, const u64* p1, const u64* p2, u64* p3, size_t n, u64 s0, u64 s1, u64 q, u64 qinv,
               const u64 bred[2]) {
  if (op < 0 || op >= RH_OP_COUNT) return -1;
  u64 q2 = 2 * q;
  for (size_t j = 0; j < n; j++) {
    u64 x = p1 ? p1[j] : 0, y = p2 ? p2[j] : 0, z = p3[j];
    switch (op) {
      case RH_OP_ADD: z = orc_cred(x + y, q); break;
      case RH_OP_ADD_LAZY: z = x + y; break;
      case RH_OP_SUB: z = orc_cred((x + q) - y, q); break;
      case RH_OP_SUB_LAZY: z = x + q - y; break;
      case RH_OP_NEG: z = q - x; break;
      case RH_OP_REDUCE: z = orc_bred_add(x, q, bred); break;
      case RH_OP_REDUCE_LAZY: z = orc_bred_add_lazy(x, q, bred); break;
      case RH_OP_MUL_LAZY: z = x * y; break;
      case RH_OP_MUL_LAZY_THEN_ADD_LAZY: z += x * y; break;
      case RH_OP_MUL_BARRETT: z = orc_bred(x, y, q, bred); break;
      case RH_OP_MUL_BARRETT_LAZY: z = orc_bred_lazy(x, y, q, bred); break;
      case RH_OP_MUL_BARRETT_THEN_ADD: z = orc_cred(z + orc_bred(x, y, q, bred), q); break;
      case RH_OP_MUL_BARRETT_THEN_ADD_LAZY: z += orc_bred(x, y, q, bred); break;
      case RH_OP_MUL_MONT: z = orc_mred(x, y, q, qinv); break;
      case RH_OP_MUL_MONT_LAZY: z = orc_mred_lazy(x, y, q, qinv); break;
      case RH_OP_MUL_MONT_THEN_ADD: z = orc_cred(z + orc_mred(x, y, q, qinv), q); break;
      case RH_OP_MUL_MONT_THEN_ADD_LAZY: z += orc_mred(x, y, q, qinv); break;
      case RH_OP_MUL_MONT_LAZY_THEN_ADD_LAZY: z += orc_mred_lazy(x, y, q, qinv); break;
      case RH_OP_MUL_MONT_THEN_SUB: z = orc_cred(z + (q - orc_mred(x, y, q, qinv)), q); break;
      case RH_OP_MUL_MONT_THEN_SUB_LAZY: z += (q - orc_mred(x, y, q, qinv)); break;
      case RH_OP_MUL_MONT_LAZY_THEN_SUB_LAZY: z += q2 - orc_mred_lazy(x, y, q, qinv); break;
      case RH_OP_MUL_MONT_LAZY_THEN_NEG: z = q2 - orc_mred_lazy(x, y, q, qinv); break;
      case RH_OP_ADD_LAZY_THEN_MUL_SCALAR_MONT: z = orc_mred(x + y, s0, q, qinv); break;
      case RH_OP_ADD_SCALAR_LAZY_THEN_MUL_SCALAR_MONT: z = orc_mred(x + s0, s1, q, qinv); break;
      case RH_OP_ADD_SCALAR: z = orc_cred(x + s0, q); break;
      case RH_OP_ADD_SCALAR_LAZY: z = x + s0; break;
      case RH_OP_ADD_SCALAR_LAZY_THEN_NEG_TWO_MODULUS_LAZY: z = s0 + q2 - x; break;
      case RH_OP_SUB_SCALAR: z = orc_cred(x + q - s0, q); break;
      case RH_OP_MUL_SCALAR_MONT: z = orc_mred(x, s0, q, qinv); break;
      case RH_OP_MUL_SCALAR_MONT_LAZY: z = orc_mred_lazy(x, s0, q, qinv); break;
      case RH_OP_MUL_SCALAR_MONT_THEN_ADD: z = orc_cred(z + orc_mred(x, s0, q, qinv), q); break;
      case RH_OP_MUL_SCALAR_MONT_THEN_ADD_SCALAR: z = orc_cred(orc_mred(x, s1, q, qinv) + s0, q); break;
      case RH_OP_SUB_THEN_MUL_SCALAR_MONT_TWO_MODULUS: z = orc_mred(q2 - y + x, s0, q, qinv); break;
      case RH_OP_MFORM: z = orc_mform(x, q, bred); break;
      case RH_OP_MFORM_LAZY: z = orc_mform_lazy(x, q, bred); break;
      case RH_OP_IMFORM: z = orc_imform(x, q, qinv); break;
      case RH_OP_ZERO: z = 0; break;
      case RH_OP_MASK: z = (x >> s0) & s1; break;
    }
    p3[j] = z;
  }
  return 0;
}

/* ------------------------------------------------------------------ RNS basis extension (ring/basis_extension.go) */

static u64 modexp_mont(u64 x, u64 e, u64 q, u64 qinv, const u64 bred[2]) { /* ring/utils.go:58-69; e is int(q-2) */
  u64 result = orc_mform(1, q, bred);
  for (u64 i = e; i > 0; i >>= 1) {
    if (i & 1) result = orc_mred(result, x, q, qinv);
    x = orc_mred(x, x, q, qinv);
  }
  return result;
}

orc_modup_constants* orc_gen_modup_constants(const u64* Q, int nq, const u64* P, int np) { /* :93-164 */
  orc_modup_constants* c = (orc_modup_constants*)calloc(1, sizeof(*c));
  c->nq = nq; c->np = np;
  c->qoverqiinvqi = (u64*)calloc((size_t)nq, 8);
  c->qoverqimodp = (u64*)calloc((size_t)np * nq, 8);
  c->vtimesqmodp = (u64*)calloc((size_t)np * (nq + 1), 8);
  u64 (*bq)[2] = (u64(*)[2])calloc((size_t)nq, 16); u64* mq = (u64*)calloc((size_t)nq, 8);
  u64 (*bp)[2] = (u64(*)[2])calloc((size_t)np, 16); u64* mp = (u64*)calloc((size_t)np, 8);
  for (int i = 0; i < nq; i++) { orc_gen_bred_constant(Q[i], bq[i]); mq[i] = orc_gen_mred_constant(Q[i]); }
  for (int j = 0; j < np; j++) { orc_gen_bred_constant(P[j], bp[j]); mp[j] = orc_gen_mred_constant(P[j]); }
  for (int i = 0; i < nq; i++) {
    u64 qi = Q[i];
    u64 star = orc_mform(1, qi, bq[i]);
    for (int j = 0; j < nq; j++) if (j != i) star = orc_mred(star, orc_mform(Q[j], qi, bq[i]), qi, mq[i]);
    c->qoverqiinvqi[i] = modexp_mont(star, qi - 2, qi, mq[i], bq[i]);                      /* :132 */
    for (int j = 0; j < np; j++) {
      u64 pj = P[j];
      u64 s = 1;                                                                           /* :136-143 */
      for (int u = 0; u < nq; u++) if (u != i) s = orc_mred(s, orc_mform(Q[u], pj, bp[j]), pj, mp[j]);
      c->qoverqimodp[(size_t)j * nq + i] = orc_mform(s, pj, bp[j]);
    }
  }
  for (int j = 0; j < np; j++) {                                                            /* :147-161 */
    u64 pj = P[j];
    u64 QmodP = 1;
    for (int i = 0; i < nq; i++) QmodP = orc_mred(QmodP, orc_mform(Q[i], pj, bp[j]), pj, mp[j]);
    u64 v = pj - QmodP;
    u64* row = c->vtimesqmodp + (size_t)j * (nq + 1);
    row[0] = 0;
    for (int i = 1; i < nq + 1; i++) row[i] = orc_cred(row[i - 1] + v, pj);
  }
  free(bq); free(mq); free(bp); free(mp);
  return c;
}
void orc_free_modup_constants(orc_modup_constants* c) {
  if (!c) return;
  free(c->qoverqiinvqi); free(c->qoverqimodp); free(c->vtimesqmodp); free(c);
}

/* one coefficient of reconstructRNS (:550-594) followed by multSum (:597-673) for every target limb.
 * src[i]: source residues (already offset by the caller when centered), nsrc <= 32.
 * tq/tqinv: target moduli; qoverqimodp row stride = rowstride; vtimes row stride = vstride. */
static void ext_one_coeff(const u64* x, int nsrc, const u64* srcQ, const u64* srcQinv, const u64* qoverqiinvqi,
                          u64* y /*[nsrc]*/, u64* vout) {
  double vi = 0.0;
  for (int i = 0; i < nsrc; i++) {
    y[i] = orc_mred(x[i], qoverqiinvqi[i], srcQ[i], srcQinv[i]);
    volatile double term = (double)y[i] / (double)srcQ[i];   /* separate rounding of the quotient, then the add */
    vi += term;
  }
  *vout = (u64)vi;
}
static u64 mult_sum(const u64* y, int nsrc, u64 v, u64 p, u64 pinv, const u64* vtimesqmodp_row,
                    const u64* qoverqimodp_row) {
  u128 acc = (u128)y[0] * qoverqimodp_row[0];
  u64 rlo = (u64)acc, rhi = (u64)(acc >> 64);
  for (int i = 1; i < nsrc; i++) {
    u128 m = (u128)y[i] * qoverqimodp_row[i];
    u64 mlo = (u64)m, mhi = (u64)(m >> 64);
    u64 s = rlo + mlo; u64 carry = s < rlo; rlo = s;
    rhi += mhi + carry;
  }
  u64 hhi = mulhi(rlo * pinv, p);
  return rhi - hhi + p + vtimesqmodp_row[v];
}

void orc_modup_exact(const u64* const* p1, u64* const* p2, size_t n, const u64* Q, const u64* P,
                     const orc_modup_constants* c) { /* :282-308 */
  int nq = c->nq, np = c->np;
  u64 qinv[32], pinv[64];
  for (int i = 0; i < nq; i++) qinv[i] = orc_gen_mred_constant(Q[i]);
  for (int j = 0; j < np; j++) pinv[j] = orc_gen_mred_constant(P[j]);
  u64 x[32], y[32];
  for (size_t k = 0; k < n; k++) {
    for (int i = 0; i < nq; i++) x[i] = p1[i][k];
    u64 v;
    ext_one_coeff(x, nq, Q, qinv, c->qoverqiinvqi, y, &v);
    for (int j = 0; j < np; j++)
      p2[j][k] = mult_sum(y, nq, v, P[j], pinv[j], c->vtimesqmodp + (size_t)j * (nq + 1),
                          c->qoverqimodp + (size_t)j * nq);
  }
}

/* small multi-precision helpers (little-endian 64-bit words) for floor(Q/2) mod p */
typedef struct { u64 w[40]; int n; } big_t;
static void big_set1(big_t* b) { memset(b, 0, sizeof(*b)); b->w[0] = 1; b->n = 1; }
static void big_mul_u64(big_t* b, u64 m) {
  u64 carry = 0;
  for (int i = 0; i < b->n; i++) { u128 t = (u128)b->w[i] * m + carry; b->w[i] = (u64)t; carry = (u64)(t >> 64); }
  if (carry) b->w[b->n++] = carry;
}
static void big_shr1(big_t* b) {
  for (int i = 0; i < b->n; i++) { b->w[i] = (b->w[i] >> 1) | ((i + 1 < b->n) ? (b->w[i + 1] << 63) : 0); }
  while (b->n > 1 && b->w[b->n - 1] == 0) b->n--;
}
static u64 big_mod_u64(const big_t* b, u64 m) {
  u128 r = 0;
  for (int i = b->n - 1; i >= 0; i--) { r = ((r << 64) | b->w[i]) % m; }
  return (u64)r;
}
static void half_product(const u64* Q, int nq, big_t* out) { big_set1(out); for (int i = 0; i < nq; i++) big_mul_u64(out, Q[i]); big_shr1(out); }

void orc_modup_centered(const u64* const* p1, u64* const* p2, size_t n, const u64* Q, int nq, const u64* P, int np) {
  /* ModUpQtoP :188-200 / ModUpPtoQ :205-217:  buff = p1 + floor(Q/2) (AddScalarBigint -> addscalarvec: CRed(x+s));
   * ModUpExact; p2 = p2 - floor(Q/2) (SubScalarBigint -> subscalarvec: CRed(x+p-s)). */
  big_t half; half_product(Q, nq, &half);
  orc_modup_constants* c = orc_gen_modup_constants(Q, nq, P, np);
  u64** buf = (u64**)calloc((size_t)nq, sizeof(u64*));
  for (int i = 0; i < nq; i++) {
    buf[i] = (u64*)malloc(n * 8);
    u64 s = big_mod_u64(&half, Q[i]);
    for (size_t k = 0; k < n; k++) buf[i][k] = orc_cred(p1[i][k] + s, Q[i]);
  }
  orc_modup_exact((const u64* const*)buf, p2, n, Q, P, c);
  for (int j = 0; j < np; j++) {
    u64 s = big_mod_u64(&half, P[j]);
    for (size_t k = 0; k < n; k++) p2[j][k] = orc_cred(p2[j][k] + P[j] - s, P[j]);
  }
  for (int i = 0; i < nq; i++) free(buf[i]);
  free(buf); orc_free_modup_constants(c);
}

/* genmodDownConstants :25-49 : constants[np-1][i] = prod_j p_j^-1 mod q_i in Montgomery form (running MRed product) */
static void moddown_constants(const u64* Q, int nq, const u64* P, int np, u64* out /*[nq]*/) {
  for (int i = 0; i < nq; i++) {
    u64 qi = Q[i]; u64 brc[2]; orc_gen_bred_constant(qi, brc); u64 qinv = orc_gen_mred_constant(qi);
    u64 acc = 0;
    for (int j = 0; j < np; j++) {
      u64 cst = orc_mform(orc_modexp(P[j], qi - 2, qi), qi, brc);
      if (j > 0) cst = orc_mred(cst, acc, qi, qinv);
      acc = cst;
    }
    out[i] = acc;
  }
}

void orc_moddown_qp_to_q(const u64* const* p1q, const u64* const* p1p, u64* const* p2q, size_t n,
                         const u64* Q, int nq, const u64* P, int np) { /* :223-234 */
  u64** buf = (u64**)calloc((size_t)nq, sizeof(u64*));
  for (int i = 0; i < nq; i++) buf[i] = (u64*)malloc(n * 8);
  orc_modup_centered(p1p, buf, n, P, np, Q, nq);
  u64 cst[64]; moddown_constants(Q, nq, P, np, cst);
  for (int i = 0; i < nq; i++) {
    u64 qi = Q[i], qinv = orc_gen_mred_constant(qi), s = qi - cst[i];
    for (size_t k = 0; k < n; k++)                 /* SubThenMulScalarMontgomeryTwoModulus(buff, p1Q, q - c, p2Q) */
      p2q[i][k] = orc_mred(2 * qi - p1q[i][k] + buf[i][k], s, qi, qinv);
    free(buf[i]);
  }
  free(buf);
}

void orc_moddown_qp_to_q_ntt(const u64* const* p1q, const u64* const* p1p, u64* const* p2q, size_t n,
                             const u64* Q, int nq, const u64* P, int np,
                             const u64* const* rootsQ_fwd, const u64* const* rootsP_bwd, const u64* ninvP) { /* :241-258 */
  u64** bp = (u64**)calloc((size_t)np, sizeof(u64*));
  u64** bq = (u64**)calloc((size_t)nq, sizeof(u64*));
  for (int j = 0; j < np; j++) {
    bp[j] = (u64*)malloc(n * 8);
    orc_intt_standard_lazy(p1p[j], bp[j], (int)n, ninvP[j], P[j], orc_gen_mred_constant(P[j]), rootsP_bwd[j]);
  }
  for (int i = 0; i < nq; i++) bq[i] = (u64*)malloc(n * 8);
  orc_modup_centered((const u64* const*)bp, bq, n, P, np, Q, nq);
  u64 cst[64]; moddown_constants(Q, nq, P, np, cst);
  for (int i = 0; i < nq; i++) {
    u64 qi = Q[i], qinv = orc_gen_mred_constant(qi), s = qi - cst[i];
    orc_ntt_standard_lazy(bq[i], bq[i], (int)n, qi, qinv, rootsQ_fwd[i]);
    for (size_t k = 0; k < n; k++) p2q[i][k] = orc_mred(2 * qi - p1q[i][k] + bq[i][k], s, qi, qinv);
    free(bq[i]);
  }
  for (int j = 0; j < np; j++) free(bp[j]);
  free(bp); free(bq);
}

void orc_decompose_and_split(int levelQ, int levelP, int nbPi, int digit, const u64* const* p0q,
                             u64* const* p1q, u64* const* p1p, size_t n,
                             const u64* Qall, int nQall, const u64* Pall, int nPall) { /* :381-502 */
  (void)nPall;
  int lvlQStart = digit * nbPi;
  int decompLvl;
  if (levelQ > nbPi * (digit + 1) - 1) decompLvl = nbPi - 2; else decompLvl = (levelQ % nbPi) - 1;   /* :394-399 */
  if (decompLvl < 0) {                                                                                /* :402-436 */
    u64 qd = Qall[lvlQStart];
    for (size_t j = 0; j < n; j++) {
      u64 coeff = p0q[lvlQStart][j];
      u64 pos = 1, neg = 0;
      if (coeff >= (qd >> 1)) { coeff = qd - coeff; pos = 0; neg = 1; }
      for (int i = 0; i < levelQ + 1; i++) {
        u64 brc[2]; orc_gen_bred_constant(Qall[i], brc);
        u64 tmp = orc_bred_add(coeff, Qall[i], brc);
        p1q[i][j] = tmp * pos + (Qall[i] - tmp) * neg;
      }
      for (int i = 0; i < levelP + 1; i++) {
        u64 brc[2]; orc_gen_bred_constant(Pall[i], brc);
        u64 tmp = orc_bred_add(coeff, Pall[i], brc);
        p1p[i][j] = tmp * pos + (Pall[i] - tmp) * neg;
      }
    }
    return;
  }
  int st = lvlQStart, ed = st + nbPi;
  if (ed > levelQ + 1) ed = levelQ + 1;
  int nsrc = ed - st;   /* == decompLvl + 2 */
  /* NewDecomposer :345-372: ModUpConstants[nbPi-2][digit][decompLvl] = GenModUpConstants(Q[digit*nbPi .. +decompLvl+2],
   * Q_all ++ P[:nbPi]) */
  int ntgt = nQall + nbPi;
  u64* tgt = (u64*)malloc((size_t)ntgt * 8);
  memcpy(tgt, Qall, (size_t)nQall * 8);
  for (int k = 0; k < nbPi; k++) tgt[nQall + k] = Pall[k];
  orc_modup_constants* c = orc_gen_modup_constants(Qall + st, nsrc, tgt, ntgt);
  big_t half; half_product(Qall + st, nsrc, &half);                                                  /* :457-468 */
  u64 halfmod[32], srcinv[32];
  for (int i = 0; i < nsrc; i++) { halfmod[i] = big_mod_u64(&half, Qall[st + i]); srcinv[i] = orc_gen_mred_constant(Qall[st + i]); }
  u64 x[32], y[32];
  for (size_t k = 0; k < n; k++) {
    for (int i = 0; i < nsrc; i++) x[i] = p0q[st + i][k] + halfmod[i];       /* reconstructRNSCentered :504-548 */
    u64 v;
    ext_one_coeff(x, nsrc, Qall + st, srcinv, c->qoverqiinvqi, y, &v);
    for (int j = 0; j < levelQ + 1; j++) {
      if (j >= st && j < ed) continue;
      p1q[j][k] = mult_sum(y, nsrc, v, Qall[j], orc_gen_mred_constant(Qall[j]), c->vtimesqmodp + (size_t)j * (nsrc + 1),
                           c->qoverqimodp + (size_t)j * nsrc);
    }
    for (int j = 0; j < levelP + 1; j++) {
      int u = nQall + j;
      p1p[j][k] = mult_sum(y, nsrc, v, Pall[j], orc_gen_mred_constant(Pall[j]), c->vtimesqmodp + (size_t)u * (nsrc + 1),
                           c->qoverqimodp + (size_t)u * nsrc);
    }
  }
  /* ringQ.SubScalarBigint(p1Q, QHalf, p1Q) at level levelQ: applies to EVERY limb 0..levelQ (incl. digit limbs) :499-500 */
  for (int j = 0; j < levelQ + 1; j++) {
    u64 s = big_mod_u64(&half, Qall[j]);
    for (size_t k = 0; k < n; k++) p1q[j][k] = orc_cred(p1q[j][k] + Qall[j] - s, Qall[j]);
  }
  for (int j = 0; j < levelP + 1; j++) {
    u64 s = big_mod_u64(&half, Pall[j]);
    for (size_t k = 0; k < n; k++) p1p[j][k] = orc_cred(p1p[j][k] + Pall[j] - s, Pall[j]);
  }
  free(tgt); orc_free_modup_constants(c);
}

/* ------------------------------------------------------------------ automorphisms (ring/automorphism.go) */

void orc_automorphism_ntt_index(int N, uint64_t nthroot, uint64_t gal, uint64_t* index) { /* :12-35 */
  int lg = 0; while (((u64)1 << (lg + 1)) < nthroot) lg++;        /* bits.Len64(NthRoot-1) - 1 */
  u64 mask = nthroot - 1;
  for (int i = 0; i < N; i++) {
    u64 t1 = 2 * bitrev64((u64)i, lg) + 1;
    u64 t2 = ((gal * t1 & mask) - 1) >> 1;
    index[i] = bitrev64(t2, lg);
  }
}
void orc_automorphism_ntt(const uint64_t* in, uint64_t* out, int N, uint64_t gal, int add_lazy) { /* :52-117 */
  u64* idx = (u64*)malloc((size_t)N * 8);
  orc_automorphism_ntt_index(N, (u64)2 * N, gal, idx);
  for (int j = 0; j < N; j++) out[j] = add_lazy ? out[j] + in[idx[j]] : in[idx[j]];
  free(idx);
}
/* NTT-domain automorphism with an explicit NthRoot (4N for conjugate-invariant rings, Ring.NthRoot() ring/ring.go:178-183) */
void orc_automorphism_ntt_nthroot(const uint64_t* in, uint64_t* out, int N, uint64_t nthroot, uint64_t gal, int add_lazy) {
  u64* idx = (u64*)malloc((size_t)N * 8);
  orc_automorphism_ntt_index(N, nthroot, gal, idx);
  for (int j = 0; j < N; j++) out[j] = add_lazy ? out[j] + in[idx[j]] : in[idx[j]];      /* caller guarantees idx[j] < N */
  free(idx);
}
/* coefficient-domain automorphism on a conjugate-invariant ring (:131-156) */
void orc_automorphism_ci(const uint64_t* in, uint64_t* out, int N, uint64_t gal, uint64_t q) {
  u64 n = (u64)N, mask = 2 * n - 1; int logN = 0; while (((u64)1 << logN) <= mask) logN++;   /* bits.Len64(mask) */
  for (u64 i = 0; i < 2 * n; i++) {
    u64 raw = i * gal, index = raw & mask, tmp = (raw >> logN) & 1;
    if (index < n) {
      u64 idx = i;
      if (idx >= n) { idx = 2 * n - idx; tmp ^= 1; }
      out[index] = in[idx] * (tmp ^ 1) | (q - in[idx]) * tmp;
    }
  }
}
void orc_automorphism(const uint64_t* in, uint64_t* out, int N, uint64_t gal, uint64_t q) { /* :162-175 */
  u64 mask = (u64)N - 1; int logN = 0; while (((u64)1 << logN) < (u64)N) logN++;
  for (u64 i = 0; i < (u64)N; i++) {
    u64 raw = i * gal, index = raw & mask, tmp = (raw >> logN) & 1;
    out[index] = in[i] * (tmp ^ 1) | (q - in[i]) * tmp;
  }
}

/* ------------------------------------------------------------------ RNS rescale (ring/scaling.go) */

/* DivFloorByLastModulus :21-28 (round = 0) / DivRoundByLastModulus :112-126 (round = 1), coefficient domain, one step at
 * `level`.  p0: level+1 limb pointers (modified in place in round mode exactly like the reference), p1: level limbs. */
void orc_div_by_last_modulus(int round, uint64_t* const* p0, uint64_t* const* p1, size_t n, const uint64_t* Q, int level) {
  u64 qL = Q[level];
  u64 pHalf = (qL - 1) >> 1;
  if (round) for (size_t k = 0; k < n; k++) p0[level][k] = orc_cred(p0[level][k] + pHalf, qL);          /* :120 */
  for (int i = 0; i < level; i++) {
    u64 qi = Q[i], qinv = orc_gen_mred_constant(qi);
    u64 brc[2]; orc_gen_bred_constant(qi, brc);
    u64 c = orc_mform(qi - orc_modexp(qL, qi - 2, qi), qi, brc);                                        /* ring/ring.go:363-380 */
    if (round) {
      u64 s = qi - orc_bred_add(pHalf, qi, brc);
      for (size_t k = 0; k < n; k++) {
        p0[i][k] = s + 2 * qi - p0[i][k];                                                                /* :123 */
        p1[i][k] = orc_mred(p0[level][k] + p0[i][k], c, qi, qinv);                                       /* :124 */
      }
    } else {
      for (size_t k = 0; k < n; k++) p1[i][k] = orc_mred(2 * qi - p0[i][k] + p0[level][k], c, qi, qinv); /* :26 */
    }
  }
}

/* ------------------------------------------------------------------ 3N-cyclotomic transform (ring/ntt_3n.go) */

static int gcd_int(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a < 0 ? -a : a; }
int orc_ntt3n_exponents(int threeN, int* out) { /* :235-243 */
  int c = 0;
  for (int e = 1; e < threeN; e++) if (gcd_int(e, threeN) == 1) out[c++] = e;
  return c;
}
void orc_ntt3n_forward_def(const u64* p1, u64* p2, int N, u64 q, u64 omega) { /* :82-109 */
  int* E = (int*)malloc((size_t)3 * N * sizeof(int));
  int cnt = orc_ntt3n_exponents(3 * N, E);
  u64* tmp = (u64*)malloc((size_t)N * 8);
  for (int k = 0; k < cnt && k < N; k++) {
    u64 xk = powmod(omega, (u64)E[k], q);
    u64 acc = 0;
    for (int j = N - 1; j >= 0; j--) { acc = mulmod(acc, xk, q); acc = orc_cred(acc + p1[j], q); }
    tmp[k] = acc;
  }
  memcpy(p2, tmp, (size_t)N * 8);
  free(tmp); free(E);
}
int orc_ntt3n_backward_def(const u64* p1, u64* p2, int N, u64 q, u64 omega) { /* :118-151 + :170-222 */
  int* E = (int*)malloc((size_t)3 * N * sizeof(int));
  orc_ntt3n_exponents(3 * N, E);
  u64* V = (u64*)malloc((size_t)N * N * 8);
  u64* y = (u64*)malloc((size_t)N * 8);
  for (int i = 0; i < N; i++) {
    u64 xi = powmod(omega, (u64)E[i], q);
    V[(size_t)i * N] = 1;
    for (int j = 1; j < N; j++) V[(size_t)i * N + j] = mulmod(V[(size_t)i * N + j - 1], xi, q);
    y[i] = p1[i];
  }
  for (int col = 0; col < N; col++) {
    int piv = col;
    while (piv < N && V[(size_t)piv * N + col] == 0) piv++;
    if (piv == N) { memset(p2, 0, (size_t)N * 8); free(E); free(V); free(y); return -1; }
    if (piv != col) {
      for (int j = 0; j < N; j++) { u64 t = V[(size_t)col * N + j]; V[(size_t)col * N + j] = V[(size_t)piv * N + j]; V[(size_t)piv * N + j] = t; }
      u64 t = y[col]; y[col] = y[piv]; y[piv] = t;
    }
    u64 inv = powmod(V[(size_t)col * N + col] % q, q - 2, q);
    for (int j = col; j < N; j++) V[(size_t)col * N + j] = mulmod(V[(size_t)col * N + j], inv, q);
    y[col] = mulmod(y[col], inv, q);
    for (int row = col + 1; row < N; row++) {
      u64 f = V[(size_t)row * N + col];
      if (f == 0) continue;
      for (int j = col; j < N; j++) V[(size_t)row * N + j] = orc_cred(V[(size_t)row * N + j] + q - mulmod(f, V[(size_t)col * N + j], q), q);
      y[row] = orc_cred(y[row] + q - mulmod(f, y[col], q), q);
    }
  }
  u64* a = (u64*)calloc((size_t)N, 8);
  for (int i = N - 1; i >= 0; i--) {
    u64 sum = 0;
    for (int j = i + 1; j < N; j++) sum = orc_cred(sum + mulmod(V[(size_t)i * N + j], a[j], q), q);
    a[i] = orc_cred(y[i] + q - sum, q);
  }
  for (int i = 0; i < N; i++) p2[i] = a[i] % q;
  free(a); free(E); free(V); free(y);
  return 0;
}

/* Fast transform, restating references/integer_dft.py.
 * Twiddle tree (:150-183): level 0 exponent 3N; level 1 {N/2, 5N/2}; radix-3 child exponents e/3, e/3+N, e/3+2N;
 * radix-2 child exponents e/2, e/2+3N/2.  Slot s of the last level evaluates at w^tree[last][s]; the Go
 * transformer's order is ascending exponent, i.e. rank(e) = 2*(e/6) + (e%6==5) among the totatives of 3N. */
typedef struct { int a, b; int levels; int* tree; /* (levels+1) x N */ } tree3n;
static int build_tree3n(int N, tree3n* t) {
  int a = 0, b = 0, m = N;
  while (m % 2 == 0) { m /= 2; a++; }
  while (m % 3 == 0) { m /= 3; b++; }
  if (m != 1 || a < 1) return -1;
  t->a = a; t->b = b; t->levels = a + b;
  t->tree = (int*)calloc((size_t)(t->levels + 1) * N, sizeof(int));
  int* T = t->tree;
  T[0] = 3 * N;
  T[N + 0] = 3 * N / 6; T[N + 1] = 5 * (3 * N) / 6;
  int cnt = 2;
  for (int ll = 1; ll <= b; ll++) {
    for (int ii = 0; ii < cnt; ii++) {
      int e = T[ll * N + ii] / 3;
      T[(ll + 1) * N + 3 * ii] = e;
      T[(ll + 1) * N + 3 * ii + 1] = e + N;
      T[(ll + 1) * N + 3 * ii + 2] = e + 2 * N;
    }
    cnt *= 3;
  }
  for (int ll = b + 1; ll < t->levels; ll++) {
    for (int ii = 0; ii < cnt; ii++) {
      int e = T[ll * N + ii] / 2;
      T[(ll + 1) * N + 2 * ii] = e;
      T[(ll + 1) * N + 2 * ii + 1] = e + 3 * N / 2;
    }
    cnt *= 2;
  }
  return 0;
}
static inline int rank3n(int e) { return 2 * (e / 6) + ((e % 6) == 5); }
static inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static inline u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }

int orc_ntt3n_forward_fast(const u64* p1, u64* p2, int N, u64 q, u64 omega) { /* integer_dft.py:266-348 */
  tree3n t; if (build_tree3n(N, &t)) return -1;
  u64* b = (u64*)malloc((size_t)N * 8);
  for (int i = 0; i < N; i++) b[i] = p1[i] % q;
  const int* T = t.tree;
  u64 w3 = powmod(omega, (u64)N, q);   /* primitive cube root: w^(3N/3) */
  /* level 1: X^N - X^(N/2) + 1 = (X^(N/2) - z)(X^(N/2) - z^5), z = w^(N/2) (:289-297) */
  {
    u64 zeta = powmod(omega, (u64)T[N + 0], q);
    int h = N / 2;
    for (int i = 0; i < h; i++) {
      u64 tt = mulmod(zeta, b[i + h], q);
      u64 lo = addmod(b[i], tt, q);
      u64 hi = submod(addmod(b[i], b[i + h], q), tt, q);
      b[i] = lo; b[i + h] = hi;
    }
  }
  /* radix-3 layers (:300-322) */
  int step = N / 6, cnt = 2, lvl = 1;
  for (int l = 0; l < t.b; l++, lvl++) {
    for (int blk = 0; blk < cnt; blk++) {
      int e = T[(lvl + 1) * N + 3 * blk];
      u64 z1 = powmod(omega, (u64)e, q), z2 = powmod(omega, (u64)2 * e, q);
      int start = blk * 3 * step;
      for (int i = start; i < start + step; i++) {
        u64 t1 = mulmod(z1, b[i + step], q), t2 = mulmod(z2, b[i + 2 * step], q);
        u64 t3 = mulmod(w3, submod(t1, t2, q), q);
        u64 b0 = b[i];
        b[i + 2 * step] = submod(submod(b0, t1, q), t3, q);
        b[i + step] = addmod(submod(b0, t2, q), t3, q);
        b[i] = addmod(addmod(b0, t1, q), t2, q);
      }
    }
    cnt *= 3; step /= 3;
  }
  /* radix-2 layers (:325-343) */
  step = 1 << (t.a - 2 >= 0 ? t.a - 2 : 0);
  if (t.a >= 2) {
    for (; step >= 1; step >>= 1, lvl++) {
      for (int blk = 0; blk < cnt; blk++) {
        u64 z = powmod(omega, (u64)T[(lvl + 1) * N + 2 * blk], q);
        int start = blk * 2 * step;
        for (int i = start; i < start + step; i++) {
          u64 tt = mulmod(z, b[i + step], q);
          u64 u = b[i];
          b[i] = addmod(u, tt, q); b[i + step] = submod(u, tt, q);
        }
      }
      cnt *= 2;
      if (step == 1) { lvl++; break; }
    }
  }
  u64* out = (u64*)malloc((size_t)N * 8);
  for (int s = 0; s < N; s++) out[rank3n(T[t.levels * N + s])] = b[s];
  memcpy(p2, out, (size_t)N * 8);
  free(out); free(b); free(t.tree);
  return 0;
}

int orc_ntt3n_backward_fast(const u64* p1, u64* p2, int N, u64 q, u64 omega) { /* integer_dft.py:350-432 */
  tree3n t; if (build_tree3n(N, &t)) return -1;
  const int* T = t.tree;
  u64* b = (u64*)malloc((size_t)N * 8);
  for (int s = 0; s < N; s++) b[s] = p1[rank3n(T[t.levels * N + s])] % q;
  u64 w3 = powmod(omega, (u64)N, q);
  int lvl = t.levels;           /* level whose children we are merging is lvl-1 -> uses T[lvl] exponents */
  int cnt = N;                  /* number of slots at current level (each of size 1) */
  int step = 1;
  /* inverse radix-2 layers (:372-389): b[i+step] = (t - b[i]) * zeta ; b[i] = t + b[i]  with t = old b[i+step]
   * -- this is the un-normalised inverse using zeta = w^e of the merged node (zetas list walked backwards). */
  for (int l = 0; l < t.a - 1; l++) {
    cnt /= 2;
    for (int blk = 0; blk < cnt; blk++) {
      /* forward used zeta = w^(T[lvl][2*blk]); inverse multiplies by its modular inverse */
      u64 z = powmod(omega, (u64)T[lvl * N + 2 * blk], q);
      u64 zi = powmod(z, q - 2, q);
      int start = blk * 2 * step;
      for (int i = start; i < start + step; i++) {
        u64 u = b[i], v = b[i + step];
        b[i] = addmod(u, v, q);
        b[i + step] = mulmod(submod(u, v, q), zi, q);
      }
    }
    step *= 2; lvl--;
  }
  /* inverse radix-3 layers: invert  [b0';b1';b2'] = M [b0; z1 b1; z2 b2]  exactly */
  u64 w3sq = mulmod(w3, w3, q);
  for (int l = 0; l < t.b; l++) {
    cnt /= 3;
    for (int blk = 0; blk < cnt; blk++) {
      int e = T[lvl * N + 3 * blk];
      u64 z1i = powmod(powmod(omega, (u64)e, q), q - 2, q), z2i = powmod(powmod(omega, (u64)2 * e, q), q - 2, q);
      int start = blk * 3 * step;
      for (int i = start; i < start + step; i++) {
        /* forward: B0 = b0+t1+t2; B1 = b0 - t2 + w3(t1-t2) = b0 + w3 t1 + w3^2 t2 ; B2 = b0 - t1 - w3(t1-t2) = b0 + w3^2 t1 + w3 t2
         * (since 1 + w3 + w3^2 = 0).  Inverse DFT-3: b0 = (B0+B1+B2); t1 = B0 + w3^2 B1 + w3 B2 ; t2 = B0 + w3 B1 + w3^2 B2 (each /3,
         * folded into the final scaling). */
        u64 B0 = b[i], B1 = b[i + step], B2 = b[i + 2 * step];
        u64 s0 = addmod(addmod(B0, B1, q), B2, q);
        u64 s1 = addmod(addmod(B0, mulmod(w3sq, B1, q), q), mulmod(w3, B2, q), q);
        u64 s2 = addmod(addmod(B0, mulmod(w3, B1, q), q), mulmod(w3sq, B2, q), q);
        b[i] = s0; b[i + step] = mulmod(s1, z1i, q); b[i + 2 * step] = mulmod(s2, z2i, q);
      }
    }
    step *= 3; lvl--;
  }
  /* final layer: forward was lo = b0 + z b1 ; hi = b0 + (1 - z) b1 = b0 + z^5... (z + z^5 = 1).  Solve, then scale:
   * accumulated scale so far is 2^(a-1) * 3^b = N/2, and this layer's determinant adds (z5 - z)^-1. */
  {
    int h = N / 2;
    u64 z = powmod(omega, (u64)(N / 2), q), z5 = powmod(z, 5, q);
    u64 dinv = powmod(submod(z5, z, q), q - 2, q);          /* (hi - lo) = (z5 - z) b1 */
    u64 sinv = powmod((u64)(N / 2) % q, q - 2, q);
    for (int i = 0; i < h; i++) {
      u64 lo = b[i], hi = b[i + h];
      u64 b1 = mulmod(submod(hi, lo, q), dinv, q);
      u64 b0 = submod(lo, mulmod(z, b1, q), q);
      b[i] = mulmod(b0, sinv, q); b[i + h] = mulmod(b1, sinv, q);
    }
  }
  memcpy(p2, b, (size_t)N * 8);
  free(b); free(t.tree);
  return 0;
}

/* ------------------------------------------------------------------ cpu_baseline timing (bench.py only) */

typedef struct { int N, nlimbs, reps, tid, nthreads; const u64* moduli; u64** roots; u64** data; u64* qinv; u64 (*bred)[2]; } tw_t;
static void* time_worker(void* arg) {
  tw_t* w = (tw_t*)arg;
  for (int r = 0; r < w->reps; r++)
    for (int l = w->tid; l < w->nlimbs; l += w->nthreads)
      orc_ntt_standard(w->data[l], w->data[l], w->N, w->moduli[l], w->qinv[l], w->bred[l], w->roots[l]);
  return NULL;
}
double orc_time_ntt_forward(int N, int nlimbs, const u64* moduli, int reps, int threads) {
  u64** roots = (u64**)calloc((size_t)nlimbs, sizeof(u64*));
  u64** data = (u64**)calloc((size_t)nlimbs, sizeof(u64*));
  u64* qinv = (u64*)calloc((size_t)nlimbs, 8);
  u64 (*bred)[2] = (u64(*)[2])calloc((size_t)nlimbs, 16);
  u64* rb = (u64*)malloc((size_t)N * 8);
  for (int l = 0; l < nlimbs; l++) {
    roots[l] = (u64*)malloc((size_t)N * 8); data[l] = (u64*)malloc((size_t)N * 8);
    u64 ninv;
    if (orc_gen_ntt_tables(moduli[l], (u64)2 * N, roots[l], rb, &ninv, NULL)) return -1.0;
    qinv[l] = orc_gen_mred_constant(moduli[l]); orc_gen_bred_constant(moduli[l], bred[l]);
    u64 s = 0x5eed + (u64)l;
    for (int i = 0; i < N; i++) { s += 0x9e3779b97f4a7c15ull; u64 z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31; data[l][i] = z % moduli[l]; }
  }
  free(rb);
  if (threads < 1) threads = 1;
  pthread_t th[256]; tw_t args[256];
  if (threads > 256) threads = 256;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < threads; t++) {
    args[t] = (tw_t){N, nlimbs, reps, t, threads, moduli, roots, data, qinv, bred};
    pthread_create(&th[t], NULL, time_worker, &args[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (int l = 0; l < nlimbs; l++) { free(roots[l]); free(data[l]); }
  free(roots); free(data); free(qinv); free(bred);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* The same work spread over `threads` host threads by (poly, limb) unit: npolys independent polys of nlimbs limbs, each
 * unit with its own data, units dealt round-robin.  What a goroutine-per-(poly, limb) caller of the reference could reach on
 * the cores this process may use.  Returns seconds for reps passes over all units. */
typedef struct { int N, nlimbs, nunits, reps, tid, nthreads; const u64* moduli; u64** roots; u64** data; u64* qinv; u64 (*bred)[2]; } twp_t;
static void* time_worker_polys(void* arg) {
  twp_t* w = (twp_t*)arg;
  for (int r = 0; r < w->reps; r++)
    for (int u = w->tid; u < w->nunits; u += w->nthreads) {
      const int l = u % w->nlimbs;
      orc_ntt_standard(w->data[u], w->data[u], w->N, w->moduli[l], w->qinv[l], w->bred[l], w->roots[l]);
    }
  return NULL;
}
double orc_time_ntt_forward_polys(int N, int nlimbs, const u64* moduli, int npolys, int reps, int threads) {
  if (npolys < 1 || nlimbs < 1) return -1.0;
  const int nunits = npolys * nlimbs;
  u64** roots = (u64**)calloc((size_t)nlimbs, sizeof(u64*));
  u64** data = (u64**)calloc((size_t)nunits, sizeof(u64*));
  u64* qinv = (u64*)calloc((size_t)nlimbs, 8);
  u64 (*bred)[2] = (u64(*)[2])calloc((size_t)nlimbs, 16);
  u64* rb = (u64*)malloc((size_t)N * 8);
  for (int l = 0; l < nlimbs; l++) {
    roots[l] = (u64*)malloc((size_t)N * 8);
    u64 ninv;
    if (orc_gen_ntt_tables(moduli[l], (u64)2 * N, roots[l], rb, &ninv, NULL)) return -1.0;
    qinv[l] = orc_gen_mred_constant(moduli[l]); orc_gen_bred_constant(moduli[l], bred[l]);
  }
  for (int u = 0; u < nunits; u++) {
    data[u] = (u64*)malloc((size_t)N * 8);
    u64 s = 0x5eed + (u64)u;
    for (int i = 0; i < N; i++) { s += 0x9e3779b97f4a7c15ull; u64 z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31; data[u][i] = z % moduli[u % nlimbs]; }
  }
  free(rb);
  if (threads < 1) threads = 1;
  if (threads > 1024) threads = 1024;
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  twp_t* args = (twp_t*)calloc((size_t)threads, sizeof(twp_t));
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < threads; t++) {
    args[t] = (twp_t){N, nlimbs, nunits, reps, t, threads, moduli, roots, data, qinv, bred};
    pthread_create(&th[t], NULL, time_worker_polys, &args[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (int l = 0; l < nlimbs; l++) free(roots[l]);
  for (int u = 0; u < nunits; u++) free(data[u]);
  free(roots); free(data); free(qinv); free(bred); free(th); free(args);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
