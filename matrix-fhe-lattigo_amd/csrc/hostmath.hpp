// hostmath.hpp -- host-side number theory for ring construction (product code, not the oracle).
//
// Mirrors what the reference computes once per modulus when a Ring is built: GenMRedConstant / GenBRedConstant
// (ring/modular_reduction.go:68-75, :99-107), IsPrime, PrimitiveRoot (ring/subring.go:218-251: smallest g >= 3 that is
// a generator), and generateNTTConstants (ring/subring.go:129-214).  All results are exact integers, so any correct
// method yields the values the Go code yields.
#pragma once
#include <cstdint>
#include <vector>
#include <algorithm>

namespace rh {
typedef unsigned __int128 u128;
typedef uint64_t u64;

inline u64 mulmod(u64 a, u64 b, u64 m) { return (u64)(((u128)a * b) % m); }
inline u64 powmod(u64 b, u64 e, u64 m) {
  u64 r = 1 % m; b %= m;
  while (e) { if (e & 1) r = mulmod(r, b, m); b = mulmod(b, b, m); e >>= 1; }
  return r;
}
inline u64 invmod_prime(u64 a, u64 p) { return powmod(a, p - 2, p); }

inline u64 gen_mred_constant(u64 q) {  // q^-1 mod 2^64 (Newton iteration; same value as the 63-squaring form)
  u64 x = q;                           // correct to 3 bits
  for (int i = 0; i < 6; ++i) x *= 2 - q * x;
  return x;
}
inline void gen_bred_constant(u64 q, u64 out[2]) {  // floor(2^128/q) -> {hi, lo}
  u128 top = (u128)1 << 64;
  u64 hi = (u64)(top / q);
  u128 rem = top % q;
  u64 lo = (u64)((rem << 64) / q);
  out[0] = hi; out[1] = lo;
}
inline u64 mform(u64 a, u64 q) { return (u64)((((u128)(a % q)) << 64) % q); }          // a*2^64 mod q
inline u64 imform(u64 a, u64 q) {                                                      // a*2^-64 mod q
  u64 r64 = (u64)((((u128)1) << 64) % q);
  return mulmod(a % q, invmod_prime(r64, q), q);
}
inline u64 shoup_quotient(u64 w, u64 q) { return (u64)((((u128)w) << 64) / q); }       // floor(w*2^64/q), w < q

inline bool is_prime(u64 n) {
  if (n < 2) return false;
  static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (u64 p : bases) { if (n % p == 0) return n == p; }
  u64 d = n - 1; int s = 0;
  while ((d & 1) == 0) { d >>= 1; ++s; }
  for (u64 a : bases) {
    u64 x = powmod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool comp = true;
    for (int r = 1; r < s; ++r) { x = mulmod(x, x, n); if (x == n - 1) { comp = false; break; } }
    if (comp) return false;
  }
  return true;
}
inline u64 gcd(u64 a, u64 b) { while (b) { u64 t = a % b; a = b; b = t; } return a; }
inline u64 pollard_rho(u64 n) {
  if ((n & 1) == 0) return 2;
  for (u64 c = 1;; ++c) {
    u64 x = 2, y = 2, d = 1;
    auto f = [&](u64 v) { return (u64)(((u128)v * v + c) % n); };
    while (d == 1) { x = f(x); y = f(f(y)); d = gcd(x > y ? x - y : y - x, n); }
    if (d != n) return d;
  }
}
inline void factor(u64 n, std::vector<u64>& out) {
  if (n == 1) return;
  if (is_prime(n)) { if (std::find(out.begin(), out.end(), n) == out.end()) out.push_back(n); return; }
  u64 d = pollard_rho(n);
  factor(d, out); factor(n / d, out);
}
inline u64 primitive_root(u64 q) {          // smallest generator >= 3 (the reference starts at g = 2 and pre-increments)
  std::vector<u64> fs; factor(q - 1, fs);
  for (u64 g = 3;; ++g) {
    bool ok = true;
    for (u64 f : fs) if (powmod(g, (q - 1) / f, q) == 1) { ok = false; break; }
    if (ok) return g;
  }
}
inline u64 bitrev(u64 x, int bits) { u64 r = 0; for (int i = 0; i < bits; ++i) r = (r << 1) | ((x >> i) & 1); return r; }

// generateNTTConstants for a power-of-two NthRoot/2: RootsForward[bitrev(j)] = psi^j * 2^64 mod q, same for psi^-1.
inline int gen_ntt_tables(u64 q, u64 nthroot, u64* rf, u64* rb, u64* ninv_mont) {
  if (!is_prime(q)) return -1;
  if (q % nthroot != 1) return -2;
  u64 g = primitive_root(q);
  u64 psi = powmod(g, (q - 1) / nthroot, q);
  u64 psiinv = invmod_prime(psi, q);
  u64 half = nthroot >> 1;
  int lg = 0; while (((u64)1 << lg) < half) ++lg;
  *ninv_mont = mform(invmod_prime(half % q, q), q);
  u64 r64 = mform(1, q);
  u64 cf = r64, cb = r64;
  for (u64 j = 0; j < half; ++j) {
    u64 idx = bitrev(j, lg);
    rf[idx] = cf; rb[idx] = cb;
    cf = mulmod(cf, psi, q); cb = mulmod(cb, psiinv, q);
  }
  return 0;
}
}  // namespace rh
