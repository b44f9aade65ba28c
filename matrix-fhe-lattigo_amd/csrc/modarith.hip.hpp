// modarith.hip.hpp -- 64-bit modular arithmetic for gfx950 device code.
//
// Two families:
//  (1) the reference's primitives, formula for formula (ring/modular_reduction.go): used wherever the result is not
//      a canonical residue (lazy forms) and bit-exactness therefore depends on the exact formula;
//  (2) a Shoup-form multiply by a precomputed constant (w, w' = floor(w*2^64/q)) used inside the NTT butterflies
//      whose final outputs are canonical ([0,q)) and therefore independent of how intermediate values are
//      represented.  Measured on MI355X (profiles/r01_micro_butterfly_rates.txt): 1.85-1.9 T butterflies/s
//      against 1.27-1.33 T for the Montgomery butterfly.
//
// gfx950 facts behind the formulations (profiles/r01_micro_valu_issue_rates.txt): v_mad_u64_u32, v_mul_lo_u32,
// v_mul_hi_u32, v_lshl_add_u64, v_add3_u32, v_bfi_b32 all issue at ~4.3 cycles per wave64; only
// v_add_u32/v_sub_u32/v_and/v_xor/v_mov/v_ashrrev issue at ~2.4.  So the cost of a butterfly is its VALU
// instruction count, and 64-bit adds go through v_lshl_add_u64 (one instruction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef unsigned int u32;
typedef unsigned __int128 u128;

#define RH_DEV __device__ __forceinline__

RH_DEV u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }

// wave-uniform values the compiler cannot prove uniform -> SGPRs (operands of the hand-scheduled bodies)
RH_DEV u32 uni32(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }
RH_DEV u64 uni64(u64 x) { return ((u64)uni32((u32)(x >> 32)) << 32) | uni32((u32)x); }

// ---- reference primitives (ring/modular_reduction.go) --------------------------------------------------------

RH_DEV u64 cred(u64 a, u64 q) { return a >= q ? a - q : a; }                       // :200-205
RH_DEV u64 mred_lazy(u64 x, u64 y, u64 q, u64 qinv) {                              // :90-95
  u128 p = (u128)x * y;
  u64 alo = (u64)p, ahi = (u64)(p >> 64);
  u64 H = mulhi64(alo * qinv, q);
  return ahi - H + q;
}
RH_DEV u64 mred(u64 x, u64 y, u64 q, u64 qinv) { return cred(mred_lazy(x, y, q, qinv), q); }   // :78-86
RH_DEV u64 bred_add_lazy(u64 a, u64 q, u64 b0) { return a - mulhi64(a, b0) * q; }  // :121-124
RH_DEV u64 bred_add(u64 a, u64 q, u64 b0) { return cred(bred_add_lazy(a, q, b0), q); }          // :110-117
RH_DEV u64 bred_lazy(u64 x, u64 y, u64 q, u64 b0, u64 b1) {                        // :166-197
  u128 m = (u128)x * y;
  u64 mhi = (u64)(m >> 64), mlo = (u64)m;
  u64 r = mhi * b0;
  u128 t = (u128)mlo * b0;
  u64 hhi = (u64)(t >> 64), hlo = (u64)t;
  r += hhi;
  u64 lhi = mulhi64(mlo, b1);
  u64 s0 = hlo + lhi;
  r += (u64)(s0 < hlo);
  t = (u128)mhi * b1;
  hhi = (u64)(t >> 64); hlo = (u64)t;
  r += hhi;
  u64 s1 = hlo + s0;
  r += (u64)(s1 < hlo);
  return mlo - r * q;
}
RH_DEV u64 bred(u64 x, u64 y, u64 q, u64 b0, u64 b1) { return cred(bred_lazy(x, y, q, b0, b1), q); }  // :127-162
RH_DEV u64 mform_lazy(u64 a, u64 q, u64 b0, u64 b1) {                              // :40-45
  u64 mhi = mulhi64(a, b1);
  return (u64)0 - (a * b0 + mhi) * q;
}
RH_DEV u64 mform(u64 a, u64 q, u64 b0, u64 b1) { return cred(mform_lazy(a, q, b0, b1), q); }    // :11-35
RH_DEV u64 imform_lazy(u64 a, u64 q, u64 qinv) { return q - mulhi64(a * qinv, q); }             // :61-65
RH_DEV u64 imform(u64 a, u64 q, u64 qinv) { return cred(imform_lazy(a, q, qinv), q); }          // :49-56

// ---- Shoup-form multiply by a constant ------------------------------------------------------------------------

// r == V*w (mod q), 0 <= r < 4q, for ANY 64-bit V.  wp = floor(w*2^64/q), nq = 2^64 - q.
// Q' = V1*wp1 + hi32(V1*wp0) + hi32(V0*wp1) under-estimates floor(V*wp/2^64) by at most 2 (the dropped low partial
// products sum to < 3 units), and the exact Shoup remainder is < 2q, hence r < 4q.  9 multiplies.
// `acc` is added to the result (the mad chain takes a 64-bit addend for free): returns acc + r (mod 2^64).
RH_DEV u64 shoup_mul_acc(u64 V, u64 w, u64 wp, u64 nq, u64 acc) {
  u32 V0 = (u32)V, V1 = (u32)(V >> 32), p0 = (u32)wp, p1 = (u32)(wp >> 32);
  u32 a = __umulhi(V1, p0), b = __umulhi(V0, p1);
  u64 Q = (u64)V1 * p1 + a;
  Q += b;
  u32 Q0 = (u32)Q, Q1 = (u32)(Q >> 32), w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
  u64 t = (u64)V0 * w0 + acc;
  t = (u64)Q0 * n0 + t;
  u32 hi = (u32)(t >> 32);
  hi = hi + V0 * w1 + V1 * w0;
  hi = hi + Q0 * n1 + Q1 * n0;
  return ((u64)hi << 32) | (u32)t;
}
RH_DEV u64 shoup_mul(u64 V, u64 w, u64 wp, u64 nq) { return shoup_mul_acc(V, w, wp, nq, 0); }

// exact Shoup: r == V*w (mod q), 0 <= r < 2q for any 64-bit V
RH_DEV u64 shoup_mul_exact(u64 V, u64 w, u64 wp, u64 q) {
  u64 Q = mulhi64(V, wp);
  return V * w - Q * q;
}

// x in [0, 2*bound) -> [0, bound): branch-free conditional subtract through the sign of (x - bound).
// Requires 2*bound <= 2^64 and bound < 2^63 (bound is 4q, 2q or q with q < 2^61).
RH_DEV u64 csub(u64 x, u64 bound) {
  u64 t = x - bound;
  return ((long long)t < 0) ? x : t;
}
// any x < 8q -> canonical [0,q)
RH_DEV u64 canon8(u64 x, u64 q) { x = csub(x, 4 * q); x = csub(x, 2 * q); return csub(x, q); }
RH_DEV u64 canon4(u64 x, u64 q) { x = csub(x, 2 * q); return csub(x, q); }
