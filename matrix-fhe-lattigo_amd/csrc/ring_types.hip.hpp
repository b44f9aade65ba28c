// ring_types.hip.hpp -- per-limb constant block, twiddle pair and tile geometry shared by host and device code.
#pragma once
#include "modarith.hip.hpp"

#define RH_MAX_LIMBS_K 64        // = RH_MAX_LIMBS (engine_internal.hpp): per-limb scalars passed to kernels by value
struct LimbConsts {
  u64 q, qinv, bred0, bred1;   // Modulus, MRedConstant, BRedConstant[0], [1]
  u64 nq;                      // 2^64 - q
  u64 ninv_w, ninv_wp;         // N^-1 mod q (standard form) and its Shoup quotient
  u64 ninv_mont;               // NInv as the reference stores it (Montgomery form)
};

struct tw2 { u64 w, wp; };     // Shoup pair: root in standard form, floor(root*2^64/q)

#define LT 12
#define TILE (1 << LT)
#define LDS_PAD(j) ((j) + ((j) >> 4))
#define LDS_WORDS (TILE + (TILE >> 4))

// per-limb constants of one rescale step at a given level (rescale.hip; also read by ntt_fwd_cols_expand)
struct RescaleLimb { u64 q, qinv, bred0, c /* MForm(q - qL^-1) */, s /* q - (h mod q) */; };

// 3N transform (ntt3n.hip)
struct Limb3N {          // per-limb constants of the non-radix-2 layers (Shoup pairs, standard form)
  tw2 zeta;              // omega^(N/2)
  tw2 w3;                // omega^N (primitive cube root)
  tw2 inv_b1;            // (z^5 - z)^-1 * (N/2)^-1
  tw2 inv_b0z;           // z * (z^5 - z)^-1 * (N/2)^-1
  tw2 inv_s;             // (N/2)^-1
};
struct N3Layer {          // arguments of the fused (column stages + radix-3 + split / merge) layer kernels of a 3N limb range
  const tw2* r3; int r3_stride; const Limb3N* l3; const LimbConsts* c; int L; const tw2* stw; int N;
};
