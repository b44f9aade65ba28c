// ntt3n_kernels_asm.hip.hpp -- 3N transform, b = 1: the layer kernels with hand-scheduled bodies (tools/gen_tile_asm.py:
// gen_3n_pre_cols_fwd / gen_3n_cols_post_inv -> ntt3n_asm.inc).  Same arithmetic as ntt3n_pre_cols_fwd / ntt3n_cols_post_inv
// (ntt3n.hip): Shoup butterflies with the approximate quotient, values < 4q between layers, canonical inverse outputs.
#pragma once
#include "ntt_kernels_asm.hip.hpp"
#include "ntt3n_asm.inc"

#define RH_3N_LAYER_ASM(BODY)                                                                                                    \
  asm volatile(BODY : : [wbase] "s"(wbase), [pin] "s"(pin), [pout] "s"(pout), [tw] "s"(tw), [l3p] "s"(l3p), [tp] "s"(tp),      \
               [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)), [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4)      \
               : NTT3N_ASM_CLOBBERS)
// unit b = (limb, column block of 256, poly): 6 * 2^S1 coefficients per thread, rows of N = 6 * 2^(12 + S1) words
template <int S1, bool INV, bool NT = false>
RH_DEV void n3_layer_asm_body(const u32 b, const u64* in, u64* out, const N3Layer& a) {
  constexpr int log_n2 = 12 + S1;
  const u32 limb = b % (u32)a.L, rr = b / (u32)a.L;
  const size_t base = ((size_t)(rr >> 4) * a.L + limb) * a.N + (rr & 15) * 256;
  const u64 pin = uni64((u64)(size_t)(in + base)), pout = uni64((u64)(size_t)(out + base));
  const u64 tw = uni64((u64)(size_t)(a.stw + ((size_t)(limb * 6) << log_n2)));
  const u64 l3p = uni64((u64)(size_t)(a.l3 + limb)), tp = uni64((u64)(size_t)(a.r3 + (size_t)limb * a.r3_stride));
  const u64 q = uni64(a.c[limb].q);
  const u64 nq = (u64)0 - q, nq2 = (u64)0 - 2 * q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const u32 wbase = uni32(threadIdx.x & ~63u);
  if constexpr (INV && NT) {
    if constexpr (S1 == 3) RH_3N_LAYER_ASM(NTT3N_COLS_POST_INV8_ASM_BODY_NT);
    else if constexpr (S1 == 2) RH_3N_LAYER_ASM(NTT3N_COLS_POST_INV4_ASM_BODY_NT);
    else RH_3N_LAYER_ASM(NTT3N_COLS_POST_INV2_ASM_BODY_NT);
  } else if constexpr (INV) {
    if constexpr (S1 == 3) RH_3N_LAYER_ASM(NTT3N_COLS_POST_INV8_ASM_BODY);
    else if constexpr (S1 == 2) RH_3N_LAYER_ASM(NTT3N_COLS_POST_INV4_ASM_BODY);
    else RH_3N_LAYER_ASM(NTT3N_COLS_POST_INV2_ASM_BODY);
  } else if constexpr (NT) {
    if constexpr (S1 == 3) RH_3N_LAYER_ASM(NTT3N_PRE_COLS_FWD8_ASM_BODY_NT);
    else if constexpr (S1 == 2) RH_3N_LAYER_ASM(NTT3N_PRE_COLS_FWD4_ASM_BODY_NT);
    else RH_3N_LAYER_ASM(NTT3N_PRE_COLS_FWD2_ASM_BODY_NT);
  } else {
    if constexpr (S1 == 3) RH_3N_LAYER_ASM(NTT3N_PRE_COLS_FWD8_ASM_BODY);
    else if constexpr (S1 == 2) RH_3N_LAYER_ASM(NTT3N_PRE_COLS_FWD4_ASM_BODY);
    else RH_3N_LAYER_ASM(NTT3N_PRE_COLS_FWD2_ASM_BODY);
  }
}
#undef RH_3N_LAYER_ASM

template <int S1, bool INV, bool NT>
__global__ void __launch_bounds__(256)
ntt3n_layer_asm(const u64* in, u64* out, N3Layer a) { n3_layer_asm_body<S1, INV, NT>(blockIdx.x, in, out, a); }
// (A software-pipelined fusion with the sub-transforms' tile stages, the 3N counterpart of ntt_fwd_fused_asm, was measured and removed:
// both halves hold ~128 VGPRs, and spans of 48 ... 384 rows ran 0-12 % slower than the two separate launches.)
