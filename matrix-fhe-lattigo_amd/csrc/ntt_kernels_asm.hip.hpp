// ntt_kernels_asm.hip.hpp -- forward 4096-tile NTT kernel with a hand-scheduled gfx950 body (tools/gen_tile_asm.py).
//
// Same contract as ntt_fwd_tile<ShoupPolicy> with canonical output (ntt_kernels.hip.hpp): last 12 stages of the
// forward negacyclic NTT (ring/ntt.go:209-552 + reducevec) on one contiguous 4096-coefficient tile.  The C++ wrapper
// resolves (poly, limb, tile), loads the per-limb constants through the scalar cache and hands everything to one asm
// statement that owns v0..v123, s36..s99 and vcc.
#pragma once
#include "ring_types.hip.hpp"
#include "ntt_kernels.hip.hpp"
#include "ntt_tile_asm.inc"
#include "ntt_ci_asm.inc"


// gap_len > 0: the L transformed rows of a poly skip the limbs [gap0, gap0 + gap_len) of its Ls rows (the digit's own limbs of a
// hybrid key-switch decomposition): row l is limb l + (l >= gap0 ? gap_len : 0)
// LAZY: no final canonical reduction, outputs < 8q (internal consumers only)
// NT: the body with non-temporal data streams (tools/gen_tile_asm.py: for working sets far beyond the Infinity Cache -- the pipelined launches)
template <bool LAZY = false, bool NT = false>
RH_DEV void fwd_tile_asm_body(u64* lds, const u32 b, const u64* in, u64* out, const tw2* __restrict__ twk,
                              const LimbConsts* __restrict__ consts, int L, int logN, int npoly, int Ls = 0, u32 gap0 = 0, u32 gap_len = 0, int Lso = 0) {
  if (Ls == 0) Ls = L;                     // rows per poly of the block (>= L)
  if (Lso == 0) Lso = Ls;                  // ... of the output block, when it differs (AtLevel views with two strides)
  const u32 lrow = b % (u32)L;
  const u32 limb = lrow + (lrow >= gap0 ? gap_len : 0u);
  const u32 r = b / (u32)L;
  const u32 poly = r % (u32)npoly;
  const u32 tile = r / (u32)npoly;
  const size_t base = (((size_t)poly * Ls + limb) << logN) + ((size_t)tile << LT);
  const u64 pin = uni64((u64)(size_t)(in + base));
  const u64 pout = uni64((u64)(size_t)(out + (((size_t)poly * Lso + limb) << logN) + ((size_t)tile << LT)));
  const u64 tw = uni64((u64)(size_t)(twk + ((size_t)limb << logN) + ((size_t)tile << LT)));
  const u64 q = uni64(consts[limb].q);
  const u64 nq = (u64)0 - q, nq2 = (u64)0 - 2 * q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const u32 lds_off = uni32((u32)(size_t)lds);
  const u32 tid = threadIdx.x;
#define RH_TILE_FWD_ASM(BODY)                                                                                          \
  asm volatile(BODY : : [tid] "v"(tid), [lds] "s"(lds_off), [pin] "s"(pin), [pout] "s"(pout), [tw] "s"(tw),             \
               [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)), [nq0] "s"((u32)nq),                    \
               [nq1] "s"((u32)(nq >> 32)), [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4) : NTT_TILE_ASM_CLOBBERS)
  if constexpr (LAZY && NT) RH_TILE_FWD_ASM(NTT_TILE_LAZY_ASM_BODY_NT);
  else if constexpr (LAZY) RH_TILE_FWD_ASM(NTT_TILE_LAZY_ASM_BODY);
  else if constexpr (NT) RH_TILE_FWD_ASM(NTT_TILE_ASM_BODY_NT);
  else RH_TILE_FWD_ASM(NTT_TILE_ASM_BODY);
#undef RH_TILE_FWD_ASM
}

// column stages for N = 2^14 .. 2^16 (S1 = 2..4): hand-scheduled radix-2^S1 register round with wave-uniform twiddles;
// same contract as fwd_cols_body<ShoupPolicy, S1> (outputs < 8q, any representative: the tile stages reduce canonically)
constexpr bool has_asm_cols(int S1) { return S1 >= 2 && S1 <= 4; }
#define RH_COLS_FWD_ASM(BODY)                                                                                       \
  asm volatile(BODY : : [tid] "v"(tid), [pin] "s"(pin), [pout] "s"(pout), [tw] "s"(tw), [nq0] "s"((u32)nq),        \
               [nq1] "s"((u32)(nq >> 32)), [nq4] "s"(nq4), [q4] "s"(q4) : NTT_TILE_ASM_CLOBBERS)
template <int S1, bool NT = false>
RH_DEV void fwd_cols_asm_body(const u32 b, const u64* in, u64* out, const tw2* __restrict__ twn,
                              const LimbConsts* __restrict__ consts, int L, int Ls = 0, u32 gap0 = 0, u32 gap_len = 0, int Lso = 0) {
  if (Ls == 0) Ls = L;                     // rows per poly of the block (>= L)
  if (Lso == 0) Lso = Ls;                  // ... of the output block, when it differs
  static_assert(has_asm_cols(S1), "asm column stages exist for S1 = 2..4");
  constexpr int logN = LT + S1;
  const u32 lrow = b % (u32)L;
  const u32 limb = lrow + (lrow >= gap0 ? gap_len : 0u);         // see fwd_tile_asm_body
  const u32 r = b / (u32)L;
  const size_t base = (((size_t)(r >> 4) * Ls + limb) << logN) + (r & 15) * 256;
  const u64 pin = uni64((u64)(size_t)(in + base));
  const u64 pout = uni64((u64)(size_t)(out + (((size_t)(r >> 4) * Lso + limb) << logN) + (r & 15) * 256));
  const u64 tw = uni64((u64)(size_t)(twn + ((size_t)limb << logN)));
  const u64 q = uni64(consts[limb].q);
  const u64 nq = (u64)0 - q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const u32 tid = threadIdx.x;
  if constexpr (S1 == 4 && NT) RH_COLS_FWD_ASM(NTT_COLS16_ASM_BODY_NT);
  else if constexpr (S1 == 3 && NT) RH_COLS_FWD_ASM(NTT_COLS8_ASM_BODY_NT);
  else if constexpr (NT) RH_COLS_FWD_ASM(NTT_COLS4_ASM_BODY_NT);
  else if constexpr (S1 == 4) RH_COLS_FWD_ASM(NTT_COLS16_ASM_BODY);
  else if constexpr (S1 == 3) RH_COLS_FWD_ASM(NTT_COLS8_ASM_BODY);
  else RH_COLS_FWD_ASM(NTT_COLS4_ASM_BODY);
}
// Column stages fed by the re-expansion of a rescale step, hand-scheduled (same outputs contract as ntt_fwd_cols_expand,
// ntt_kernels.hip.hpp: values < 8q congruent to the expanded limb): x = cred(t + hq, qL) + s is NOT reduced modulo the limb's q --
// the launcher checks qL + q <= 8q for every limb, which the first stage's conditional subtraction needs.
template <int S1, bool NT = false>
__global__ void __launch_bounds__(256)
ntt_fwd_cols_expand_asm(const u64* tmp, u64* out, const tw2* __restrict__ twn, const LimbConsts* __restrict__ consts,
                        const RescaleLimb* __restrict__ T, int L, int mode, u64 qL) {
  static_assert(has_asm_cols(S1), "asm column stages exist for S1 = 2..4");
  constexpr int logN = LT + S1;
  const u32 b = blockIdx.x;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u64 pin = uni64((u64)(size_t)(tmp + ((size_t)(r >> 4) << logN) + (r & 15) * 256));
  const u64 pout = uni64((u64)(size_t)(out + (((size_t)(r >> 4) * L + limb) << logN) + (r & 15) * 256));
  const u64 tw = uni64((u64)(size_t)(twn + ((size_t)limb << logN)));
  const u64 q = uni64(consts[limb].q);
  const u64 nq = (u64)0 - q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const u64 hq = mode == 1 ? (qL - 1) >> 1 : 0, sadd = mode == 1 ? uni64(T[limb].s) : 0, nqL = (u64)0 - qL;
  const u32 tid = threadIdx.x;
#define RH_COLS_EXP_ASM(BODY)                                                                                       \
  asm volatile(BODY : : [tid] "v"(tid), [pin] "s"(pin), [pout] "s"(pout), [tw] "s"(tw), [nq0] "s"((u32)nq),        \
               [nq1] "s"((u32)(nq >> 32)), [nq4] "s"(nq4), [q4] "s"(q4), [hq] "s"(hq), [nqL] "s"(nqL), [sadd] "s"(sadd) : NTT_TILE_ASM_CLOBBERS)
  if constexpr (S1 == 4 && NT) RH_COLS_EXP_ASM(NTT_COLS16_EXPAND_ASM_BODY_NT);
  else if constexpr (S1 == 3 && NT) RH_COLS_EXP_ASM(NTT_COLS8_EXPAND_ASM_BODY_NT);
  else if constexpr (NT) RH_COLS_EXP_ASM(NTT_COLS4_EXPAND_ASM_BODY_NT);
  else if constexpr (S1 == 4) RH_COLS_EXP_ASM(NTT_COLS16_EXPAND_ASM_BODY);
  else if constexpr (S1 == 3) RH_COLS_EXP_ASM(NTT_COLS8_EXPAND_ASM_BODY);
  else RH_COLS_EXP_ASM(NTT_COLS4_EXPAND_ASM_BODY);
#undef RH_COLS_EXP_ASM
}
template <int S1, bool ASMCOLS, bool NT = false>
RH_DEV void fwd_cols_best(const u32 b, const u64* in, u64* out, const tw2* __restrict__ twn,
                          const LimbConsts* __restrict__ consts, int L, int logN) {
  if constexpr (has_asm_cols(S1) && ASMCOLS) fwd_cols_asm_body<S1, NT>(b, in, out, twn, consts, L);
  else fwd_cols_body<ShoupPolicy, S1>(b, in, out, twn, consts, L, logN);
}
template <int S1, bool NT = false>
__global__ void __launch_bounds__(256)
ntt_fwd_cols_asm(const u64* in, u64* out, const tw2* __restrict__ twn, const LimbConsts* __restrict__ consts, int L, int Ls, int Lso = 0) {
  fwd_cols_asm_body<S1, NT>(blockIdx.x, in, out, twn, consts, L, Ls, 0, 0, Lso);
}

template <bool NT = false>
__global__ void __launch_bounds__(256)
ntt_fwd_tile_asm(const u64* in, u64* out, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts,
                 int L, int logN, int npoly, int Ls, int Lso = 0) {
  __shared__ u64 lds[LDS_WORDS];
  fwd_tile_asm_body<false, NT>(lds, blockIdx.x, in, out, twk, consts, L, logN, npoly, Ls, 0, 0, Lso);
}

// Forward tile stages with the subtract-multiply epilogue, hand-scheduled: same contract as ntt_fwd_tile_submul (ntt_kernels.hip.hpp),
//   out = [z +] MRed(2q - y + NTT(in), s_limb),
// with MRed by the wave-uniform scalar done as a Shoup multiply by s*2^-64 mod q (sw / sp: that constant and its quotient, per limb).
struct LimbShoup { u64 w[RH_MAX_LIMBS_K], wp[RH_MAX_LIMBS_K]; };
template <bool ADD, bool NT = false>
__global__ void __launch_bounds__(256)
ntt_fwd_tile_submul_asm(const u64* in, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts, int L, int logN, int npoly,
                        const u64* y, int y_rows, u64* out, int out_rows, LimbShoup sc, const u64* z, int z_rows,
                        int npoly_a = 0, const u64* y_b = nullptr, u64* out_b = nullptr, const u64* z_b = nullptr) {
  // npoly_a > 0: TWO operand sets in one launch (both components of a ModDown): polys [0, npoly_a) of `in` go with (y, out, z), the rest with (y_b, out_b, z_b)
  __shared__ u64 lds[LDS_WORDS];
  const u32 b = blockIdx.x;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u32 poly_in = r % (u32)npoly;
  const u32 tile = r / (u32)npoly;
  const size_t toff = (size_t)tile << LT;
  const u64 pin = uni64((u64)(size_t)(in + (((size_t)poly_in * L + limb) << logN) + toff));
  u32 poly = poly_in;
  if (npoly_a > 0 && poly_in >= (u32)npoly_a) { poly = poly_in - (u32)npoly_a; y = y_b; out = out_b; z = z_b; }
  const u64 py = uni64((u64)(size_t)(y + (((size_t)poly * y_rows + limb) << logN) + toff));
  const u64 pout = uni64((u64)(size_t)(out + (((size_t)poly * out_rows + limb) << logN) + toff));
  const u64 pz = ADD ? uni64((u64)(size_t)(z + (((size_t)poly * z_rows + limb) << logN) + toff)) : 0;
  const u64 tw = uni64((u64)(size_t)(twk + ((size_t)limb << logN) + toff));
  const u64 q = uni64(consts[limb].q);
  const u64 sw = uni64(sc.w[limb]), sp = uni64(sc.wp[limb]);
  const u64 nq = (u64)0 - q, nq2 = (u64)0 - 2 * q, nq4 = (u64)0 - 4 * q, q4 = 4 * q, q2 = 2 * q;
  const u32 lds_off = uni32((u32)(size_t)lds);
  const u32 tid = threadIdx.x;
#define RH_SUBMUL_ADD_ASM(BODY)                                                                                                      \
  asm volatile(BODY : : [tid] "v"(tid), [lds] "s"(lds_off), [pin] "s"(pin), [pout] "s"(pout), [py] "s"(py), [pz] "s"(pz), [tw] "s"(tw),     \
               [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)), [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)),            \
               [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4), [q2] "s"(q2), [sw0] "s"((u32)sw), [sw1] "s"((u32)(sw >> 32)),    \
               [sp0] "s"((u32)sp), [sp1] "s"((u32)(sp >> 32)) : NTT_TILE_ASM_CLOBBERS)
#define RH_SUBMUL_ASM(BODY)                                                                                                          \
  asm volatile(BODY : : [tid] "v"(tid), [lds] "s"(lds_off), [pin] "s"(pin), [pout] "s"(pout), [py] "s"(py), [tw] "s"(tw),                   \
               [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)), [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)),            \
               [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4), [q2] "s"(q2), [sw0] "s"((u32)sw), [sw1] "s"((u32)(sw >> 32)),    \
               [sp0] "s"((u32)sp), [sp1] "s"((u32)(sp >> 32)) : NTT_TILE_ASM_CLOBBERS)
  if constexpr (ADD && NT) RH_SUBMUL_ADD_ASM(NTT_TILE_SUBMUL_ADD_ASM_BODY_NT);
  else if constexpr (ADD) RH_SUBMUL_ADD_ASM(NTT_TILE_SUBMUL_ADD_ASM_BODY);
  else if constexpr (NT) RH_SUBMUL_ASM(NTT_TILE_SUBMUL_ASM_BODY_NT);
  else RH_SUBMUL_ASM(NTT_TILE_SUBMUL_ASM_BODY);
#undef RH_SUBMUL_ADD_ASM
#undef RH_SUBMUL_ASM
}

// software-pipelined launch (see ntt_fwd_fused): column stages of span j, then the asm tile body of span j-1
template <int S1, bool ASMCOLS = false, bool NT = true>
__global__ void __launch_bounds__(256)
ntt_fwd_fused_asm(const u64* in1, u64* out1, unsigned n1, u64* data2, unsigned n2, int npoly2,
                  const tw2* __restrict__ twn, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts, int L, int logN) {
  __shared__ u64 lds[LDS_WORDS];
  if (blockIdx.x < n1) fwd_cols_best<S1, ASMCOLS, NT>(blockIdx.x, in1, out1, twn, consts, L, logN);
  if (blockIdx.x < n2) fwd_tile_asm_body<false, NT>(lds, blockIdx.x, data2, data2, twk, consts, L, logN, npoly2);
}

// The same pipeline over the digit blocks of a hybrid key-switch decomposition (rh_std_ntt_fwd_digits): launch j runs the
// column stages of digit j's non-digit limbs and the tile stages of digit j-1's; each digit skips its own limbs (GapRows).
struct GapRows { int L, Ls; u32 gap0, gap_len; };
template <int S1, bool LAZY, bool NT>
__global__ void __launch_bounds__(256)
ntt_fwd_fused_gap_asm(u64* data1, unsigned n1, GapRows g1, u64* data2, unsigned n2, int npoly2, GapRows g2,
                      const tw2* __restrict__ twn, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts) {
  __shared__ u64 lds[LDS_WORDS];
  if (blockIdx.x < n1) fwd_cols_asm_body<S1, NT>(blockIdx.x, data1, data1, twn, consts, g1.L, g1.Ls, g1.gap0, g1.gap_len);
  if (blockIdx.x < n2) fwd_tile_asm_body<LAZY, NT>(lds, blockIdx.x, data2, data2, twk, consts, g2.L, LT + S1, npoly2, g2.Ls, g2.gap0, g2.gap_len);
}

// Small batches (a key switch of a few ciphertexts: one digit block fills a fraction of the chip, and the pipelined stream above is a chain of
// nblocks + 1 dependent launches): ALL the blocks in one launch of the column stages and one of the tile stages, blockIdx.y = the block.
struct GapBlocks { int L[8]; u32 gap0[8], gap_len[8]; int Ls; };
// one ring's blocks of such a launch; a launch carries up to TWO sets (the Q blocks and the P blocks of a key switch: two rings, one launch pair)
struct BlockSet { u64* data; size_t stride; const tw2* tw; const LimbConsts* consts; GapBlocks g; int nblocks; };
template <int S1>
__global__ void __launch_bounds__(256)
ntt_fwd_cols_blocks_asm(BlockSet a, BlockSet b, int npoly) {
  const bool first = blockIdx.y < (unsigned)a.nblocks;
  const u32 y = first ? blockIdx.y : blockIdx.y - (unsigned)a.nblocks;
  const GapBlocks& g = first ? a.g : b.g;
  const int L = g.L[y];
  if (blockIdx.x >= (unsigned)npoly * (unsigned)L * 16u) return;
  u64* d = (first ? a.data : b.data) + (size_t)y * (first ? a.stride : b.stride);
  fwd_cols_asm_body<S1, false>(blockIdx.x, d, d, first ? a.tw : b.tw, first ? a.consts : b.consts, L, g.Ls, g.gap0[y], g.gap_len[y]);
}
template <int S1, bool LAZY>
__global__ void __launch_bounds__(256)
ntt_fwd_tile_blocks_asm(BlockSet a, BlockSet b, int npoly) {
  __shared__ u64 lds[LDS_WORDS];
  const bool first = blockIdx.y < (unsigned)a.nblocks;
  const u32 y = first ? blockIdx.y : blockIdx.y - (unsigned)a.nblocks;
  const GapBlocks& g = first ? a.g : b.g;
  const int L = g.L[y];
  if (blockIdx.x >= ((unsigned)npoly * (unsigned)L) << S1) return;
  u64* d = (first ? a.data : b.data) + (size_t)y * (first ? a.stride : b.stride);
  fwd_tile_asm_body<LAZY, false>(lds, blockIdx.x, d, d, first ? a.tw : b.tw, first ? a.consts : b.consts, L, LT + S1, npoly, g.Ls, g.gap0[y], g.gap_len[y]);
}

// ---- inverse: first 12 stages (t = 1..2048) on a 4096-tile, values leave < 4q (N^-1 is applied by ntt_inv_cols).
// Same contract as ntt_inv_tile(last = 0).  twk = kernel-order table built from RootsBackward.
// MUL: the tile's input is MRedLazy(in, in2) formed on load (rh_ring_intt_mul; same contract as inv_tile_body<true>)
template <bool MUL = false, bool NT = false>
RH_DEV void inv_tile_asm_body(u64* lds, const u32 b, const u64* in, const u64* in2, u64* out, const tw2* __restrict__ twk,
                              const LimbConsts* __restrict__ consts, int L, int logN, int npoly, int in_Ls = 0, int out_Ls = 0) {
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u32 poly = r % (u32)npoly;
  const u32 tile = r / (u32)npoly;
  const size_t base = (((size_t)poly * L + limb) << logN) + ((size_t)tile << LT);
  // in_Ls > 0: the input block has in_Ls rows per poly (a limb gathered out of a larger block); the output is dense
  const u64 pin = uni64((u64)(size_t)(in + (in_Ls ? (((size_t)poly * in_Ls + limb) << logN) + ((size_t)tile << LT) : base)));
  const u64 pout = uni64((u64)(size_t)(out + (out_Ls ? (((size_t)poly * out_Ls + limb) << logN) + ((size_t)tile << LT) : base)));   // out_Ls: as in_Ls
  const u64 tw = uni64((u64)(size_t)(twk + ((size_t)limb << logN) + ((size_t)tile << LT)));
  const u64 q = uni64(consts[limb].q);
  const u64 nq = (u64)0 - q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const u32 lds_off = uni32((u32)(size_t)lds);
  const u32 tid = threadIdx.x;
#define RH_TILE_INV_MUL_ASM(BODY)                                                                                                 \
  asm volatile(BODY : : [tid] "v"(tid), [lds] "s"(lds_off), [pin] "s"(pin), [pin2] "s"(pin2), [pout] "s"(pout), [tw] "s"(tw),        \
               [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)), [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)),     \
               [nq4] "s"(nq4), [q4] "s"(q4), [q] "s"(q), [q0] "s"((u32)q), [q1] "s"((u32)(q >> 32)), [qi0] "s"((u32)qinv),           \
               [qi1] "s"((u32)(qinv >> 32)) : NTT_TILE_ASM_CLOBBERS)
#define RH_TILE_INV_ASM(BODY)                                                                                                     \
  asm volatile(BODY : : [tid] "v"(tid), [lds] "s"(lds_off), [pin] "s"(pin), [pout] "s"(pout), [tw] "s"(tw),                          \
               [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)), [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)),     \
               [nq4] "s"(nq4), [q4] "s"(q4) : NTT_TILE_ASM_CLOBBERS)
  if constexpr (MUL) {
    const u64 pin2 = uni64((u64)(size_t)(in2 + base));
    const u64 qinv = uni64(consts[limb].qinv);
    if constexpr (NT) RH_TILE_INV_MUL_ASM(NTT_TILE_INV_MUL_ASM_BODY_NT);
    else RH_TILE_INV_MUL_ASM(NTT_TILE_INV_MUL_ASM_BODY);
  } else {
    if constexpr (NT) RH_TILE_INV_ASM(NTT_TILE_INV_ASM_BODY_NT);
    else RH_TILE_INV_ASM(NTT_TILE_INV_ASM_BODY);
  }
#undef RH_TILE_INV_MUL_ASM
#undef RH_TILE_INV_ASM
}
// inverse column stages with N^-1 folded in, S1 = 2..4: same contract as inv_cols_body<S1>(scale = 1)
#define RH_COLS_INV_ASM(BODY)                                                                                                     \
  asm volatile(BODY : : [tid] "v"(tid), [pin] "s"(pin), [tw] "s"(tw), [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)),            \
               [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4), [iw0] "s"((u32)iw), [iw1] "s"((u32)(iw >> 32)),       \
               [ip0] "s"((u32)ip), [ip1] "s"((u32)(ip >> 32)), [lw0] "s"((u32)lw), [lw1] "s"((u32)(lw >> 32)),                   \
               [lp0] "s"((u32)lp), [lp1] "s"((u32)(lp >> 32)) : NTT_TILE_ASM_CLOBBERS)
template <int S1, bool NT = false>
RH_DEV void inv_cols_asm_body(const u32 b, u64* data, const tw2* __restrict__ twn, const tw2* __restrict__ lastw,
                              const LimbConsts* __restrict__ consts, int L, int Ls = 0) {
  if (Ls == 0) Ls = L;                     // rows per poly of the block (>= L)
  static_assert(has_asm_cols(S1), "asm column stages exist for S1 = 2..4");
  constexpr int logN = LT + S1;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const size_t base = (((size_t)(r >> 4) * Ls + limb) << logN) + (r & 15) * 256;
  const u64 pin = uni64((u64)(size_t)(data + base));
  const u64 tw = uni64((u64)(size_t)(twn + ((size_t)limb << logN)));
  const u64 q = uni64(consts[limb].q);
  const u64 iw = uni64(consts[limb].ninv_w), ip = uni64(consts[limb].ninv_wp);
  const u64 lw = uni64(lastw[limb].w), lp = uni64(lastw[limb].wp);
  const u64 nq = (u64)0 - q, nq2 = (u64)0 - 2 * q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const u32 tid = threadIdx.x;
  if constexpr (S1 == 4 && NT) RH_COLS_INV_ASM(NTT_COLS16_INV_ASM_BODY_NT);
  else if constexpr (S1 == 3 && NT) RH_COLS_INV_ASM(NTT_COLS8_INV_ASM_BODY_NT);
  else if constexpr (NT) RH_COLS_INV_ASM(NTT_COLS4_INV_ASM_BODY_NT);
  else if constexpr (S1 == 4) RH_COLS_INV_ASM(NTT_COLS16_INV_ASM_BODY);
  else if constexpr (S1 == 3) RH_COLS_INV_ASM(NTT_COLS8_INV_ASM_BODY);
  else RH_COLS_INV_ASM(NTT_COLS4_INV_ASM_BODY);
}
template <int S1, bool NT = false>
__global__ void __launch_bounds__(256)
ntt_inv_cols_asm(u64* data, const tw2* __restrict__ twn, const tw2* __restrict__ lastw, const LimbConsts* __restrict__ consts, int L, int Ls) {
  inv_cols_asm_body<S1, NT>(blockIdx.x, data, twn, lastw, consts, L, Ls);
}

template <bool NT = false>
__global__ void __launch_bounds__(256)
ntt_inv_tile_asm(const u64* in, u64* out, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts,
                 int L, int logN, int npoly, int in_Ls, int out_Ls) {
  __shared__ u64 lds[LDS_WORDS];
  inv_tile_asm_body<false, NT>(lds, blockIdx.x, in, nullptr, out, twk, consts, L, logN, npoly, in_Ls, out_Ls);
}
__global__ void __launch_bounds__(256)
ntt_inv_tile_mul_asm(const u64* in, const u64* in2, u64* out, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts,
                     int L, int logN, int npoly) {
  __shared__ u64 lds[LDS_WORDS];
  inv_tile_asm_body<true>(lds, blockIdx.x, in, in2, out, twk, consts, L, logN, npoly);
}
// ---- c = INTT(NTT(a) . NTT(b)), the tile-stage middle in ONE kernel (BASELINE config 3; tools/gen_tile_asm.py: gen_polymul): forward tile
// stages of a's and of b's column-stage outputs, MRedLazy product, inverse tile stages -- NTT(a), NTT(b) and their product never reach memory
// (72 instead of 104 bytes per coefficient of a poly-mul).  The forward body leaves thread tid with coefficients 16 tid + k, where the inverse
// body's first round starts, so two LDS transposes drop out too.  156 VGPRs: 3 workgroups per CU.  out may alias a or b (a tile is read whole
// before it is written).  consts: qinv for the product; the factor 2^64 is restored by the inverse column stages' constants (d_consts_r).
template <bool NT = false>
RH_DEV void polymul_tile_asm_body(u64* lds, const u32 b, const u64* a, const u64* b2, u64* out, const tw2* __restrict__ twk_fwd,
                                  const tw2* __restrict__ twk_inv, const LimbConsts* __restrict__ consts, int L, int logN, int npoly) {
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u32 poly = r % (u32)npoly;
  const u32 tile = r / (u32)npoly;
  const size_t base = (((size_t)poly * L + limb) << logN) + ((size_t)tile << LT);
  const size_t toff = ((size_t)limb << logN) + ((size_t)tile << LT);
  const u64 pin = uni64((u64)(size_t)(a + base)), pin2 = uni64((u64)(size_t)(b2 + base)), pout = uni64((u64)(size_t)(out + base));
  const u64 tw = uni64((u64)(size_t)(twk_fwd + toff)), twi = uni64((u64)(size_t)(twk_inv + toff));
  const u64 q = uni64(consts[limb].q), qinv = uni64(consts[limb].qinv);
  const u64 nq = (u64)0 - q, nq2 = (u64)0 - 2 * q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const u32 lds_off = uni32((u32)(size_t)lds);
  const u32 tid = threadIdx.x;
#define RH_TILE_POLYMUL_ASM(BODY)                                                                                                          \
  asm volatile(BODY : : [tid] "v"(tid), [lds] "s"(lds_off), [pin] "s"(pin), [pin2] "s"(pin2), [pout] "s"(pout), [tw] "s"(tw),               \
               [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)), [twi] "s"(twi), [twilo] "s"((u32)(size_t)twi),            \
               [twihi] "s"((u32)((size_t)twi >> 32)), [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)), [nq] "s"(nq), [nq2] "s"(nq2),       \
               [nq4] "s"(nq4), [q4] "s"(q4), [q] "s"(q), [q0] "s"((u32)q), [q1] "s"((u32)(q >> 32)), [qi0] "s"((u32)qinv),                 \
               [qi1] "s"((u32)(qinv >> 32)) : NTT_TILE_POLYMUL_ASM_CLOBBERS)
  if constexpr (NT) RH_TILE_POLYMUL_ASM(NTT_TILE_POLYMUL_ASM_BODY_NT);
  else RH_TILE_POLYMUL_ASM(NTT_TILE_POLYMUL_ASM_BODY);
#undef RH_TILE_POLYMUL_ASM
}
template <bool NT = false>
__global__ void __launch_bounds__(256)
ntt_polymul_tile_asm(const u64* a, const u64* b2, u64* out, const tw2* __restrict__ twk_fwd, const tw2* __restrict__ twk_inv,
                     const LimbConsts* __restrict__ consts, int L, int logN, int npoly) {
  __shared__ u64 lds[LDS_WORDS];
  polymul_tile_asm_body<NT>(lds, blockIdx.x, a, b2, out, twk_fwd, twk_inv, consts, L, logN, npoly);
}
// The software pipeline of a large poly-mul batch (rh_ring_polymul, spans of ~2048 rows): launch j runs the forward column stages of span j
// (both operands, in place: the memory-bound work), the one-kernel tile middle of span j-1 (the VALU-bound work) and the inverse column stages of
// span j-2, so the column passes run under the tile passes like in ntt_fwd_fused_asm.  n1: column blocks per operand of span j, n2: tiles of span
// j-1, n3: column blocks of span j-2 (0: that part is absent in this launch).
template <int S1, bool NT>
__global__ void __launch_bounds__(256)
ntt_polymul_fused_asm(u64* a1, u64* b1, unsigned n1, const u64* a2, const u64* b2, u64* out2, unsigned n2, int npoly2, u64* out3, unsigned n3,
                      const tw2* __restrict__ twn_f, const tw2* __restrict__ twk_f, const tw2* __restrict__ twk_i, const tw2* __restrict__ twn_i,
                      const tw2* __restrict__ lastw_r, const LimbConsts* __restrict__ consts_r, int L, int logN) {
  __shared__ u64 lds[LDS_WORDS];
  const u32 b = blockIdx.x;
  if (b < n1) {
    fwd_cols_best<S1, true, NT>(b, a1, a1, twn_f, consts_r, L, logN);               // (the column stages read q only: either constant block does)
    fwd_cols_best<S1, true, NT>(b, b1, b1, twn_f, consts_r, L, logN);
  }
  if (b < n2) polymul_tile_asm_body<NT>(lds, b, a2, b2, out2, twk_f, twk_i, consts_r, L, logN, npoly2);
  if (b < n3) {
    if constexpr (has_asm_cols(S1)) inv_cols_asm_body<S1, NT>(b, out3, twn_i, lastw_r, consts_r, L);
    else inv_cols_body<S1>(b, out3, twn_i, lastw_r, consts_r, L, logN, 1);
  }
}

// software-pipelined inverse: tile stages of span j (in -> out), then column stages + N^-1 of span j-1 (in place)
template <int S1, bool ASMCOLS = false, bool MUL = false, bool NT = true>
__global__ void __launch_bounds__(256)
ntt_inv_fused_asm(const u64* in1, const u64* in1b, u64* out1, unsigned n1, int npoly1, u64* data2, unsigned n2,
                  const tw2* __restrict__ twk, const tw2* __restrict__ twn, const tw2* __restrict__ lastw,
                  const LimbConsts* __restrict__ consts, int L, int logN) {
  __shared__ u64 lds[LDS_WORDS];
  if (blockIdx.x < n2) {
    if constexpr (has_asm_cols(S1) && ASMCOLS) inv_cols_asm_body<S1, NT>(blockIdx.x, data2, twn, lastw, consts, L);
    else inv_cols_body<S1>(blockIdx.x, data2, twn, lastw, consts, L, logN, 1);
  }
  if (blockIdx.x < n1) inv_tile_asm_body<MUL, NT>(lds, blockIdx.x, in1, in1b, out1, twk, consts, L, logN, npoly1);
}

// ---- conjugate-invariant ring (ring/ntt.go:716-1311): the fold fused with the column stages (tools/gen_tile_asm.py: gen_cols_ci).
// Unit b = (limb, group g < 8, poly): thread t owns columns c = 256 g + t + 1 and 4096 - c, whose elements the fold couples.
// Column 0 (its own mirror image, with the two special coefficients 0 and N/2) is ci_col0_kernel.
struct CiFoldTw { tw2 f, b; };            // = CiFold (engine_internal.hpp): fold twiddles roots_fwd[1] / roots_bwd[1]
template <int S1, bool INV>
RH_DEV void cols_ci_asm_body(const u32 b, const u64* in, u64* out, const tw2* __restrict__ twn, const tw2* __restrict__ lastw,
                             const CiFoldTw* __restrict__ fold, const LimbConsts* __restrict__ consts, int L) {
  static_assert(has_asm_cols(S1), "asm column stages exist for S1 = 2..4");
  constexpr int logN = LT + S1;
  const u32 limb = b % (u32)L, rr = b / (u32)L, g = rr & 7;
  const size_t row = ((size_t)(rr >> 3) * L + limb) << logN;
  const size_t offa = row + 256 * g + 1, offb = row + 256 * (15 - g);
  const u64 pina = uni64((u64)(size_t)(in + offa)), pinb = uni64((u64)(size_t)(in + offb));
  const u64 pouta = uni64((u64)(size_t)(out + offa)), poutb = uni64((u64)(size_t)(out + offb));
  const u64 tw = uni64((u64)(size_t)(twn + ((size_t)limb << logN)));
  const u64 q = uni64(consts[limb].q);
  const u64 nq = (u64)0 - q, nq2 = (u64)0 - 2 * q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const tw2 F = INV ? fold[limb].b : fold[limb].f;
  const u64 fw = uni64(F.w), fp = uni64(F.wp);
  const u32 tid = threadIdx.x;
  if constexpr (INV) {
    const u64 iw = uni64(consts[limb].ninv_w), ip = uni64(consts[limb].ninv_wp);
    const u64 lw = uni64(lastw[limb].w), lp = uni64(lastw[limb].wp);
#define RH_CI_INV_ASM(BODY)                                                                                                      \
    asm volatile(BODY : : [tid] "v"(tid), [pina] "s"(pina), [pinb] "s"(pinb), [pouta] "s"(pouta), [poutb] "s"(poutb), [tw] "s"(tw), \
                 [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)), [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4),     \
                 [fw0] "s"((u32)fw), [fw1] "s"((u32)(fw >> 32)), [fp0] "s"((u32)fp), [fp1] "s"((u32)(fp >> 32)),                \
                 [iw0] "s"((u32)iw), [iw1] "s"((u32)(iw >> 32)), [ip0] "s"((u32)ip), [ip1] "s"((u32)(ip >> 32)),                \
                 [lw0] "s"((u32)lw), [lw1] "s"((u32)(lw >> 32)), [lp0] "s"((u32)lp), [lp1] "s"((u32)(lp >> 32)) : NTT_TILE_ASM_CLOBBERS)
    if constexpr (S1 == 4) RH_CI_INV_ASM(NTT_CI_COLS16_INV_ASM_BODY);
    else if constexpr (S1 == 3) RH_CI_INV_ASM(NTT_CI_COLS8_INV_ASM_BODY);
    else RH_CI_INV_ASM(NTT_CI_COLS4_INV_ASM_BODY);
#undef RH_CI_INV_ASM
  } else {
#define RH_CI_FWD_ASM(BODY)                                                                                                      \
    asm volatile(BODY : : [tid] "v"(tid), [pina] "s"(pina), [pinb] "s"(pinb), [pouta] "s"(pouta), [poutb] "s"(poutb), [tw] "s"(tw), \
                 [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)), [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4),     \
                 [fw0] "s"((u32)fw), [fw1] "s"((u32)(fw >> 32)), [fp0] "s"((u32)fp), [fp1] "s"((u32)(fp >> 32)) : NTT_TILE_ASM_CLOBBERS)
    if constexpr (S1 == 4) RH_CI_FWD_ASM(NTT_CI_COLS16_FWD_ASM_BODY);
    else if constexpr (S1 == 3) RH_CI_FWD_ASM(NTT_CI_COLS8_FWD_ASM_BODY);
    else RH_CI_FWD_ASM(NTT_CI_COLS4_FWD_ASM_BODY);
#undef RH_CI_FWD_ASM
  }
}
template <int S1, bool INV>
__global__ void __launch_bounds__(256)
ntt_cols_ci_asm(const u64* in, u64* out, const tw2* __restrict__ twn, const tw2* __restrict__ lastw, const CiFoldTw* __restrict__ fold,
                const LimbConsts* __restrict__ consts, int L) {
  cols_ci_asm_body<S1, INV>(blockIdx.x, in, out, twn, lastw, fold, consts, L);
}
// The software pipeline of large conjugate-invariant batches (round 3), as ntt_fwd_fused_asm / ntt_inv_fused_asm for the standard ring:
// forward launch j = fold + column stages of span j (n1 = rows * 8 units) + tile stages of span j-1; inverse launch j = tile stages of span j
// + column stages + fold of span j-1.  Column 0 of a span is ci_col0_kernel, launched beside it (a one-thread-per-row kernel).
template <int S1, bool NT>
__global__ void __launch_bounds__(256)
ntt_ci_fwd_fused_asm(const u64* in1, u64* out1, unsigned n1, u64* data2, unsigned n2, int npoly2, const tw2* __restrict__ twn,
                     const tw2* __restrict__ twk, const CiFoldTw* __restrict__ fold, const LimbConsts* __restrict__ consts, int L) {
  __shared__ u64 lds[LDS_WORDS];
  if (blockIdx.x < n1) cols_ci_asm_body<S1, false>(blockIdx.x, in1, out1, twn, nullptr, fold, consts, L);
  if (blockIdx.x < n2) fwd_tile_asm_body<false, NT>(lds, blockIdx.x, data2, data2, twk, consts, L, LT + S1, npoly2);
}
template <int S1, bool NT>
__global__ void __launch_bounds__(256)
ntt_ci_inv_fused_asm(const u64* in1, u64* out1, unsigned n1, int npoly1, u64* data2, unsigned n2, const tw2* __restrict__ twk,
                     const tw2* __restrict__ twn, const tw2* __restrict__ lastw, const CiFoldTw* __restrict__ fold,
                     const LimbConsts* __restrict__ consts, int L) {
  __shared__ u64 lds[LDS_WORDS];
  if (blockIdx.x < n2) cols_ci_asm_body<S1, true>(blockIdx.x, data2, data2, twn, lastw, fold, consts, L);
  if (blockIdx.x < n1) inv_tile_asm_body<false, NT>(lds, blockIdx.x, in1, nullptr, out1, twk, consts, L, LT + S1, npoly1);
}
// column 0 of every (poly, limb) row: coefficients 4096 k, k < R.  Coefficient 0 is not folded (forward) / doubled (inverse), N/2 folds
// with itself, k folds with R - k; column stages as fwd_cols_body<ShoupPolicy> / inv_cols_body(scale = 1).  One thread per row.
template <int S1, bool INV>
__global__ void __launch_bounds__(64)
ci_col0_kernel(const u64* in, u64* out, const tw2* __restrict__ twn, const tw2* __restrict__ lastw, const CiFoldTw* __restrict__ fold,
               const LimbConsts* __restrict__ consts, int L, unsigned rows) {
  constexpr int R = 1 << S1, logN = LT + S1;
  const unsigned row = blockIdx.x * 64 + threadIdx.x;
  if (row >= rows) return;
  const u32 limb = row % (u32)L;
  const LimbConsts c = consts[limb];
  const tw2* tw = twn + ((size_t)limb << logN);
  const tw2 F = INV ? fold[limb].b : fold[limb].f;
  ShoupPolicy p; p.init(c);
  const u64 q4 = 4 * c.q;
  const size_t base = (size_t)row << logN;
  u64 x[R];
#pragma unroll
  for (int k = 0; k < R; ++k) x[k] = in[base + ((size_t)k << LT)];
  auto fold_all = [&](bool canonical) {
    for (int k = 1; k < R / 2; ++k) {
      const u64 a = csub(x[k], q4), b = csub(x[R - k], q4);
      const u64 va = a + q4 - shoup_mul(b, F.w, F.wp, c.nq), vb = b + q4 - shoup_mul(a, F.w, F.wp, c.nq);
      x[k] = canonical ? canon8(va, c.q) : va; x[R - k] = canonical ? canon8(vb, c.q) : vb;
    }
    const u64 m = csub(x[R / 2], q4);
    const u64 vm = m + q4 - shoup_mul(m, F.w, F.wp, c.nq);
    x[R / 2] = canonical ? canon8(vm, c.q) : vm;
  };
  if constexpr (!INV) {
    fold_all(false);                                            // x[0] unchanged (ring/ntt.go:770)
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      const int h = R >> (s + 1);
#pragma unroll
      for (int g = 0; g < (1 << s); ++g) {
        const tw2 w = tw[(1 << s) + g];
#pragma unroll
        for (int e = 0; e < h; ++e) p.fwd(x[g * 2 * h + e], x[g * 2 * h + e + h], w, false);
      }
    }
  } else {
#pragma unroll
    for (int s = S1 - 1; s >= 1; --s) {
      const int hh = R >> (s + 1);
#pragma unroll
      for (int g = 0; g < (1 << s); ++g) {
        const tw2 w = tw[(1 << s) + g];
#pragma unroll
        for (int e = 0; e < hh; ++e) p.inv(x[g * 2 * hh + e], x[g * 2 * hh + e + hh], w);
      }
    }
    const tw2 wl = lastw[limb];
#pragma unroll
    for (int e = 0; e < R / 2; ++e) {
      const u64 U = x[e], V = x[e + R / 2], d = U + q4 - V;
      x[e] = canon4(shoup_mul(U + V, c.ninv_w, c.ninv_wp, c.nq), c.q);
      x[e + R / 2] = canon4(shoup_mul(d, wl.w, wl.wp, c.nq), c.q);
    }
    x[0] = cred(2 * x[0], c.q);                                 // p2[0] = CRed(p2[0] << 1) (:1157)
    fold_all(true);
  }
#pragma unroll
  for (int k = 0; k < R; ++k) out[base + ((size_t)k << LT)] = x[k];
}

// ---------------------------------------------------------------------------------------------------------------
// One-pass transforms of N = 2^13 and 2^14 (S1 = 1, 2): a whole limb row (64 / 128 KiB) lives in ONE workgroup's LDS, so the hand-over between
// the column stages and the tile stages never leaves the CU: 16 bytes of HBM traffic per coefficient instead of 32.  The workgroup has 256 << S1
// threads: all of them run the S1 column stages (compiled C++, 16 >> S1 column positions each) and leave the 2^S1 tiles in LDS in the tile body's
// layout A (word t + (t >> 4) + 272 k of a tile holds its coefficient t + 256 k); then each group of 256 threads runs the hand-scheduled tile body
// on its own tile, reading its first round from LDS (tools/gen_tile_asm.py: gen(lds_in=True)) -- every thread overwrites only the slots it read,
// so one __syncthreads separates the two phases.  The body's s_barriers span the whole workgroup; every group executes the same number.
// Inverse: mirror image (gen_inverse(lds_out=True)); S1 = 0 (N = 4096) is the hand-scheduled inverse body followed by the N^-1 scaling out of LDS.  Occupancy as for the tile kernels: 4 waves per SIMD (<= 128 VGPRs), 136 KiB of LDS per CU.
// Contract: ring/ntt.go:209-552 + reducevec / :554-714 + NInv, outputs canonical -- identical to the two-pass launches (tests: every N = 2^13 / 2^14 case).
// ---------------------------------------------------------------------------------------------------------------
RH_DEV u32 lds_word_a(u32 j) { const u32 t = j & 255u; return t + (t >> 4) + 272u * (j >> 8); }
template <int S1, bool NT>
__global__ void __launch_bounds__(256 << S1)
ntt_fwd_onepass_asm(const u64* in, u64* out, const tw2* __restrict__ twn, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts,
                    int L, int Ls, int Lso) {
  extern __shared__ u64 op_lds[];                     // (1 << S1) * LDS_WORDS
  constexpr int R = 1 << S1, logN = LT + S1, T = 256 << S1, PER = 16 >> S1;
  if (Ls == 0) Ls = L;
  if (Lso == 0) Lso = Ls;
  {
    const u32 b = blockIdx.x;
    const u32 limb = b % (u32)L, poly = b / (u32)L;
    const size_t base = ((size_t)poly * Ls + limb) << logN;
    {
      const tw2* tw = twn + ((size_t)limb << logN);
      ShoupPolicy p; p.init(consts[limb]);
      u64 x[PER][R];
#pragma unroll
      for (int i = 0; i < PER; ++i)
#pragma unroll
        for (int c = 0; c < R; ++c) {
          const u64* src = in + base + ((size_t)c << LT) + i * T + threadIdx.x;
          x[i][c] = NT ? __builtin_nontemporal_load(src) : *src;
        }
#pragma unroll
      for (int s = 0; s < S1; ++s) {
        const int h = R >> (s + 1);
#pragma unroll
        for (int g = 0; g < (1 << s); ++g) {
          const tw2 w = tw[(1 << s) + g];
#pragma unroll
          for (int e = 0; e < h; ++e)
#pragma unroll
            for (int i = 0; i < PER; ++i) p.fwd(x[i][g * 2 * h + e], x[i][g * 2 * h + e + h], w, false);
        }
      }
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const u32 wa = lds_word_a((u32)(i * T) + threadIdx.x);
#pragma unroll
        for (int c = 0; c < R; ++c) op_lds[c * LDS_WORDS + wa] = x[i][c];
      }
    }
    __syncthreads();
    const u32 g = uni32(threadIdx.x >> 8);              // this wave's tile
    const u32 tid = threadIdx.x & 255u;
    const u64 pout = uni64((u64)(size_t)(out + (((size_t)poly * Lso + limb) << logN) + ((size_t)g << LT)));
    const u64 pin = pout;                               // (unused by the body)
    const u64 tw = uni64((u64)(size_t)(twk + ((size_t)limb << logN) + ((size_t)g << LT)));
    const u64 q = uni64(consts[limb].q);
    const u64 nq = (u64)0 - q, nq2 = (u64)0 - 2 * q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
    const u32 lds_off = uni32((u32)(size_t)(op_lds + g * LDS_WORDS));
#define RH_TILE_LDSIN_ASM(BODY)                                                                                         \
  asm volatile(BODY : : [tid] "v"(tid), [lds] "s"(lds_off), [pin] "s"(pin), [pout] "s"(pout), [tw] "s"(tw),             \
               [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)), [nq0] "s"((u32)nq),                    \
               [nq1] "s"((u32)(nq >> 32)), [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4) : NTT_TILE_ASM_CLOBBERS)
    if constexpr (NT) RH_TILE_LDSIN_ASM(NTT_TILE_LDSIN_ASM_BODY_NT);
    else RH_TILE_LDSIN_ASM(NTT_TILE_LDSIN_ASM_BODY);
#undef RH_TILE_LDSIN_ASM
  }
}

// scale: 1 = the last stage folds N^-1 in and reduces canonically (inv_cols_body); 0 = plain last stage, values stay < 4q
template <int S1, bool NT>
__global__ void __launch_bounds__(256 << S1)
ntt_inv_onepass_asm(const u64* in, u64* out, const tw2* __restrict__ twk, const tw2* __restrict__ twn, const tw2* __restrict__ lastw,
                    const LimbConsts* __restrict__ consts, int L, int Ls, int Lso, int scale) {
  extern __shared__ u64 op_lds[];
  constexpr int R = 1 << S1, logN = LT + S1, T = 256 << S1, PER = 16 >> S1;
  if (Ls == 0) Ls = L;
  if (Lso == 0) Lso = Ls;
  {
    const u32 b = blockIdx.x;
    const u32 limb = b % (u32)L, poly = b / (u32)L;
    {
      const u32 g = uni32(threadIdx.x >> 8);
      const u32 tid = threadIdx.x & 255u;
      const u64 pin = uni64((u64)(size_t)(in + (((size_t)poly * Ls + limb) << logN) + ((size_t)g << LT)));
      const u64 pout = pin;                             // (unused by the body)
      const u64 tw = uni64((u64)(size_t)(twk + ((size_t)limb << logN) + ((size_t)g << LT)));
      const u64 q = uni64(consts[limb].q);
      const u64 nq = (u64)0 - q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
      const u32 lds_off = uni32((u32)(size_t)(op_lds + g * LDS_WORDS));
#define RH_TILE_INV_LDSOUT_ASM(BODY)                                                                                              \
  asm volatile(BODY : : [tid] "v"(tid), [lds] "s"(lds_off), [pin] "s"(pin), [pout] "s"(pout), [tw] "s"(tw),                          \
               [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)), [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)),     \
               [nq4] "s"(nq4), [q4] "s"(q4) : NTT_TILE_ASM_CLOBBERS)
      if constexpr (NT) RH_TILE_INV_LDSOUT_ASM(NTT_TILE_INV_LDSOUT_ASM_BODY_NT);
      else RH_TILE_INV_LDSOUT_ASM(NTT_TILE_INV_LDSOUT_ASM_BODY);
#undef RH_TILE_INV_LDSOUT_ASM
    }
    __syncthreads();
    const size_t obase = ((size_t)poly * Lso + limb) << logN;
    const tw2* tw = twn + ((size_t)limb << logN);
    const LimbConsts c = consts[limb];
    ShoupPolicy p; p.init(c);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const u32 j = (u32)(i * T) + threadIdx.x;
      const u32 wa = lds_word_a(j);
      u64 x[R];
#pragma unroll
      for (int k = 0; k < R; ++k) x[k] = op_lds[k * LDS_WORDS + wa];
#pragma unroll
      for (int s = S1 - 1; s >= 1; --s) {
        const int hh = R >> (s + 1);
#pragma unroll
        for (int g = 0; g < (1 << s); ++g) {
          const tw2 w = tw[(1 << s) + g];
#pragma unroll
          for (int e = 0; e < hh; ++e) p.inv(x[g * 2 * hh + e], x[g * 2 * hh + e + hh], w);
        }
      }
      constexpr int hh = R >> 1;
      if constexpr (S1 == 0) {        // N = 4096: the body ran ALL twelve stages; what is left is N^-1 and the canonical reduction (scale = 1 only)
        x[0] = canon4(shoup_mul(x[0], c.ninv_w, c.ninv_wp, c.nq), c.q);
      } else if (scale) {
        const tw2 wl = lastw[limb];
#pragma unroll
        for (int e = 0; e < hh; ++e) {
          const u64 U = x[e], V = x[e + hh];
          const u64 d = U + p.q4 - V;
          x[e] = canon4(shoup_mul(U + V, c.ninv_w, c.ninv_wp, c.nq), c.q);
          x[e + hh] = canon4(shoup_mul(d, wl.w, wl.wp, c.nq), c.q);
        }
      } else {
        const tw2 w1 = tw[1];
#pragma unroll
        for (int e = 0; e < hh; ++e) p.inv(x[e], x[e + hh], w1);
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        u64* dst = out + obase + ((size_t)k << LT) + j;
        if (NT) __builtin_nontemporal_store(x[k], dst); else *dst = x[k];
      }
    }
  }
}
