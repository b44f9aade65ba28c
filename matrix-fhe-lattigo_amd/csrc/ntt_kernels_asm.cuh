// ntt_kernels_asm.cuh -- forward 4096-tile NTT kernel with a hand-scheduled gfx950 body (tools/gen_tile_asm.py).
//
// Same contract as ntt_fwd_tile<ShoupPolicy> with canonical output (ntt_kernels.cuh): last 12 stages of the
// forward negacyclic NTT (ring/ntt.go:209-552 + reducevec) on one contiguous 4096-coefficient tile.  The C++ wrapper
// resolves (poly, limb, tile), loads the per-limb constants through the scalar cache and hands everything to one asm
// statement that owns v0..v123 and s36..s101.
#pragma once
#include "ring_types.cuh"
#include "ntt_kernels.cuh"
#include "ntt_tile_asm.inc"

// wave-uniform values the compiler cannot prove uniform (loop-carried item index of the persistent kernel) -> SGPRs
RH_DEV u32 uni32(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }
RH_DEV u64 uni64(u64 x) { return ((u64)uni32((u32)(x >> 32)) << 32) | uni32((u32)x); }

RH_DEV void fwd_tile_asm_body(u64* lds, const u32 b, const u64* in, u64* out, const tw2* __restrict__ twk,
                              const LimbConsts* __restrict__ consts, int L, int logN, int npoly) {
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u32 poly = r % (u32)npoly;
  const u32 tile = r / (u32)npoly;
  const size_t base = (((size_t)poly * L + limb) << logN) + ((size_t)tile << LT);
  const u64 pin = uni64((u64)(size_t)(in + base));
  const u64 pout = uni64((u64)(size_t)(out + base));
  const u64 tw = uni64((u64)(size_t)(twk + ((size_t)limb << logN) + ((size_t)tile << LT)));
  const u64 q = uni64(consts[limb].q);
  const u64 nq = (u64)0 - q, nq2 = (u64)0 - 2 * q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const u32 lds_off = uni32((u32)(size_t)lds);
  const u32 tid = threadIdx.x;
  asm volatile(NTT_TILE_ASM_BODY
               :
               : [tid] "v"(tid), [lds] "s"(lds_off), [pin] "s"(pin), [pout] "s"(pout), [tw] "s"(tw),
                 [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)),
                 [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)), [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4)
               : NTT_TILE_ASM_CLOBBERS);
}

__global__ void __launch_bounds__(256)
ntt_fwd_tile_asm(const u64* in, u64* out, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts,
                 int L, int logN, int npoly) {
  __shared__ u64 lds[LDS_WORDS];
  fwd_tile_asm_body(lds, blockIdx.x, in, out, twk, consts, L, logN, npoly);
}

// software-pipelined launch (see ntt_fwd_fused): column stages of span j (C++ body), then the asm tile body of span j-1
template <int S1>
__global__ void __launch_bounds__(256)
ntt_fwd_fused_asm(const u64* in1, u64* out1, unsigned n1, u64* data2, unsigned n2, int npoly2,
                  const tw2* __restrict__ twn, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts, int L, int logN,
                  int order_mix) {
  __shared__ u64 lds[LDS_WORDS];
  // the two items of a launch are independent: alternate their order between the workgroup slots of a CU
  // (blocks b, b+256, ... share a CU slot under round-robin placement; speed only) so that memory-phase and
  // compute-phase workgroups are co-resident from the first instant of the launch
  if (order_mix && ((blockIdx.x >> 8) & 1u)) {
    if (blockIdx.x < n2) fwd_tile_asm_body(lds, blockIdx.x, data2, data2, twk, consts, L, logN, npoly2);
    if (blockIdx.x < n1) fwd_cols_body<ShoupPolicy, S1>(blockIdx.x, in1, out1, twn, consts, L, logN);
  } else {
    if (blockIdx.x < n1) fwd_cols_body<ShoupPolicy, S1>(blockIdx.x, in1, out1, twn, consts, L, logN);
    if (blockIdx.x < n2) fwd_tile_asm_body(lds, blockIdx.x, data2, data2, twk, consts, L, logN, npoly2);
  }
}

// ---- inverse: first 12 stages (t = 1..2048) on a 4096-tile, values leave < 4q (N^-1 is applied by ntt_inv_cols).
// Same contract as ntt_inv_tile(last = 0).  twk = kernel-order table built from RootsBackward.
RH_DEV void inv_tile_asm_body(u64* lds, const u32 b, const u64* in, u64* out, const tw2* __restrict__ twk,
                              const LimbConsts* __restrict__ consts, int L, int logN, int npoly) {
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u32 poly = r % (u32)npoly;
  const u32 tile = r / (u32)npoly;
  const size_t base = (((size_t)poly * L + limb) << logN) + ((size_t)tile << LT);
  const u64 pin = uni64((u64)(size_t)(in + base));
  const u64 pout = uni64((u64)(size_t)(out + base));
  const u64 tw = uni64((u64)(size_t)(twk + ((size_t)limb << logN) + ((size_t)tile << LT)));
  const u64 q = uni64(consts[limb].q);
  const u64 nq = (u64)0 - q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const u32 lds_off = uni32((u32)(size_t)lds);
  const u32 tid = threadIdx.x;
  asm volatile(NTT_TILE_INV_ASM_BODY
               :
               : [tid] "v"(tid), [lds] "s"(lds_off), [pin] "s"(pin), [pout] "s"(pout), [tw] "s"(tw),
                 [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)),
                 [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)), [nq4] "s"(nq4), [q4] "s"(q4)
               : NTT_TILE_ASM_CLOBBERS);
}
__global__ void __launch_bounds__(256)
ntt_inv_tile_asm(const u64* in, u64* out, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts,
                 int L, int logN, int npoly) {
  __shared__ u64 lds[LDS_WORDS];
  inv_tile_asm_body(lds, blockIdx.x, in, out, twk, consts, L, logN, npoly);
}
// software-pipelined inverse: tile stages of span j (in -> out), then column stages + N^-1 of span j-1 (in place)
template <int S1>
__global__ void __launch_bounds__(256)
ntt_inv_fused_asm(const u64* in1, u64* out1, unsigned n1, int npoly1, u64* data2, unsigned n2,
                  const tw2* __restrict__ twk, const tw2* __restrict__ twn, const tw2* __restrict__ lastw,
                  const LimbConsts* __restrict__ consts, int L, int logN) {
  __shared__ u64 lds[LDS_WORDS];
  if (blockIdx.x < n2) inv_cols_body<S1>(blockIdx.x, data2, twn, lastw, consts, L, logN, 1);
  if (blockIdx.x < n1) inv_tile_asm_body(lds, blockIdx.x, in1, out1, twk, consts, L, logN, npoly1);
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent single-launch forward transform (N >= 8192): gridDim.x resident workgroups walk a static schedule that
// alternates a column unit of poly-group g with a tile of poly-group g-1.  A tile of row (poly, limb) may start once
// all 16 column units of that row have published (rowcnt[row] == 16).  Groups are small enough (a few polys) that what
// the column stages wrote is still in the 256 MiB Infinity Cache when the tile stages read it.
// EXPERIMENTAL, off by default ("persistent" tuning key): measured 7.9 ms per 1024 polys against 7.2 ms for the
// launch-granular pipeline (ntt_fwd_fused_asm) -- the fabric between L2 and the memory side, not HBM itself, is the
// limit, so Infinity-Cache hits buy little (tools/mall_probe.py), and the static schedule phase-locks the CUs.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms"):
//   producer: sc1 (write-through) stores -> every wave s_waitcnt vmcnt(0) -> __syncthreads -> lane 0 agent-scope add
//   consumer: lane 0 relaxed agent poll (bounded) -> agent acquire fence -> s_waitcnt vmcnt(0) -> __syncthreads -> loads
// No wait can deadlock: a workgroup publishes its group-g column unit BEFORE it waits on any group-g row, column units
// never wait, and the host launches at most the resident number of workgroups; the poll is bounded anyway and raises
// *err (the host then redoes the batch with the two-launch path).
// ---------------------------------------------------------------------------------------------------------------
template <int S1>
__global__ void __launch_bounds__(256)
ntt_fwd_persistent(const u64* in, u64* out, int npoly, int gpolys, const tw2* __restrict__ twn, const tw2* __restrict__ twk,
                   const LimbConsts* __restrict__ consts, int L, int logN, unsigned* rowcnt, unsigned* err, int unsafe_timing_mode) {
  __shared__ u64 lds[LDS_WORDS];
  const unsigned W = gridDim.x, w = blockIdx.x;
  const int ngroups = (npoly + gpolys - 1) / gpolys;
  const size_t poly_words = (size_t)L << logN;
  for (int g = 0; g <= ngroups; ++g) {
    const int p1 = g < ngroups ? ((npoly - g * gpolys < gpolys) ? npoly - g * gpolys : gpolys) : 0;      // polys with column work
    const int p2 = g >= 1 ? ((npoly - (g - 1) * gpolys < gpolys) ? npoly - (g - 1) * gpolys : gpolys) : 0;  // polys with tile work
    const unsigned R1 = (unsigned)p1 * L, R2 = (unsigned)p2 * L;
    const unsigned n1 = R1 * 16, n2 = R2 << S1;
    const unsigned nmax = n1 > n2 ? n1 : n2;
    for (unsigned i = w; i < nmax; i += W) {
      if (i < n1) {
        const unsigned row = i % R1, unit = i / R1;
        const unsigned poly = row / (unsigned)L, limb = row % (unsigned)L;
        const unsigned b = (poly * 16 + unit) * (unsigned)L + limb;
        const size_t goff = (size_t)g * gpolys * poly_words;
        if (unsafe_timing_mode) fwd_cols_body<ShoupPolicy, S1, false>(b, in + goff, out + goff, twn, consts, L, logN);
        else fwd_cols_body<ShoupPolicy, S1, true>(b, in + goff, out + goff, twn, consts, L, logN);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&rowcnt[(size_t)g * gpolys * L + row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (i < n2) {
        const unsigned row = i % R2;
        if (threadIdx.x == 0) {
          const unsigned* c = &rowcnt[(size_t)(g - 1) * gpolys * L + row];
          unsigned spins = 0;
          while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 22)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          }
          if (!unsafe_timing_mode) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
        __syncthreads();
        const size_t goff = (size_t)(g - 1) * gpolys * poly_words;
        fwd_tile_asm_body(lds, i, out + goff, out + goff, twk, consts, L, logN, p2);
        __syncthreads();          // LDS is reused by the next tile item
      }
    }
  }
}
