// ntt_team.hip.hpp -- EXPERIMENTAL one-pass forward transform (N = 2^16 .. 2^14): the column stages' outputs go to a small, RE-USED
// exchange buffer that stays in the XCD's write-back L2 (0.02 nJ per byte against 0.14 nJ over the fabric), the tile
// stages read it from there.  Off by default (tuning key "team_slots"); included by engine.hip.
//
// A row (one limb of one poly) is handled by 16 workgroups that sit on the same XCD by construction: every workgroup reads its own
// HW_REG_XCC_ID and draws (row, unit) tickets from that XCD's queue (rows in limb-major order, an eighth of them per XCD).  Unit u
//   1. waits until the exchange buffer (slot = row index modulo S, per XCD) has been released by the row that used it S rows ago,
//   2. runs the column stages of columns [256u, 256u + 256) and stores them into the slot (plain stores: dirty lines of this XCD's L2),
//   3. publishes, waits until all 16 units of the row have published,
//   4. loads tile u of the row from the slot (agent-scope loads: served by the L2, not by this CU's L1), releases the slot once the
//      whole workgroup holds its 16 coefficients per thread, and runs the 12 tile stages from registers (NTT_TILE_PRE_ASM_BODY),
//      storing the result to the caller's block.
// A slot is live only from the column stores to the tile loads, so a pool of S slots of 8 N bytes per XCD serves the rows in flight.
// Every wait is for OLDER tickets of the same queue (progress as in round 1's ntt_fwd_cluster) and is bounded: on a time-out *err is
// raised and the host redoes the batch with the two-pass launches.
#pragma once

#define RH_TEAM_PRE_OP(k) "+{v[" #k "]}"
// the column stages of fwd_cols_asm_body<4> with STREAMING (nt) loads of the input row: what passes through once must not displace the
// exchange buffers from the L2 (the stores, into the exchange buffer, keep the default policy)
RH_DEV void team_cols16_nt(const u32 b, const u64* in, u64* out, const tw2* __restrict__ twn, const LimbConsts* __restrict__ consts, int L) {
  constexpr int logN = LT + 4;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const size_t base = (((size_t)(r >> 4) * L + limb) << logN) + (r & 15) * 256;
  const u64 pin = uni64((u64)(size_t)(in + base));
  const u64 pout = uni64((u64)(size_t)(out + base));
  const u64 tw = uni64((u64)(size_t)(twn + ((size_t)limb << logN)));
  const u64 q = uni64(consts[limb].q);
  const u64 nq = (u64)0 - q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
  const u32 tid = threadIdx.x;
  asm volatile(NTT_COLS16_NT_ASM_BODY : : [tid] "v"(tid), [pin] "s"(pin), [pout] "s"(pout), [tw] "s"(tw), [nq0] "s"((u32)nq),
               [nq1] "s"((u32)(nq >> 32)), [nq4] "s"(nq4), [q4] "s"(q4) : NTT_TILE_ASM_CLOBBERS);
}
template <int S1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
ntt_fwd_team(const u64* in, u64* out, unsigned nrows, const tw2* __restrict__ twn, const tw2* __restrict__ twk,
             const LimbConsts* __restrict__ consts, int L, u64* scratch, int S, unsigned* head, unsigned* rowcnt, unsigned* slotrel, unsigned* err) {
  extern __shared__ u64 lds_dyn[];                          // LDS_WORDS words + the host's occupancy padding
  u64* lds = lds_dyn;
  __shared__ unsigned s_ticket;
  constexpr int logN = LT + S1;
  static_assert(S1 == 4, "the exchange-buffer layout below is written for N = 2^16 (16 tiles per row)");
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const unsigned npoly = nrows / (unsigned)L;
  const unsigned lo = (unsigned)(((unsigned long long)nrows * xcc) >> 3), hi = (unsigned)(((unsigned long long)nrows * (xcc + 1)) >> 3);
  const unsigned ntick = (hi - lo) * 16;
  const u32 tid = threadIdx.x;
  for (;;) {
    if (tid == 0)                                           // after a time-out anywhere: stop drawing work (the host redoes the batch)
      s_ticket = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0xFFFFFFFFu
                                                                                     : __hip_atomic_fetch_add(&head[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned t = s_ticket;
    __syncthreads();
    if (t >= ntick) break;
    const unsigned k = t >> 4, unit = t & 15;               // k: row index within this XCD's queue
    const unsigned row = lo + k;
    const unsigned limb = row / npoly, poly = row % npoly;
    const unsigned slot = k % (unsigned)S, gen = k / (unsigned)S;
    u64* sl = scratch + (((size_t)xcc * S + slot) << logN);
    unsigned* rel = &slotrel[xcc * (unsigned)S + slot];
    if (tid == 0) {                                         // 1. the slot's previous user (row k - S) has loaded all its tiles
      unsigned spins = 0;
      while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u * gen) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 20)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;      // someone timed out: drain
      }
    }
    __syncthreads();
    // 2. column stages; block index of the column body = (poly * 16 + unit) * L + limb, its row base subtracted so that it lands in the slot
    const size_t rowbase = ((size_t)poly * L + limb) << logN;
    team_cols16_nt((poly * 16 + unit) * (unsigned)L + limb, in, sl - rowbase, twn, consts, L);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {                                         // 3. publish, wait for the row
      __hip_atomic_fetch_add(&rowcnt[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(&rowcnt[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 20)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
    }
    __syncthreads();
    // 4. tile `unit`: x[j] = slot[unit * 4096 + tid + 256 j] through the L2 (agent scope: not this CU's L1)
    const u64* pd = sl + ((size_t)unit << LT) + tid;
#define RH_TEAM_LD(j) __hip_atomic_load(pd + 256 * (j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
    u64 d0 = RH_TEAM_LD(0), d1 = RH_TEAM_LD(1), d2 = RH_TEAM_LD(2), d3 = RH_TEAM_LD(3), d4 = RH_TEAM_LD(4), d5 = RH_TEAM_LD(5),
        d6 = RH_TEAM_LD(6), d7 = RH_TEAM_LD(7), d8 = RH_TEAM_LD(8), d9 = RH_TEAM_LD(9), d10 = RH_TEAM_LD(10), d11 = RH_TEAM_LD(11),
        d12 = RH_TEAM_LD(12), d13 = RH_TEAM_LD(13), d14 = RH_TEAM_LD(14), d15 = RH_TEAM_LD(15);
#undef RH_TEAM_LD
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(d8), "+v"(d9), "+v"(d10),
                 "+v"(d11), "+v"(d12), "+v"(d13), "+v"(d14), "+v"(d15) : : "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(rel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
      const size_t obase = rowbase + ((size_t)unit << LT);
      const u64 pout = uni64((u64)(size_t)(out + obase));
      const u64 tw = uni64((u64)(size_t)(twk + ((size_t)limb << logN) + ((size_t)unit << LT)));
      const u64 q = uni64(consts[limb].q);
      const u64 nq = (u64)0 - q, nq2 = (u64)0 - 2 * q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
      const u32 lds_off = uni32((u32)(size_t)lds);
      asm volatile(NTT_TILE_PRE_ASM_BODY
                   : RH_TEAM_PRE_OP(0:1)(d0), RH_TEAM_PRE_OP(2:3)(d1), RH_TEAM_PRE_OP(4:5)(d2), RH_TEAM_PRE_OP(6:7)(d3), RH_TEAM_PRE_OP(8:9)(d4),
                     RH_TEAM_PRE_OP(10:11)(d5), RH_TEAM_PRE_OP(12:13)(d6), RH_TEAM_PRE_OP(14:15)(d7), RH_TEAM_PRE_OP(16:17)(d8), RH_TEAM_PRE_OP(18:19)(d9),
                     RH_TEAM_PRE_OP(20:21)(d10), RH_TEAM_PRE_OP(22:23)(d11), RH_TEAM_PRE_OP(24:25)(d12), RH_TEAM_PRE_OP(26:27)(d13),
                     RH_TEAM_PRE_OP(28:29)(d14), RH_TEAM_PRE_OP(30:31)(d15)
                   : [tid] "v"(tid), [lds] "s"(lds_off), [pout] "s"(pout), [tw] "s"(tw),
                     [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)),
                     [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)), [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4)
                   : NTT_TILE_PRE_ASM_CLOBBERS);
    }
    __syncthreads();                                        // the LDS tile is re-used by the next ticket
  }
}
