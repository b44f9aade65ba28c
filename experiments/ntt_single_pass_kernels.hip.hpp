// experiments/ntt_single_pass_kernels.hip.hpp -- forward-NTT variants that were measured on MI355X in round 1 and lost
// ("what did not work").  NOT part of libringhip.so and not built by the Makefile: kept as source so the
// measurements can be repeated.  They compiled against csrc/ntt_kernels_asm.hip.hpp at commit 27fe980 (which also carried the
// sc1-load and pre-loaded variants of the generated tile body, tools/gen_tile_asm.py at that commit) and were driven by the
// tuning keys persistent / cluster / prefetch / cols2 / order_mix of that commit's rh_ring_set_tuning.
//
//   ntt_fwd_fused_pre   tile loads issued ahead of the column stages            7.20 -> 7.21 ms per 1024 polys (no gain)
//   ntt_fwd_persistent  single launch, hand-off through the Infinity Cache      7.9 ms
//   ntt_fwd_cluster     single pass through one XCD's L2 (HW_REG_XCC_ID queues)  10.5 ms, fetched bytes 17.3 -> 9.1 GB
//   ntt_fwd_cols2       16-byte column accesses                                  no change
#if 0
// Same pipeline step with the tile's 16 data loads issued BEFORE the column stages run: by the time the column
// butterflies and stores are done the tile data has landed, so a workgroup exposes one memory latency per step instead
// of two.  The loads are ordinary C++ loads (the compiler tracks their vmcnt); the values enter the assembly body as
// read-write operands pinned to v[2k:2k+1] (NTT_TILE_PRE_ASM_BODY = the forward body without its own data loads).
#define RH_PRE_OP(k) "+{v[" #k "]}"
template <int S1>
__global__ void __launch_bounds__(256)
ntt_fwd_fused_pre(const u64* in1, u64* out1, unsigned n1, u64* data2, unsigned n2, int npoly2,
                  const tw2* __restrict__ twn, const tw2* __restrict__ twk, const LimbConsts* __restrict__ consts, int L, int logN) {
  __shared__ u64 lds[LDS_WORDS];
  constexpr int R = 1 << S1;
  const u32 b = blockIdx.x;
  // first / last launch of a pipeline (only one of the two items exists): the plain bodies
  if (b >= n1) { if (b < n2) fwd_tile_asm_body(lds, b, data2, data2, twk, consts, L, logN, npoly2); return; }
  if (b >= n2) { fwd_cols_body<ShoupPolicy, S1>(b, in1, out1, twn, consts, L, logN); return; }
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u32 tid = threadIdx.x;
  const size_t base1 = (((size_t)(r >> 4) * L + limb) << logN) + (r & 15) * 256 + tid;     // column unit (fwd_cols_body)
  const u32 poly2 = r % (u32)npoly2, tile = r / (u32)npoly2;                                // tile (fwd_tile_asm_body)
  const size_t base2 = (((size_t)poly2 * L + limb) << logN) + ((size_t)tile << LT);
  u64 x[R];
#pragma unroll
  for (int k = 0; k < R; ++k) x[k] = in1[base1 + ((size_t)k << LT)];
  const u64* pd = data2 + base2 + tid;
  u64 d0 = pd[0], d1 = pd[256], d2 = pd[512], d3 = pd[768], d4 = pd[1024], d5 = pd[1280], d6 = pd[1536], d7 = pd[1792],
      d8 = pd[2048], d9 = pd[2304], d10 = pd[2560], d11 = pd[2816], d12 = pd[3072], d13 = pd[3328], d14 = pd[3584], d15 = pd[3840];
  asm volatile("" ::: "memory");             // keep both load groups above the column stores
  {
    const tw2* tw = twn + ((size_t)limb << logN);
    ShoupPolicy p; p.init(consts[limb]);
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      const int h = R >> (s + 1);
      const bool red = ref_reduce(s, logN);
#pragma unroll
      for (int g = 0; g < (1 << s); ++g) {
        tw2 w = tw[(1 << s) + g];
#pragma unroll
        for (int e = 0; e < h; ++e) p.fwd(x[g * 2 * h + e], x[g * 2 * h + e + h], w, red);
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) out1[base1 + ((size_t)k << LT)] = x[k];
  }
  {
    const u64 pout = uni64((u64)(size_t)(data2 + base2));
    const u64 tw = uni64((u64)(size_t)(twk + ((size_t)limb << logN) + ((size_t)tile << LT)));
    const u64 q = uni64(consts[limb].q);
    const u64 nq = (u64)0 - q, nq2 = (u64)0 - 2 * q, nq4 = (u64)0 - 4 * q, q4 = 4 * q;
    const u32 lds_off = uni32((u32)(size_t)lds);
    asm volatile(NTT_TILE_PRE_ASM_BODY
                 : RH_PRE_OP(0:1)(d0), RH_PRE_OP(2:3)(d1), RH_PRE_OP(4:5)(d2), RH_PRE_OP(6:7)(d3), RH_PRE_OP(8:9)(d4),
                   RH_PRE_OP(10:11)(d5), RH_PRE_OP(12:13)(d6), RH_PRE_OP(14:15)(d7), RH_PRE_OP(16:17)(d8), RH_PRE_OP(18:19)(d9),
                   RH_PRE_OP(20:21)(d10), RH_PRE_OP(22:23)(d11), RH_PRE_OP(24:25)(d12), RH_PRE_OP(26:27)(d13),
                   RH_PRE_OP(28:29)(d14), RH_PRE_OP(30:31)(d15)
                 : [tid] "v"(tid), [lds] "s"(lds_off), [pout] "s"(pout), [tw] "s"(tw),
                   [twlo] "s"((u32)(size_t)tw), [twhi] "s"((u32)((size_t)tw >> 32)),
                   [nq0] "s"((u32)nq), [nq1] "s"((u32)(nq >> 32)), [nq] "s"(nq), [nq2] "s"(nq2), [nq4] "s"(nq4), [q4] "s"(q4)
                 : NTT_TILE_PRE_ASM_CLOBBERS);
  }
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

// ---------------------------------------------------------------------------------------------------------------
// Single-pass forward transform through the XCD's L2 (N >= 8192).
//
// A row (one limb of one poly, 8*N bytes) is handled by 16 workgroups THAT SIT ON THE SAME XCD BY CONSTRUCTION: every
// workgroup reads its own HW_REG_XCC_ID and draws (row, unit) tickets from that XCD's queue, which only holds rows
// r with r % 8 == xcc.  Unit u first runs the column stages of columns [256u, 256u+256) (plain stores: the lines
// stay dirty in this XCD's write-back L2), publishes, waits until all 16 units of the row have published, and then
// runs the 12 tile stages of tile u, loading the row with sc1 loads (bypass the CU's L1, served by the shared L2).
// The final stores overwrite the same lines, so each coefficient crosses the HBM interface once in and once out.
//
// Correctness does not depend on how the dispatcher places workgroups: co-location is established at run time from
// the hardware id, and within one XCD the L2 is the single point of coherence for every CU (completed stores are in
// L2 once the storing wave's vmcnt reaches 0).  Progress: tickets are drawn in order by running workgroups only, so
// the oldest incomplete row always has all of its ticket holders resident as soon as >= 16 workgroups of the launch
// live on that XCD; the wait is bounded and raises *err otherwise (host falls back to the two-pass path).
// ---------------------------------------------------------------------------------------------------------------
template <int S1>
__global__ void __launch_bounds__(256)
ntt_fwd_cluster(const u64* in, u64* out, unsigned nrows, const tw2* __restrict__ twn, const tw2* __restrict__ twk,
                const LimbConsts* __restrict__ consts, int L, int logN, unsigned* head, unsigned* rowcnt, unsigned* err,
                int dbg_skip) {
  __shared__ u64 lds[LDS_WORDS];
  __shared__ unsigned s_ticket;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  // rows in limb-major order rho = limb*npoly + poly; XCD x owns the contiguous range [x*nrows/8, (x+1)*nrows/8): successive
  // rows of an XCD share a limb, so that limb's 16*N-byte tile twiddle table stays in the XCD's L2 beside the rows in flight
  const unsigned npoly = nrows / (unsigned)L;
  const unsigned lo = (unsigned)(((unsigned long long)nrows * xcc) >> 3), hi = (unsigned)(((unsigned long long)nrows * (xcc + 1)) >> 3);
  const unsigned ntick = (hi - lo) * 16;
  for (;;) {
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&head[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned t = s_ticket;
    __syncthreads();
    if (t >= ntick) break;
    const unsigned row = lo + (t >> 4), unit = t & 15;
    const unsigned limb = row / npoly, poly = row % npoly;
    // column stages: block index of fwd_cols_body = (poly*16 + unit)*L + limb
    if (!(dbg_skip & 1)) fwd_cols_body<ShoupPolicy, S1>((poly * 16 + unit) * (unsigned)L + limb, in, out, twn, consts, L, logN);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&rowcnt[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(&rowcnt[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 22)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
    // tile stages of tile `unit` of this row; block index of the tile body with npoly = 1: limb = b % L, r = b / L = tile
    // a row has 2^S1 tiles but always 16 column units: unit u takes tiles u, u+16, ... (none when u >= 2^S1)
    if (!(dbg_skip & 2)) {
      for (unsigned tile = unit; tile < (1u << S1); tile += 16) {
        fwd_tile_asm_body<true>(lds, tile * (unsigned)L + limb, out + ((size_t)poly * L << logN), out + ((size_t)poly * L << logN), twk, consts, L, logN, 1);
        __syncthreads();
      }
    }
    __syncthreads();
  }
}


// two adjacent columns per thread: every global access is 16 B per lane (1 KiB per wave instruction)
template <class P, int S1>
RH_DEV void fwd_cols2_body(const u32 b, const u64* in, u64* out, const typename P::tw_t* __restrict__ twn,
                           const LimbConsts* __restrict__ consts, int L, int logN) {
  constexpr int R = 1 << S1;
  const u32 limb = b % (u32)L;
  const u32 r = b / (u32)L;
  const u32 cb = r & 7;             // 8 blocks of 512 columns per limb
  const u32 poly = r >> 3;
  const size_t base = (((size_t)poly * L + limb) << logN) + cb * 512 + 2 * threadIdx.x;
  const typename P::tw_t* tw = twn + ((size_t)limb << logN);
  P p; p.init(consts[limb]);
  u64 x[R], y[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(in + base + ((size_t)k << LT));
    x[k] = v.x; y[k] = v.y;
  }
#pragma unroll
  for (int s = 0; s < S1; ++s) {
    const int h = R >> (s + 1);
    const bool red = ref_reduce(s, logN);
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      typename P::tw_t w = tw[(1 << s) + g];
#pragma unroll
      for (int e = 0; e < h; ++e) {
        p.fwd(x[g * 2 * h + e], x[g * 2 * h + e + h], w, red);
        p.fwd(y[g * 2 * h + e], y[g * 2 * h + e + h], w, red);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    ulonglong2 v; v.x = x[k]; v.y = y[k];
    *reinterpret_cast<ulonglong2*>(out + base + ((size_t)k << LT)) = v;
  }
}
template <class P, int S1>
__global__ void __launch_bounds__(256)
ntt_fwd_cols2(const u64* in, u64* out, const typename P::tw_t* __restrict__ twn,
              const LimbConsts* __restrict__ consts, int L, int logN) {
  fwd_cols2_body<P, S1>(blockIdx.x, in, out, twn, consts, L, logN);
}

#endif
