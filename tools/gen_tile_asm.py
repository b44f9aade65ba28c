#!/usr/bin/env python3
"""Generates matrix-fhe-lattigo_amd/csrc/ntt_tile_asm.inc: the body of the forward 4096-tile NTT kernel (the metric's
dominant kernel) as ONE hand-scheduled gfx950 assembly block, wrapped by ntt_fwd_tile_asm in ntt_kernels_asm.hip.hpp.

Why assembly: the butterfly is VALU-issue bound and hipcc's code for it is ~21 slow + 3 fast
instructions; the sequence below is 16 slow + 2 fast, uses no VCC in the hot chain, and fits 4 waves/SIMD.

Butterfly (Shoup form, r = V*w - Q'*q in [0,4q), values < 8q), registers are even-aligned pairs:
   T  = U - 4q ; M = T.hi >>s 31 ; u = M ? U : T                        (3 slow + 1 fast, no vcc)
   Q' = V1*p1 + hi(V1*p0) + hi(V0*p1)                                    (4 slow; zero-extended pairs {a,0},{b,0})
   h  = lo32(V0*w1 + V1*w0 + Q0*nq1 + Q1*nq0)                            (4 mads, each adds into the LOW word)
   X  = u + (h<<32) + V0*w0 + Q0*nq0                                     (1 fast + 2 slow)
   Y  = (2u + 4q) - X                                                    (1 + 2 slow)

Same math as ShoupPolicy::fwd / shoup_mul_acc (modarith.hip.hpp); outputs are canonical so results are bit-identical to
the C++ kernel and to the reference's Forward.  Layouts (LDS padding j + j/16, kernel-order twiddles) as in
ntt_kernels.hip.hpp: fwd_tile_body.
"""
import os
import sys

EXP = int(os.environ.get("RH_ASM_EXP", "0"))     # TIMING EXPERIMENTS ONLY (wrong results): 1 no LDS exchanges, 2 no round-B/C twiddle loads, 4 no final reduction, 8 no barriers, 16 no round C
NOTAIL = int(os.environ.get("RH_ASM_NOTAIL", "0"))   # A/B: forward tile bodies end with their stores in flight (no trailing s_waitcnt vmcnt(0))
PRIO = int(os.environ.get("RH_ASM_PRIO", "0"))   # s_setprio around the load-issue and store phases (0 = off, for A/B runs)
out = []


# Cache policy of the DATA streams (every 8-byte-per-lane global access of these bodies moves coefficients that pass through once; twiddles
# are the 16-byte accesses and the scalar loads): " nt" = non-temporal, the streams do not age out the twiddle tables and each other's lines in
# the caches.  Every body is rendered twice: NAME (default policy) and NAME_NT; the pipelined launches of large batches (working sets far
# beyond the 256 MiB Infinity Cache) use the _NT bodies: 7.00 -> 6.74 ms per step of the metric, -2 .. -3 % on the key switch; batches that fit the
# Infinity Cache (32 .. 256 MiB) run 3 - 6 % SLOWER with them (tools/exp_nt_sizes.sh), hence the two sets.  KEEP_CACHED: loads that ARE
# re-used (the one coefficient-domain row every limb of a rescale re-expands) keep the default policy in both.
DATA_FLAGS = ""          # set per generation pass below ("" and " nt")
KEEP_CACHED = [False]


def emit(s):
    global DATA_FLAGS
    if DATA_FLAGS and (s.startswith("global_load_dwordx2 ") or s.startswith("global_store_dwordx2 ")) and not (KEEP_CACHED[0] and s.startswith("global_load")):
        if not s.rstrip().endswith(("nt", "sc0", "sc1")):
            s = s + DATA_FLAGS
    out.append(s)


# ---------------------------------------------------------------------------------------------- register map (VGPR)
def X(k):            # data pair k (0..15)
    return 2 * k


TW0 = 32             # 15 twiddles x 4 VGPR (w.lo, w.hi, wp.lo, wp.hi) = v32..v91
TMP0 = 92            # two temp sets of 12: T(2) M(1) pad(1) Q(2) H(2: lo, ZERO) G(2: lo, ZERO) R(2) -> and S(2) = 14
NTMP = 14
ADDR = 120           # v120..v123 address temporaries
SCR = TMP0           # v92..v95: global offsets (only live before the first and after the last butterfly)
NVGPR_USED = 124     # v0..v123 clobbered; the compiler keeps its own operands (tid) above that


def pair(r):
    return "v[%d:%d]" % (r, r + 1)


class Tmp:
    def __init__(self, base):
        self.T = base
        self.M = base + 2
        self.Q = base + 4
        self.H = base + 6      # H.lo = a, H.hi = 0 (kept zero)
        self.G = base + 8      # G.lo = b, G.hi = 0 (kept zero)
        self.R = base + 10
        self.S = base + 12
        self.cc = None         # carry pair name


T0 = Tmp(TMP0)
T1 = Tmp(TMP0 + NTMP)
T0.cc = "s[96:97]"
T1.cc = "s[98:99]"
DUMMY = "vcc"           # carry-out sink of v_mad_u64_u32 (vcc is otherwise only address scratch between rounds)


def butterfly_steps(u, v, w, t, sgpr_tw=None):
    """returns the instruction list of one forward butterfly on data pairs u (U) and v (V).
    w: VGPR index of the twiddle quad, or None when sgpr_tw = (s_w_lo, s_w_hi, s_p_lo, s_p_hi) names SGPRs."""
    if sgpr_tw is None:
        w0, w1, p0, p1 = "v%d" % w, "v%d" % (w + 1), "v%d" % (w + 2), "v%d" % (w + 3)
    else:
        w0, w1, p0, p1 = sgpr_tw
    U, V = pair(u), pair(v)
    ul, uh, vl, vh = "v%d" % u, "v%d" % (u + 1), "v%d" % v, "v%d" % (v + 1)
    T, Q, H, G, R, S = pair(t.T), pair(t.Q), pair(t.H), pair(t.G), pair(t.R), pair(t.S)
    tl, th, m = "v%d" % t.T, "v%d" % (t.T + 1), "v%d" % t.M
    ql, qh = "v%d" % t.Q, "v%d" % (t.Q + 1)
    return [
        "v_lshl_add_u64 %s, %s, 0, %%[nq4]" % (T, U),
        "v_mul_hi_u32 v%d, %s, %s" % (t.H, vh, p0),
        "v_mul_hi_u32 v%d, %s, %s" % (t.G, vl, p1),
        "v_ashrrev_i32 %s, 31, %s" % (m, th),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (Q, DUMMY, vh, p1, H),
        "v_mad_u64_u32 %s, %s, %s, %s, 0" % (R, DUMMY, vl, w1),
        "v_bfi_b32 %s, %s, %s, %s" % (ul, m, ul, tl),
        "v_bfi_b32 %s, %s, %s, %s" % (uh, m, uh, th),
        "v_lshl_add_u64 %s, %s, 0, %s" % (Q, Q, G),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (S, DUMMY, vh, w0, R),
        "v_lshl_add_u64 %s, %s, 1, %%[q4]" % (T, U),
        "v_mad_u64_u32 %s, %s, %s, %%[nq1], %s" % (R, DUMMY, ql, S),
        "v_mad_u64_u32 %s, %s, %s, %%[nq0], %s" % (S, DUMMY, qh, R),
        "v_add_u32 %s, v%d, %s" % (uh, t.S, uh),                   # u + (h << 32), in place (V_LSHL_ADD_U64 shifts <= 7 only)
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (R, DUMMY, vl, w0, U),
        "v_mad_u64_u32 %s, %s, %s, %%[nq0], %s" % (U, DUMMY, ql, R),
        "v_sub_co_u32 %s, %s, %s, %s" % (vl, t.cc, tl, ul),
        "@CARRY",                                                   # marker: >= 2 wait states to the consumer
        "v_subb_co_u32 %s, %s, %s, %s, %s" % (vh, t.cc, th, uh, t.cc),
    ]


def inv_butterfly_steps(u, v, w, t, sgpr_tw=None):
    """inverse (Gentleman-Sande) butterfly, U,V < 4q:  X = csub(U+V, 4q) -> U ;  Y = (U + 4q - V)*w in [0,4q) -> V.
    Same math as ShoupPolicy::inv (ntt_kernels.hip.hpp).  17 slow + 2 fast."""
    if sgpr_tw is None:
        w0, w1, p0, p1 = "v%d" % w, "v%d" % (w + 1), "v%d" % (w + 2), "v%d" % (w + 3)
    else:
        w0, w1, p0, p1 = sgpr_tw
    U, V = pair(u), pair(v)
    ul, uh, vl, vh = "v%d" % u, "v%d" % (u + 1), "v%d" % v, "v%d" % (v + 1)
    T, Q, H, G, R, S = pair(t.T), pair(t.Q), pair(t.H), pair(t.G), pair(t.R), pair(t.S)
    tl, th, m = "v%d" % t.T, "v%d" % (t.T + 1), "v%d" % t.M
    ql, qh = "v%d" % t.Q, "v%d" % (t.Q + 1)
    return [
        "v_lshl_add_u64 %s, %s, 0, %s" % (T, U, V),                 # T = U + V
        "v_lshl_add_u64 %s, %s, 0, %%[q4]" % (U, U),                # U = U + 4q
        "v_sub_co_u32 %s, %s, %s, %s" % (vl, t.cc, ul, vl),         # V = d = U + 4q - V
        "@CARRY",
        "v_subb_co_u32 %s, %s, %s, %s, %s" % (vh, t.cc, uh, vh, t.cc),
        "v_lshl_add_u64 %s, %s, 0, %%[nq4]" % (U, T),               # U = T - 4q
        "v_mul_hi_u32 v%d, %s, %s" % (t.H, vh, p0),
        "v_mul_hi_u32 v%d, %s, %s" % (t.G, vl, p1),
        "v_ashrrev_i32 %s, 31, %s" % (m, uh),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (Q, DUMMY, vh, p1, H),
        "v_mad_u64_u32 %s, %s, %s, %s, 0" % (R, DUMMY, vl, w1),
        "v_bfi_b32 %s, %s, %s, %s" % (ul, m, tl, ul),               # X = (T - 4q < 0) ? T : T - 4q
        "v_bfi_b32 %s, %s, %s, %s" % (uh, m, th, uh),
        "v_lshl_add_u64 %s, %s, 0, %s" % (Q, Q, G),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (S, DUMMY, vh, w0, R),
        "v_mad_u64_u32 %s, %s, %s, %%[nq1], %s" % (R, DUMMY, ql, S),
        "v_mad_u64_u32 %s, %s, %s, %%[nq0], %s" % (S, DUMMY, qh, R),
        "v_mad_u64_u32 %s, %s, %s, %s, 0" % (R, DUMMY, vl, w0),
        "v_add_u32 v%d, v%d, v%d" % (t.R + 1, t.S, t.R + 1),        # + (h << 32)
        "v_mad_u64_u32 %s, %s, %s, %%[nq0], %s" % (V, DUMMY, ql, R),
    ]


def mred_lazy_steps(x, y, t):
    """x <- MRedLazy(x, y) = x*y*2^-64 mod q, in [0, 2q) for x, y < 2q (ring/modular_reduction.go:90-95: ahi - hi64((alo*qinv)*q) + q);
    x, y: data pairs; operands qi0/qi1 (MRedConstant), q0/q1 and the pair q.  16 slow + 10 fast."""
    xl, xh, yl, yh = "v%d" % x, "v%d" % (x + 1), "v%d" % y, "v%d" % (y + 1)
    R, S, Q, T, H, G = pair(t.R), pair(t.S), pair(t.Q), pair(t.T), pair(t.H), pair(t.G)
    rl, rh_, sl, sh, ql, qh = "v%d" % t.R, "v%d" % (t.R + 1), "v%d" % t.S, "v%d" % (t.S + 1), "v%d" % t.Q, "v%d" % (t.Q + 1)
    tl, th, hl, gl, m = "v%d" % t.T, "v%d" % (t.T + 1), "v%d" % t.H, "v%d" % t.G, "v%d" % t.M
    return [
        "v_mad_u64_u32 %s, %s, %s, %s, 0" % (R, DUMMY, xl, yl),            # x0*y0: p0 = R.lo
        "v_mov_b32 %s, %s" % (hl, rh_),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (S, DUMMY, xl, yh, H),        # x0*y1 + carry
        "v_mov_b32 %s, %s" % (gl, sl),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (Q, DUMMY, xh, yl, G),        # x1*y0 + ...: p1 = Q.lo
        "v_mov_b32 %s, %s" % (hl, sh),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (T, DUMMY, xh, yh, H),        # x1*y1 + carry
        "v_mov_b32 %s, %s" % (gl, qh),
        "v_lshl_add_u64 %s, %s, 0, %s" % (T, T, G),                        # ahi = T
        "v_mad_u64_u32 %s, %s, %s, %%[qi0], 0" % (S, DUMMY, rl),           # m = alo * qinv mod 2^64 -> S
        "v_mul_lo_u32 %s, %s, %%[qi1]" % (m, rl),
        "v_add_u32 %s, %s, %s" % (sh, sh, m),
        "v_mul_lo_u32 %s, %s, %%[qi0]" % (m, ql),
        "v_add_u32 %s, %s, %s" % (sh, sh, m),
        "v_mad_u64_u32 %s, %s, %s, %%[q0], 0" % (R, DUMMY, sl),            # hi64(m * q) -> Q
        "v_mov_b32 %s, %s" % (hl, rh_),
        "v_mad_u64_u32 %s, %s, %s, %%[q1], %s" % (Q, DUMMY, sl, H),
        "v_mov_b32 %s, %s" % (gl, ql),
        "v_mad_u64_u32 %s, %s, %s, %%[q0], %s" % (R, DUMMY, sh, G),
        "v_mov_b32 %s, %s" % (hl, qh),
        "v_mad_u64_u32 %s, %s, %s, %%[q1], %s" % (Q, DUMMY, sh, H),
        "v_mov_b32 %s, %s" % (gl, rh_),
        "v_lshl_add_u64 %s, %s, 0, %s" % (Q, Q, G),
        "v_sub_co_u32 %s, %s, %s, %s" % (xl, t.cc, tl, ql),
        "@CARRY",
        "v_subb_co_u32 %s, %s, %s, %s, %s" % (xh, t.cc, th, qh, t.cc),
        "v_lshl_add_u64 %s, %s, 0, %%[q]" % (pair(x), pair(x)),
    ]


def interleave(a, b):
    """instruction-wise interleave of two independent butterflies; resolves @CARRY markers (the partner's
    instructions provide the wait states between v_sub_co and v_subb_co; pad with s_nop when they do not)"""
    res = []
    i = j = 0
    while i < len(a) or j < len(b):
        if i < len(a):
            res.append(("a", a[i])); i += 1
        if j < len(b):
            res.append(("b", b[j])); j += 1
    final = []
    pending = {}
    for who, ins in res:
        if ins == "@CARRY":
            pending[who] = 0
            continue
        if ins.startswith("v_subb_co_u32") and who in pending:
            gap = pending.pop(who)
            if gap < 2:
                final.append("s_nop %d" % (1 - gap))
        for k in pending:
            pending[k] += 1
        final.append(ins)
    return final


def single(a):
    final = []
    for ins in a:
        if ins == "@CARRY":
            final.append("s_nop 1")
        else:
            final.append(ins)
    return final


def round16_inv(tw_of_slot, stage_hook=None):
    for u in (3, 2, 1, 0):
        h = 8 >> u
        if stage_hook:
            stage_hook(u)
        bfs = []
        for g in range(1 << u):
            slot = (1 << u) - 1 + g
            w, sg = tw_of_slot(slot)
            for e in range(h):
                k = g * 2 * h + e
                bfs.append((X(k), X(k + h), w, sg))
        for i in range(0, len(bfs), 2):
            a = inv_butterfly_steps(bfs[i][0], bfs[i][1], bfs[i][2], T0, bfs[i][3])
            b = inv_butterfly_steps(bfs[i + 1][0], bfs[i + 1][1], bfs[i + 1][2], T1, bfs[i + 1][3])
            for ins in interleave(a, b):
                emit(ins)


INV_SLOT_ORDER = list(range(7, 15)) + list(range(3, 7)) + [1, 2] + [0]      # consumption order of an inverse round
INV_NEED = {3: 8, 2: 12, 1: 14, 0: 15}                                      # twiddle quads needed before stage u


def gen_inverse(mul=False, from_registers=False, lds_out=False):
    """first 12 stages (t = 1..2048) of the inverse transform on a 4096-tile; values leave < 4q (no scaling):
    mirror image of gen(); same LDS layout and kernel-order twiddle table (built from RootsBackward).
    mul: the input is the pointwise Montgomery product of two blocks (pin, pin2), formed on load (rh_ring_intt_mul); the second
    operand lands in the twiddle registers v32..v63, so the round-C' twiddle loads are issued after the products."""
    A0, A1, A2, A3 = ADDR, ADDR + 1, ADDR + 2, ADDR + 3
    for t in (T0, T1):
        emit("v_mov_b32 v%d, 0" % (t.H + 1))
        emit("v_mov_b32 v%d, 0" % (t.G + 1))
    for slot in range(15):
        emit("s_load_dwordx4 s[%d:%d], %%[tw], %d" % (36 + 4 * slot, 39 + 4 * slot, 16 * slot))
    emit("v_lshlrev_b32 v%d, 3, %%[tid]" % A0)
    for j in range(4):
        emit("v_add_u32 v%d, %d, v%d" % (SCR + j, 8192 * j + 4096, A0))
    for k in range(16):
        if from_registers:                                    # gen_polymul: x[k] already holds coefficient 16 tid + k (the layout of round C')
            break
        j, rem = divmod(k, 4)
        emit("global_load_dwordx2 %s, v%d, %%[pin] offset:%d" % (pair(X(k)), SCR + j, rem * 2048 - 4096))
    if mul:
        for k in range(16):
            j, rem = divmod(k, 4)
            emit("global_load_dwordx2 %s, v%d, %%[pin2] offset:%d" % (pair(TW0 + 2 * k), SCR + j, rem * 2048 - 4096))
        for k in range(0, 16, 2):                              # x[k] is load k, y[k] load 16 + k of the 32 issued
            emit("s_waitcnt vmcnt(%d)" % (14 - k))
            for ins in interleave(mred_lazy_steps(X(k), TW0 + 2 * k, T0), mred_lazy_steps(X(k + 1), TW0 + 2 * (k + 1), T1)):
                emit(ins)
    # round C' twiddles: tw[256 + slot*256 + tid]
    emit("v_lshlrev_b32 v%d, 4, %%[tid]" % A2)
    for slot in INV_SLOT_ORDER:
        emit("s_add_u32 vcc_lo, %%[twlo], %d" % ((256 + 256 * slot) * 16))
        emit("s_addc_u32 vcc_hi, %[twhi], 0")
        emit("global_load_dwordx4 v[%d:%d], v%d, vcc" % (TW0 + 4 * slot, TW0 + 4 * slot + 3, A2))
    emit("v_lshrrev_b32 v%d, 4, %%[tid]" % A1)                # hi4
    emit("v_add_u32 v%d, %%[tid], v%d" % (A3, A1))
    emit("v_lshlrev_b32 v%d, 3, v%d" % (A3, A3))
    emit("v_add_u32 v%d, %%[lds], v%d" % (A3, A3))            # addrA
    emit("v_mul_u32_u24 v%d, 136, %%[tid]" % A0)
    emit("v_add_u32 v%d, %%[lds], v%d" % (A0, A0))            # addrC
    if not from_registers:
        emit("s_waitcnt vmcnt(15)")                           # the 16 data loads
        for k in range(16):
            emit("ds_write_b64 v%d, %s offset:%d" % (A3, pair(X(k)), 2176 * k))
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_barrier")
        for k in range(16):
            emit("ds_read_b64 %s, v%d offset:%d" % (pair(X(k)), A0, 8 * k))
        emit("s_waitcnt lgkmcnt(0)")
    emit("; ---- round C' (t = 1, 2, 4, 8)")
    round16_inv(lambda slot: (TW0 + 4 * slot, None), stage_hook=lambda u: emit("s_waitcnt vmcnt(%d)" % (15 - INV_NEED[u])))
    # round B' twiddles: tw[16 + slot*16 + hi4]
    emit("v_lshlrev_b32 v%d, 4, v%d" % (A2, A1))              # hi4*16 bytes
    for slot in INV_SLOT_ORDER:
        emit("global_load_dwordx4 v[%d:%d], v%d, %%[tw] offset:%d" % (TW0 + 4 * slot, TW0 + 4 * slot + 3, A2, 256 + 256 * slot))
    for k in range(16):
        emit("ds_write_b64 v%d, %s offset:%d" % (A0, pair(X(k)), 8 * k))
    emit("v_and_b32 v%d, 15, %%[tid]" % A0)                   # lo4
    emit("v_mul_u32_u24 v%d, 272, v%d" % (A1, A1))
    emit("v_add_u32 v%d, v%d, v%d" % (A0, A0, A1))
    emit("v_lshlrev_b32 v%d, 3, v%d" % (A0, A0))
    emit("v_add_u32 v%d, %%[lds], v%d" % (A0, A0))            # addrB
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    for k in range(16):
        emit("ds_read_b64 %s, v%d offset:%d" % (pair(X(k)), A0, 136 * k))
    emit("s_waitcnt lgkmcnt(0)")
    emit("; ---- round B' (t = 16 .. 128)")
    round16_inv(lambda slot: (TW0 + 4 * slot, None), stage_hook=lambda u: emit("s_waitcnt vmcnt(%d)" % (15 - INV_NEED[u])))
    for k in range(16):
        emit("ds_write_b64 v%d, %s offset:%d" % (A0, pair(X(k)), 136 * k))
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    for k in range(16):
        emit("ds_read_b64 %s, v%d offset:%d" % (pair(X(k)), A3, 2176 * k))
    emit("s_waitcnt lgkmcnt(0)")
    emit("; ---- round A' (t = 256 .. 2048), twiddles in SGPRs")
    round16_inv(lambda slot: (None, ("s%d" % (36 + 4 * slot), "s%d" % (37 + 4 * slot), "s%d" % (38 + 4 * slot), "s%d" % (39 + 4 * slot))))
    if lds_out:          # one-pass transform: the tile stays in LDS (layout A, the thread's own slots of round A') for the column stages of the same kernel
        for k in range(16):
            emit("ds_write_b64 v%d, %s offset:%d" % (A3, pair(X(k)), 2176 * k))
        emit("s_waitcnt lgkmcnt(0)")
        return
    emit("v_lshlrev_b32 v%d, 3, %%[tid]" % A0)
    for j in range(4):
        emit("v_add_u32 v%d, %d, v%d" % (SCR + j, 8192 * j + 4096, A0))
    for k in range(16):
        j, rem = divmod(k, 4)
        emit("global_store_dwordx2 v%d, %s, %%[pout] offset:%d" % (SCR + j, pair(X(k)), rem * 2048 - 4096))
    emit("s_waitcnt vmcnt(0)")


SAVE0 = 124          # gen_polymul: the first operand's 16 transformed coefficients wait in v124..v155 while the second is transformed
NVGPR_POLYMUL = 156


def gen_polymul():
    """One workgroup, one 4096-tile: forward tile stages of a (pin) and of b (pin2), their pointwise Montgomery product, the inverse
    tile stages, store (pout) -- the middle of c = INTT(NTT(a) . NTT(b)) (BASELINE config 3) without NTT(a), NTT(b) or their product
    ever reaching memory.  The forward body ends with thread tid holding coefficients 16 tid + k, which is where the inverse body's
    first round starts: the forward transpose + store and the inverse load + transpose drop out with the memory passes.
    Operands: the forward body's, plus pin2, twi / twilo / twihi (kernel-order inverse table) and qi0 / qi1 / q0 / q1 / q (MRedLazy)."""
    gen(stop_after_reduce=True, reduce_below_2q=True)
    for k in range(16):
        emit("v_mov_b32 v%d, v%d" % (SAVE0 + 2 * k, X(k)))
        emit("v_mov_b32 v%d, v%d" % (SAVE0 + 2 * k + 1, X(k) + 1))
    mark = len(out)
    gen(stop_after_reduce=True, barrier_before_lds=True, reduce_below_2q=True)
    for i in range(mark, len(out)):
        out[i] = out[i].replace("%[pin]", "%[pin2]")
    emit("; ---- x[k] = MRedLazy(NTT(b)[k], NTT(a)[k]) < 2q  (both operands < 2q, congruent to the canonical values)")
    for k in range(0, 16, 2):
        for ins in interleave(mred_lazy_steps(X(k), SAVE0 + 2 * k, T0), mred_lazy_steps(X(k + 1), SAVE0 + 2 * (k + 1), T1)):
            emit(ins)
    mark = len(out)
    gen_inverse(from_registers=True)
    for i in range(mark, len(out)):
        out[i] = out[i].replace("%[tw]", "%[twi]").replace("%[twlo]", "%[twilo]").replace("%[twhi]", "%[twihi]")


def round16(tw_of_slot, stage_hook=None, pair_hook=None, stages=4):
    """`stages` stages over x[0 .. 2^stages - 1] (4 -> the radix-16 round); tw_of_slot(slot) -> (vgpr_quad_index or None,
    sgpr tuple or None).  stage_hook(u) / pair_hook(u, i) emit waits just before the first consumer (guide G15)"""
    for u in range(stages):
        h = (1 << (stages - 1)) >> u
        if stage_hook:
            stage_hook(u)
        bfs = []
        for g in range(1 << u):
            slot = (1 << u) - 1 + g
            w, sg = tw_of_slot(slot)
            for e in range(h):
                k = g * 2 * h + e
                bfs.append((X(k), X(k + h), w, sg))
        for i in range(0, len(bfs), 2):
            if pair_hook:
                pair_hook(u, i)
            a = butterfly_steps(bfs[i][0], bfs[i][1], bfs[i][2], T0, bfs[i][3])
            b = butterfly_steps(bfs[i + 1][0], bfs[i + 1][1], bfs[i + 1][2], T1, bfs[i + 1][3])
            for ins in interleave(a, b):
                emit(ins)


def csub_all(const_name):
    """x[k] = csub(x[k], bound) for all 16, bound given as the SGPR pair holding -bound"""
    for k in range(0, 16, 2):
        for kk, t in ((k, T0), (k + 1, T1)):
            emit("v_lshl_add_u64 %s, %s, 0, %%[%s]" % (pair(t.T), pair(X(kk)), const_name))
        for kk, t in ((k, T0), (k + 1, T1)):
            emit("v_ashrrev_i32 v%d, 31, v%d" % (t.M, t.T + 1))
        for kk, t in ((k, T0), (k + 1, T1)):
            emit("v_bfi_b32 v%d, v%d, v%d, v%d" % (X(kk), t.M, X(kk), t.T))
            emit("v_bfi_b32 v%d, v%d, v%d, v%d" % (X(kk) + 1, t.M, X(kk) + 1, t.T + 1))


def gen(load_flags="", store_flags="", preloaded=False, tail_wait=True, epilogue=None, lazy_out=False, stop_after_reduce=False, barrier_before_lds=False, reduce_below_2q=False,
        lds_in=False):
    """lds_in: the tile already sits in the workgroup's LDS in layout A (word tid + (tid >> 4) + 272 k holds coefficient tid + 256 k), written
    by the column stages of the same kernel (one-pass transform of N = 2^13 / 2^14): the 16 data loads become ds_reads of the thread's OWN slots,
    which round A then overwrites -- no barrier in between"""
    A0, A1, A2, A3 = ADDR, ADDR + 1, ADDR + 2, ADDR + 3
    if lds_in:
        preloaded = True
    emit("; ---- prologue: zero halves of the zero-extended pairs, addresses")
    if PRIO in (1, 2):
        emit("s_setprio 3")                                   # a young wave issues its loads ahead of the older waves' butterflies
    for t in (T0, T1):
        emit("v_mov_b32 v%d, 0" % (t.H + 1))
        emit("v_mov_b32 v%d, 0" % (t.G + 1))
    # round A twiddles: 15 uniform Shoup pairs -> s[36:95]
    for slot in range(15):
        emit("s_load_dwordx4 s[%d:%d], %%[tw], %d" % (36 + 4 * slot, 39 + 4 * slot, 16 * slot))
    # data loads: x[k] = in[tid + 256k]; byte offset tid*8 + 2048k; 4 offset registers cover imm -4096..2048
    emit("v_lshlrev_b32 v%d, 3, %%[tid]" % A0)
    for j in range(4):
        emit("v_add_u32 v%d, %d, v%d" % (SCR + j, 8192 * j + 4096, A0))
    if lds_in:
        emit("v_lshrrev_b32 v%d, 4, %%[tid]" % A1)            # hi4
        emit("v_add_u32 v%d, %%[tid], v%d" % (A3, A1))
        emit("v_lshlrev_b32 v%d, 3, v%d" % (A3, A3))          # addrA
        emit("v_add_u32 v%d, %%[lds], v%d" % (A3, A3))
        for k in range(16):
            emit("ds_read_b64 %s, v%d offset:%d" % (pair(X(k)), A3, 2176 * k))
    for kk in range(16):
        if preloaded:
            break                                             # x[k] already sits in v[2k:2k+1] (loaded by the caller)
        k = (kk >> 1) + 8 * (kk & 1)
        j, rem = divmod(k, 4)
        emit("global_load_dwordx2 %s, v%d, %%[pin] offset:%d%s" % (pair(X(k)), SCR + j, rem * 2048 - 4096, load_flags))
    # round B twiddles: tw[16 + slot*16 + hi4] -> byte (16+16*slot)*16 + hi4*16
    emit("v_lshrrev_b32 v%d, 4, %%[tid]" % A1)                # hi4
    emit("v_lshlrev_b32 v%d, 4, v%d" % (A2, A1))              # hi4*16 bytes
    for slot in range(15):
        emit("global_load_dwordx4 v[%d:%d], v%d, %%[tw] offset:%d" % (TW0 + 4 * slot, TW0 + 4 * slot + 3, A2, 256 + 256 * slot))
    # LDS addresses
    #   A: 8*(tid + (tid>>4))          + 2176*k
    #   B: 8*(272*hi4 + lo4)           + 136*k
    #   C: 136*tid                     + 8*k
    emit("v_add_u32 v%d, %%[tid], v%d" % (A3, A1))
    emit("v_lshlrev_b32 v%d, 3, v%d" % (A3, A3))              # addrA
    emit("v_add_u32 v%d, %%[lds], v%d" % (A3, A3))
    emit("s_waitcnt lgkmcnt(0)")                              # round A twiddles (scalar loads)
    if PRIO in (1, 2):
        emit("s_setprio 0")
    emit("; ---- round A (twiddles in SGPRs)")

    def a_pair_hook(u, i):
        # stage 0, butterflies i and i+1 consume data loads 0 .. 2*i+3 of the 31 issued (16 data then 15 twiddle quads)
        if u == 0 and not preloaded:
            emit("s_waitcnt vmcnt(%d)" % (31 - (2 * i + 4)))
    round16(lambda slot: (None, ("s%d" % (36 + 4 * slot), "s%d" % (37 + 4 * slot), "s%d" % (38 + 4 * slot), "s%d" % (39 + 4 * slot))),
            pair_hook=a_pair_hook)
    if barrier_before_lds:                                    # a second transform in the same workgroup (gen_polymul): every wave must have
        emit("s_barrier")                                     # finished the previous transform's LDS reads before the tile is written again
    for k in range(16):
        emit("ds_write_b64 v%d, %s offset:%d" % (A3, pair(X(k)), 2176 * k))
    # addrB
    emit("v_and_b32 v%d, 15, %%[tid]" % A0)                   # lo4
    emit("v_mul_u32_u24 v%d, 272, v%d" % (A1, A1))            # 272*hi4
    emit("v_add_u32 v%d, v%d, v%d" % (A0, A0, A1))
    emit("v_lshlrev_b32 v%d, 3, v%d" % (A0, A0))              # addrB
    emit("v_add_u32 v%d, %%[lds], v%d" % (A0, A0))
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    for k in range(16):
        emit("ds_read_b64 %s, v%d offset:%d" % (pair(X(k)), A0, 136 * k))
    emit("s_waitcnt vmcnt(0)")                                # round B twiddles
    emit("s_waitcnt lgkmcnt(0)")
    emit("; ---- round B")
    round16(lambda slot: (TW0 + 4 * slot, None))
    # round C twiddles into the same registers: tw[256 + slot*256 + tid], 16 B each
    emit("v_lshlrev_b32 v%d, 4, %%[tid]" % A2)
    for slot in range(15):
        emit("s_add_u32 vcc_lo, %%[twlo], %d" % ((256 + 256 * slot) * 16))
        emit("s_addc_u32 vcc_hi, %[twhi], 0")
        emit("global_load_dwordx4 v[%d:%d], v%d, vcc" % (TW0 + 4 * slot, TW0 + 4 * slot + 3, A2))
    for k in range(16):
        emit("ds_write_b64 v%d, %s offset:%d" % (A0, pair(X(k)), 136 * k))
    emit("v_mul_u32_u24 v%d, 136, %%[tid]" % A1)              # addrC
    emit("v_add_u32 v%d, %%[lds], v%d" % (A1, A1))
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    for k in range(16):
        emit("ds_read_b64 %s, v%d offset:%d" % (pair(X(k)), A1, 8 * k))
    emit("s_waitcnt lgkmcnt(0)")
    emit("; ---- round C (twiddle quads arrive in slot order; stage u needs slots < 2^(u+1) - 1)")
    if EXP & 16 and not epilogue:      # TIMING ONLY: an 8-stage pass (rounds A and B) -- what a balanced 8 + 8 split's pass would cost
        emit("s_waitcnt vmcnt(0)")
    else:
        round16(lambda slot: (TW0 + 4 * slot, None), stage_hook=lambda u: emit("s_waitcnt vmcnt(%d)" % (15 - ((2 << u) - 1))))
    if epilogue:
        return gen_submul_epilogue(A0, A1, A3, with_z=(epilogue == "submul_add"))
    emit("; ---- canonical reduction: x < 8q -> [0,q)")
    if PRIO in (1, 3):
        emit("s_setprio 2")                                   # finish: reduce, transpose, store -> frees the CU slot sooner
    if not lazy_out:                                          # lazy_out: values leave < 8q (consumers that reduce anyway: the key
        csub_all("nq4")                                       # multiply-accumulate of the gadget product takes any 64-bit operand)
        csub_all("nq2")
        if not reduce_below_2q:                               # gen_polymul: MRedLazy takes operands < 2q (4 q^2 < q 2^64), the last step is not needed
            csub_all("nq")
    if stop_after_reduce:                                     # gen_polymul: thread tid now holds the canonical NTT values of coefficients
        return                                                # 16 tid + k in x[k] -- the layout the inverse tile's first round starts from
    for k in range(16):
        emit("ds_write_b64 v%d, %s offset:%d" % (A1, pair(X(k)), 8 * k))
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    for k in range(16):
        emit("ds_read_b64 %s, v%d offset:%d" % (pair(X(k)), A3, 2176 * k))
    emit("v_lshlrev_b32 v%d, 3, %%[tid]" % A0)
    for j in range(4):
        emit("v_add_u32 v%d, %d, v%d" % (SCR + j, 8192 * j + 4096, A0))
    for k in range(16):
        j, rem = divmod(k, 4)
        emit("s_waitcnt lgkmcnt(%d)" % (15 - k))
        emit("global_store_dwordx2 v%d, %s, %%[pout] offset:%d%s" % (SCR + j, pair(X(k)), rem * 2048 - 4096, store_flags))
    if tail_wait and not NOTAIL:
        emit("s_waitcnt vmcnt(0)")


class VmTracker:
    """issue order of vector-memory operations -> the vmcnt that guarantees a given one has completed (they retire in order)"""

    def __init__(self):
        self.n = 0
        self.idx = {}

    def issue(self, tag):
        self.idx[tag] = self.n
        self.n += 1

    def wait(self, *tags):
        need = max(self.idx[t] for t in tags)
        emit("s_waitcnt vmcnt(%d)" % min(self.n - 1 - need, 63))


def gen_submul_epilogue(A0, A1, A3, with_z):
    """tail of the forward tile body with the epilogue of ntt_fwd_tile_submul (ntt_kernels.hip.hpp):
         out = [z +] MRed(2q - y + NTT(x), s)        (ring/basis_extension.go:255-257, ring/scaling.go:120-124; z: the ring.Add that follows)
    MRed by the wave-uniform scalar s is a Shoup multiply by s*2^-64 mod q (operands sw0/sw1/sp0/sp1): the canonical result is the
    same residue.  The transform's values are only brought below 4q before the subtraction (one conditional subtraction, not three).
    y -> v32..v63 (the twiddle registers, free after round C), z -> v64..v79 in two halves, per-lane offsets v80..v83."""
    Y0, Z0, EOFF = TW0, TW0 + 32, TW0 + 48
    vm = VmTracker()
    emit("; ---- epilogue: out = [z +] MRed(2q - y + NTT(x), s)")
    emit("v_lshlrev_b32 v%d, 3, %%[tid]" % A0)
    for j in range(4):
        emit("v_add_u32 v%d, %d, v%d" % (EOFF + j, 8192 * j + 4096, A0))
    for k in range(16):
        j, rem = divmod(k, 4)
        emit("global_load_dwordx2 %s, v%d, %%[py] offset:%d" % (pair(Y0 + 2 * k), EOFF + j, rem * 2048 - 4096))
        vm.issue(("y", k))
    if with_z:
        for k in range(8):
            j, rem = divmod(k, 4)
            emit("global_load_dwordx2 %s, v%d, %%[pz] offset:%d" % (pair(Z0 + 2 * k), EOFF + j, rem * 2048 - 4096))
            vm.issue(("z", k))
    csub_all("nq4")
    for k in range(16):
        emit("ds_write_b64 v%d, %s offset:%d" % (A1, pair(X(k)), 8 * k))
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    for k in range(16):
        emit("ds_read_b64 %s, v%d offset:%d" % (pair(X(k)), A3, 2176 * k))
    emit("s_waitcnt lgkmcnt(0)")
    sw = ("%[sw0]", "%[sw1]", "%[sp0]", "%[sp1]")

    def one(k, t):
        x, y = X(k), Y0 + 2 * k
        st = ["v_lshl_add_u64 %s, %s, 0, %%[q2]" % (pair(x), pair(x)),
              "v_sub_co_u32 v%d, %s, v%d, v%d" % (x, t.cc, x, y),
              "@CARRY",
              "v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (x + 1, t.cc, x + 1, y + 1, t.cc)]
        st += shoup_mul_steps(x, sw, t) + csub_steps(x, "nq2", t) + csub_steps(x, "nq", t)
        if with_z:
            st += ["v_lshl_add_u64 %s, %s, 0, %s" % (pair(x), pair(x), pair(Z0 + 2 * (k % 8)))] + csub_steps(x, "nq", t)
        return st
    for k in range(0, 16, 2):
        if with_z and k == 8:                                  # second half of z into the same registers (their readers have issued)
            for kk in range(8, 16):
                j, rem = divmod(kk, 4)
                emit("global_load_dwordx2 %s, v%d, %%[pz] offset:%d" % (pair(Z0 + 2 * (kk - 8)), EOFF + j, rem * 2048 - 4096))
                vm.issue(("z", kk))
        tags = [("y", k + 1)] + ([("z", k + 1)] if with_z else [])
        vm.wait(*tags)
        for ins in interleave(one(k, T0), one(k + 1, T1)):
            emit(ins)
        for kk in (k, k + 1):
            j, rem = divmod(kk, 4)
            emit("global_store_dwordx2 v%d, %s, %%[pout] offset:%d" % (EOFF + j, pair(X(kk)), rem * 2048 - 4096))
            vm.issue(("o", kk))
    emit("s_waitcnt vmcnt(0)")


def gen_cols(S1=4, expand=False):
    """expand: the inputs are the coefficient-domain last limb t of a rescale step (ring/scaling.go:97-118), re-expanded on load as
    x = cred(t + hq, qL) + s  (operands hq, nqL = -qL, sadd: SGPR pairs; DivFloor passes hq = sadd = 0), NOT reduced modulo the limb's
    q: the host guarantees qL + q <= 8q, the first stage's conditional subtraction and the Shoup multiply take it from there.
    Column stages for N = 2^(12+S1), S1 = 2..4: x[k] = in[c + 4096 k], k < R = 2^S1, one radix-R register round with the
    wave-uniform twiddles tw[1..R-1] (natural order: stage s, group g -> tw[2^s + g] = slot 2^s - 1 + g), outputs < 8q
    stored back.  Same contract as fwd_cols_body<ShoupPolicy, S1> (ntt_kernels.hip.hpp); operands: tid, pin, pout (row base +
    256*cb columns, bytes), tw (limb's natural-order table), nq0, nq1, nq4, q4."""
    R = 1 << S1
    if PRIO in (1, 2):
        emit("s_setprio 3")
    for t in (T0, T1):
        emit("v_mov_b32 v%d, 0" % (t.H + 1))
        emit("v_mov_b32 v%d, 0" % (t.G + 1))
    for slot in range(R - 1):
        emit("s_load_dwordx4 s[%d:%d], %%[tw], %d" % (36 + 4 * slot, 39 + 4 * slot, 16 * (slot + 1)))
    emit("v_lshlrev_b32 v%d, 3, %%[tid]" % TW0)                # byte offset of column c; element k adds 32768*k
    for k in range(1, R):
        emit("v_add_u32 v%d, %d, v%d" % (TW0 + k, 32768 * k, TW0))
    for kk in range(R):
        k = (kk >> 1) + (R >> 1) * (kk & 1)
        KEEP_CACHED[0] = expand
        emit("global_load_dwordx2 %s, v%d, %%[pin]" % (pair(X(k)), TW0 + k))
        KEEP_CACHED[0] = False
    emit("s_waitcnt lgkmcnt(0)")
    if PRIO in (1, 2):
        emit("s_setprio 0")
    if expand:
        for kk in range(0, R, 2):
            ks = [((kk + j) >> 1) + (R >> 1) * ((kk + j) & 1) for j in (0, 1)]
            emit("s_waitcnt vmcnt(%d)" % (R - 2 - kk))
            a, b = [], []
            for lst, k, t in ((a, ks[0], T0), (b, ks[1], T1)):
                lst.append("v_lshl_add_u64 %s, %s, 0, %%[hq]" % (pair(X(k)), pair(X(k))))
                lst.extend(csub_steps(X(k), "nqL", t))
                lst.append("v_lshl_add_u64 %s, %s, 0, %%[sadd]" % (pair(X(k)), pair(X(k))))
            for ins in interleave(a, b):
                emit(ins)

    def pair_hook(u, i):
        if u == 0 and not expand:
            emit("s_waitcnt vmcnt(%d)" % (R - (2 * i + 4)))
    round16(lambda slot: (None, ("s%d" % (36 + 4 * slot), "s%d" % (37 + 4 * slot), "s%d" % (38 + 4 * slot), "s%d" % (39 + 4 * slot))),
            pair_hook=pair_hook, stages=S1)
    for k in range(R):
        emit("global_store_dwordx2 v%d, %s, %%[pout]" % (TW0 + k, pair(X(k))))


def shoup_mul_steps(v, tw, t):
    """v <- v * w in [0,4q) for the data pair v (any 64-bit value), tw = (w0, w1, p0, p1) operand names (SGPRs): the
    multiplication half of inv_butterfly_steps.  11 slow + 1 fast."""
    w0, w1, p0, p1 = tw
    V = pair(v)
    vl, vh = "v%d" % v, "v%d" % (v + 1)
    Q, H, G, R, S = pair(t.Q), pair(t.H), pair(t.G), pair(t.R), pair(t.S)
    ql, qh = "v%d" % t.Q, "v%d" % (t.Q + 1)
    return [
        "v_mul_hi_u32 v%d, %s, %s" % (t.H, vh, p0),
        "v_mul_hi_u32 v%d, %s, %s" % (t.G, vl, p1),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (Q, DUMMY, vh, p1, H),
        "v_mad_u64_u32 %s, %s, %s, %s, 0" % (R, DUMMY, vl, w1),
        "v_lshl_add_u64 %s, %s, 0, %s" % (Q, Q, G),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (S, DUMMY, vh, w0, R),
        "v_mad_u64_u32 %s, %s, %s, %%[nq1], %s" % (R, DUMMY, ql, S),
        "v_mad_u64_u32 %s, %s, %s, %%[nq0], %s" % (S, DUMMY, qh, R),
        "v_mad_u64_u32 %s, %s, %s, %s, 0" % (R, DUMMY, vl, w0),
        "v_add_u32 v%d, v%d, v%d" % (t.R + 1, t.S, t.R + 1),
        "v_mad_u64_u32 %s, %s, %s, %%[nq0], %s" % (V, DUMMY, ql, R),
    ]


def csub_steps(x, const_name, t):
    """x <- x - bound if x >= bound (bound given as the SGPR pair holding -bound); 3 slow + 1 fast"""
    return [
        "v_lshl_add_u64 %s, %s, 0, %%[%s]" % (pair(t.T), pair(x), const_name),
        "v_ashrrev_i32 v%d, 31, v%d" % (t.M, t.T + 1),
        "v_bfi_b32 v%d, v%d, v%d, v%d" % (x, t.M, x, t.T),
        "v_bfi_b32 v%d, v%d, v%d, v%d" % (x + 1, t.M, x + 1, t.T + 1),
    ]


def scaled_last_butterfly_steps(u, v, t):
    """last inverse stage with N^-1 folded in (inv_cols_body, ntt_kernels.hip.hpp): U, V < 4q ->
    X = canon((U + V) * ninv) -> U,  Y = canon((U + 4q - V) * (psi * ninv)) -> V, both in [0, q)."""
    U, V = pair(u), pair(v)
    ul, uh, vl, vh = "v%d" % u, "v%d" % (u + 1), "v%d" % v, "v%d" % (v + 1)
    ninv = ("%[iw0]", "%[iw1]", "%[ip0]", "%[ip1]")
    last = ("%[lw0]", "%[lw1]", "%[lp0]", "%[lp1]")
    steps = [
        "v_lshl_add_u64 %s, %s, 0, %s" % (pair(t.T), U, V),          # T = U + V
        "v_lshl_add_u64 %s, %s, 0, %%[q4]" % (U, U),                 # U = U + 4q
        "v_sub_co_u32 %s, %s, %s, %s" % (vl, t.cc, ul, vl),          # V = U + 4q - V
        "@CARRY",
        "v_subb_co_u32 %s, %s, %s, %s, %s" % (vh, t.cc, uh, vh, t.cc),
        "v_mov_b32 %s, v%d" % (ul, t.T),                             # U = T
        "v_mov_b32 %s, v%d" % (uh, t.T + 1),
    ]
    steps += shoup_mul_steps(u, ninv, t) + csub_steps(u, "nq2", t) + csub_steps(u, "nq", t)
    steps += shoup_mul_steps(v, last, t) + csub_steps(v, "nq2", t) + csub_steps(v, "nq", t)
    return steps


def gen_cols_inv(S1=4):
    """Inverse column stages for N = 2^(12+S1), S1 = 2..4, scaled: the stages with 2^(S1-1) .. 2 blocks (wave-uniform
    twiddles tw[2^u + g]), then the last stage with N^-1 folded into both multipliers; canonical outputs.  Same contract as
    inv_cols_body<S1>(scale = 1).  In place."""
    R = 1 << S1
    if PRIO in (1, 2):
        emit("s_setprio 3")
    for t in (T0, T1):
        emit("v_mov_b32 v%d, 0" % (t.H + 1))
        emit("v_mov_b32 v%d, 0" % (t.G + 1))
    for slot in range(1, R - 1):
        emit("s_load_dwordx4 s[%d:%d], %%[tw], %d" % (36 + 4 * slot, 39 + 4 * slot, 16 * (slot + 1)))
    emit("v_lshlrev_b32 v%d, 3, %%[tid]" % TW0)
    for k in range(1, R):
        emit("v_add_u32 v%d, %d, v%d" % (TW0 + k, 32768 * k, TW0))
    for k in range(R):
        emit("global_load_dwordx2 %s, v%d, %%[pin]" % (pair(X(k)), TW0 + k))
    emit("s_waitcnt lgkmcnt(0)")
    if PRIO in (1, 2):
        emit("s_setprio 0")
    emit("s_waitcnt vmcnt(0)")
    for u in range(S1 - 1, 0, -1):
        h = (R >> 1) >> u
        bfs = []
        for g in range(1 << u):
            slot = (1 << u) - 1 + g
            sg = ("s%d" % (36 + 4 * slot), "s%d" % (37 + 4 * slot), "s%d" % (38 + 4 * slot), "s%d" % (39 + 4 * slot))
            for e in range(h):
                k = g * 2 * h + e
                bfs.append((X(k), X(k + h), sg))
        for i in range(0, len(bfs), 2):
            a = inv_butterfly_steps(bfs[i][0], bfs[i][1], None, T0, bfs[i][2])
            b = inv_butterfly_steps(bfs[i + 1][0], bfs[i + 1][1], None, T1, bfs[i + 1][2])
            for ins in interleave(a, b):
                emit(ins)
    for e in range(0, R >> 1, 2):
        a = scaled_last_butterfly_steps(X(e), X(e + (R >> 1)), T0)
        b = scaled_last_butterfly_steps(X(e + 1), X(e + 1 + (R >> 1)), T1)
        for ins in interleave(a, b):
            emit(ins)
    for k in range(R):
        emit("global_store_dwordx2 v%d, %s, %%[pin]" % (TW0 + k, pair(X(k))))


def gen_3n_cols_post_inv(S1):
    """3N transform, inverse, b = 1 (ntt3n.hip: ntt3n_cols_post_inv<S1>): the column stages (unscaled, values < 4q) of the six
    sub-transforms of a limb, then the radix-3 layer and the split merge (post_b1), 6 * 2^S1 coefficients per thread.
    Data: x[j][k] = row[j * n2 + col + 4096 k] in v[2 (j R + k)]; two temp sets and one spare pair each above them; v0..v127 in all.
    Operands: wbase (wave's first thread id, SGPR), pin / pout (row base + 256 cb columns, bytes), tw (the limb's six sub-ring
    natural-order tables, n2 entries apart), l3p (Limb3N of the limb), tp (radix-3 inverse pairs z1, z2 of halves 0 and 1),
    nq0, nq1, nq, nq2, nq4, q4.  SGPRs: s[36:63] / s[64:91] twiddle buffers, s[92:95] running pointers, s[96:99] carries."""
    R = 1 << S1
    logn2 = 12 + S1
    sB, twB = (1 << logn2) * 8, (1 << logn2) * 16
    XJ = lambda j, k: 2 * (j * R + k)
    base = 12 * R
    Ta, Tb = Tmp(base), Tmp(base + NTMP)
    Ta.cc, Tb.cc = "s[96:97]", "s[98:99]"
    E = (base + 2 * NTMP, base + 2 * NTMP + 2)
    OFF = Tb.T                                                  # R <= 8 offset registers, live only while loads / stores issue
    BUF = (36, 64)
    quad = lambda b, i: ("s%d" % (b + 4 * i), "s%d" % (b + 4 * i + 1), "s%d" % (b + 4 * i + 2), "s%d" % (b + 4 * i + 3))

    def tid_offsets():
        emit("v_mbcnt_lo_u32_b32 v%d, -1, 0" % OFF)
        emit("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (OFF, OFF))
        emit("v_add_u32 v%d, %%[wbase], v%d" % (OFF, OFF))
        emit("v_lshlrev_b32 v%d, 3, v%d" % (OFF, OFF))
        for k in range(1, R):
            emit("v_add_u32 v%d, %d, v%d" % (OFF + k, 32768 * k, OFF))

    def load_tw(buf):                                           # natural-order entries 1 .. R-1 of the table at s[94:95]
        for i in range(1, R):
            emit("s_load_dwordx4 s[%d:%d], s[94:95], %d" % (buf + 4 * (i - 1), buf + 4 * (i - 1) + 3, 16 * i))

    def sub_steps(x, y, t):                                     # x <- x - y (64-bit, borrow through the temp set's carry pair)
        return ["v_sub_co_u32 v%d, %s, v%d, v%d" % (x, t.cc, x, y), "@CARRY",
                "v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (x + 1, t.cc, x + 1, y + 1, t.cc)]

    def emit_pairs(items, fn):
        """fn(item, tempset, spare) -> instruction list; consecutive items run interleaved on the two temp sets"""
        for i in range(0, len(items), 2):
            a = fn(items[i], Ta, E[0])
            if i + 1 < len(items):
                for ins in interleave(a, fn(items[i + 1], Tb, E[1])):
                    emit(ins)
            else:
                for ins in single(a):
                    emit(ins)

    tid_offsets()
    emit("s_mov_b64 s[92:93], %[pin]")
    emit("s_mov_b64 s[94:95], %[tw]")
    emit("s_nop 4")
    load_tw(BUF[0])
    for j in range(6):
        for k in range(R):
            emit("global_load_dwordx2 %s, v%d, s[92:93]" % (pair(XJ(j, k)), OFF + k))
        if j < 5:
            emit("s_add_u32 s92, s92, %d" % sB)
            emit("s_addc_u32 s93, s93, 0")
            emit("s_nop 4")
    for t in (Ta, Tb):
        emit("v_mov_b32 v%d, 0" % (t.H + 1))
        emit("v_mov_b32 v%d, 0" % (t.G + 1))
    # ---- column stages of sub-transform j: inv_cols_body(scale = 0), twiddle tw[2^st + g]
    for j in range(6):
        buf = BUF[j & 1]
        emit("s_waitcnt lgkmcnt(0)")
        if j < 5:
            emit("s_add_u32 s94, s94, %d" % twB)
            emit("s_addc_u32 s95, s95, 0")
            emit("s_nop 4")
            load_tw(BUF[(j + 1) & 1])
        else:                                                   # the radix-3 constants ride behind the last sub-transform
            emit("s_load_dwordx4 s[36:39], %[l3p], 16")         # w3
            for i in range(4):
                emit("s_load_dwordx4 s[%d:%d], %%[tp], %d" % (40 + 4 * i, 43 + 4 * i, 16 * i))
        emit("s_waitcnt vmcnt(%d)" % (R * (5 - j)))
        for it in range(S1):
            st = S1 - 1 - it
            h = R >> (st + 1)
            bfs = []
            for g in range(1 << st):
                sg = quad(buf, (1 << st) + g - 1)
                for e in range(h):
                    bfs.append((XJ(j, g * 2 * h + e), XJ(j, g * 2 * h + e + h), sg))
            emit_pairs(bfs, lambda b, t, sp: inv_butterfly_steps(b[0], b[1], None, t, b[2]))
    # ---- radix-3 layer (post_b1, first loop), all columns; the merge constants load meanwhile into the second buffer
    emit("s_waitcnt lgkmcnt(0)")
    for i, off in enumerate((32, 48, 64)):                      # inv_b1, inv_b0z, inv_s
        emit("s_load_dwordx4 s[%d:%d], %%[l3p], %d" % (64 + 4 * i, 67 + 4 * i, off))
    W3 = quad(36, 0)

    def radix3(item, t, sp):
        k, h = item
        B0, B1, B2 = XJ(3 * h, k), XJ(3 * h + 1, k), XJ(3 * h + 2, k)
        z1, z2 = quad(40, 2 * h), quad(40, 2 * h + 1)
        ins = ["v_lshl_add_u64 %s, %s, 0, %%[q4]" % (pair(sp), pair(B1))] + sub_steps(sp, B2, t)     # sp = B1 + 4q - B2
        ins += shoup_mul_steps(sp, W3, t)                                                            # tt
        ins += ["v_mov_b32 v%d, v%d" % (t.R, B0), "v_mov_b32 v%d, v%d" % (t.R + 1, B0 + 1)]          # R = B0
        ins += ["v_lshl_add_u64 %s, %s, 0, %s" % (pair(B0), pair(B0), pair(B1))] + csub_steps(B0, "nq4", t)
        ins += ["v_lshl_add_u64 %s, %s, 0, %s" % (pair(B0), pair(B0), pair(B2))] + csub_steps(B0, "nq4", t)
        ins += ["v_lshl_add_u64 %s, %s, 0, %%[q4]" % (pair(t.S), pair(t.R))]                         # S = B0 + 4q
        ins += ["v_sub_co_u32 v%d, %s, v%d, v%d" % (B1, t.cc, t.S, B1), "@CARRY",
                "v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (B1 + 1, t.cc, t.S + 1, B1 + 1, t.cc)] + csub_steps(B1, "nq4", t)   # B0 - B1
        ins += ["v_sub_co_u32 v%d, %s, v%d, v%d" % (B2, t.cc, t.S, B2), "@CARRY",
                "v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (B2 + 1, t.cc, t.S + 1, B2 + 1, t.cc)] + csub_steps(B2, "nq4", t)   # B0 - B2
        ins += ["v_lshl_add_u64 %s, %s, 0, %%[q4]" % (pair(B1), pair(B1))] + sub_steps(B1, sp, t) + csub_steps(B1, "nq4", t)  # s1
        ins += ["v_lshl_add_u64 %s, %s, 0, %s" % (pair(B2), pair(B2), pair(sp))] + csub_steps(B2, "nq4", t)                    # s2
        ins += shoup_mul_steps(B1, z1, t) + shoup_mul_steps(B2, z2, t)
        return ins
    emit_pairs([(k, h) for k in range(R) for h in (0, 1)], radix3)
    # ---- split merge (post_b1, second loop): canonical outputs
    emit("s_waitcnt lgkmcnt(0)")
    IB1, IB0Z, IS = quad(64, 0), quad(64, 1), quad(64, 2)

    def merge(item, t, sp):
        k, jj = item
        lo, hi = XJ(jj, k), XJ(jj + 3, k)
        ins = ["v_lshl_add_u64 %s, %s, 0, %%[q4]" % (pair(hi), pair(hi))] + sub_steps(hi, lo, t)      # d = hi + 4q - lo
        ins += ["v_mov_b32 v%d, v%d" % (sp, hi), "v_mov_b32 v%d, v%d" % (sp + 1, hi + 1)]
        ins += shoup_mul_steps(hi, IB1, t) + shoup_mul_steps(sp, IB0Z, t) + shoup_mul_steps(lo, IS, t)
        ins += ["v_lshl_add_u64 %s, %s, 0, %%[q4]" % (pair(lo), pair(lo))] + sub_steps(lo, sp, t)
        ins += csub_steps(lo, "nq4", t) + csub_steps(lo, "nq2", t) + csub_steps(lo, "nq", t)
        ins += csub_steps(hi, "nq2", t) + csub_steps(hi, "nq", t)
        return ins
    emit_pairs([(k, jj) for k in range(R) for jj in range(3)], merge)
    # ---- stores
    tid_offsets()
    emit("s_mov_b64 s[92:93], %[pout]")
    emit("s_nop 4")
    for j in range(6):
        for k in range(R):
            emit("global_store_dwordx2 v%d, %s, s[92:93]" % (OFF + k, pair(XJ(j, k))))
        if j < 5:
            emit("s_add_u32 s92, s92, %d" % sB)
            emit("s_addc_u32 s93, s93, 0")
            emit("s_nop 4")
    emit("s_waitcnt vmcnt(0)")


def gen_3n_pre_cols_fwd(S1):
    """3N transform, forward, b = 1 (ntt3n.hip: ntt3n_pre_cols_fwd<S1>): split + radix-3 layer (pre_b1) on the six values of every column,
    then the column stages of the six sub-transforms (outputs < 8q, any representative: the tile stages reduce).  Same register map,
    operands and SGPR use as gen_3n_cols_post_inv; tp = the forward radix-3 pairs."""
    R = 1 << S1
    logn2 = 12 + S1
    sB, twB = (1 << logn2) * 8, (1 << logn2) * 16
    loc = {(j, k): 2 * (j * R + k) for j in range(6) for k in range(R)}        # register of x[j][k]; the radix-3 layer renames
    base = 12 * R
    Ta, Tb = Tmp(base), Tmp(base + NTMP)
    Ta.cc, Tb.cc = "s[96:97]", "s[98:99]"
    E = (base + 2 * NTMP, base + 2 * NTMP + 2)
    OFF = Tb.T
    quad = lambda b, i: ("s%d" % (b + 4 * i), "s%d" % (b + 4 * i + 1), "s%d" % (b + 4 * i + 2), "s%d" % (b + 4 * i + 3))

    def tid_offsets():
        emit("v_mbcnt_lo_u32_b32 v%d, -1, 0" % OFF)
        emit("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (OFF, OFF))
        emit("v_add_u32 v%d, %%[wbase], v%d" % (OFF, OFF))
        emit("v_lshlrev_b32 v%d, 3, v%d" % (OFF, OFF))
        for k in range(1, R):
            emit("v_add_u32 v%d, %d, v%d" % (OFF + k, 32768 * k, OFF))

    def load_tw(buf):
        for i in range(1, R):
            emit("s_load_dwordx4 s[%d:%d], s[94:95], %d" % (buf + 4 * (i - 1), buf + 4 * (i - 1) + 3, 16 * i))

    def sub_steps(x, y, t):
        return ["v_sub_co_u32 v%d, %s, v%d, v%d" % (x, t.cc, x, y), "@CARRY",
                "v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (x + 1, t.cc, x + 1, y + 1, t.cc)]

    def rsub_steps(x, a, t):                                    # x <- a - x
        return ["v_sub_co_u32 v%d, %s, v%d, v%d" % (x, t.cc, a, x), "@CARRY",
                "v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (x + 1, t.cc, a + 1, x + 1, t.cc)]

    def add_steps(x, y):
        return ["v_lshl_add_u64 %s, %s, 0, %s" % (pair(x), pair(x), pair(y))]

    def addq4(x):
        return ["v_lshl_add_u64 %s, %s, 0, %%[q4]" % (pair(x), pair(x))]

    def emit_pairs(items, fn):
        for i in range(0, len(items), 2):
            a = fn(items[i], Ta, E[0])
            if i + 1 < len(items):
                for ins in interleave(a, fn(items[i + 1], Tb, E[1])):
                    emit(ins)
            else:
                for ins in single(a):
                    emit(ins)

    tid_offsets()
    emit("s_mov_b64 s[92:93], %[pin]")
    emit("s_mov_b64 s[94:95], %[tw]")
    emit("s_nop 4")
    emit("s_load_dwordx4 s[36:39], %[l3p], 0")                 # zeta
    emit("s_load_dwordx4 s[40:43], %[l3p], 16")                # w3
    for i in range(4):
        emit("s_load_dwordx4 s[%d:%d], %%[tp], %d" % (44 + 4 * i, 47 + 4 * i, 16 * i))
    load_tw(64)                                                 # sub-transform 0's column twiddles
    for j in range(6):
        for k in range(R):
            emit("global_load_dwordx2 %s, v%d, s[92:93]" % (pair(loc[(j, k)]), OFF + k))
        if j < 5:
            emit("s_add_u32 s92, s92, %d" % sB)
            emit("s_addc_u32 s93, s93, 0")
            emit("s_nop 4")
    for t in (Ta, Tb):
        emit("v_mov_b32 v%d, 0" % (t.H + 1))
        emit("v_mov_b32 v%d, 0" % (t.G + 1))
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    ZETA, W3 = quad(36, 0), quad(36, 1)
    # ---- inputs < 8q -> < 4q, then the split: lo' = lo + z hi, hi' = lo + hi - z hi
    emit_pairs([(j, k) for k in range(R) for j in range(6)], lambda it, t, sp: csub_steps(loc[it], "nq4", t))

    def split(item, t, sp):
        k, jj = item
        lo, hi = loc[(jj, k)], loc[(jj + 3, k)]
        ins = ["v_mov_b32 v%d, v%d" % (sp, hi), "v_mov_b32 v%d, v%d" % (sp + 1, hi + 1)] + shoup_mul_steps(sp, ZETA, t)   # tt
        ins += add_steps(hi, lo) + csub_steps(hi, "nq4", t)                                  # lo + hi
        ins += addq4(hi) + sub_steps(hi, sp, t) + csub_steps(hi, "nq4", t)                   # - tt
        ins += add_steps(lo, sp) + csub_steps(lo, "nq4", t)                                  # lo + tt
        return ins
    emit_pairs([(k, jj) for k in range(R) for jj in range(3)], split)

    def radix3(item, t, sp):
        k, h = item
        b0, b1, b2 = loc[(3 * h, k)], loc[(3 * h + 1, k)], loc[(3 * h + 2, k)]
        z1, z2 = quad(44, 2 * h), quad(44, 2 * h + 1)
        ins = shoup_mul_steps(b1, z1, t) + shoup_mul_steps(b2, z2, t)                         # t1, t2 in place
        ins += ["v_lshl_add_u64 %s, %s, 0, %%[q4]" % (pair(sp), pair(b1))] + sub_steps(sp, b2, t) + shoup_mul_steps(sp, W3, t)   # t3
        ins += ["v_mov_b32 v%d, v%d" % (t.R, b0), "v_mov_b32 v%d, v%d" % (t.R + 1, b0 + 1)]  # R = b0
        ins += add_steps(b0, b1) + csub_steps(b0, "nq4", t) + add_steps(b0, b2) + csub_steps(b0, "nq4", t)      # x0
        ins += ["v_lshl_add_u64 %s, %s, 0, %%[q4]" % (pair(t.S), pair(t.R))]                 # S = b0 + 4q
        ins += rsub_steps(b2, t.S, t) + csub_steps(b2, "nq4", t) + add_steps(b2, sp) + csub_steps(b2, "nq4", t)  # x1 = (b0 - t2) + t3, in b2's registers
        ins += rsub_steps(b1, t.S, t) + csub_steps(b1, "nq4", t) + addq4(b1) + sub_steps(b1, sp, t) + csub_steps(b1, "nq4", t)   # x2 = (b0 - t1) - t3, in b1's
        return ins
    emit_pairs([(k, h) for k in range(R) for h in (0, 1)], radix3)
    for k in range(R):                                          # x1 / x2 sit in each other's registers
        for h in (0, 1):
            loc[(3 * h + 1, k)], loc[(3 * h + 2, k)] = loc[(3 * h + 2, k)], loc[(3 * h + 1, k)]
    # ---- column stages of sub-transform j (fwd_cols_body: stage st, group g -> tw[2^st + g])
    BUF = (64, 36)
    for j in range(6):
        emit("s_waitcnt lgkmcnt(0)")
        if j < 5:
            emit("s_add_u32 s94, s94, %d" % twB)
            emit("s_addc_u32 s95, s95, 0")
            emit("s_nop 4")
            load_tw(BUF[(j + 1) & 1])
        buf = BUF[j & 1]
        for st in range(S1):
            h = R >> (st + 1)
            bfs = []
            for g in range(1 << st):
                sg = quad(buf, (1 << st) + g - 1)
                for e in range(h):
                    bfs.append((loc[(j, g * 2 * h + e)], loc[(j, g * 2 * h + e + h)], sg))
            emit_pairs(bfs, lambda b, t, sp: butterfly_steps(b[0], b[1], None, t, b[2]))
    # ---- stores
    tid_offsets()
    emit("s_mov_b64 s[92:93], %[pout]")
    emit("s_nop 4")
    for j in range(6):
        for k in range(R):
            emit("global_store_dwordx2 v%d, %s, s[92:93]" % (OFF + k, pair(loc[(j, k)])))
        if j < 5:
            emit("s_add_u32 s92, s92, %d" % sB)
            emit("s_addc_u32 s93, s93, 0")
            emit("s_nop 4")
    emit("s_waitcnt vmcnt(0)")


def gen_cols_ci(S1, inverse):
    """Conjugate-invariant ring (ring/ntt.go:716-1311): the fold fused with the column stages.  The fold couples coefficient jx with N - jx,
    i.e. element k of column c with element R-1-k of column 4096 - c: a thread owns BOTH columns (c = 256 g + t + 1 and 4096 - c, g < 8),
    A[k] in v[2k], B[k] in v[2 (R + k)].  Forward: va = a + 4q - F b, vb = b + 4q - F a (ci_fold_kernel, engine.hip), then the column
    stages of gen_cols on both columns.  Inverse: the stages of gen_cols_inv (N^-1 folded in, canonical), then the fold with canonical
    outputs.  Column 0 (its own mirror) is a separate small kernel.  Operands: tid, pina / pinb (bytes: row base + (256 g + 1) columns,
    row base + 256 (15 - g) columns), pouta / poutb, tw, nq0, nq1, nq, nq2, nq4, q4, fw0 fw1 fp0 fp1 (the fold twiddle, Shoup pair),
    inverse also iw0 iw1 ip0 ip1 lw0 lw1 lp0 lp1."""
    R = 1 << S1
    A = lambda k: 2 * k
    B = lambda k: 2 * (R + k)
    base = 4 * R
    Ta, Tb = Tmp(base), Tmp(base + NTMP)
    Ta.cc, Tb.cc = "s[96:97]", "s[98:99]"
    E = (base + 2 * NTMP, base + 2 * NTMP + 2, base + 2 * NTMP + 4, base + 2 * NTMP + 6)
    OA, OB = base, base + R                                     # 2R <= 32 offset registers in the temp area while loads / stores issue
    F = ("%[fw0]", "%[fw1]", "%[fp0]", "%[fp1]")

    def offsets():
        emit("v_lshlrev_b32 v%d, 3, %%[tid]" % OA)
        emit("v_sub_u32 v%d, 2040, v%d" % (OB, OA))            # (255 - t) * 8
        for k in range(1, R):
            emit("v_add_u32 v%d, %d, v%d" % (OA + k, 32768 * k, OA))
            emit("v_add_u32 v%d, %d, v%d" % (OB + k, 32768 * k, OB))

    def sub_steps(x, y, t):
        return ["v_sub_co_u32 v%d, %s, v%d, v%d" % (x, t.cc, x, y), "@CARRY",
                "v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (x + 1, t.cc, x + 1, y + 1, t.cc)]

    def emit_pairs(items, fn):
        for i in range(0, len(items), 2):
            a = fn(items[i], Ta, (E[0], E[1]))
            if i + 1 < len(items):
                for ins in interleave(a, fn(items[i + 1], Tb, (E[2], E[3]))):
                    emit(ins)
            else:
                for ins in single(a):
                    emit(ins)

    def fold(item, t, sp):
        a, b = A(item), B(R - 1 - item)
        ea, eb = sp
        ins = [] if inverse else csub_steps(a, "nq4", t) + csub_steps(b, "nq4", t)
        ins += ["v_mov_b32 v%d, v%d" % (ea, a), "v_mov_b32 v%d, v%d" % (ea + 1, a + 1),
                "v_mov_b32 v%d, v%d" % (eb, b), "v_mov_b32 v%d, v%d" % (eb + 1, b + 1)]
        ins += shoup_mul_steps(ea, F, t) + shoup_mul_steps(eb, F, t)
        ins += ["v_lshl_add_u64 %s, %s, 0, %%[q4]" % (pair(a), pair(a))] + sub_steps(a, eb, t)
        ins += ["v_lshl_add_u64 %s, %s, 0, %%[q4]" % (pair(b), pair(b))] + sub_steps(b, ea, t)
        if inverse:
            for x in (a, b):
                ins += csub_steps(x, "nq4", t) + csub_steps(x, "nq2", t) + csub_steps(x, "nq", t)
        return ins

    if inverse:
        for slot in range(1, R - 1):
            emit("s_load_dwordx4 s[%d:%d], %%[tw], %d" % (36 + 4 * slot, 39 + 4 * slot, 16 * (slot + 1)))
    else:
        for slot in range(R - 1):
            emit("s_load_dwordx4 s[%d:%d], %%[tw], %d" % (36 + 4 * slot, 39 + 4 * slot, 16 * (slot + 1)))
    offsets()
    for k in range(R):
        emit("global_load_dwordx2 %s, v%d, %%[pina]" % (pair(A(k)), OA + k))
        emit("global_load_dwordx2 %s, v%d, %%[pinb]" % (pair(B(R - 1 - k)), OB + R - 1 - k))
    for t in (Ta, Tb):
        emit("v_mov_b32 v%d, 0" % (t.H + 1))
        emit("v_mov_b32 v%d, 0" % (t.G + 1))
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    sgq = lambda slot: ("s%d" % (36 + 4 * slot), "s%d" % (37 + 4 * slot), "s%d" % (38 + 4 * slot), "s%d" % (39 + 4 * slot))
    if not inverse:
        emit_pairs(list(range(R)), fold)
        for st in range(S1):
            h = R >> (st + 1)
            bfs = []
            for g in range(1 << st):
                sg = sgq((1 << st) - 1 + g)
                for e in range(h):
                    for X_ in (A, B):                           # the two columns alternate on the two temp sets
                        bfs.append((X_(g * 2 * h + e), X_(g * 2 * h + e + h), sg))
            emit_pairs(bfs, lambda b, t, sp: butterfly_steps(b[0], b[1], None, t, b[2]))
    else:
        for u in range(S1 - 1, 0, -1):
            h = (R >> 1) >> u
            bfs = []
            for g in range(1 << u):
                sg = sgq((1 << u) - 1 + g)
                for e in range(h):
                    for X_ in (A, B):
                        bfs.append((X_(g * 2 * h + e), X_(g * 2 * h + e + h), sg))
            emit_pairs(bfs, lambda b, t, sp: inv_butterfly_steps(b[0], b[1], None, t, b[2]))
        last = []
        for e in range(R >> 1):
            for X_ in (A, B):
                last.append((X_(e), X_(e + (R >> 1))))
        emit_pairs(last, lambda b, t, sp: scaled_last_butterfly_steps(b[0], b[1], t))
        emit_pairs(list(range(R)), fold)
    offsets()
    for k in range(R):
        emit("global_store_dwordx2 v%d, %s, %%[pouta]" % (OA + k, pair(A(k))))
        emit("global_store_dwordx2 v%d, %s, %%[poutb]" % (OB + k, pair(B(k))))
    emit("s_waitcnt vmcnt(0)")


def render(name, lines):
    body = "\n".join('  "%s\\n\\t"' % l for l in lines)
    return "#define %s \\\n%s\n" % (name, body.replace("\n", " \\\n"))


bodies_text = ""
for _suffix, _flags in (("", ""), ("_NT", " nt")):
    DATA_FLAGS = _flags
    del out[:]
    gen()
    fwd = list(out)
    if EXP:
        keep, seen_epi = [], False
        for l in fwd:
            if "canonical reduction" in l:
                seen_epi = True
            if (EXP & 1) and (l.startswith("ds_") or l == "s_barrier"):
                continue
            if (EXP & 8) and l == "s_barrier":
                continue
            if (EXP & 2) and l.startswith("global_load_dwordx4"):
                continue
            if (EXP & 4) and seen_epi and (l.startswith("v_lshl_add_u64") or l.startswith("v_ashrrev") or l.startswith("v_bfi")):
                continue
            keep.append(l)
        fwd = keep
    del out[:]
    cols, cols_inv, cols_exp = {}, {}, {}
    for s1 in (2, 3, 4):
        gen_cols(s1)
        cols[s1] = list(out)
        del out[:]
        gen_cols_inv(s1)
        cols_inv[s1] = list(out)
        del out[:]
        gen_cols(s1, expand=True)
        cols_exp[s1] = list(out)
        del out[:]
    gen(lazy_out=True)
    fwd_lazy = list(out)
    del out[:]
    gen(epilogue="submul")
    fwd_sm = list(out)
    del out[:]
    gen(epilogue="submul_add")
    fwd_sma = list(out)
    del out[:]
    gen_inverse()
    inv = list(out)
    del out[:]
    gen_inverse(mul=True)
    inv_mul = list(out)
    del out[:]
    gen_polymul()
    polymul = list(out)
    del out[:]
    gen(lds_in=True, tail_wait=False)                     # (the persistent one-pass kernel goes on to its next row while these stores drain)
    fwd_ldsin = list(out)
    del out[:]
    gen_inverse(lds_out=True)
    inv_ldsout = list(out)
    bodies_text += render("NTT_TILE_LDSIN_ASM_BODY" + _suffix, fwd_ldsin) + render("NTT_TILE_INV_LDSOUT_ASM_BODY" + _suffix, inv_ldsout)
    bodies_text += render("NTT_TILE_POLYMUL_ASM_BODY" + _suffix, polymul)
    bodies_text += (render("NTT_TILE_ASM_BODY" + _suffix, fwd) + render("NTT_TILE_LAZY_ASM_BODY" + _suffix, fwd_lazy) + render("NTT_TILE_SUBMUL_ASM_BODY" + _suffix, fwd_sm)
                    + render("NTT_TILE_SUBMUL_ADD_ASM_BODY" + _suffix, fwd_sma)
                    + "".join(render("NTT_COLS%d_ASM_BODY%s" % (1 << k, _suffix), cols[k]) + render("NTT_COLS%d_INV_ASM_BODY%s" % (1 << k, _suffix), cols_inv[k])
                              + render("NTT_COLS%d_EXPAND_ASM_BODY%s" % (1 << k, _suffix), cols_exp[k]) for k in (2, 3, 4))
                    + render("NTT_TILE_INV_ASM_BODY" + _suffix, inv) + render("NTT_TILE_INV_MUL_ASM_BODY" + _suffix, inv_mul))
DATA_FLAGS = ""
clob_v = ", ".join('"v%d"' % i for i in range(NVGPR_USED))
clob_s = ", ".join('"s%d"' % i for i in range(36, 100))
text = "// GENERATED by tools/gen_tile_asm.py -- do not edit.  forward: %d instructions, inverse: %d; NAME_NT = the same body with non-temporal data streams.\n" % (len(fwd), len(inv))
text += bodies_text
text += "#define NTT_TILE_ASM_CLOBBERS %s, %s, \"vcc\", \"scc\", \"memory\"\n" % (clob_v, clob_s)
text += "#define NTT_TILE_POLYMUL_ASM_CLOBBERS %s, %s, \"vcc\", \"scc\", \"memory\"\n" % (", ".join('"v%d"' % i for i in range(NVGPR_POLYMUL)), clob_s)
path = sys.argv[1] if len(sys.argv) > 1 else "ntt_tile_asm.inc"
open(path, "w").write(text)
text3 = "// GENERATED by tools/gen_tile_asm.py -- do not edit.  3N transform: column stages fused with the radix-3 layer and the split / merge, both directions.\n"
for _suffix, _flags in (("", ""), ("_NT", " nt")):           # both cache policies, as for the power-of-two bodies
    DATA_FLAGS = _flags
    for s1 in (1, 2, 3):
        del out[:]
        gen_3n_cols_post_inv(s1)
        text3 += render("NTT3N_COLS_POST_INV%d_ASM_BODY%s" % (1 << s1, _suffix), list(out))
        del out[:]
        gen_3n_pre_cols_fwd(s1)
        text3 += render("NTT3N_PRE_COLS_FWD%d_ASM_BODY%s" % (1 << s1, _suffix), list(out))
DATA_FLAGS = ""
text3 += "#define NTT3N_ASM_CLOBBERS %s, %s, \"vcc\", \"scc\", \"memory\"\n" % (", ".join('"v%d"' % i for i in range(128)), clob_s)
open(os.path.join(os.path.dirname(path), "ntt3n_asm.inc"), "w").write(text3)
textc = "// GENERATED by tools/gen_tile_asm.py -- do not edit.  Conjugate-invariant ring: fold fused with the column stages, both directions.\n"
for s1 in (2, 3, 4):
    for ci_inv in (False, True):
        del out[:]
        gen_cols_ci(s1, ci_inv)
        textc += render("NTT_CI_COLS%d_%s_ASM_BODY" % (1 << s1, "INV" if ci_inv else "FWD"), list(out))
open(os.path.join(os.path.dirname(path), "ntt_ci_asm.inc"), "w").write(textc)
print("wrote", path, "forward:", len(fwd), "VALU", sum(1 for l in fwd if l.startswith("v_")), "| inverse:", len(inv), "VALU", sum(1 for l in inv if l.startswith("v_")))
