/*
 * ringhip.h -- C ABI of the MI355X-native full-RNS polynomial-ring engine.
 *
 * Drop-in boundary for the `ring` hot path of swanhong/matrix-fhe-lattigo (paths below are relative to the reference
 * root).  Plain pointers and sizes only; no C++/torch types.  Every entry point returns 0 on success or a negative
 * rh_status; rh_last_error() returns a thread-local message.  Nothing aborts or throws across this boundary: the Go
 * wrapper turns a non-zero status into the same panic/error the reference raises (ring/ntt.go:212-214,
 * ring/ring.go:321-331).  The cgo binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Data model.  A "limb" is N uint64 residues; device polynomials are (poly, limb, coefficient) contiguous blocks:
 * word index ((poly*L)+limb)*N + j, with L = level+1 limbs per poly for the level a call names (the *_rows entry points take
 * blocks with more limbs per poly: views at a lower level of max-level polys).  This replaces Poly.Coeffs [][]uint64 (ring/poly.go:13-24) for device-resident
 * data; host-pointer entry points take one limb at a time exactly like the NumberTheoreticTransformer interface.
 */
#ifndef RINGHIP_H
#define RINGHIP_H
#include <stddef.h>
#include <stdint.h>
#include "ringhip_ops.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef enum rh_status {
  RH_OK = 0,
  RH_ERR_ARG = -1,        /* bad argument (short slice / bad level / N not supported) -> Go side panics like ntt.go:212 */
  RH_ERR_MODULUS = -2,    /* modulus not prime / not 1 mod NthRoot (ring/subring.go:139-145)                          */
  RH_ERR_DEVICE = -3,     /* HIP runtime error                                                                        */
  RH_ERR_NOMEM = -4,
  RH_ERR_UNSUPPORTED = -5
} rh_status;

typedef enum rh_ring_kind {
  RH_RING_STANDARD = 0,   /* Z_q[X]/(X^N+1), NumberTheoreticTransformerStandard (ring/ntt.go:31-78); N = 2^k, 8 <= N <=
                             2^17 (MinimumRingDegreeForLoopUnrolledOperations = 8, ring/ring.go:21-23, :318); for N < 16
                             BackwardLazy returns the reference's non-canonical values in [0, 2q) (ring/ntt.go:197-202) */
  RH_RING_CI = 1,         /* Z_q[X+X^-1]/(X^2N+1), NumberTheoreticTransformerConjugateInvariant (ring/ntt.go:80-124,
                             716-1311); NthRoot = 4N: root tables have 2N entries per limb; ForwardLazy returns the
                             canonical residues (inside the documented range, congruent)                             */
  RH_RING_3N = 2          /* Z_q[X]/(X^N-X^(N/2)+1), NumberTheoreticTransformer3N (ring/ntt_3n.go:21-156)             */
} rh_ring_kind;

typedef struct rh_ring rh_ring;   /* replaces ring.Ring + its SubRings' NTT state (ring/ring.go:76-89, subring.go:35-55) */

const char* rh_last_error(void);
int rh_device_count(void);

/* ---- ring construction -------------------------------------------------------------------------------------------
 * rh_ring_create: the constants handoff of NewRingWithCustomNTT / NewSubRingWithCustomNTT (ring/ring.go:314-356,
 * ring/subring.go:74-111, generateNTTConstants :129-214).  The engine never re-derives roots: it receives what the Go
 * side generated, so it is bit-identical to that Ring, including the random omega of a 3N ring.
 *   moduli[L], mred[L] (MRedConstant), bred[2L] (BRedConstant hi,lo), ninv[L] (NTTTable.NInv, Montgomery form),
 *   roots_fwd/roots_bwd: L tables of N words (NTTTable.RootsForward/RootsBackward), STANDARD rings only,
 *   omega3n[L]: primitive 3N-th root per limb (NumberTheoreticTransformer3N.psi3N), 3N rings only.            */
int rh_ring_create(rh_ring** out, int device, int kind, int N, int L, const uint64_t* moduli, const uint64_t* mred,
                   const uint64_t* bred, const uint64_t* ninv, const uint64_t* roots_fwd, const uint64_t* roots_bwd,
                   const uint64_t* omega3n);
/* rh_ring_create_auto: NewRing(N, moduli) (ring/ring.go:264-272): generates every constant on the host with the
 * reference's rules (smallest primitive root g >= 3, psi = g^((q-1)/2N), bit-reversed Montgomery tables).  For a 3N
 * ring omega = g^((q-1)/3N) (Find3NPrimitiveRoot, ring/subring.go:255-290) unless omega3n != NULL.               */
int rh_ring_create_auto(rh_ring** out, int device, int kind, int N, int L, const uint64_t* moduli,
                        const uint64_t* omega3n);
void rh_ring_destroy(rh_ring* r);
int rh_ring_n(const rh_ring* r);
int rh_ring_limbs(const rh_ring* r);
/* copies the per-limb constants back (any pointer may be NULL): what SubRing exposes (Modulus, MRedConstant,
 * BRedConstant, NInv, RootsForward, RootsBackward / omega) */
int rh_ring_get_constants(const rh_ring* r, uint64_t* moduli, uint64_t* mred, uint64_t* bred, uint64_t* ninv,
                          uint64_t* roots_fwd, uint64_t* roots_bwd, uint64_t* omega3n);
/* all device work of this ring is enqueued on `hip_stream` (a hipStream_t; NULL = the null stream) */
int rh_ring_set_stream(rh_ring* r, void* hip_stream);
int rh_ring_sync(rh_ring* r);
/* Optional: pre-sizes every lazily grown scratch of the ring (rescale, 3N transform) for batches of up to npoly polys of all
 * limbs, so that no later call allocates (hipMalloc / hipFree synchronise the device and cannot be captured in a HIP graph). */
int rh_ring_reserve(rh_ring* r, int npoly);

/* ---- concurrency (ring/ring.go:192-194: transformers are immutable, AtLevel views are concurrency-safe) ------------------
 * A ring handle may be used by any number of OS threads at once (goroutines migrate between threads):
 *   - rh_ntt_forward/_lazy/backward/_lazy: every call runs on a (stream, scratch) slot of its own taken from a pool in the
 *     handle; calls on the same or different limbs proceed in parallel and never share device memory;
 *   - the device-batched entry points only ENQUEUE on the ring's stream: concurrent callers are serialised in stream order;
 *     the few that use lazily built shared state (rescale, the 3N transform's workspace) take a lock for the enqueue;
 *   - tables and twiddles are read-only after creation; rh_ring_set_stream / rh_ring_set_tuning configure the handle and are
 *     not meant to race with calls.
 * rh_bext / rh_kshard objects carry scratch and plans like the reference's BasisExtender (one ShallowCopy per goroutine,
 * ring/basis_extension.go:166-183): one object per thread is the intended use; their entry points still lock the object,
 * so sharing one is safe, merely serial.  Several such objects may share the same rings.
 * rh_last_error() is thread-local: read it on the thread that got the status (the cgo wrapper locks the OS thread). */

/* ---- device memory (plain hipMalloc'd words; any device pointer from another allocator, e.g. torch, is accepted) */
int rh_dev_alloc(rh_ring* r, size_t words, uint64_t** dptr);
int rh_dev_free(rh_ring* r, uint64_t* dptr);
int rh_dev_upload(rh_ring* r, uint64_t* dst_dev, const uint64_t* src_host, size_t words);
int rh_dev_download(rh_ring* r, uint64_t* dst_host, const uint64_t* src_dev, size_t words);
/* Poly.CopyLvl (ring/poly.go) on device blocks: limbs 0..level of every poly, blocks with src_rows / dst_rows >= level+1 limbs per poly;
 * asynchronous on the ring's stream */
int rh_ring_copy_rows(rh_ring* r, uint64_t* dst_dev, int dst_rows, const uint64_t* src_dev, int src_rows, int npoly, int level);

/* ---- NumberTheoreticTransformer interface, one limb, host pointers (ring/ntt.go:17-22; SubRing.NTT/NTTLazy/INTT/
 * INTTLazy ring/subring_ops.go:235-252).  p1 and p2 hold N words each and may alias.  Synchronous.
 *   Forward      -> canonical [0,q)                       (NTTStandard, ntt.go:174-177 / 3N Forward ntt_3n.go:82-109)
 *   ForwardLazy  -> exactly the reference's lazy representatives in [0,6q-2] (NTTStandardLazy :180-182)
 *   Backward, BackwardLazy -> canonical [0,q)             (INTTStandard :185-194, INTTStandardLazy :197-206, N>=16) */
int rh_ntt_forward(rh_ring* r, int limb, const uint64_t* p1, uint64_t* p2);
int rh_ntt_forward_lazy(rh_ring* r, int limb, const uint64_t* p1, uint64_t* p2);
int rh_ntt_backward(rh_ring* r, int limb, const uint64_t* p1, uint64_t* p2);
int rh_ntt_backward_lazy(rh_ring* r, int limb, const uint64_t* p1, uint64_t* p2);

/* ---- Ring.NTT / NTTLazy / INTT / INTTLazy on a whole host Poly in ONE call (ring/ntt.go:127-152: the loop over
 * r.SubRings[:level+1] calling s.NTT(p1.Coeffs[i], p2.Coeffs[i]); Poly.Coeffs is [][]uint64, ring/poly.go:13-24).
 * p1 / p2: host arrays of level+1 limb pointers (N words each; p2[i] may alias p1[i]); limb i uses modulus i.  Same results as level+1
 * calls of rh_ntt_forward / _lazy / backward / _lazy, but pipelined: up to four limb groups alternate between two streams (upload of
 * group g under the transform and download of group g-1), one batched launch pair per group, one host synchronisation.  Limbs in
 * page-locked memory (rh_host_alloc / rh_host_register) are DMA'd where they lie; pageable limbs are staged through page-locked buffers
 * owned by the handle.  Synchronous; any number of OS threads may call it on one handle (each call takes a slot of its own).
 * The cgo wrapper pins the Go slices for the call (runtime.Pinner) and passes a C array of their data pointers. */
int rh_ntt_poly_forward(rh_ring* r, int level, const uint64_t* const* p1, uint64_t* const* p2, int lazy);
int rh_ntt_poly_backward(rh_ring* r, int level, const uint64_t* const* p1, uint64_t* const* p2, int lazy);
/* Page-locked host memory for Poly.Coeffs backing arrays (a Go slice over it: unsafe.Slice): such limbs skip the staging copy.
 * rh_host_register page-locks memory the caller already owns (it must stay allocated and unmoved until rh_host_unregister).  WHOLE 4 KiB PAGES only
 * (pointer and byte count multiples of 4096, else RH_ERR_ARG), pages
 * the caller owns exclusively (a page-aligned mapping or allocation: a Go slice of a megabyte or more sits in spans of its own): the runtime pins and
 * maps every page the range touches, and a page shared with unrelated heap objects drags those into the GPU mapping for the registration's lifetime. */
int rh_host_alloc(size_t words, uint64_t** hptr);
int rh_host_free(uint64_t* hptr);
int rh_host_register(uint64_t* hptr, size_t words);
int rh_host_unregister(uint64_t* hptr);

/* ---- Ring.NTT / NTTLazy / INTT / INTTLazy on device-resident batches (ring/ntt.go:127-152).
 * in/out: npoly polys of (level+1) limbs each, limbs 0..level of the ring (Ring.AtLevel view); may alias.
 * Asynchronous on the ring's stream. */
int rh_ring_ntt(rh_ring* r, const uint64_t* in_dev, uint64_t* out_dev, int npoly, int level, int lazy);
int rh_ring_intt(rh_ring* r, const uint64_t* in_dev, uint64_t* out_dev, int npoly, int level, int lazy);

/* Ring.NTT on several blocks in one call (device-API extension: e.g. both operands of a product, schemes/ckks/evaluator.go:821-834
 * after the two NTTs).  in[k] / out[k]: block k, npoly[k] polys of (level+1) limbs; each pair may alias.  Same results as nblocks
 * rh_ring_ntt calls; for two-pass standard rings (N >= 8192) the software pipeline of the fused launches runs through the block
 * boundaries.  The pointer arrays are host arrays of device pointers, read before the call returns. */
int rh_ring_ntt_many(rh_ring* r, const uint64_t* const* in_dev, uint64_t* const* out_dev, const int* npoly, int nblocks, int level);

/* The same on blocks that carry MORE limbs per poly than the level they are used at (in_rows / out_rows limbs per poly, each
 * >= level+1): ring.AtLevel(level) on max-level polys and buffers (ring/ring.go:192-213), the idiomatic use inside the reference's
 * evaluators.  Limbs 0..level of every poly are transformed, the others are untouched.  With rows == level+1 this IS rh_ring_ntt /
 * rh_ring_intt. */
/* (round 3: every shape is ONE batched transform.  Standard rings take both row strides inside the kernels (every N, forward and inverse);
 * conjugate-invariant and 3N rings compact the leading limbs with one strided device copy on the way in and / or out.  rh_ring_stats(ring, "rows_direct" | "rows_compacted" |
 * "rows_poly_by_poly", &n) counts how the calls of a handle were served.) */
int rh_ring_stats(const rh_ring* r, const char* key, long* value);
int rh_ring_ntt_rows(rh_ring* r, const uint64_t* in_dev, int in_rows, uint64_t* out_dev, int out_rows, int npoly, int level, int lazy);
int rh_ring_intt_rows(rh_ring* r, const uint64_t* in_dev, int in_rows, uint64_t* out_dev, int out_rows, int npoly, int level, int lazy);

/* INTT of a pointwise product, NTT-domain inputs: out = INTT(a . b), the values of ring.MForm(a, t); ring.MulCoeffsMontgomery(t, b, c);
 * ring.INTT(c, c) (the degree-0 part of ckks mulRelin, schemes/ckks/evaluator.go:821-834, and BASELINE config 3) -- canonical, hence
 * bit-identical -- with the product formed on load by the inverse transform's first kernel: 24 bytes per coefficient less traffic
 * than the three calls.  a, b: npoly polys of level+1 limbs, canonical or lazy (< 2q); out may alias either.  Standard rings. */
int rh_ring_intt_mul(rh_ring* r, const uint64_t* a_dev, const uint64_t* b_dev, uint64_t* out_dev, int npoly, int level);

/* The whole of BASELINE config 3 from COEFFICIENT-domain operands: out = INTT(NTT(a) . NTT(b)), the values of ring.NTT(a); ring.NTT(b);
 * ring.MForm; ring.MulCoeffsMontgomery; ring.INTT (schemes/ckks/evaluator.go:821-834 around a fresh product) -- canonical, hence bit-identical.
 * The forward tile stages of both operands, their product and the inverse tile stages run as ONE kernel per 4096-coefficient tile: NTT(a), NTT(b)
 * and the product never reach memory (72 instead of 104 bytes per coefficient).  a and b are CONSUMED: they are transformed in place as far as
 * their column stages and do not hold NTT(a) / NTT(b) afterwards (a caller that needs those keeps rh_ring_ntt_many + rh_ring_intt_mul).  out may
 * alias a or b; a != b.  Standard rings, 2^13 <= N <= 2^17 (otherwise RH_ERR_UNSUPPORTED). */
int rh_ring_polymul(rh_ring* r, uint64_t* a_dev, uint64_t* b_dev, uint64_t* out_dev, int npoly, int level);

/* 3N rings: NTT-domain blocks between the reference's ascending-totative order and block order (see ntt3n_block_order):
 * to_reference = 1: block order -> reference order, 0: the reverse.  Out of place (in != out). */
int rh_ring_ntt3n_reorder(rh_ring* r, const uint64_t* in_dev, uint64_t* out_dev, int npoly, int level, int to_reference);

/* 3N rings, for hosts that TAG their device polys with the NTT-domain layout (matrix-fhe-lattigo_amd/ringhip.py DevicePoly.layout, the Go
 * wrapper's DevicePoly.BlockOrder) instead of switching the whole handle with the tuning key: the layout of the NTT side of the call is an
 * argument (block_order 1: block order, 0: the reference's order), so concurrent callers of one handle may differ.  rows as rh_ring_ntt_rows;
 * canonical outputs.  rh_ring_ntt3n_block_order_supported: 1 for 3N rings with N = 3 * 2^k, k >= 13.  Other ring kinds ignore the argument. */
int rh_ring_ntt3n_block_order_supported(const rh_ring* r);
int rh_ring_ntt_layout(rh_ring* r, const uint64_t* in_dev, int in_rows, uint64_t* out_dev, int out_rows, int npoly, int level, int inverse, int block_order);
int rh_ring_div_by_last_modulus_many_ntt_layout(rh_ring* r, int round, int level, int nb, const uint64_t* p0_dev, uint64_t* p1_dev, int p1_rows, int npoly,
                                                int block_order);

/* profiling aid: phase 0 = whole transform, 1 = column kernel only, 2 = tile kernel only (N >= 8192) */
int rh_ring_ntt_phase(rh_ring* r, const uint64_t* in_dev, uint64_t* out_dev, int npoly, int level, int inverse, int phase);
/* Tuning knobs: performance only, never results (each non-default setting is covered by a parity test).  Unknown keys
 * return RH_ERR_ARG.  Defaults are the measured optimum on MI355X.  Set them before the handle is
 * shared between threads: they are plain fields read by every call.
 *   chunk_polys     polys per span of the fused (column + tile) pipeline: -1 auto (auto_span_rows), 0 = two launches per batch
 *   auto_span_rows  span size of the auto rule in (poly, limb) rows (2048)
 *   asm_tile / asm_cols   1: generated hand-scheduled bodies (default), 0: the C++ kernels
 *   fuse_submul     1: ModDown / rescale subtract-multiply in the forward tile kernel's epilogue (default)
 *   ks_small_rows   n: key switches whose blocks have at most n (poly, limb) rows (up to ~20 ciphertexts at 24 limbs) run every digit of the decomposition in ONE
 *                   basis-extension launch and the transforms of all digit blocks of BOTH rings in one launch pair instead of digit by digit (default 512; 0: never).
 *                   One ciphertext at N = 2^16, 24 + 6 limbs: 0.38 -> 0.24 ms; eight: 1.26 -> 1.15 ms; from ~24 ciphertexts on the pipelined stream of launches
 *                   is as fast or faster (set on ringQ)
 *   one_pass        1: N = 2^13 / 2^14 (and the inverse at N = 4096) keep the whole limb row in one workgroup's LDS between the column and the tile
 *                   stages: one HBM pass per transform instead of two (default); 0: the two-pass launches
 *   fuse_ci         1: conjugate-invariant rings fold inside the column stages instead of in a pass of their own (default; N = 2^14..2^16)
 *   digit_pipeline  1: key switch transforms all digit blocks with one software-pipelined stream of launches (default; N = 2^14..2^16)
 *   fuse3n          1: 3N rings, split + radix-3 layer fused with the sub-transforms' column stages (default)
 *   perm_fwd_shape / perm_inv_shape  3N permutation tile as 10*A + B: block-order runs of 2^A words, rank-order runs of nb * 2^B words
 * One key changes a LAYOUT, not a value (3N rings, N = 3*2^k >= 24576, default 0):
 *   ntt3n_block_order  1: rh_ring_ntt / rh_ring_intt keep the NTT domain in "block order" (slot j of block c of the radix-2
 *                   sub-transforms at word c*N/6 + j) instead of the Go transformer's ascending-totative order
 *                   (ring/ntt_3n.go:82-109, :235-243).  Every NTT-domain operation of a ring is coefficient-wise, so NTT ->
 *                   pointwise -> INTT chains (matrix_ckks.Evaluator.Mul) give the same coefficient-domain bits with one pass less
 *                   per transform; NTT-domain data that crosses the host boundary is converted with rh_ring_ntt3n_reorder.  The
 *                   per-limb host interface (rh_ntt_*) always speaks the reference order. */
int rh_ring_set_tuning(rh_ring* r, const char* key, long value);

/* ---- element-wise family (ring/vec_ops.go via ring/operations.go loops): p3 = op(p1, p2 [, p3]) on npoly polys of
 * (level+1) limbs.  s0/s1: per-limb scalar arrays of level+1 words on the HOST (NULL = unused), e.g. the RNSScalar of
 * MulRNSScalarMontgomery (ring/operations.go).  Asynchronous. */
int rh_ring_vec_op(rh_ring* r, int opcode, const uint64_t* p1_dev, const uint64_t* p2_dev, uint64_t* p3_dev, int npoly,
                   int level, const uint64_t* s0_host, const uint64_t* s1_host);

/* rows-per-poly form (see rh_ring_ntt_rows): every operand block with its own limb count >= level+1; one launch, the three row
 * strides inside the kernel */
int rh_ring_vec_op_rows(rh_ring* r, int opcode, const uint64_t* p1_dev, int rows1, const uint64_t* p2_dev, int rows2, uint64_t* p3_dev,
                        int rows3, int npoly, int level, const uint64_t* s0_host, const uint64_t* s1_host);

/* p2 = ONE row of N words on the device used for every (poly, limb): Ring.MulByVectorMontgomery / ...ThenAddLazy (ring/operations.go:366-377)
 * with RH_OP_MUL_MONT / RH_OP_MUL_MONT_THEN_ADD_LAZY (any opcode with a second operand) */
int rh_ring_vec_op_bcast(rh_ring* r, int opcode, const uint64_t* p1_dev, int rows1, const uint64_t* vector_dev, uint64_t* p3_dev, int rows3,
                         int npoly, int level);
/* a one-operand scalar opcode with one RNS scalar for coefficients [0, N/2) and another for [N/2, N): Ring.AddDoubleRNSScalar,
 * SubDoubleRNSScalar, MulDoubleRNSScalar(ThenAdd) (ring/operations.go:167-184, 250-266); scalars as the opcode takes them */
int rh_ring_vec_op_halves(rh_ring* r, int opcode, const uint64_t* p1_dev, uint64_t* p3_dev, int npoly, int level,
                          const uint64_t* s_lo_host, const uint64_t* s_hi_host);
/* Ring.Shift (:278-282): p2[j] = p1[(j + k) mod N]; Ring.MultByMonomial (:306-363): p2 = p1 * X^k with the reference's representatives
 * (q - 0 is written as q).  Every limb 0..level of dense blocks; out of place. */
int rh_ring_shift(rh_ring* r, int level, const uint64_t* in_dev, uint64_t* out_dev, int k, int npoly);
int rh_ring_mult_by_monomial(rh_ring* r, int level, const uint64_t* in_dev, uint64_t* out_dev, int k, int npoly);
/* Ring.AutomorphismNTTWithIndex / AutomorphismNTTWithIndexThenAddLazy (ring/automorphism.go:50-117): out[j] (=|+=) in[index[j]] on every limb
 * 0..level; index_dev: the caller's lookup table of N words ON THE DEVICE (e.g. AutomorphismNTTIndex, :12-34); out of place */
int rh_ring_automorphism_ntt_index(rh_ring* r, int level, const uint64_t* in_dev, const uint64_t* index_dev, uint64_t* out_dev, int npoly,
                                   int add_lazy);
/* Standard <-> conjugate-invariant bridges (ring/conjugate_invariant.go; callers: schemes/ckks/bridge.go:82-83, 116-117,
 * core/rlwe/keygenerator.go:213).  Dense blocks, every limb 0..level, out of place; n = the SMALLER of the two degrees.
 *   rh_ring_unfold_ci_to_standard (:8-26)   r: the standard ring of degree 2n.  ci: rows of n words, std: rows of 2n words;
 *                                           std[j] = ci[j], std[n + k] = ci[n - 1 - k]
 *   rh_ring_fold_standard_to_ci (:31-49)    r: the conjugate-invariant ring of degree n.  std: rows of 2n words, ci: rows of n words,
 *                                           index_dev: n words on the device with values < 2n (AutomorphismNTTIndex of the standard ring);
 *                                           ci[j] = CRed(std[index[j]] + std[j])
 *   rh_ring_pad_default_to_ci (:52-80)      r: a ring of degree n (level and moduli).  std: rows of n words, ci: rows of 2n words of which only
 *                                           the first n are written, with the reference's in-place loop reproduced: is_ntt: ci[k] = std[k],
 *                                           ci[n-1-k] = std[k] for k < n/2;  else ci[0] = 0, ci[k] = std[k] (1 <= k < n/2),
 *                                           ci[n/2] = q - std[n/2], ci[n-k] = q - std[k] (1 <= k < n/2; q - 0 is written as q) */
int rh_ring_unfold_ci_to_standard(rh_ring* r, int level, const uint64_t* ci_dev, uint64_t* std_dev, int npoly);
int rh_ring_fold_standard_to_ci(rh_ring* r, int level, const uint64_t* std_dev, const uint64_t* index_dev, uint64_t* ci_dev, int npoly);
int rh_ring_pad_default_to_ci(rh_ring* r, int level, const uint64_t* std_dev, int is_ntt, uint64_t* ci_dev, int npoly);

/* ---- RNS rescale (ring/scaling.go): divide by the last modulus, `nb` times.  round = 0: floored, 1: rounded.
 * p0: npoly polys of level+1 limbs; p1: npoly polys of p1_rows >= level+1-nb limbs (limbs 0..level-nb are written).
 *   rh_ring_div_by_last_modulus_many      coefficient domain: DivFloorByLastModulus(:21-28) / DivRoundByLastModulus
 *                                          (:112-126) and their Many forms (:56-88, :160-192); p0 is modified in
 *                                          place exactly as the reference modifies its input / buffer
 *   rh_ring_div_by_last_modulus_many_ntt  NTT domain: DivFloorByLastModulusManyNTT(:32-52), DivRoundByLastModulusNTT
 *                                          (:92-108), DivRoundByLastModulusManyNTT(:130-156); p0 is not modified
 * All three ring types (the 3N ring is what schemes/matrix_ckks/evaluator.go:235 rescales on); the fused kernels are the standard ring's. */
int rh_ring_div_by_last_modulus_many(rh_ring* r, int round, int level, int nb, uint64_t* p0_dev, uint64_t* p1_dev, int p1_rows, int npoly);
int rh_ring_div_by_last_modulus_many_ntt(rh_ring* r, int round, int level, int nb, const uint64_t* p0_dev, uint64_t* p1_dev, int p1_rows, int npoly);

/* Degree-1 x degree-1 tensoring in one pass.  mform_first = 1: ckks mulRelin (schemes/ckks/evaluator.go:821-834),
 * c0 = MRed(MForm(a0), b0), c1 = CRed(MRed(MForm(a0), b1) + MRed(MForm(a1), b0)), c2 = MRed(MForm(a1), b1) -- the values the six
 * ring calls of the reference produce.  mform_first = 0: matrix_ckks.Evaluator.Mul (schemes/matrix_ckks/evaluator.go:166-173),
 * the same without the MForm (its four ring calls).  All blocks: npoly polys of level+1 limbs, NTT domain; any ring kind. */
int rh_ring_tensor_degree1(rh_ring* r, const uint64_t* a0_dev, const uint64_t* a1_dev, const uint64_t* b0_dev, const uint64_t* b1_dev,
                           uint64_t* c0_dev, uint64_t* c1_dev, uint64_t* c2_dev, int npoly, int level, int mform_first);

/* ---- Galois automorphisms X -> X^gen (ring/automorphism.go), standard and conjugate-invariant rings, never in place.
 *   conjugate-invariant rings: the NTT-domain table is built over NthRoot = 4N (gen must be 1 mod 4, else the reference's look-up
 *   runs out of range); the coefficient-domain form is the Z[X+X^-1] branch (:131-156)
 *   rh_ring_automorphism_ntt: AutomorphismNTT (:39-47) / AutomorphismNTTWithIndex (:52-81); add_lazy != 0:
 *                             AutomorphismNTTWithIndexThenAddLazy (:86-117) (out += permuted in, wrapping)
 *   rh_ring_automorphism:     Automorphism (:121-176, standard ring branch), coefficient domain with sign flips   */
int rh_ring_automorphism_ntt(rh_ring* r, int level, const uint64_t* in_dev, uint64_t gen, uint64_t* out_dev, int npoly, int add_lazy);
int rh_ring_automorphism(rh_ring* r, int level, const uint64_t* in_dev, uint64_t gen, uint64_t* out_dev, int npoly);

/* ---- RNS basis extension (ring/basis_extension.go).  A basis extender pairs a Q ring and a P ring
 * (NewBasisExtender :52-79).  All polys device-resident, limbs 0..levelQ / 0..levelP, npoly polys.  Asynchronous. */
typedef struct rh_bext rh_bext;
int rh_bext_create(rh_bext** out, rh_ring* ringQ, rh_ring* ringP);
void rh_bext_destroy(rh_bext* be);
/* Optional: pre-sizes the extender's scratch for key switches / ModDowns of up to npoly polys at the rings' top levels. */
int rh_bext_reserve(rh_bext* be, int npoly);
int rh_bext_modup_q_to_p(rh_bext* be, int levelQ, int levelP, const uint64_t* polQ, uint64_t* polP, int npoly);   /* :188-200 */
int rh_bext_modup_p_to_q(rh_bext* be, int levelP, int levelQ, const uint64_t* polP, uint64_t* polQ, int npoly);   /* :205-217 */
int rh_bext_moddown_qp_to_q(rh_bext* be, int levelQ, int levelP, const uint64_t* p1Q, const uint64_t* p1P,
                            uint64_t* p2Q, int npoly);                                                            /* :223-234 */
int rh_bext_moddown_qp_to_q_ntt(rh_bext* be, int levelQ, int levelP, const uint64_t* p1Q, const uint64_t* p1P,
                                uint64_t* p2Q, int npoly);                                                        /* :241-258 */
int rh_bext_moddown_qp_to_p(rh_bext* be, int levelQ, int levelP, const uint64_t* p1Q, const uint64_t* p1P,
                            uint64_t* p2P, int npoly);                                                            /* :264-278 */
/* Decomposer.DecomposeAndSplit (:381-502): digit `digit` of p0Q -> p1Q (levelQ+1 limbs), p1P (levelP+1 limbs) */
int rh_bext_decompose_and_split(rh_bext* be, int levelQ, int levelP, int nbPi, int digit, const uint64_t* p0Q,
                                uint64_t* p1Q, uint64_t* p1P, int npoly);


/* ---- hybrid key-switch gadget product: rlwe.Evaluator.GadgetProduct.  This entry: NTT-domain ciphertext, levelP >= 1
 * (core/rlwe/evaluator_gadget_product.go:16-30) = gadgetProductMultiplePLazy (:122-188) + ModDown NTT->NTT (:33-46).
 * cx: npoly polys of levelQ+1 limbs (NTT domain).  evkQ / evkP: GadgetCiphertext.Value[i][0][c].Q / .P
 * (core/rlwe/gadgetciphertext.go:17-45) laid out [digit i < beta_key][component c < 2][all limbs of the ring][N], NTT
 * domain, Montgomery form, shared by every poly of the batch.  ct0/ct1: npoly polys of levelQ+1 limbs (NTT domain). */
int rh_bext_gadget_product(rh_bext* be, int levelQ, int levelP, const uint64_t* cx_dev, const uint64_t* evkQ_dev,
                           const uint64_t* evkP_dev, int beta_key, uint64_t* ct0_dev, uint64_t* ct1_dev, int npoly);

/* The same with the ring.Add that follows it in Relinearize (core/rlwe/evaluator_evaluationkey.go:144-146), mulRelin
 * (schemes/ckks/evaluator.go:850-852), applyEvaluationKey (:105-112) and Automorphism (evaluator_automorphism.go:42-44):
 * ct_c = add_c + product_c, canonical.  add0 / add1 may be NULL and may alias ct0 / ct1. */
int rh_bext_gadget_product_then_add(rh_bext* be, int levelQ, int levelP, const uint64_t* cx_dev, const uint64_t* evkQ_dev,
                                    const uint64_t* evkP_dev, int beta_key, const uint64_t* add0_dev, const uint64_t* add1_dev,
                                    uint64_t* ct0_dev, uint64_t* ct1_dev, int npoly);

/* The same for a COEFFICIENT-domain ciphertext (ct.IsNTT == false; :114-118, :139-143 and ModDown INTT -> INTT :62-66): cx, ct0, ct1 in
 * the coefficient domain, levelP >= 1. */
int rh_bext_gadget_product_coeff(rh_bext* be, int levelQ, int levelP, const uint64_t* cx_dev, const uint64_t* evkQ_dev,
                                 const uint64_t* evkP_dev, int beta_key, uint64_t* ct0_dev, uint64_t* ct1_dev, int npoly);

/* Gadget ciphertexts with at most ONE P modulus, optionally with a power-of-two decomposition on top of the RNS one:
 * gadgetProductSinglePAndBitDecompLazy (core/rlwe/evaluator_gadget_product.go:190-324) + ModDown (:33-98).
 *   levelP = 0: one P modulus (be has a P ring);  levelP = -1: none (be created with ringP = NULL, evkP = NULL, needs pw2 > 0)
 *   pw2 = GadgetCiphertext.BaseTwoDecomposition (0: RNS digits only); digits_per_limb[i] = len(Value[i]) for i <= levelQ (host array,
 *   NULL when pw2 = 0); key rows: row e = (sum of digits_per_limb before limb i) + j holds Value[i][j]: evkQ / evkP are
 *   [row][component < 2][all limbs of the ring][N] like rh_bext_gadget_product, key_rows = number of rows supplied
 *   cx_is_ntt: the ciphertext's domain (cx, ct0 and ct1 alike), as ct.IsNTT in the reference. */
int rh_bext_gadget_product_single_p(rh_bext* be, int levelQ, int levelP, const uint64_t* cx_dev, int cx_is_ntt, int pw2,
                                    const int* digits_per_limb, const uint64_t* evkQ_dev, const uint64_t* evkP_dev, int key_rows,
                                    uint64_t* ct0_dev, uint64_t* ct1_dev, int npoly);
/* The same WITHOUT the closing ModDown / CopyLvl (the accumulators after the closing Reduce: canonical residues modulo Q in ctQ0 / ctQ1 and,
 * for levelP = 0, modulo P in ctP0 / ctP1; NTT domain), for callers that sum several products under one ModDown: rgsw's external product.
 * raw_limb_digits != 0: the digit form of externalProductInPlaceSinglePAndBitDecomp (core/rgsw/evaluator.go:119-186) -- MaskVec of limb i
 * under every modulus even when pw2 = 0 (mask = all ones), where rlwe's routine calls DecomposeAndSplit. */
int rh_bext_gadget_product_single_p_lazy(rh_bext* be, int levelQ, int levelP, const uint64_t* cx_dev, int cx_is_ntt, int pw2,
                                         const int* digits_per_limb, const uint64_t* evkQ_dev, const uint64_t* evkP_dev, int key_rows,
                                         int raw_limb_digits, uint64_t* ctQ0_dev, uint64_t* ctQ1_dev, uint64_t* ctP0_dev, uint64_t* ctP1_dev, int npoly);

/* Hoisted form (rotations of one ciphertext share the decomposition).  Evaluator.DecomposeNTT
 * (core/rlwe/evaluator_gadget_product.go:431-453): c2 (levelQ+1 limbs, NTT or coefficient domain per c2_is_ntt) ->
 * decompQ [beta][npoly][levelQ+1][N], decompP [beta][npoly][levelP+1][N], NTT domain, beta = BaseRNSDecompositionVectorSize. */
int rh_bext_decompose_ntt(rh_bext* be, int levelQ, int levelP, const uint64_t* c2_dev, int c2_is_ntt, uint64_t* decompQ_dev,
                          uint64_t* decompP_dev, int npoly);
/* Evaluator.GadgetProductHoisted (:326-349, :373-429) on such a decomposition; key and outputs as rh_bext_gadget_product */
int rh_bext_gadget_product_hoisted(rh_bext* be, int levelQ, int levelP, const uint64_t* decompQ_dev, const uint64_t* decompP_dev,
                                   const uint64_t* evkQ_dev, const uint64_t* evkP_dev, int beta_key, uint64_t* ct0_dev,
                                   uint64_t* ct1_dev, int npoly);

/* Evaluator.GadgetProductHoistedLazy (core/rlwe/evaluator_gadget_product.go:351-429): the product WITHOUT the ModDown -- accumulators
 * modulo Q (ctQ0 / ctQ1, levelQ+1 limbs) and modulo P (ctP0 / ctP1, levelP+1 limbs), canonical, NTT domain, still scaled by P: what
 * AutomorphismHoistedLazy and the linear transformations accumulate before ONE ModDown.
 * rh_bext_moddown_qp_to_q_ntt_pair: Evaluator.ModDown (:33-46) NTT -> NTT on both components (ct_c may alias ctQ_c). */
int rh_bext_gadget_product_hoisted_lazy(rh_bext* be, int levelQ, int levelP, const uint64_t* decompQ_dev, const uint64_t* decompP_dev,
                                        const uint64_t* evkQ_dev, const uint64_t* evkP_dev, int beta_key, uint64_t* ctQ0_dev,
                                        uint64_t* ctQ1_dev, uint64_t* ctP0_dev, uint64_t* ctP1_dev, int npoly);
int rh_bext_moddown_qp_to_q_ntt_pair(rh_bext* be, int levelQ, int levelP, const uint64_t* ctQ0_dev, const uint64_t* ctQ1_dev,
                                     const uint64_t* ctP0_dev, const uint64_t* ctP1_dev, uint64_t* ct0_dev, uint64_t* ct1_dev, int npoly);

/* the same with the ring.Add of AutomorphismHoisted (core/rlwe/evaluator_automorphism.go:88-89): ct_c = add_c + product_c */
int rh_bext_gadget_product_hoisted_then_add(rh_bext* be, int levelQ, int levelP, const uint64_t* decompQ_dev, const uint64_t* decompP_dev,
                                            const uint64_t* evkQ_dev, const uint64_t* evkP_dev, int beta_key, const uint64_t* add0_dev,
                                            const uint64_t* add1_dev, uint64_t* ct0_dev, uint64_t* ct1_dev, int npoly);

/* ---- limb-sharded hybrid key switch (BASELINE config 5): one process per GPU owns a subset of the limbs
 * of Q and P and the matching slice of the evaluation key.  Same arithmetic as rh_bext_gadget_product, cut where
 * reconstructRNS (ring/basis_extension.go:550-594) needs limbs of other owners; the exchange (an all-gather of the
 * digit's source limbs, and of the P part before ModDown) is the caller's -- RCCL through torch.distributed in
 * matrix-fhe-lattigo_amd/sharding.py -- and this library never communicates.
 * ringQ_loc / ringP_loc: rings over the OWNED moduli only, ascending global order (ringP_loc NULL iff nownP == 0);
 * allQ / allP: moduli 0..levelQ / 0..levelP of the full chain; ownQ / ownP: global indices of the owned limbs.
 * Local blocks are (poly, owned limb, N); gathered source blocks are (poly, source limb in global order, N). */
typedef struct rh_kshard rh_kshard;
int rh_kshard_create(rh_kshard** out, rh_ring* ringQ_loc, rh_ring* ringP_loc, const uint64_t* allQ, int levelQ,
                     const uint64_t* allP, int levelP, const int* ownQ, int nownQ, const int* ownP, int nownP);
void rh_kshard_destroy(rh_kshard* ks);
int rh_kshard_num_digits(const rh_kshard* ks);                             /* core/rlwe/params.go:635-642 */
int rh_kshard_digit_range(const rh_kshard* ks, int digit, int* st, int* ed); /* global limbs [st, ed) of the digit */
/* one digit of gadgetProductMultiplePLazy (core/rlwe/evaluator_gadget_product.go:154-187, DecomposeSingleNTT :455-478)
 * for the owned limbs.  src_dev: limbs [st, ed) of INTT(cx) gathered from their owners; cx_loc: owned limbs of the
 * NTT-domain input; evkQ_loc / evkP_loc: [digit][component < 2][owned limb][N].  Feed digits 0 .. beta-1 in order. */
int rh_kshard_digit(rh_kshard* ks, int digit, const uint64_t* src_dev, const uint64_t* cx_loc, const uint64_t* evkQ_loc,
                    const uint64_t* evkP_loc, uint64_t* ct0_loc, uint64_t* ct1_loc, uint64_t* accP0_loc, uint64_t* accP1_loc,
                    int npoly);

/* All digits in one call: src_all_dev = EVERY limb of INTT(cx), (npoly, levelQ + 1, N) in chain order (ONE all-gather of the owners'
 * limbs instead of one per digit); the other arguments as for rh_kshard_digit.  Same results as the beta rh_kshard_digit calls, with
 * the structure of the single-GPU product: one pipelined transform of all digit blocks, one multiply-accumulate over all digits. */
int rh_kshard_product(rh_kshard* ks, const uint64_t* src_all_dev, const uint64_t* cx_loc, const uint64_t* evkQ_loc,
                      const uint64_t* evkP_loc, uint64_t* ct0_loc, uint64_t* ct1_loc, uint64_t* accP0_loc, uint64_t* accP1_loc, int npoly);
/* ModDownQPtoQNTT (ring/basis_extension.go:241-258) for the owned Q limbs.  srcP_dev: (npoly, levelP+1, N), the
 * INTTLazy of the P part gathered from its owners. */
int rh_kshard_moddown(rh_kshard* ks, const uint64_t* srcP_dev, const uint64_t* ctQ_in_loc, uint64_t* ctQ_out_loc, int npoly);


/* The WHOLE limb-sharded product in one call, orchestration included: ringQ.INTT(cx) -> exchange of every limb of INTT(cx) ->
 * gadgetProductMultiplePLazy on the owned limbs -> INTTLazy of the owned P limbs -> exchange of the P part of both accumulators ->
 * ModDownQPtoQNTT on the owned Q limbs (core/rlwe/evaluator_gadget_product.go:16-30, 33-46, 122-188, 455-478).  The library still never
 * communicates: both exchanges are calls of `allgather`, which the HOST supplies -- ncclAllGather on the node's RCCL communicator from a cgo
 * host, torch.distributed from Python (sharding.LimbShardedKeySwitch).
 *   rh_allgather_fn(ctx, send_dev, recv_dev, send_words, hip_stream): every rank contributes send_words uint64 at send_dev and receives
 *       world * send_words at recv_dev, rank-major (the semantics of ncclAllGather), ENQUEUED on hip_stream; 0 on success.  All ranks make
 *       the same sequence of calls (two per chunk), so the collectives match up.
 *   rh_kshard_set_world: the owner of every limb of Q ++ P (owner[i], i <= levelQ: Q limb i; owner[levelQ+1+j]: P limb j); the entries equal
 *       to `rank` must be exactly the owned limbs the handle was created with.  A handle that owns every limb needs no map (world = 1).
 *   chunks: the batch is cut into this many chunks alternating between two side streams, so that one chunk's exchange runs under the other's
 *       arithmetic (<= 0: auto = 4 for npoly >= 4 on more than one rank, else 1).  The call returns with everything enqueued; the ring's
 *       stream (rh_ring_set_stream of the local Q ring) waits for the side streams, so the caller synchronises as for any other entry.
 *   rh_kshard_exchange_words / rh_kshard_set_exchange: optional.  The exchange blocks are carved from memory the library allocates, unless
 *       the host registers an arena of at least rh_kshard_exchange_words(...) words (a host whose all-gather must map the pointers it is
 *       handed back to its own buffer objects, e.g. torch tensors).
 * cx_loc, ct0_loc, ct1_loc: (npoly, owned Q limbs, N); evkQ_loc / evkP_loc: [digit][component < 2][owned limb][N].  Outputs stay limb-sharded,
 * every owned limb bit-identical to the same limb of rh_bext_gadget_product. */
typedef int (*rh_allgather_fn)(void* ctx, const uint64_t* send_dev, uint64_t* recv_dev, size_t send_words, void* hip_stream);
int rh_kshard_set_world(rh_kshard* ks, int world, int rank, const int* owner);
int rh_kshard_exchange_words(const rh_kshard* ks, int npoly, int chunks, size_t* words);
int rh_kshard_set_exchange(rh_kshard* ks, uint64_t* arena_dev, size_t words);
int rh_kshard_gadget_product(rh_kshard* ks, const uint64_t* cx_loc, const uint64_t* evkQ_loc, const uint64_t* evkP_loc,
                             uint64_t* ct0_loc, uint64_t* ct1_loc, int npoly, rh_allgather_fn allgather, void* ctx, int chunks);

#ifdef __cplusplus
}
#endif
#endif
