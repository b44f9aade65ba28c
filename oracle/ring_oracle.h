/*
 * ring_oracle.h -- CPU restatement of the reference `ring` hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the parity oracle.  It is NOT part of the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library (matrix-fhe-lattigo_amd/csrc) never links,
 * loads or calls anything declared here.
 *
 * Every function restates (does not copy) the algorithm of the cited reference file:line in
 * swanhong/matrix-fhe-lattigo (paths relative to the reference root).  Pinning: the NTT is checked against the
 * reference's own known-answer vectors (ring/ntt_test.go:10-89 -> tests/golden/ntt_kat.json); basis extension
 * and vec ops are checked against big-integer ground truth the way ring/ring_test.go:534-884 does; the 3N
 * transform is checked against vectors generated from references/integer_dft.py (tests/golden/ntt3n_*.json).
 */
#ifndef RING_ORACLE_H
#define RING_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar primitives: ring/modular_reduction.go ---- */
uint64_t orc_mform(uint64_t a, uint64_t q, const uint64_t bred[2]);       /* :11-35  */
uint64_t orc_mform_lazy(uint64_t a, uint64_t q, const uint64_t bred[2]);  /* :40-45  */
uint64_t orc_imform(uint64_t a, uint64_t q, uint64_t qinv);               /* :49-56  */
uint64_t orc_imform_lazy(uint64_t a, uint64_t q, uint64_t qinv);          /* :61-65  */
uint64_t orc_gen_mred_constant(uint64_t q);                               /* :68-75  */
void     orc_gen_bred_constant(uint64_t q, uint64_t out[2]);              /* :99-107 */
uint64_t orc_mred(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv);     /* :78-86  */
uint64_t orc_mred_lazy(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv);/* :90-95  */
uint64_t orc_bred_add(uint64_t a, uint64_t q, const uint64_t bred[2]);    /* :110-117 */
uint64_t orc_bred_add_lazy(uint64_t a, uint64_t q, const uint64_t bred[2]);/* :121-124 */
uint64_t orc_bred(uint64_t x, uint64_t y, uint64_t q, const uint64_t bred[2]);     /* :127-162 */
uint64_t orc_bred_lazy(uint64_t x, uint64_t y, uint64_t q, const uint64_t bred[2]);/* :166-197 */
uint64_t orc_cred(uint64_t a, uint64_t q);                                /* :200-205 */
uint64_t orc_modexp(uint64_t x, uint64_t e, uint64_t p);                  /* ring/utils.go:30-40 */

/* ---- number theory used by table generation ---- */
int      orc_is_prime(uint64_t n);
/* smallest primitive root >= 3 of prime q: ring/subring.go:218-251 (g starts at 2 and is pre-incremented) */
uint64_t orc_primitive_root(uint64_t q);

/* ---- NTT tables: ring/subring.go:129-214 (power-of-two NthRoot/2 branch, bit-reversed Montgomery tables) ----
 * roots_fwd/roots_bwd have nthroot/2 entries; *ninv = MForm((nthroot/2)^-1).  Returns 0 on success, <0 on error
 * (-1: q not prime, -2: q != 1 mod nthroot). */
int orc_gen_ntt_tables(uint64_t q, uint64_t nthroot, uint64_t* roots_fwd, uint64_t* roots_bwd,
                       uint64_t* ninv, uint64_t* primitive_root);

/* ---- negacyclic NTT: ring/ntt.go ---- */
/* nttCoreLazy :209-221 (N<16 -> nttLazy :223-257, else the reduce schedule of nttUnrolled16Lazy :258-552) */
void orc_ntt_core_lazy(const uint64_t* p1, uint64_t* p2, int N, uint64_t q, uint64_t qinv, const uint64_t* roots);
/* inttCoreLazy :554-714 */
void orc_intt_core_lazy(const uint64_t* p1, uint64_t* p2, int N, uint64_t q, uint64_t qinv, const uint64_t* roots);
void orc_ntt_standard(const uint64_t* p1, uint64_t* p2, int N, uint64_t q, uint64_t qinv, const uint64_t bred[2],
                      const uint64_t* roots);                                                   /* :174-177 */
void orc_ntt_standard_lazy(const uint64_t* p1, uint64_t* p2, int N, uint64_t q, uint64_t qinv,
                           const uint64_t* roots);                                              /* :180-182 */
void orc_intt_standard(const uint64_t* p1, uint64_t* p2, int N, uint64_t ninv, uint64_t q, uint64_t qinv,
                       const uint64_t* roots);                                                  /* :185-194 */
void orc_intt_standard_lazy(const uint64_t* p1, uint64_t* p2, int N, uint64_t ninv, uint64_t q, uint64_t qinv,
                            const uint64_t* roots);                                             /* :197-206 */

/* conjugate-invariant NTT (ring/ntt.go:716-1311), canonical outputs; roots = tables of the 4N-th root (2N entries) */
void orc_ntt_ci(const uint64_t* p1, uint64_t* p2, int N, uint64_t q, uint64_t qinv, const uint64_t bred[2], const uint64_t* roots);
void orc_intt_ci(const uint64_t* p1, uint64_t* p2, int N, uint64_t ninv, uint64_t q, uint64_t qinv, const uint64_t* roots);

/* ---- element-wise kernels: ring/vec_ops.go.  Opcodes are shared with include/ringhip.h (RH_OP_*). ----
 * p1,p2: inputs (p2 may be NULL for unary/scalar ops); p3: output (read-modify-write for the *then* ops);
 * s0,s1: scalars.  n must be a multiple of 8 (reference contract). */
int orc_vec_op(int opcode, const uint64_t* p1, const uint64_t* p2, uint64_t* p3, size_t n,
               uint64_t s0, uint64_t s1, uint64_t q, uint64_t qinv, const uint64_t bred[2]);

/* ---- RNS basis extension: ring/basis_extension.go ---- */
typedef struct {
  int nq, np;
  uint64_t* qoverqiinvqi;   /* [nq]            */
  uint64_t* qoverqimodp;    /* [np][nq]        */
  uint64_t* vtimesqmodp;    /* [np][nq+1]      */
} orc_modup_constants;
/* GenModUpConstants :93-164 */
orc_modup_constants* orc_gen_modup_constants(const uint64_t* Q, int nq, const uint64_t* P, int np);
void orc_free_modup_constants(orc_modup_constants* c);
/* ModUpExact :282-308 with reconstructRNS :550-594 and multSum :597-673.  p1: nq limb pointers, p2: np limb pointers */
void orc_modup_exact(const uint64_t* const* p1, uint64_t* const* p2, size_t n, const uint64_t* Q, const uint64_t* P,
                     const orc_modup_constants* c);
/* ModUpQtoP/ModUpPtoQ :188-217 (add floor(Q/2), ModUpExact, subtract floor(Q/2) mod each target) */
void orc_modup_centered(const uint64_t* const* p1, uint64_t* const* p2, size_t n, const uint64_t* Q, int nq,
                        const uint64_t* P, int np);
/* ModDownQPtoQ :223-234 (coefficient domain).  p1q: nq limbs, p1p: np limbs, p2q: nq limbs out */
void orc_moddown_qp_to_q(const uint64_t* const* p1q, const uint64_t* const* p1p, uint64_t* const* p2q, size_t n,
                         const uint64_t* Q, int nq, const uint64_t* P, int np);
/* ModDownQPtoQNTT :241-258.  Needs NTT tables for every Q and P limb (N = n). */
void orc_moddown_qp_to_q_ntt(const uint64_t* const* p1q, const uint64_t* const* p1p, uint64_t* const* p2q, size_t n,
                             const uint64_t* Q, int nq, const uint64_t* P, int np,
                             const uint64_t* const* rootsQ_fwd, const uint64_t* const* rootsP_bwd,
                             const uint64_t* ninvP);
/* DecomposeAndSplit :381-502.  Qall: all moduli of ringQ (nQall), Pall: all of ringP (nPall).
 * p0q: levelQ+1 limbs in; p1q: levelQ+1 limbs out; p1p: levelP+1 limbs out. */
void orc_decompose_and_split(int levelQ, int levelP, int nbPi, int digit, const uint64_t* const* p0q,
                             uint64_t* const* p1q, uint64_t* const* p1p, size_t n,
                             const uint64_t* Qall, int nQall, const uint64_t* Pall, int nPall);

/* ---- automorphisms: ring/automorphism.go:12-35 (index), :52-117 (NTT domain, optional += lazy), :162-175 (coefficients) ---- */
void orc_automorphism_ntt_index(int N, uint64_t nthroot, uint64_t gal, uint64_t* index);
void orc_automorphism_ntt(const uint64_t* in, uint64_t* out, int N, uint64_t gal, int add_lazy);
void orc_automorphism(const uint64_t* in, uint64_t* out, int N, uint64_t gal, uint64_t q);

/* ---- RNS rescale: ring/scaling.go:21-28 (floor) / :112-126 (round); one step at `level`, coefficient domain ---- */
void orc_div_by_last_modulus(int round, uint64_t* const* p0, uint64_t* const* p1, size_t n, const uint64_t* Q, int level);

/* ---- 3N-cyclotomic transform: ring/ntt_3n.go ---- */
/* ascending totatives of 3N :235-243; returns count written (= N) */
int  orc_ntt3n_exponents(int threeN, int* out);
/* Forward by definition (Horner at x_k = omega^E[k]) :82-109.  O(N^2). */
void orc_ntt3n_forward_def(const uint64_t* p1, uint64_t* p2, int N, uint64_t q, uint64_t omega);
/* Backward by definition (Vandermonde solve) :118-151,170-222.  O(N^3): small N only. */
int  orc_ntt3n_backward_def(const uint64_t* p1, uint64_t* p2, int N, uint64_t q, uint64_t omega);
/* O(N log N) restatement of references/integer_dft.py:266-348 / :350-432 with outputs permuted into the
 * ascending-totative order of the Go transformer (N = 2^a 3^b, a>=1). */
int  orc_ntt3n_forward_fast(const uint64_t* p1, uint64_t* p2, int N, uint64_t q, uint64_t omega);
int  orc_ntt3n_backward_fast(const uint64_t* p1, uint64_t* p2, int N, uint64_t q, uint64_t omega);

/* ---- timing helper for bench.py's cpu_baseline leg: runs `reps` forward NTTs of `nlimbs` limbs, returns seconds */
void orc_automorphism_ntt_nthroot(const uint64_t* in, uint64_t* out, int N, uint64_t nthroot, uint64_t gal, int add_lazy);
void orc_automorphism_ci(const uint64_t* in, uint64_t* out, int N, uint64_t gal, uint64_t q);
double orc_time_ntt_forward(int N, int nlimbs, const uint64_t* moduli, int reps, int threads);
double orc_time_ntt_forward_polys(int N, int nlimbs, const uint64_t* moduli, int npolys, int reps, int threads);

#ifdef __cplusplus
}
#endif
#endif
