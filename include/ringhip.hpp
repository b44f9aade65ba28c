// ringhip.hpp -- C++ host-side mirror of the reference's `ring` interface over the C ABI (ringhip.h).
//
// The reference is compiled Go; this image has no Go toolchain, so the compiled-language host side is C++ (the Go/cgo
// binding is go/ringhip/).  Names, argument order and error behaviour follow the reference: methods that panic in Go
// throw ringhip::Panic here (ring/ntt.go:212-214), constructors that return `error` throw ringhip::Error
// (ring/ring.go:321-331).  Header-only; link with -lringhip.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "ringhip.h"

namespace ringhip {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };   // Go `error`
struct Panic : std::logic_error { using std::logic_error::logic_error; };         // Go `panic`

inline void check(int rc) {
  if (rc == RH_OK) return;
  const std::string msg = rh_last_error();
  if (rc == RH_ERR_ARG) throw Panic(msg);
  throw Error(msg);
}

enum class Type { Standard = RH_RING_STANDARD, ConjugateInvariant = RH_RING_CI, Matrix = RH_RING_3N };   // ring.Type

class Ring;

// Device-resident block of `npoly` polynomials with `limbs` limbs: the device form of ring.Poly (ring/poly.go:13-24)
class Poly {
 public:
  Poly(const Ring& r, int npoly, int limbs);
  ~Poly();
  Poly(const Poly&) = delete;
  Poly& operator=(const Poly&) = delete;
  Poly(Poly&& o) noexcept : ring_(o.ring_), ptr_(o.ptr_), npoly_(o.npoly_), limbs_(o.limbs_) { o.ptr_ = nullptr; }
  uint64_t* data() const { return ptr_; }
  int npoly() const { return npoly_; }
  int limbs() const { return limbs_; }
  size_t words() const;
  void upload(const std::vector<uint64_t>& host);          // host layout: [poly][limb][coeff]
  std::vector<uint64_t> download() const;
 private:
  const Ring* ring_; uint64_t* ptr_ = nullptr; int npoly_, limbs_;
};

// one modulus of a Ring: the NumberTheoreticTransformer seam (ring/ntt.go:17-22), host slices, one limb per call
class SubRing {
 public:
  SubRing(rh_ring* h, int idx, int n, uint64_t q) : N(n), Modulus(q), h_(h), idx_(idx) {}
  const int N; const uint64_t Modulus;
  void NTT(const std::vector<uint64_t>& p1, std::vector<uint64_t>& p2) const { call(rh_ntt_forward, p1, p2); }
  void NTTLazy(const std::vector<uint64_t>& p1, std::vector<uint64_t>& p2) const { call(rh_ntt_forward_lazy, p1, p2); }
  void INTT(const std::vector<uint64_t>& p1, std::vector<uint64_t>& p2) const { call(rh_ntt_backward, p1, p2); }
  void INTTLazy(const std::vector<uint64_t>& p1, std::vector<uint64_t>& p2) const { call(rh_ntt_backward_lazy, p1, p2); }
 private:
  template <class F> void call(F f, const std::vector<uint64_t>& p1, std::vector<uint64_t>& p2) const {
    if ((int)p1.size() < N || (int)p2.size() < N)
      throw Panic("cannot NTT: ensure that len(p1)=" + std::to_string(p1.size()) + ", len(p2)=" + std::to_string(p2.size()) + " >= N=" + std::to_string(N));
    check(f(h_, idx_, p1.data(), p2.data()));
  }
  rh_ring* h_; int idx_;
};

// what a SubRing holds per modulus (ring/subring.go:35-55, NTTTable ring/ntt.go:38-44), concatenated over the limbs:
// roots_* have N words per limb (Standard) or 2N (ConjugateInvariant, NthRoot = 4N); omega3n is the 3N transformer's psi3N
struct Constants { std::vector<uint64_t> mred, bred, ninv, roots_fwd, roots_bwd, omega3n; };

class Ring {
 public:
  // NewRing / NewRingConjugateInvariant / NewRingFromType(.., Matrix) (ring/ring.go:264-308): constants generated with the
  // reference's rules.  Type::Matrix: omega3n (one primitive 3N-th root per modulus) may be given, as the Go transformer draws
  // its own at random (ring/ntt_3n.go:39, primes_3n.go:127-149); default g^((q-1)/3N).
  Ring(int N, const std::vector<uint64_t>& moduli, Type type = Type::Standard, int device = 0, const std::vector<uint64_t>* omega3n = nullptr)
      : N_(N), type_(type), moduli_(moduli), level_((int)moduli.size() - 1) {
    rh_ring* h = nullptr;
    check(rh_ring_create_auto(&h, device, (int)type, N, (int)moduli.size(), moduli.data(), omega3n ? omega3n->data() : nullptr));
    adopt(h);
  }
  // NewRingWithCustomNTT (ring/ring.go:314-356): the constants handoff -- the engine receives what the host side generated and
  // never re-derives a root (this is the constructor the cgo factories Factory / FactoryCI / Factory3N use, go/ringhip)
  Ring(int N, const std::vector<uint64_t>& moduli, Type type, const Constants& c, int device = 0)
      : N_(N), type_(type), moduli_(moduli), level_((int)moduli.size() - 1) {
    rh_ring* h = nullptr;
    auto ptr = [](const std::vector<uint64_t>& v) { return v.empty() ? nullptr : v.data(); };
    check(rh_ring_create(&h, device, (int)type, N, (int)moduli.size(), moduli.data(), ptr(c.mred), ptr(c.bred), ptr(c.ninv), ptr(c.roots_fwd),
                         ptr(c.roots_bwd), ptr(c.omega3n)));
    adopt(h);
  }
  Type GetType() const { return type_; }
  Constants GetConstants() const {
    const size_t L = moduli_.size(), TN = type_ == Type::ConjugateInvariant ? (size_t)2 * N_ : (size_t)N_;
    Constants c;
    c.mred.resize(L); c.bred.resize(2 * L); c.ninv.resize(L);
    if (type_ == Type::Matrix) {
      c.omega3n.resize(L);
      check(rh_ring_get_constants(h_.get(), nullptr, c.mred.data(), c.bred.data(), nullptr, nullptr, nullptr, c.omega3n.data()));
    } else {
      c.roots_fwd.resize(L * TN); c.roots_bwd.resize(L * TN);
      check(rh_ring_get_constants(h_.get(), nullptr, c.mred.data(), c.bred.data(), c.ninv.data(), c.roots_fwd.data(), c.roots_bwd.data(), nullptr));
    }
    return c;
  }
  void Reserve(int npoly) const { check(rh_ring_reserve(h_.get(), npoly)); }
  int N() const { return N_; }
  int Level() const { return level_; }
  int ModuliChainLength() const { return (int)moduli_.size(); }
  const std::vector<uint64_t>& ModuliChain() const { return moduli_; }
  rh_ring* handle() const { return h_.get(); }
  // view restricted to limbs 0..level (ring/ring.go:194-213); shares the engine handle
  Ring AtLevel(int level) const {
    if (level < 0 || level >= (int)moduli_.size()) throw Panic("level out of range");
    Ring v(*this); v.level_ = level; return v;
  }
  Poly NewPoly(int npoly = 1) const { return Poly(*this, npoly, level_ + 1); }
  void Sync() const { check(rh_ring_sync(h_.get())); }

  // polys may carry more limbs than the view's level (AtLevel on max-level polys, ring/ring.go:192-213): the *_rows entry points
  void NTT(const Poly& p1, Poly& p2) const { check(rh_ring_ntt_rows(h_.get(), p1.data(), p1.limbs(), p2.data(), p2.limbs(), p1.npoly(), level_, 0)); }
  void NTTLazy(const Poly& p1, Poly& p2) const { check(rh_ring_ntt_rows(h_.get(), p1.data(), p1.limbs(), p2.data(), p2.limbs(), p1.npoly(), level_, 1)); }
  void INTT(const Poly& p1, Poly& p2) const { check(rh_ring_intt_rows(h_.get(), p1.data(), p1.limbs(), p2.data(), p2.limbs(), p1.npoly(), level_, 0)); }
  void INTTLazy(const Poly& p1, Poly& p2) const { check(rh_ring_intt_rows(h_.get(), p1.data(), p1.limbs(), p2.data(), p2.limbs(), p1.npoly(), level_, 1)); }
  // Ring.NTT / NTTLazy / INTT / INTTLazy on a whole HOST poly in one call (ring/ntt.go:127-152: the loop over Poly.Coeffs [][]uint64):
  // coeffs[i] = limb i's slice (>= N words; pageable, or page-locked memory from rh_host_alloc / rh_host_register), limbs 0..level of this view
  void NTT(const std::vector<const uint64_t*>& p1, const std::vector<uint64_t*>& p2) const { hostPoly(rh_ntt_poly_forward, p1, p2, 0); }
  void NTTLazy(const std::vector<const uint64_t*>& p1, const std::vector<uint64_t*>& p2) const { hostPoly(rh_ntt_poly_forward, p1, p2, 1); }
  void INTT(const std::vector<const uint64_t*>& p1, const std::vector<uint64_t*>& p2) const { hostPoly(rh_ntt_poly_backward, p1, p2, 0); }
  void INTTLazy(const std::vector<const uint64_t*>& p1, const std::vector<uint64_t*>& p2) const { hostPoly(rh_ntt_poly_backward, p1, p2, 1); }
  // 3N rings: the NTT-domain layout as an argument of the call (the host tags its blocks: DevicePoly.layout in Python, DevPoly.BlockOrder in Go)
  bool BlockOrderSupported() const { return rh_ring_ntt3n_block_order_supported(h_.get()) != 0; }
  void NTTLayout(const Poly& p1, Poly& p2, bool blockOrder) const {
    check(rh_ring_ntt_layout(h_.get(), p1.data(), p1.limbs(), p2.data(), p2.limbs(), p1.npoly(), level_, 0, blockOrder ? 1 : 0));
  }
  void INTTLayout(const Poly& p1, Poly& p2, bool blockOrder) const {
    check(rh_ring_ntt_layout(h_.get(), p1.data(), p1.limbs(), p2.data(), p2.limbs(), p1.npoly(), level_, 1, blockOrder ? 1 : 0));
  }
  void NTT3NReorder(const Poly& p1, Poly& p2, bool toReference) const { check(rh_ring_ntt3n_reorder(h_.get(), p1.data(), p2.data(), p1.npoly(), level_, toReference ? 1 : 0)); }
  long Stats(const char* key) const { long v = 0; check(rh_ring_stats(h_.get(), key, &v)); return v; }
  // c = INTT(NTT(a) . NTT(b)) from coefficient-domain operands (BASELINE config 3) with the tile stages of all three transforms as one kernel;
  // a and b are consumed (rh_ring_polymul)
  void PolyMul(Poly& a, Poly& b, Poly& c) const { check(rh_ring_polymul(h_.get(), a.data(), b.data(), c.data(), a.npoly(), level_)); }
  // Ring.NTT on several blocks in one call (rh_ring_ntt_many): every poly of a block at this view's level (limbs() == level + 1)
  void NTTMany(const std::vector<std::pair<const Poly*, Poly*>>& blocks) const {
    std::vector<const uint64_t*> in; std::vector<uint64_t*> out; std::vector<int> cnt;
    for (const auto& b : blocks) {
      if (b.first->limbs() != level_ + 1 || b.second->limbs() != level_ + 1) throw std::invalid_argument("NTTMany: blocks must hold level+1 limbs per poly");
      in.push_back(b.first->data()); out.push_back(b.second->data()); cnt.push_back(b.first->npoly());
    }
    check(rh_ring_ntt_many(h_.get(), in.data(), out.data(), cnt.data(), (int)in.size(), level_));
  }

  // ring/operations.go -> ring/vec_ops.go
  void VecOp(int op, const Poly* p1, const Poly* p2, Poly& p3, const uint64_t* s0 = nullptr, const uint64_t* s1 = nullptr) const {
    check(rh_ring_vec_op_rows(h_.get(), op, p1 ? p1->data() : nullptr, p1 ? p1->limbs() : 0, p2 ? p2->data() : nullptr, p2 ? p2->limbs() : 0,
                              p3.data(), p3.limbs(), p3.npoly(), level_, s0, s1));
  }
  void Add(const Poly& a, const Poly& b, Poly& c) const { VecOp(RH_OP_ADD, &a, &b, c); }
  void Sub(const Poly& a, const Poly& b, Poly& c) const { VecOp(RH_OP_SUB, &a, &b, c); }
  void Neg(const Poly& a, Poly& c) const { VecOp(RH_OP_NEG, &a, nullptr, c); }
  void Reduce(const Poly& a, Poly& c) const { VecOp(RH_OP_REDUCE, &a, nullptr, c); }
  void MForm(const Poly& a, Poly& c) const { VecOp(RH_OP_MFORM, &a, nullptr, c); }
  void IMForm(const Poly& a, Poly& c) const { VecOp(RH_OP_IMFORM, &a, nullptr, c); }
  void MulCoeffsBarrett(const Poly& a, const Poly& b, Poly& c) const { VecOp(RH_OP_MUL_BARRETT, &a, &b, c); }
  void MulCoeffsMontgomery(const Poly& a, const Poly& b, Poly& c) const { VecOp(RH_OP_MUL_MONT, &a, &b, c); }
  void MulCoeffsMontgomeryThenAdd(const Poly& a, const Poly& b, Poly& c) const { VecOp(RH_OP_MUL_MONT_THEN_ADD, &a, &b, c); }
  void MulCoeffsMontgomeryLazy(const Poly& a, const Poly& b, Poly& c) const { VecOp(RH_OP_MUL_MONT_LAZY, &a, &b, c); }
  void MulCoeffsMontgomeryLazyThenAddLazy(const Poly& a, const Poly& b, Poly& c) const { VecOp(RH_OP_MUL_MONT_LAZY_THEN_ADD_LAZY, &a, &b, c); }
  // ring/scaling.go
  void DivRoundByLastModulusManyNTT(int nb, const Poly& p0, Poly& p1) const {
    check(rh_ring_div_by_last_modulus_many_ntt(h_.get(), 1, level_, nb, p0.data(), p1.data(), p1.limbs(), p0.npoly()));
  }
  void DivFloorByLastModulusManyNTT(int nb, const Poly& p0, Poly& p1) const {
    check(rh_ring_div_by_last_modulus_many_ntt(h_.get(), 0, level_, nb, p0.data(), p1.data(), p1.limbs(), p0.npoly()));
  }
  // ring/automorphism.go
  // ckks mulRelin's degree-1 x degree-1 tensoring as one kernel (schemes/ckks/evaluator.go:821-834)
  void TensorDegree1(const Poly& a0, const Poly& a1, const Poly& b0, const Poly& b1, Poly& c0, Poly& c1, Poly& c2, bool mformFirst = true) const {
    check(rh_ring_tensor_degree1(h_.get(), a0.data(), a1.data(), b0.data(), b1.data(), c0.data(), c1.data(), c2.data(), a0.npoly(), level_, mformFirst ? 1 : 0));
  }
  void AutomorphismNTT(const Poly& in, uint64_t gen, Poly& out) const { check(rh_ring_automorphism_ntt(h_.get(), level_, in.data(), gen, out.data(), in.npoly(), 0)); }
  void Automorphism(const Poly& in, uint64_t gen, Poly& out) const { check(rh_ring_automorphism(h_.get(), level_, in.data(), gen, out.data(), in.npoly())); }
  // AutomorphismNTTWithIndex / ...ThenAddLazy (ring/automorphism.go:50-117): `index` = a 1-poly, 1-limb device block holding the lookup table
  void AutomorphismNTTWithIndex(const Poly& in, const Poly& index, Poly& out, bool thenAddLazy = false) const {
    check(rh_ring_automorphism_ntt_index(h_.get(), level_, in.data(), index.data(), out.data(), in.npoly(), thenAddLazy ? 1 : 0));
  }
  // ring/operations.go: Poly.CopyLvl, Shift (:278-282), MultByMonomial (:306-363), MulByVectorMontgomery(ThenAddLazy) (:366-377); out of place
  void CopyLvl(const Poly& p1, Poly& p2) const { check(rh_ring_copy_rows(h_.get(), p2.data(), p2.limbs(), p1.data(), p1.limbs(), p1.npoly(), level_)); }
  // ring/conjugate_invariant.go: UnfoldConjugateInvariantToStandard (:8-26, receiver: standard ring of degree 2n), FoldStandardToConjugateInvariant
  // (:31-49, receiver: conjugate-invariant ring of degree n; index: device table of n words < 2n), PadDefaultRingToConjugateInvariant (:52-80)
  void UnfoldConjugateInvariantToStandard(const Poly& ci, Poly& std_) const { check(rh_ring_unfold_ci_to_standard(h_.get(), level_, ci.data(), std_.data(), std_.npoly())); }
  void FoldStandardToConjugateInvariant(const Poly& std_, const Poly& index, Poly& ci) const {
    check(rh_ring_fold_standard_to_ci(h_.get(), level_, std_.data(), index.data(), ci.data(), std_.npoly()));
  }
  void PadDefaultRingToConjugateInvariant(const Poly& std_, bool isNTT, Poly& ci) const {
    check(rh_ring_pad_default_to_ci(h_.get(), level_, std_.data(), isNTT ? 1 : 0, ci.data(), std_.npoly()));
  }
  void Shift(const Poly& p1, int k, Poly& p2) const { check(rh_ring_shift(h_.get(), level_, p1.data(), p2.data(), k, p1.npoly())); }
  void MultByMonomial(const Poly& p1, int k, Poly& p2) const { check(rh_ring_mult_by_monomial(h_.get(), level_, p1.data(), p2.data(), k, p1.npoly())); }
  void MulByVectorMontgomery(const Poly& p1, const Poly& vector, Poly& p2, bool thenAddLazy = false) const {
    check(rh_ring_vec_op_bcast(h_.get(), thenAddLazy ? RH_OP_MUL_MONT_THEN_ADD_LAZY : RH_OP_MUL_MONT, p1.data(), p1.limbs(), vector.data(), p2.data(),
                               p2.limbs(), p1.npoly(), level_));
  }
  // MForm + MulCoeffsMontgomery + INTT as one call (schemes/ckks/evaluator.go:821-834 + INTT)
  void INTTMul(const Poly& a, const Poly& b, Poly& out) const { check(rh_ring_intt_mul(h_.get(), a.data(), b.data(), out.data(), a.npoly(), level_)); }

  std::vector<SubRing> SubRings;
 private:
  template <class F> void hostPoly(F f, const std::vector<const uint64_t*>& p1, const std::vector<uint64_t*>& p2, int lazy) const {
    if ((int)p1.size() < level_ + 1 || (int)p2.size() < level_ + 1)
      throw Panic("cannot NTT: poly has " + std::to_string(p1.size()) + " / " + std::to_string(p2.size()) + " limbs, ring level needs " + std::to_string(level_ + 1));
    check(f(h_.get(), level_, p1.data(), p2.data(), lazy));
  }
  void adopt(rh_ring* h) {
    h_.reset(h, rh_ring_destroy);
    for (int i = 0; i < (int)moduli_.size(); ++i) SubRings.emplace_back(h, i, N_, moduli_[i]);
  }
  int N_; Type type_; std::vector<uint64_t> moduli_; int level_;
  std::shared_ptr<rh_ring> h_;
};

inline Poly::Poly(const Ring& r, int npoly, int limbs) : ring_(&r), npoly_(npoly), limbs_(limbs) {
  check(rh_dev_alloc(r.handle(), words() ? words() : 1, &ptr_));
}
inline Poly::~Poly() { if (ptr_) rh_dev_free(nullptr, ptr_); }
inline size_t Poly::words() const { return (size_t)npoly_ * limbs_ * ring_->N(); }
inline void Poly::upload(const std::vector<uint64_t>& host) {
  if (host.size() < words()) throw Panic("upload: host vector too short");
  check(rh_dev_upload(ring_->handle(), ptr_, host.data(), words()));
}
inline std::vector<uint64_t> Poly::download() const {
  std::vector<uint64_t> out(words());
  check(rh_dev_download(ring_->handle(), out.data(), ptr_, words()));
  return out;
}

// ring.BasisExtender (ring/basis_extension.go:13-79) + the key-switch gadget product built on it
class BasisExtender {
 public:
  BasisExtender(const Ring& q, const Ring& p) { rh_bext* h = nullptr; check(rh_bext_create(&h, q.handle(), p.handle())); h_.reset(h, rh_bext_destroy); }
  void Reserve(int npoly) const { check(rh_bext_reserve(h_.get(), npoly)); }
  void ModUpQtoP(int lq, int lp, const Poly& polQ, Poly& polP) const { check(rh_bext_modup_q_to_p(h_.get(), lq, lp, polQ.data(), polP.data(), polQ.npoly())); }
  void ModUpPtoQ(int lp, int lq, const Poly& polP, Poly& polQ) const { check(rh_bext_modup_p_to_q(h_.get(), lp, lq, polP.data(), polQ.data(), polP.npoly())); }
  void ModDownQPtoQ(int lq, int lp, const Poly& q1, const Poly& p1, Poly& q2) const { check(rh_bext_moddown_qp_to_q(h_.get(), lq, lp, q1.data(), p1.data(), q2.data(), q1.npoly())); }
  void ModDownQPtoQNTT(int lq, int lp, const Poly& q1, const Poly& p1, Poly& q2) const { check(rh_bext_moddown_qp_to_q_ntt(h_.get(), lq, lp, q1.data(), p1.data(), q2.data(), q1.npoly())); }
  void ModDownQPtoP(int lq, int lp, const Poly& q1, const Poly& p1, Poly& p2) const { check(rh_bext_moddown_qp_to_p(h_.get(), lq, lp, q1.data(), p1.data(), p2.data(), q1.npoly())); }
  void DecomposeAndSplit(int lq, int lp, int nbPi, int digit, const Poly& p0Q, Poly& p1Q, Poly& p1P) const {
    check(rh_bext_decompose_and_split(h_.get(), lq, lp, nbPi, digit, p0Q.data(), p1Q.data(), p1P.data(), p0Q.npoly()));
  }
  void GadgetProduct(int lq, int lp, const Poly& cx, const Poly& evkQ, const Poly& evkP, int beta, Poly& ct0, Poly& ct1) const {
    check(rh_bext_gadget_product(h_.get(), lq, lp, cx.data(), evkQ.data(), evkP.data(), beta, ct0.data(), ct1.data(), cx.npoly()));
  }
  // ct_c = add_c + GadgetProduct(cx)_c: the ring.Add that follows the product in Relinearize / mulRelin / Automorphism,
  // folded into ModDown's epilogue (add0 / add1 may be null and may alias ct0 / ct1)
  void GadgetProductThenAdd(int lq, int lp, const Poly& cx, const Poly& evkQ, const Poly& evkP, int beta, const Poly* add0, const Poly* add1,
                            Poly& ct0, Poly& ct1) const {
    check(rh_bext_gadget_product_then_add(h_.get(), lq, lp, cx.data(), evkQ.data(), evkP.data(), beta, add0 ? add0->data() : nullptr,
                                          add1 ? add1->data() : nullptr, ct0.data(), ct1.data(), cx.npoly()));
  }
  // Evaluator.DecomposeNTT / GadgetProductHoisted (core/rlwe/evaluator_gadget_product.go:431-453, 326-429): decompQ / decompP
  // hold beta * npoly polys (digit i of poly k = poly i*npoly + k)
  void DecomposeNTT(int lq, int lp, const Poly& c2, bool c2IsNTT, Poly& decompQ, Poly& decompP) const {
    check(rh_bext_decompose_ntt(h_.get(), lq, lp, c2.data(), c2IsNTT ? 1 : 0, decompQ.data(), decompP.data(), c2.npoly()));
  }
  void GadgetProductHoisted(int lq, int lp, const Poly& decompQ, const Poly& decompP, const Poly& evkQ, const Poly& evkP, int beta,
                            Poly& ct0, Poly& ct1) const {
    check(rh_bext_gadget_product_hoisted(h_.get(), lq, lp, decompQ.data(), decompP.data(), evkQ.data(), evkP.data(), beta, ct0.data(),
                                         ct1.data(), ct0.npoly()));
  }
  // GadgetProductHoistedLazy (:351-371): the accumulators modulo Q and modulo P, no ModDown; ModDownPair: Evaluator.ModDown (:33-46) NTT -> NTT
  void GadgetProductHoistedLazy(int lq, int lp, const Poly& decompQ, const Poly& decompP, const Poly& evkQ, const Poly& evkP, int beta,
                                Poly& ctQ0, Poly& ctQ1, Poly& ctP0, Poly& ctP1) const {
    check(rh_bext_gadget_product_hoisted_lazy(h_.get(), lq, lp, decompQ.data(), decompP.data(), evkQ.data(), evkP.data(), beta, ctQ0.data(),
                                              ctQ1.data(), ctP0.data(), ctP1.data(), ctQ0.npoly()));
  }
  void ModDownPair(int lq, int lp, const Poly& ctQ0, const Poly& ctQ1, const Poly& ctP0, const Poly& ctP1, Poly& ct0, Poly& ct1) const {
    check(rh_bext_moddown_qp_to_q_ntt_pair(h_.get(), lq, lp, ctQ0.data(), ctQ1.data(), ctP0.data(), ctP1.data(), ct0.data(), ct1.data(), ct0.npoly()));
  }
 private:
  std::shared_ptr<rh_bext> h_;
};

// limb-sharded key switch (rh_kshard_*): this rank's limbs of the gadget product; the caller moves the
// gathered source limbs (RCCL all-gather) between Digit / ModDown calls
class KeySwitchShard {
 public:
  KeySwitchShard(const Ring& qLoc, const Ring* pLoc, const std::vector<uint64_t>& allQ, const std::vector<uint64_t>& allP,
                 const std::vector<int>& ownQ, const std::vector<int>& ownP) {
    rh_kshard* h = nullptr;
    check(rh_kshard_create(&h, qLoc.handle(), pLoc ? pLoc->handle() : nullptr, allQ.data(), (int)allQ.size() - 1, allP.data(),
                           (int)allP.size() - 1, ownQ.data(), (int)ownQ.size(), ownP.empty() ? ownQ.data() : ownP.data(), (int)ownP.size()));
    h_.reset(h, rh_kshard_destroy);
  }
  int NumDigits() const { return rh_kshard_num_digits(h_.get()); }
  std::pair<int, int> DigitRange(int digit) const { int st = 0, ed = 0; check(rh_kshard_digit_range(h_.get(), digit, &st, &ed)); return {st, ed}; }
  void Digit(int digit, const uint64_t* srcGathered, const Poly& cxLoc, const uint64_t* evkQLoc, const uint64_t* evkPLoc, Poly& ct0, Poly& ct1,
             uint64_t* accP0, uint64_t* accP1) const {
    check(rh_kshard_digit(h_.get(), digit, srcGathered, cxLoc.data(), evkQLoc, evkPLoc, ct0.data(), ct1.data(), accP0, accP1, cxLoc.npoly()));
  }
  // all digits in one call: srcAll = every limb of INTT(cx) in chain order (one all-gather per product)
  void Product(const uint64_t* srcAll, const Poly& cxLoc, const uint64_t* evkQLoc, const uint64_t* evkPLoc, Poly& ct0, Poly& ct1,
               uint64_t* accP0, uint64_t* accP1) const {
    check(rh_kshard_product(h_.get(), srcAll, cxLoc.data(), evkQLoc, evkPLoc, ct0.data(), ct1.data(), accP0, accP1, cxLoc.npoly()));
  }
  void ModDown(const uint64_t* srcPGathered, const Poly& ctIn, Poly& ctOut) const {
    check(rh_kshard_moddown(h_.get(), srcPGathered, ctIn.data(), ctOut.data(), ctIn.npoly()));
  }
  // the whole product in one call (rlwe.Evaluator.GadgetProduct on the owned limbs): the host supplies the all-gather -- ncclAllGather on a node
  // (tests/cpp/test_sharded_host.cpp runs it with threads as ranks).  owner[i]: the rank of limb i of Q ++ P.
  void SetWorld(int world, int rank, const std::vector<int>& owner) const { check(rh_kshard_set_world(h_.get(), world, rank, owner.data())); }
  void GadgetProduct(const Poly& cxLoc, const uint64_t* evkQLoc, const uint64_t* evkPLoc, Poly& ct0, Poly& ct1, rh_allgather_fn allgather, void* ctx,
                     int chunks = 0) const {
    check(rh_kshard_gadget_product(h_.get(), cxLoc.data(), evkQLoc, evkPLoc, ct0.data(), ct1.data(), cxLoc.npoly(), allgather, ctx, chunks));
  }
 private:
  std::shared_ptr<rh_kshard> h_;
};

}  // namespace ringhip
