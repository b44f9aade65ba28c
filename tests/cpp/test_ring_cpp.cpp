// C++ conformance test of the host-side mirror (include/ringhip.hpp), written the way the reference's TestNTT is
// (ring/ntt_test.go:91-121): NewRing, NTT(poly) == polyNTT, INTT(NTT(poly)) == poly -- on the N=16 known-answer vector
// of ring/ntt_test.go (both limbs, tests/cpp/golden_vectors.inc), the 3N (Matrix) and conjugate-invariant ring types with the
// constants handoff the cgo factories use, a device-resident poly-mul, the key switch, and the panic/error behaviour.
#include <algorithm>
#include <cstdio>
#include <utility>
#include <vector>
#include "ringhip.hpp"
#include "golden_vectors.inc"

using namespace ringhip;

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main() {
  // TestNTT (ring/ntt_test.go:91-121) on its N = 16 known-answer vector, both limbs: NTT(poly) == polyNTT, INTT back
  const uint64_t q = KAT16_Q[0];
  Ring r(16, KAT16_Q);
  {
    const std::vector<uint64_t>* in[2] = {&KAT16_POLY_0, &KAT16_POLY_1};
    const std::vector<uint64_t>* want[2] = {&KAT16_NTT_0, &KAT16_NTT_1};
    for (int i = 0; i < 2; ++i) {
      std::vector<uint64_t> y(16), back(16), lz(16);
      r.SubRings[i].NTT(*in[i], y);
      EXPECT(y == *want[i]);
      r.SubRings[i].INTT(y, back);
      EXPECT(back == *in[i]);
      r.SubRings[i].NTTLazy(*in[i], lz);
      for (int j = 0; j < 16; ++j) EXPECT(lz[j] % KAT16_Q[i] == y[j]);
    }
    // the device-resident batched path on the same vector: (1 poly, 2 limbs, 16)
    std::vector<uint64_t> host(KAT16_POLY_0); host.insert(host.end(), KAT16_POLY_1.begin(), KAT16_POLY_1.end());
    std::vector<uint64_t> want2(KAT16_NTT_0); want2.insert(want2.end(), KAT16_NTT_1.begin(), KAT16_NTT_1.end());
    Poly p = r.NewPoly();
    p.upload(host);
    r.NTT(p, p);
    EXPECT(p.download() == want2);
    // NewRingWithCustomNTT's constants handoff (ring/ring.go:314-356): a ring built from received constants gives the same bits
    Ring r2(16, KAT16_Q, Type::Standard, r.GetConstants());
    std::vector<uint64_t> y2(16);
    r2.SubRings[1].NTT(KAT16_POLY_1, y2);
    EXPECT(y2 == KAT16_NTT_1);
  }
  // testShift (ring/ring_test.go:904-916): N = 16, q = 97, p1 = 0..15, Shift by 3; then MultByMonomial X^1 * X^8 == X^9 (:880-899) and CopyLvl
  {
    Ring s(16, {97});
    std::vector<uint64_t> iota(16); for (int i = 0; i < 16; ++i) iota[i] = (uint64_t)i;
    Poly p1 = s.NewPoly(), p2 = s.NewPoly(), p3 = s.NewPoly(), p4 = s.NewPoly();
    p1.upload(iota);
    s.Shift(p1, 3, p2);
    EXPECT((p2.download() == std::vector<uint64_t>{3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2}));
    std::vector<uint64_t> ones(16); for (int i = 0; i < 16; ++i) ones[i] = (uint64_t)(i + 1);
    p1.upload(ones);
    s.MultByMonomial(p1, 1, p2); s.MultByMonomial(p2, 8, p3); s.MultByMonomial(p1, 9, p4);
    EXPECT(p3.download() == p4.download());
    s.CopyLvl(p4, p2);
    EXPECT(p2.download() == p4.download());
  }
  // ring/conjugate_invariant.go: Unfold (:8-26), Fold (:31-49), Pad (:52-80) on n = 8 / 2n = 16, q = 97, by hand
  {
    Ring small(8, {97}), big(16, {97});
    std::vector<uint64_t> c8 = {1, 2, 3, 4, 5, 6, 7, 8}, s16(16);
    for (int i = 0; i < 16; ++i) s16[i] = (uint64_t)(90 + i) % 97;          // 90..96, 0..8
    Poly pc = small.NewPoly(), ps = big.NewPoly(), idx = small.NewPoly(), po = small.NewPoly();
    pc.upload(c8);
    big.UnfoldConjugateInvariantToStandard(pc, ps);
    EXPECT((ps.download() == std::vector<uint64_t>{1, 2, 3, 4, 5, 6, 7, 8, 8, 7, 6, 5, 4, 3, 2, 1}));
    ps.upload(s16);
    idx.upload(std::vector<uint64_t>{15, 14, 13, 12, 11, 10, 9, 8});
    small.FoldStandardToConjugateInvariant(ps, idx, po);                     // (s16[15 - j] + s16[j]) mod 97
    EXPECT((po.download() == std::vector<uint64_t>{(8 + 90) % 97, (7 + 91) % 97, (6 + 92) % 97, (5 + 93) % 97, (4 + 94) % 97, (3 + 95) % 97, (2 + 96) % 97, 1}));
    small.PadDefaultRingToConjugateInvariant(pc, true, ps);                 // first n words: c[0..3], reversed c[0..3]; the rest untouched
    std::vector<uint64_t> w = s16; const uint64_t a[8] = {1, 2, 3, 4, 4, 3, 2, 1}; for (int i = 0; i < 8; ++i) w[i] = a[i];
    EXPECT(ps.download() == w);
    small.PadDefaultRingToConjugateInvariant(pc, false, ps);                // 0, c1, c2, c3, q - c4, q - c3, q - c2, q - c1
    const uint64_t b[8] = {0, 2, 3, 4, 97 - 5, 97 - 4, 97 - 3, 97 - 2}; for (int i = 0; i < 8; ++i) w[i] = b[i];
    EXPECT(ps.download() == w);
  }
  // 3N-cyclotomic ring, Type::Matrix (ring/ntt_3n.go:21-156, ring/ring.go:299-304): omega handed over like the Go factory
  // does; vectors from references/integer_dft.py in the Go transformer's ascending-totative order
  {
    struct V { int N; uint64_t p, w; const std::vector<uint64_t>* in; const std::vector<uint64_t>* out; };
    const V vs[2] = {{12, V3N_12_P, V3N_12_W, &V3N_12_IN, &V3N_12_OUT}, {24, V3N_24_P, V3N_24_W, &V3N_24_IN, &V3N_24_OUT}};
    for (const V& v : vs) {
      const std::vector<uint64_t> om = {v.w};
      Ring m(v.N, {v.p}, Type::Matrix, 0, &om);
      std::vector<uint64_t> y(v.N), back(v.N);
      m.SubRings[0].NTT(*v.in, y);
      EXPECT(y == *v.out);
      m.SubRings[0].INTT(y, back);
      EXPECT(back == *v.in);
      Constants c = m.GetConstants();
      EXPECT(c.omega3n.size() == 1 && c.omega3n[0] == v.w);
      Ring m2(v.N, {v.p}, Type::Matrix, c);                  // constants handoff incl. omega
      Poly pd = m2.NewPoly();
      pd.upload(*v.in);
      m2.NTT(pd, pd);
      EXPECT(pd.download() == *v.out);
    }
    bool noroot = false;                                     // modulus without a primitive 3N-th root: the Go ctor panics (ntt_3n.go:41)
    try { Ring bad(12, {65539ull}, Type::Matrix); } catch (const Error&) { noroot = true; } catch (const Panic&) { noroot = true; }
    EXPECT(noroot);
  }
  // conjugate-invariant ring (ring/ntt.go:80-124, NewRingConjugateInvariant ring/ring.go:276-284): 4N-th-root tables of 2N entries
  {
    const int n = 64;
    const std::vector<uint64_t> mods = {0x1fffffffffe00001ull, 0x1fffffffffc80001ull};
    Ring ci(n, mods, Type::ConjugateInvariant);
    Constants c = ci.GetConstants();
    EXPECT(c.roots_fwd.size() == (size_t)2 * 2 * n);
    Ring ci2(n, mods, Type::ConjugateInvariant, c);
    std::vector<uint64_t> a(n), y(n), y2(n), back(n);
    uint64_t sd = 7;
    for (int j = 0; j < n; ++j) { sd = sd * 6364136223846793005ull + 1442695040888963407ull; a[j] = sd % mods[1]; }
    ci.SubRings[1].NTT(a, y); ci2.SubRings[1].NTT(a, y2);
    EXPECT(y == y2);
    ci.SubRings[1].INTT(y, back);
    EXPECT(back == a);
  }

  // device-resident: c = INTT(NTT(a) * NTT(b)) with MForm + MulCoeffsMontgomery (schemes/ckks/evaluator.go:821-834)
  const int N = 1 << 13;
  const std::vector<uint64_t> mods = {0x1fffffffffe00001ull, 0x1fffffffffc80001ull};
  Ring R(N, mods);
  std::vector<uint64_t> a(2 * N, 0), b(2 * N, 0);
  for (int l = 0; l < 2; ++l) { a[l * N + N - 1] = 3; b[l * N + 2] = 5; }      // 3 X^(N-1) * 5 X^2 = -15 X
  Poly pa = R.NewPoly(), pb = R.NewPoly();
  pa.upload(a); pb.upload(b);
  R.NTT(pa, pa); R.NTT(pb, pb); R.MForm(pa, pa); R.MulCoeffsMontgomery(pa, pb, pa); R.INTT(pa, pa);
  std::vector<uint64_t> c = pa.download();
  for (int l = 0; l < 2; ++l)
    for (int j = 0; j < N; ++j) EXPECT(c[l * N + j] == (j == 1 ? mods[l] - 15 : 0));
  // the same product through Ring.PolyMul (the tile stages of all three transforms as one kernel; operands consumed)
  {
    Poly qa = R.NewPoly(), qb = R.NewPoly(), qc = R.NewPoly();
    qa.upload(a); qb.upload(b);
    R.PolyMul(qa, qb, qc);
    EXPECT(qc.download() == c);
  }

  // Ring.AtLevel(l) on polys allocated at the top level (ring/ring.go:192-213): limbs 0..l transformed, the rest untouched
  {
    Ring lo = R.AtLevel(0);
    std::vector<uint64_t> h(2 * N);
    for (int j = 0; j < 2 * N; ++j) h[j] = (uint64_t)(j * 2654435761u) % mods[j / N];
    Poly full = R.NewPoly(), ref = lo.NewPoly();
    full.upload(h);
    ref.upload(std::vector<uint64_t>(h.begin(), h.begin() + N));
    lo.NTT(full, full); lo.NTT(ref, ref);
    std::vector<uint64_t> g = full.download(), e = ref.download();
    EXPECT(std::equal(e.begin(), e.end(), g.begin()));
    EXPECT(std::equal(h.begin() + N, h.end(), g.begin() + N));
  }

  // key switch (core/rlwe/evaluator_gadget_product.go): direct == hoisted == the shard path with every limb owned
  {
    const int n = 4096;
    const std::vector<uint64_t> Q = {0x1fffffffffe00001ull, 0x1fffffffffc80001ull, 0x1fffffffffb40001ull, 0x1fffffffff500001ull, 0x1fffffffff380001ull};
    const std::vector<uint64_t> P = {0x1ffffffff6c80001ull, 0x1ffffffff6140001ull};
    Ring rq(n, Q), rp(n, P);
    BasisExtender be(rq, rp);
    const int lq = 4, lp = 1, beta = (lq + lp + 1) / (lp + 1), np = 2;
    uint64_t seed = 0x5eed;
    auto fill = [&](std::vector<uint64_t>& v, const std::vector<uint64_t>& mods, int polys) {
      v.resize((size_t)polys * mods.size() * n);
      for (int k = 0; k < polys; ++k) for (size_t l = 0; l < mods.size(); ++l) for (int j = 0; j < n; ++j) {
        seed += 0x9e3779b97f4a7c15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31;
        v[((size_t)k * mods.size() + l) * n + j] = z % mods[l];
      }
    };
    std::vector<uint64_t> hcx, hkq, hkp;
    fill(hcx, Q, np); fill(hkq, Q, beta * 2); fill(hkp, P, beta * 2);
    Poly cx = rq.NewPoly(np), kq = rq.NewPoly(beta * 2), kp = rp.NewPoly(beta * 2);
    cx.upload(hcx); kq.upload(hkq); kp.upload(hkp);
    Poly d0 = rq.NewPoly(np), d1 = rq.NewPoly(np), h0 = rq.NewPoly(np), h1 = rq.NewPoly(np), s0 = rq.NewPoly(np), s1 = rq.NewPoly(np);
    be.GadgetProduct(lq, lp, cx, kq, kp, beta, d0, d1);
    Poly dq = rq.NewPoly(beta * np), dp = rp.NewPoly(beta * np);
    be.DecomposeNTT(lq, lp, cx, true, dq, dp);
    be.GadgetProductHoisted(lq, lp, dq, dp, kq, kp, beta, h0, h1);
    EXPECT(h0.download() == d0.download());
    EXPECT(h1.download() == d1.download());
    KeySwitchShard ks(rq, &rp, Q, P, {0, 1, 2, 3, 4}, {0, 1});
    EXPECT(ks.NumDigits() == beta);
    Poly cxinv = rq.NewPoly(np), a0 = rp.NewPoly(np), a1 = rp.NewPoly(np);
    rq.INTT(cx, cxinv);
    std::vector<uint64_t> inv = cxinv.download();
    for (int d = 0; d < beta; ++d) {
      auto [st, ed] = ks.DigitRange(d);
      std::vector<uint64_t> src((size_t)np * (ed - st) * n);          // with one rank the "gather" is a strided copy
      for (int k = 0; k < np; ++k) for (int l = st; l < ed; ++l)
        std::copy(inv.begin() + ((size_t)k * Q.size() + l) * n, inv.begin() + ((size_t)k * Q.size() + l + 1) * n, src.begin() + ((size_t)k * (ed - st) + (l - st)) * n);
      Ring rsrc(n, std::vector<uint64_t>(Q.begin() + st, Q.begin() + ed));
      Poly psrc = rsrc.NewPoly(np); psrc.upload(src);
      ks.Digit(d, psrc.data(), cx, kq.data(), kp.data(), s0, s1, a0.data(), a1.data());
      rq.Sync();
    }
    for (auto pr : {std::make_pair(&a0, &s0), std::make_pair(&a1, &s1)}) {
      rp.INTTLazy(*pr.first, *pr.first);
      ks.ModDown(pr.first->data(), *pr.second, *pr.second);           // all P limbs are local: the gathered block is the accumulator
    }
    EXPECT(s0.download() == d0.download());
    EXPECT(s1.download() == d1.download());
    // the whole sharded product in ONE call (round 3): a handle that owns every limb needs no map and no all-gather
    Poly w0 = rq.NewPoly(np), w1 = rq.NewPoly(np);
    ks.GadgetProduct(cx, kq.data(), kp.data(), w0, w1, nullptr, nullptr, 2);      // two chunks on the side streams
    rq.Sync();
    EXPECT(w0.download() == d0.download());
    EXPECT(w1.download() == d1.download());
  }

  // round 3: Ring.NTT(p1, p2 Poly) on a whole host poly in one call, the KAT again (ring/ntt_test.go:10-89, N = 16, both limbs)
  {
    std::vector<uint64_t> o0(16), o1(16);
    r.NTT(std::vector<const uint64_t*>{KAT16_POLY_0.data(), KAT16_POLY_1.data()}, std::vector<uint64_t*>{o0.data(), o1.data()});
    EXPECT(o0 == KAT16_NTT_0);
    EXPECT(o1 == KAT16_NTT_1);
    r.INTT(std::vector<const uint64_t*>{o0.data(), o1.data()}, std::vector<uint64_t*>{o0.data(), o1.data()});      // in place
    EXPECT(o0 == KAT16_POLY_0);
    EXPECT(o1 == KAT16_POLY_1);
    bool short_poly = false;
    try { r.NTT(std::vector<const uint64_t*>{KAT16_POLY_0.data()}, std::vector<uint64_t*>{o0.data()}); } catch (const Panic&) { short_poly = true; }
    EXPECT(short_poly);
    EXPECT(r.Stats("rows_poly_by_poly") == 0);
  }

  // error behaviour
  bool panicked = false;
  try { std::vector<uint64_t> s(4), t(16); r.SubRings[0].NTT(s, t); } catch (const Panic&) { panicked = true; }
  EXPECT(panicked);
  bool errored = false;
  try { Ring bad(16, {q + 2}); } catch (const Error&) { errored = true; }
  EXPECT(errored);
  bool lvl = false;
  try { R.AtLevel(5); } catch (const Panic&) { lvl = true; }
  EXPECT(lvl);
  std::printf(fails ? "C++ mirror: %d failure(s)\n" : "C++ mirror: all checks passed\n", fails);
  return fails ? 1 : 0;
}
