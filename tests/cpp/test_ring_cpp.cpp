// C++ conformance test of the host-side mirror (include/ringhip.hpp), written the way the reference's TestNTT is
// (ring/ntt_test.go:91-121): NewRing, NTT(poly) == polyNTT, INTT(NTT(poly)) == poly -- on the N=16 known-answer vector
// of ring/ntt_test.go (first limb), plus a device-resident poly-mul and the panic/error behaviour.
#include <algorithm>
#include <cstdio>
#include <utility>
#include <vector>
#include "ringhip.hpp"

using namespace ringhip;

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main() {
  const uint64_t q = 576460752303439873ull;       // Qis[0] of the reference's testVector
  const std::vector<uint64_t> poly = {75484882814757497ull, 568962213571011535ull, 219781953812704432ull, 431409500923750484ull,
                                      91669795593397493ull, 473049842650292367ull, 213965426418426961ull, 195843195830982574ull,
                                      359420738741905339ull, 134609403297510286ull, 546063636015790939ull, 475349391419928453ull,
                                      305803859266883227ull, 434608786379655593ull, 233020405896920653ull, 421812738743799064ull};
  Ring r(16, {q});
  std::vector<uint64_t> y(16), back(16);
  r.SubRings[0].NTT(poly, y);
  r.SubRings[0].INTT(y, back);
  EXPECT(back == poly);
  for (uint64_t v : y) EXPECT(v < q);

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
