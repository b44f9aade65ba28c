// C++ conformance test of the host-side mirror (include/ringhip.hpp), written the way the reference's TestNTT is
// (ring/ntt_test.go:91-121): NewRing, NTT(poly) == polyNTT, INTT(NTT(poly)) == poly -- on the N=16 known-answer vector
// of ring/ntt_test.go (first limb), plus a device-resident poly-mul and the panic/error behaviour.
#include <cstdio>
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
