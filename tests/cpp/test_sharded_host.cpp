// A COMPILED host driving the limb-sharded key switch through the C ABI alone -- the shape of the cgo host INTEGRATION.md 2b describes, with
// host threads standing in for the ranks of a node and a plain-C all-gather callback standing in for ncclAllGather (same semantics: every
// rank contributes `words` words and receives world * words, rank-major, enqueued on the stream the library names).  No Python, no torch.
//
// rlwe.Evaluator.GadgetProduct (core/rlwe/evaluator_gadget_product.go:16-30) of a batch: every rank owns limbs {i : i mod G = r} of Q ++ P and
// calls rh_kshard_gadget_product once; each owned limb must equal the same limb of the unsharded rh_bext_gadget_product bit for bit.
// Cases: 4 ranks over 5 + 2 limbs (two ranks own no P limb), 3 ranks over 7 + 3 limbs with digits of 3, 3, 1 limbs; 1 and 4 chunks.
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "ringhip.h"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s (%s)\n", __FILE__, __LINE__, #c, rh_last_error()); ++fails; } } while (0)
#define HIPOK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { std::printf("HIP %s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); ++fails; } } while (0)

static const uint64_t QI60[] = {0x1fffffffffe00001, 0x1fffffffffc80001, 0x1fffffffffb40001, 0x1fffffffff500001, 0x1fffffffff380001, 0x1fffffffff000001,
                                0x1ffffffffef00001, 0x1ffffffffee80001};
static const uint64_t PI60[] = {0x1ffffffff6c80001, 0x1ffffffff6140001, 0x1ffffffff5f40001};

static uint64_t sm64(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static void fill_mod(uint64_t* p, size_t n, uint64_t q, uint64_t& seed) { for (size_t i = 0; i < n; ++i) p[i] = (sm64(seed) >> 2) % q; }

struct Node {                                   // what an RCCL communicator is to a real host: the meeting point of the ranks' collectives
  int world;
  pthread_barrier_t bar;
  std::vector<const uint64_t*> send; std::vector<hipEvent_t> ready, done;
  int calls = 0;
};
struct RankCtx { Node* node; int rank; };

// rh_allgather_fn: stream-ordered all-gather among the threads of this process (device-to-device copies on the caller's stream)
static int allgather_cb(void* ctx, const uint64_t* send, uint64_t* recv, size_t words, void* stream) {
  RankCtx* c = static_cast<RankCtx*>(ctx);
  Node* n = c->node;
  hipStream_t st = static_cast<hipStream_t>(stream);
  n->send[c->rank] = send;
  if (hipEventRecord(n->ready[c->rank], st) != hipSuccess) return 1;            // my block is complete once the stream reaches this point
  pthread_barrier_wait(&n->bar);
  for (int p = 0; p < n->world; ++p) {
    if (hipStreamWaitEvent(st, n->ready[p], 0) != hipSuccess) return 2;
    if (hipMemcpyAsync(recv + (size_t)p * words, n->send[p], words * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) return 3;
  }
  if (hipEventRecord(n->done[c->rank], st) != hipSuccess) return 4;             // I have read every peer's block
  pthread_barrier_wait(&n->bar);
  for (int p = 0; p < n->world; ++p) if (hipStreamWaitEvent(st, n->done[p], 0) != hipSuccess) return 5;   // nobody overwrites a block still being read
  pthread_barrier_wait(&n->bar);
  if (c->rank == 0) ++n->calls;
  return 0;
}

static void run_case(int N, int nq, int np_, int world, int npoly, int chunks) {
  const int beta = (nq - 1 + np_) / np_;
  std::vector<uint64_t> Q(QI60, QI60 + nq), P(PI60, PI60 + np_);
  uint64_t seed = 0x5eed + (uint64_t)N + nq * 131 + world;
  std::vector<uint64_t> cx((size_t)npoly * nq * N), kq((size_t)beta * 2 * nq * N), kp((size_t)beta * 2 * np_ * N);
  for (int k = 0; k < npoly; ++k) for (int i = 0; i < nq; ++i) fill_mod(&cx[((size_t)k * nq + i) * N], N, Q[i], seed);
  for (int e = 0; e < beta * 2; ++e) {
    for (int i = 0; i < nq; ++i) fill_mod(&kq[((size_t)e * nq + i) * N], N, Q[i], seed);
    for (int j = 0; j < np_; ++j) fill_mod(&kp[((size_t)e * np_ + j) * N], N, P[j], seed);
  }
  // the unsharded product (what tests/test_gpu_keyswitch.py pins to the oracle composition)
  std::vector<uint64_t> want0(cx.size()), want1(cx.size());
  {
    rh_ring *rq = nullptr, *rp = nullptr; rh_bext* be = nullptr;
    EXPECT(rh_ring_create_auto(&rq, 0, RH_RING_STANDARD, N, nq, Q.data(), nullptr) == 0);
    EXPECT(rh_ring_create_auto(&rp, 0, RH_RING_STANDARD, N, np_, P.data(), nullptr) == 0);
    EXPECT(rh_bext_create(&be, rq, rp) == 0);
    uint64_t *dcx, *dkq, *dkp, *d0, *d1;
    EXPECT(rh_dev_alloc(rq, cx.size(), &dcx) == 0); EXPECT(rh_dev_alloc(rq, kq.size(), &dkq) == 0); EXPECT(rh_dev_alloc(rq, kp.size(), &dkp) == 0);
    EXPECT(rh_dev_alloc(rq, cx.size(), &d0) == 0); EXPECT(rh_dev_alloc(rq, cx.size(), &d1) == 0);
    EXPECT(rh_dev_upload(rq, dcx, cx.data(), cx.size()) == 0); EXPECT(rh_dev_upload(rq, dkq, kq.data(), kq.size()) == 0); EXPECT(rh_dev_upload(rq, dkp, kp.data(), kp.size()) == 0);
    EXPECT(rh_bext_gadget_product(be, nq - 1, np_ - 1, dcx, dkq, dkp, beta, d0, d1, npoly) == 0);
    EXPECT(rh_dev_download(rq, want0.data(), d0, want0.size()) == 0); EXPECT(rh_dev_download(rq, want1.data(), d1, want1.size()) == 0);
    for (uint64_t* p : {dcx, dkq, dkp, d0, d1}) rh_dev_free(rq, p);
    rh_bext_destroy(be); rh_ring_destroy(rq); rh_ring_destroy(rp);
  }
  Node node; node.world = world; node.send.assign(world, nullptr); node.ready.resize(world); node.done.resize(world);
  pthread_barrier_init(&node.bar, nullptr, (unsigned)world);
  for (int r = 0; r < world; ++r) { HIPOK(hipEventCreateWithFlags(&node.ready[r], hipEventDisableTiming)); HIPOK(hipEventCreateWithFlags(&node.done[r], hipEventDisableTiming)); }
  std::vector<int> rank_fail(world, 0);
  auto rank_main = [&](int r) {
    int bad = 0;
#define RCHK(c) do { if (!(c)) { std::printf("rank %d FAIL line %d: %s (%s)\n", r, __LINE__, #c, rh_last_error()); ++bad; } } while (0)
    HIPOK(hipSetDevice(0));
    std::vector<int> owner(nq + np_), ownQ, ownP;
    for (int i = 0; i < nq + np_; ++i) { owner[i] = i % world; if (owner[i] == r) (i < nq ? ownQ : ownP).push_back(i < nq ? i : i - nq); }
    std::vector<uint64_t> mq, mp; for (int i : ownQ) mq.push_back(Q[i]); for (int j : ownP) mp.push_back(P[j]);
    rh_ring *rq = nullptr, *rp = nullptr; rh_kshard* ks = nullptr;
    hipStream_t st; HIPOK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    RCHK(rh_ring_create_auto(&rq, 0, RH_RING_STANDARD, N, (int)mq.size(), mq.data(), nullptr) == 0);
    if (!mp.empty()) RCHK(rh_ring_create_auto(&rp, 0, RH_RING_STANDARD, N, (int)mp.size(), mp.data(), nullptr) == 0);
    RCHK(rh_ring_set_stream(rq, st) == 0); if (rp) RCHK(rh_ring_set_stream(rp, st) == 0);
    RCHK(rh_kshard_create(&ks, rq, rp, Q.data(), nq - 1, P.data(), np_ - 1, ownQ.data(), (int)ownQ.size(), ownP.empty() ? ownQ.data() : ownP.data(), (int)ownP.size()) == 0);
    RCHK(rh_kshard_set_world(ks, world, r, owner.data()) == 0);
    const size_t nQ = ownQ.size(), nP = ownP.size();
    std::vector<uint64_t> lcx((size_t)npoly * nQ * N), lkq((size_t)beta * 2 * nQ * N), lkp((size_t)beta * 2 * (nP ? nP : 1) * N);
    for (int k = 0; k < npoly; ++k) for (size_t a = 0; a < nQ; ++a) memcpy(&lcx[((size_t)k * nQ + a) * N], &cx[((size_t)k * nq + ownQ[a]) * N], (size_t)N * 8);
    for (int e = 0; e < beta * 2; ++e) {
      for (size_t a = 0; a < nQ; ++a) memcpy(&lkq[((size_t)e * nQ + a) * N], &kq[((size_t)e * nq + ownQ[a]) * N], (size_t)N * 8);
      for (size_t a = 0; a < nP; ++a) memcpy(&lkp[((size_t)e * nP + a) * N], &kp[((size_t)e * np_ + ownP[a]) * N], (size_t)N * 8);
    }
    uint64_t *dcx, *dkq, *dkp = nullptr, *d0, *d1;
    RCHK(rh_dev_alloc(rq, lcx.size(), &dcx) == 0); RCHK(rh_dev_alloc(rq, lkq.size(), &dkq) == 0); RCHK(rh_dev_alloc(rq, lcx.size(), &d0) == 0); RCHK(rh_dev_alloc(rq, lcx.size(), &d1) == 0);
    RCHK(rh_dev_upload(rq, dcx, lcx.data(), lcx.size()) == 0); RCHK(rh_dev_upload(rq, dkq, lkq.data(), lkq.size()) == 0);
    if (nP) { RCHK(rh_dev_alloc(rq, lkp.size(), &dkp) == 0); RCHK(rh_dev_upload(rq, dkp, lkp.data(), lkp.size()) == 0); }
    RankCtx ctx{&node, r};
    RCHK(rh_kshard_gadget_product(ks, dcx, dkq, dkp, d0, d1, npoly, allgather_cb, &ctx, chunks) == 0);
    RCHK(rh_ring_sync(rq) == 0);
    std::vector<uint64_t> g0(lcx.size()), g1(lcx.size());
    RCHK(rh_dev_download(rq, g0.data(), d0, g0.size()) == 0); RCHK(rh_dev_download(rq, g1.data(), d1, g1.size()) == 0);
    for (int k = 0; k < npoly; ++k) for (size_t a = 0; a < nQ; ++a) {
      const size_t lo = ((size_t)k * nQ + a) * N, go = ((size_t)k * nq + ownQ[a]) * N;
      if (memcmp(&g0[lo], &want0[go], (size_t)N * 8) || memcmp(&g1[lo], &want1[go], (size_t)N * 8)) { std::printf("rank %d: poly %d limb %d differs from the unsharded product\n", r, k, ownQ[a]); ++bad; }
    }
    for (uint64_t* p : {dcx, dkq, dkp, d0, d1}) if (p) rh_dev_free(rq, p);
    rh_kshard_destroy(ks); rh_ring_destroy(rq); if (rp) rh_ring_destroy(rp);
    HIPOK(hipStreamDestroy(st));
    rank_fail[r] = bad;
#undef RCHK
  };
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r) th.emplace_back(rank_main, r);
  for (auto& t : th) t.join();
  for (int r = 0; r < world; ++r) fails += rank_fail[r];
  const int nc = chunks > 0 ? (chunks > npoly ? npoly : chunks) : (npoly >= 4 ? 4 : 1);
  const int pc = (npoly + nc - 1) / nc, run = (npoly + pc - 1) / pc;
  EXPECT(node.calls == 2 * run);                                                  // two exchanges per chunk, on every rank, in the same order
  std::printf("N=%d Q=%d P=%d ranks=%d polys=%d chunks=%d: %d exchanges, %s\n", N, nq, np_, world, npoly, chunks, node.calls, fails ? "FAILED" : "ok");
  for (int r = 0; r < world; ++r) { (void)hipEventDestroy(node.ready[r]); (void)hipEventDestroy(node.done[r]); }
  pthread_barrier_destroy(&node.bar);
}

int main() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { std::printf("no GPU\n"); return 2; }
  run_case(4096, 5, 2, 4, 5, 0);          // auto chunks (4): chunks of 2, 2, 1; ranks 0 and 3 own no P limb
  run_case(4096, 5, 2, 4, 5, 1);
  run_case(1 << 14, 7, 3, 3, 6, 3);       // digits of 3, 3, 1 limbs; the pipelined digit-block transform
  run_case(64, 6, 2, 2, 3, 2);
  if (fails) { std::printf("%d failure(s)\n", fails); return 1; }
  std::printf("all checks passed\n");
  return 0;
}
