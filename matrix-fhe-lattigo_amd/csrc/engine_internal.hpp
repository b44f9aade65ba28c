// engine_internal.hpp -- shared between the translation units of libringhip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <mutex>
#include <atomic>
#include <cstdint>
#include "../../include/ringhip.h"
#include "ring_types.hip.hpp"

#define RH_MAX_LIMBS 64

int rh_fail(int code, const char* fmt, ...);

struct rh_ring3n_state;   // ntt3n.hip
struct CiFold { tw2 f, b; };   // conjugate-invariant fold twiddles roots_fwd[1], roots_bwd[1] (Shoup pairs)

struct rh_ring {
  int device = 0, kind = 0, N = 0, logN = 0, L = 0;
  hipStream_t stream = nullptr;
  // host copies of what the Go side handed over (SubRing fields)
  std::vector<uint64_t> moduli, mred, bred, ninv, roots_fwd, roots_bwd, omega3n;
  std::vector<LimbConsts> hconsts;
  // device tables
  LimbConsts* d_consts = nullptr;
  tw2* d_tw_fwd = nullptr;        // natural order, Shoup pairs of RootsForward (standard form)
  tw2* d_tw_inv = nullptr;        // natural order, Shoup pairs of RootsBackward
  u64* d_tw_fwd_mont = nullptr;   // natural order, RootsForward as given (Montgomery form)
  tw2* d_twk_fwd = nullptr;       // kernel order (logN >= 12)
  tw2* d_twk_inv = nullptr;
  u64* d_twk_fwd_mont = nullptr;
  u64* d_tw_inv_mont = nullptr;   // N < 16 only: RootsBackward as given, for the non-canonical BackwardLazy of tiny rings
  CiFold* d_cifold = nullptr;     // conjugate-invariant rings only
  tw2* d_lastw = nullptr;         // psi_bwd[1] * N^-1 per limb
  LimbConsts* d_consts_r = nullptr;   // standard rings: d_consts with N^-1 replaced by N^-1 * 2^64 (rh_ring_intt_mul: the inverse
  tw2* d_lastw_r = nullptr;           //   transform of a Montgomery product restores the factor in its last stage), and lastw likewise
  // host-pointer single-limb path (rh_ntt_*): a pool of (stream, scratch) slots, one per concurrent caller
  std::mutex slot_mu;
  std::vector<struct RhHostSlot*> free_slots, all_slots;
  std::vector<struct RhPolySlot*> free_poly_slots, all_poly_slots;   // whole-Poly host path (rh_ntt_poly_*): two streams + staging per concurrent caller
  // serialises the HOST side of the entry points that touch lazily built shared state (rescale tables / scratch, the 3N
  // transform's scratch); device work stays stream-ordered.  Tables, twiddles and tuning are read-only after creation.
  std::recursive_mutex mu;
  rh_ring3n_state* s3n = nullptr;
  std::vector<void*> rescale_tables;      // per level, rescale.hip
  u64* d_rs[2] = {nullptr, nullptr}; size_t rs_words[2] = {0, 0};
  u64* d_rows = nullptr; size_t rows_words = 0;   // dense scratch of the rows-per-poly transforms that compact / expand (engine.hip: ntt_rows)
  std::atomic<long> stats_rows_direct{0}, stats_rows_compacted{0};   // rh_ring_stats: how the rows-per-poly calls were served
  int fuse_submul = 1;            // ModDown / rescale: subtract-multiply fused into the forward tile kernel's epilogue
  int fuse_ci = 1;                // conjugate-invariant ring: the fold inside the column stages (N = 2^14 .. 2^16)
  int digit_pipeline = 1;         // key switch: all digit blocks transformed by one pipelined stream of launches (N = 2^14 .. 2^16)
  int perm_fwd_shape = 44;        // 3N permutation tiles as 10*A + B: block-order runs of 2^A words, rank-order runs of nb * 2^B words
  int perm_inv_shape = 44;        // (4, 4) measured best on MI355X for both directions (12 KiB tiles, 13 workgroups per CU): 1.06 -> 0.91 ms at the config 4 ring
  int fuse3n = 1;                 // 3N rings, b = 1: split + radix-3 layer fused with the sub-transforms' column stages
  int block_order3n = 0;          // 3N rings: device-batched NTT domain kept in block order (ntt3n.hip), no permutation pass
  int asm_cols = 1;               // N = 2^16: hand-scheduled column stages (fwd_cols16_asm_body) in place of the C++ body
  bool asm_tile = true;           // forward tile kernel: hand-scheduled body (ntt_kernels_asm.hip.hpp) vs the C++ one
  bool inv_scale = true;          // false: inverse leaves values < 4q without the N^-1 factor (3N sub-transform)
  int auto_span_rows = 2048;      // chunk_polys = -1: span size of the fused pipeline in (poly, limb) rows
  int ks_small_rows = 512;        // key switch (keyswitch.hip): blocks of at most this many (poly, limb) rows (a few ciphertexts) take the small-batch launches: every digit
                                  // in ONE extension launch and one launch pair of the transforms instead of a chain of ~10 dependent launches (0: never)
  bool pair_submul = true;        // ModDown of a ciphertext: both components' transform + subtract-multiply in ONE launch (false: one launch per component; A/B runs)
  bool one_pass = true;           // N = 2^13 / 2^14: whole limb row in one workgroup's LDS (ntt_fwd_onepass_asm / ntt_inv_onepass_asm); false: the two-pass launches
  bool one_pass_ready = false;    // ... their dynamic-LDS limit has been raised on this ring's device
  bool nt_streams = true;         // non-temporal data streams for launches beyond the Infinity Cache (the generated _NT bodies); false: default policy everywhere
  int chunk_polys = -1;           // -1 = auto (128-poly spans for batches >= 256), 0 = whole batch in two launches, >0 = polys per span
};

// ---- per-call context of the calling thread -----------------------------------------------------------------------
// A handle is shared by concurrent callers (ring/ring.go:192-194: transformers immutable, AtLevel views concurrency-safe), so
// nothing a call needs may live in the handle as mutable state.  The stream a launch goes to and the 3N transform's workspace
// are resolved through thread-local overrides: the host-limb path installs its slot's stream + scratch, a basis extender
// pins both of its rings to ringQ's stream for the duration of the call.
struct RhHostSlot { hipStream_t stream = nullptr; u64* buf = nullptr; size_t words = 0; };
hipStream_t rh_stream(const rh_ring* r);                     // the calling thread's override, else the ring's stream
u64* rh_ws_override(size_t words);                           // workspace of the calling thread's slot (nullptr: none installed)
struct RhCallScope {                                         // installs (stream [, workspace]) for the calling thread; restores on exit
  RhCallScope(hipStream_t st, u64* ws = nullptr, size_t ws_words = 0);
  ~RhCallScope();
  RhCallScope(const RhCallScope&) = delete;
  RhCallScope& operator=(const RhCallScope&) = delete;
 private:
  hipStream_t prev_st; bool prev_has; u64* prev_ws; size_t prev_words;
};

int rh_layout3n(const rh_ring* r);                          // 3N rings: the calling entry point's NTT-domain layout (1 block order, 0 reference), else the tuning value
struct RhLayoutScope { explicit RhLayoutScope(int layout); ~RhLayoutScope(); RhLayoutScope(const RhLayoutScope&) = delete; RhLayoutScope& operator=(const RhLayoutScope&) = delete; private: int prev; };
bool rh_ring3n_block_order_ok(const rh_ring* r);
int rh_std_ntt_launch(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, bool inverse, bool lazy, int phase = 0);
int rh_ring_ntt_any(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, bool inverse);
int rh_std_ntt_fwd_strided(rh_ring* r, u64* data, int npoly, int Lrows, int limb0, int Ls);
bool rh_can_ntt_digits(const rh_ring* r);
int rh_std_ntt_fwd_digits(rh_ring* r, u64* data, size_t digit_stride, int npoly, int beta, int LQ, int LP, bool lazy_out = false, int small = -1);
int rh_std_ntt_fwd_blocks(rh_ring* r, u64* data, size_t block_stride, int npoly, int nblocks, int Ls, const int* gap0, const int* gap_len,
                          bool lazy_out = false, int small = -1);
int rh_std_ntt_fwd_blocks_small(rh_ring* r, u64* data, size_t block_stride, int nblocks, int Ls, const int* gap0, const int* gap_len,
                                rh_ring* r2, u64* data2, size_t block_stride2, int nblocks2, int Ls2, int npoly, bool lazy_out);
bool rh_can_intt_limb_strided(const rh_ring* r);
int rh_std_intt_limb_strided(rh_ring* r, const u64* in, int in_rows, int limb, u64* out, int npoly);
bool rh_can_fuse_submul(const rh_ring* r);
int rh_std_ntt_submul_launch(rh_ring* r, u64* buf, int npoly, int Lrows, int limb0, const u64* y, int y_rows, u64* out, int out_rows,
                             const u64* scalars_host, bool cols_done = false, const u64* z = nullptr, int z_rows = 0);
int rh_std_ntt_submul_launch_pair(rh_ring* r, u64* buf, int npoly, int Lrows, const u64* y0, const u64* y1, int y_rows, u64* out0, u64* out1, int out_rows,
                                  const u64* scalars_host, const u64* z0, const u64* z1, int z_rows);
int rh_std_ntt_expand_cols_launch(rh_ring* r, const u64* tmp, u64* buf, int npoly, int Lrows, const void* table_dev, int mode, u64 qL);
int rh_vec_launch(rh_ring* r, int opcode, const u64* p1, const u64* p2, u64* p3, int npoly, int Lrows, int limb0,
                  const u64* s0, const u64* s1, int rows1 = 0, int rows2 = 0, int rows3 = 0, int half = 0);
int rh_std_intt_rows(rh_ring* r, const u64* in, int in_rows, u64* out, int out_rows, int npoly, int Lrows);
int rh_std_upload_tables(rh_ring* r, const std::vector<tw2>& fs, const std::vector<tw2>& is, const std::vector<u64>* mont,
                         const std::vector<tw2>& lastw);
int rh_upload_consts(rh_ring* r, const std::vector<LimbConsts>& hc);
void rh_rescale_teardown(rh_ring* r);
int rh_rescale_reserve(rh_ring* r, int npoly);
int rh_ring3n_reserve(rh_ring* r, int npoly);
void rh_3n_launch_layer(bool inverse, int S1, unsigned nblocks, hipStream_t st, const u64* in, u64* out, const N3Layer& a, bool nt_streams);
// 3N-cyclotomic transform (ntt3n.hip)
int rh_ring3n_setup(rh_ring* r, std::vector<LimbConsts>& hc);
void rh_ring3n_teardown(rh_ring* r);
void rh_ring3n_set_nt_streams(rh_ring* r, bool on);
int rh_ring3n_ntt_launch(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, int limb0, bool inverse, bool block_order = false);
int rh_ring3n_reorder_launch(rh_ring* r, const u64* in, u64* out, int npoly, int Lrows, bool to_reference);

// basis extender internals shared with keyswitch.hip
struct rh_bext;
rh_ring* rh_bext_ringQ(rh_bext* be);
rh_ring* rh_bext_ringP(rh_bext* be);
int rh_bext_scratch(rh_bext* be, int which, size_t words, u64** out);
std::recursive_mutex& rh_bext_mutex(rh_bext* be);
// A basis extender carries scratch and lazily built plans, like the reference's (ring/basis_extension.go:166-183: one ShallowCopy
// per goroutine); its entry points still serialise their host side on the object, and pin both rings to ringQ's stream.
struct RhBextGuard {
  std::unique_lock<std::recursive_mutex> lk; RhCallScope sc;
  explicit RhBextGuard(rh_bext* be) : lk(rh_bext_mutex(be)), sc(rh_stream(rh_bext_ringQ(be))) {}
};
// ModDownQPtoQNTT with an optional addend: p2Q = [addend +] (p1Q - ext(p1P)) / P   (addend: the ring.Add that follows a key switch)
int rh_bext_moddown_ntt_add(rh_bext* be, int levelQ, int levelP, const u64* p1Q, const u64* p1P, u64* p2Q, int npoly, const u64* addend);
int rh_bext_moddown_ntt_pair(rh_bext* be, int levelQ, int levelP, const u64* q0, const u64* q1, const u64* p1P, u64* out0, u64* out1,
                             int npoly, const u64* add0, const u64* add1);
