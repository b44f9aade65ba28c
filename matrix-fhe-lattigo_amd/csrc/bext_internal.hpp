// bext_internal.hpp -- basis-extension plan types and helpers shared by bext.hip and kshard.hip (not part of the C ABI).
#pragma once
#include <vector>
#include "engine_internal.hpp"

struct BextSource { u64 q, qinv, qstar_inv /* ((Q/q_i)^-1) Montgomery form */, half /* floor(Q/2) mod q_i */; };
struct BextTarget {
  u64 p, pinv;
  u64 half;        // floor(Q/2) mod p (centred subtraction), used when post >= 1
  u64 md_scalar;   // ModDown: p - (P^-1 mod p) in Montgomery form, used when post == 2
  int buf;         // 0 / 1: which output block
  int limb;        // limb row inside that block
  int post;        // 0 none, 1 CRed(x + p - half), 2 = 1 then MRed(2p - other + x, md_scalar)
  int skip;        // 1: no extension for this limb (digit limbs of DecomposeAndSplit); post 1 applies to prior content
};

enum { BEXT_ADD_NONE = 0, BEXT_ADD_CRED = 1, BEXT_ADD_RAW = 2 };
int rh_bext_decompose_and_split_all(rh_bext* be, int levelQ, int levelP, int nbPi, int beta, const u64* p0Q, u64* p1Q, size_t strideQ,
                                    u64* p1P, size_t strideP, int npoly);      // 0 done, 1 not applicable (go digit by digit), < 0 error

struct SignTarget { u64 p, bred0; int buf, limb; };
struct BextPlan {
  int nsrc = 0, ntgt = 0;
  int ntgt_c = 0;  // targets [0, ntgt_c) are extended, the rest skipped (digit limbs)
  int post = 0;    // the plan's post step (uniform over its targets)
  BextSource* d_S = nullptr; BextTarget* d_T = nullptr; u64* d_coef = nullptr; u64* d_vt = nullptr;
  SignTarget* d_sign = nullptr; u64 qd = 0;      // single-prime digit plan
};


u64 rh_half_product_mod(const std::vector<u64>& M, u64 m);
void rh_gen_modup(const std::vector<u64>& Qs, const std::vector<u64>& tg, std::vector<u64>& qstar_inv_mont, std::vector<u64>& coef,
                  std::vector<u64>& vt);
u64 rh_moddown_const(const std::vector<u64>& Ps, u64 qi);
int rh_bext_upload_plan(BextPlan& p, const std::vector<BextSource>& S, const std::vector<BextTarget>& T, const std::vector<u64>& coef,
                        const std::vector<u64>& vt);
int rh_bext_upload_sign_plan(BextPlan& p, const std::vector<SignTarget>& T, u64 qd);
void rh_bext_free_plan(BextPlan& p);
int rh_bext_launch_raw(hipStream_t st, int N, const BextPlan& p, const u64* in, int in_rows, int src_limb0, u64* out0, int out0_rows,
                       u64* out1, int out1_rows, const u64* other, int other_rows, int npoly, int add_mode);
int rh_bext_launch_sign(hipStream_t st, int N, const BextPlan& p, const u64* in, int in_rows, int src_limb, u64* out0, int out0_rows,
                        u64* out1, int out1_rows, int npoly);
// keyswitch.hip: acc_c (=|+=) MRedLazy(evk_c, c2) for both components
int rh_gadget_mac(rh_ring* r, const u64* c2, const u64* e0, const u64* e1, u64* a0, u64* a1, int npoly, int L, int first);
int rh_overflow_margin(const std::vector<u64>& m, int level);
// keyswitch.hip: all beta digits in one pass, accumulators in registers, the reference's Reduce schedule; see the definition
int rh_gadget_mac_all(rh_ring* r, const u64* c2, size_t digit_stride, const u64* evk, int beta, int overf, u64* a0, u64* a1, int npoly, int L,
                      const u64* cx, const int* own_digit, int digit_limbs);
