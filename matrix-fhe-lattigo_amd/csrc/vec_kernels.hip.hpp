// vec_kernels.hip.hpp -- the element-wise kernel family of ring/vec_ops.go on (poly, limb, coefficient) blocks.
//
// One template instantiation per opcode (include/ringhip_ops.h); each thread moves 16 B per operand per step
// (two coefficients), rows (= one limb of one poly) are mapped to blockIdx.y so the per-limb constants are
// wave-uniform.  Every formula is the reference's, including the NON-reduction of the lazy forms, so results are
// bit-identical.  HBM-bound: 8*(operands)*N*L bytes per poly.
#pragma once
#include "modarith.hip.hpp"
#include "ntt_kernels.hip.hpp"
#include "../../include/ringhip_ops.h"

template <int OP>
RH_DEV u64 vec_apply(u64 x, u64 y, u64 z, u64 s0, u64 s1, const LimbConsts& c) {
  const u64 q = c.q, qi = c.qinv, q2 = 2 * c.q;
  switch (OP) {
    case RH_OP_ADD: return cred(x + y, q);
    case RH_OP_ADD_LAZY: return x + y;
    case RH_OP_SUB: return cred((x + q) - y, q);
    case RH_OP_SUB_LAZY: return x + q - y;
    case RH_OP_NEG: return q - x;
    case RH_OP_REDUCE: return bred_add(x, q, c.bred0);
    case RH_OP_REDUCE_LAZY: return bred_add_lazy(x, q, c.bred0);
    case RH_OP_MUL_LAZY: return x * y;
    case RH_OP_MUL_LAZY_THEN_ADD_LAZY: return z + x * y;
    case RH_OP_MUL_BARRETT: return bred(x, y, q, c.bred0, c.bred1);
    case RH_OP_MUL_BARRETT_LAZY: return bred_lazy(x, y, q, c.bred0, c.bred1);
    case RH_OP_MUL_BARRETT_THEN_ADD: return cred(z + bred(x, y, q, c.bred0, c.bred1), q);
    case RH_OP_MUL_BARRETT_THEN_ADD_LAZY: return z + bred(x, y, q, c.bred0, c.bred1);
    case RH_OP_MUL_MONT: return mred(x, y, q, qi);
    case RH_OP_MUL_MONT_LAZY: return mred_lazy(x, y, q, qi);
    case RH_OP_MUL_MONT_THEN_ADD: return cred(z + mred(x, y, q, qi), q);
    case RH_OP_MUL_MONT_THEN_ADD_LAZY: return z + mred(x, y, q, qi);
    case RH_OP_MUL_MONT_LAZY_THEN_ADD_LAZY: return z + mred_lazy(x, y, q, qi);
    case RH_OP_MUL_MONT_THEN_SUB: return cred(z + (q - mred(x, y, q, qi)), q);
    case RH_OP_MUL_MONT_THEN_SUB_LAZY: return z + (q - mred(x, y, q, qi));
    case RH_OP_MUL_MONT_LAZY_THEN_SUB_LAZY: return z + q2 - mred_lazy(x, y, q, qi);
    case RH_OP_MUL_MONT_LAZY_THEN_NEG: return q2 - mred_lazy(x, y, q, qi);
    case RH_OP_ADD_LAZY_THEN_MUL_SCALAR_MONT: return mred(x + y, s0, q, qi);
    case RH_OP_ADD_SCALAR_LAZY_THEN_MUL_SCALAR_MONT: return mred(x + s0, s1, q, qi);
    case RH_OP_ADD_SCALAR: return cred(x + s0, q);
    case RH_OP_ADD_SCALAR_LAZY: return x + s0;
    case RH_OP_ADD_SCALAR_LAZY_THEN_NEG_TWO_MODULUS_LAZY: return s0 + q2 - x;
    case RH_OP_SUB_SCALAR: return cred(x + q - s0, q);
    case RH_OP_MUL_SCALAR_MONT: return mred(x, s0, q, qi);
    case RH_OP_MUL_SCALAR_MONT_LAZY: return mred_lazy(x, s0, q, qi);
    case RH_OP_MUL_SCALAR_MONT_THEN_ADD: return cred(z + mred(x, s0, q, qi), q);
    case RH_OP_MUL_SCALAR_MONT_THEN_ADD_SCALAR: return cred(mred(x, s1, q, qi) + s0, q);
    case RH_OP_SUB_THEN_MUL_SCALAR_MONT_TWO_MODULUS: return mred(q2 - y + x, s0, q, qi);
    case RH_OP_MFORM: return mform(x, q, c.bred0, c.bred1);
    case RH_OP_MFORM_LAZY: return mform_lazy(x, q, c.bred0, c.bred1);
    case RH_OP_IMFORM: return imform(x, q, qi);
    case RH_OP_ZERO: return 0;
    case RH_OP_MASK: return (x >> s0) & s1;
  }
  return 0;
}

constexpr bool op_reads_y(int op) {
  return op == RH_OP_ADD || op == RH_OP_ADD_LAZY || op == RH_OP_SUB || op == RH_OP_SUB_LAZY ||
         (op >= RH_OP_MUL_LAZY && op <= RH_OP_ADD_LAZY_THEN_MUL_SCALAR_MONT) ||
         op == RH_OP_SUB_THEN_MUL_SCALAR_MONT_TWO_MODULUS;
}
constexpr bool op_reads_z(int op) {
  return op == RH_OP_MUL_LAZY_THEN_ADD_LAZY || op == RH_OP_MUL_BARRETT_THEN_ADD ||
         op == RH_OP_MUL_BARRETT_THEN_ADD_LAZY || (op >= RH_OP_MUL_MONT_THEN_ADD && op <= RH_OP_MUL_MONT_LAZY_THEN_SUB_LAZY) ||
         op == RH_OP_MUL_SCALAR_MONT_THEN_ADD;
}
constexpr bool op_reads_x(int op) { return op != RH_OP_ZERO; }

// scalars: s0s/s1s are per-limb arrays (length L) or null -> 0.  n = coefficients per row (multiple of 2).
// limb of row r is (limb0 + r % L): `L` rows per poly.
template <int OP>
__global__ void __launch_bounds__(256)
vec_op_kernel(const u64* p1, const u64* p2, u64* p3, size_t n,
              const u64* __restrict__ s0s, const u64* __restrict__ s1s, const LimbConsts* __restrict__ consts, int L) {
  const u32 row = blockIdx.y;
  const u32 limb = row % (u32)L;
  const LimbConsts c = consts[limb];
  const u64 s0 = s0s ? s0s[limb] : 0, s1 = s1s ? s1s[limb] : 0;
  const size_t rowoff = (size_t)row * n;
  const size_t npairs = n >> 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = rowoff + 2 * i;
    ulonglong2 x = make_ulonglong2(0, 0), y = x, z = x;
    if (op_reads_x(OP)) x = *reinterpret_cast<const ulonglong2*>(p1 + o);
    if (op_reads_y(OP)) y = *reinterpret_cast<const ulonglong2*>(p2 + o);
    if (op_reads_z(OP)) z = *reinterpret_cast<const ulonglong2*>(p3 + o);
    ulonglong2 w;
    w.x = vec_apply<OP>(x.x, y.x, z.x, s0, s1, c);
    w.y = vec_apply<OP>(x.y, y.y, z.y, s0, s1, c);
    *reinterpret_cast<ulonglong2*>(p3 + o) = w;
  }
}
