// bext.hip -- RNS basis extension (placeholder until the kernels land in this round)
#include "engine_internal.hpp"
extern "C" int rh_bext_create(rh_bext**, rh_ring*, rh_ring*) { return rh_fail(RH_ERR_UNSUPPORTED, "basis extension not built yet"); }
extern "C" void rh_bext_destroy(rh_bext*) {}
extern "C" int rh_bext_modup_q_to_p(rh_bext*, int, int, const uint64_t*, uint64_t*, int) { return rh_fail(RH_ERR_UNSUPPORTED, "nyi"); }
extern "C" int rh_bext_modup_p_to_q(rh_bext*, int, int, const uint64_t*, uint64_t*, int) { return rh_fail(RH_ERR_UNSUPPORTED, "nyi"); }
extern "C" int rh_bext_moddown_qp_to_q(rh_bext*, int, int, const uint64_t*, const uint64_t*, uint64_t*, int) { return rh_fail(RH_ERR_UNSUPPORTED, "nyi"); }
extern "C" int rh_bext_moddown_qp_to_q_ntt(rh_bext*, int, int, const uint64_t*, const uint64_t*, uint64_t*, int) { return rh_fail(RH_ERR_UNSUPPORTED, "nyi"); }
extern "C" int rh_bext_moddown_qp_to_p(rh_bext*, int, int, const uint64_t*, const uint64_t*, uint64_t*, int) { return rh_fail(RH_ERR_UNSUPPORTED, "nyi"); }
extern "C" int rh_bext_decompose_and_split(rh_bext*, int, int, int, int, const uint64_t*, uint64_t*, uint64_t*, int) { return rh_fail(RH_ERR_UNSUPPORTED, "nyi"); }
