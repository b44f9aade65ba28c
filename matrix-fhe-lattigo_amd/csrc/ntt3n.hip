// ntt3n.hip -- 3N-cyclotomic transform (placeholder until the kernels land in this round)
#include "engine_internal.hpp"
int rh_ring3n_setup(rh_ring*, std::vector<LimbConsts>&) { return rh_fail(RH_ERR_UNSUPPORTED, "3N ring not built yet"); }
void rh_ring3n_teardown(rh_ring*) {}
int rh_ring3n_ntt_launch(rh_ring*, const u64*, u64*, int, int, int, bool) { return rh_fail(RH_ERR_UNSUPPORTED, "3N ring not built yet"); }
