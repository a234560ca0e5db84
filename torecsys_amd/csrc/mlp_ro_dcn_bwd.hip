// one instantiation of the row-owner fused MLP kernel (mlp_ro.hpp) per file: each takes minutes to compile
#include "mlp_ro.hpp"

namespace trs {

int ro_launch_dcn_bwd(const RoArgs& a, hipStream_t s) { return ro_launch<RoDcn, true, 64, RO_BWD_RT>(a, s); }

}  // namespace trs
