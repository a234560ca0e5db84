// one instantiation of the row-owner fused MLP kernel (mlp_ro.hpp) per file: each takes minutes to compile
#include "mlp_ro.hpp"

namespace trs {

int ro_launch_tail_bwd(const RoArgs& a, hipStream_t s) { return ro_launch<RoTailB, true, 8, RO_BWD_RT>(a, s); }

}  // namespace trs
