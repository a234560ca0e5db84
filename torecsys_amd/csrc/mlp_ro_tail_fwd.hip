// one instantiation of the row-owner fused MLP kernel (mlp_ro.hpp) per file: each takes minutes to compile
#include "mlp_ro.hpp"

namespace trs {

int ro_launch_tail_fwd(const RoArgs& a, hipStream_t s) { return ro_launch<RoTail, false, 416, 1>(a, s); }

}  // namespace trs
