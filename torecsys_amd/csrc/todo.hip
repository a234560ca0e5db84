// Entry points declared in include/trs_abi.h whose kernels are not written yet.
#include "trs_common.hpp"
using namespace trs;
#define TRS_TODO(name) return fail(TRS_ESHAPE, name ": not implemented in this build")
extern "C" int trs_pair_dot_fwd(const void*, int64_t, int32_t, int32_t, int32_t, void*, trs_stream_t) { TRS_TODO("pair_dot_fwd"); }
extern "C" int trs_pair_dot_bwd(const void*, const void*, int64_t, int32_t, int32_t, int32_t, void*, trs_stream_t) { TRS_TODO("pair_dot_bwd"); }
extern "C" int trs_ffm_fwd(const void*, int64_t, int32_t, int32_t, int32_t, void*, trs_stream_t) { TRS_TODO("ffm_fwd"); }
extern "C" int trs_ffm_bwd(const void*, const void*, int64_t, int32_t, int32_t, int32_t, void*, trs_stream_t) { TRS_TODO("ffm_bwd"); }
extern "C" int trs_ffm_fused_fwd(const void* const*, int64_t, int32_t, int32_t, const void*, int32_t, const int64_t*, int64_t, int32_t, void*, int32_t*, trs_stream_t) { TRS_TODO("ffm_fused_fwd"); }
extern "C" size_t trs_cross_workspace_bytes(int64_t, int32_t, int32_t, int32_t) { return 0; }
extern "C" int trs_cross_fwd(const void*, const void*, const void*, int64_t, int32_t, int32_t, int32_t, void*, trs_stream_t) { TRS_TODO("cross_fwd"); }
extern "C" int trs_cross_bwd(const void*, const void*, const void*, const void*, int64_t, int32_t, int32_t, int32_t, void*, float*, float*, void*, size_t, trs_stream_t) { TRS_TODO("cross_bwd"); }
extern "C" int trs_cin_fwd(const void*, const void*, const void*, const void*, int64_t, int32_t, int32_t, int32_t, int32_t, int32_t, void*, float*, trs_stream_t) { TRS_TODO("cin_fwd"); }
extern "C" int trs_cin_bwd(const void*, const void*, const void*, const void*, int64_t, int32_t, int32_t, int32_t, int32_t, int32_t, float*, void*, void*, int32_t, trs_stream_t) { TRS_TODO("cin_bwd"); }
extern "C" size_t trs_bucket_workspace_bytes(int64_t, int32_t) { return 0; }
extern "C" int trs_bucket_by_owner(const void*, int32_t, const int64_t*, int64_t, int32_t, int64_t, int32_t, int64_t*, int64_t*, int32_t*, void*, size_t, trs_stream_t) { TRS_TODO("bucket_by_owner"); }
