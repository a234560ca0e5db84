// Entry points declared in include/trs_abi.h whose kernels are not written yet.
#include "trs_common.hpp"
using namespace trs;
#define TRS_TODO(name) return fail(TRS_ESHAPE, name ": not implemented in this build")
extern "C" size_t trs_bucket_workspace_bytes(int64_t, int32_t) { return 0; }
extern "C" int trs_bucket_by_owner(const void*, int32_t, const int64_t*, int64_t, int32_t, int64_t, int32_t, int64_t*, int64_t*, int32_t*, void*, size_t, trs_stream_t) { TRS_TODO("bucket_by_owner"); }
