// eagle_vara_i8.hip -- vara_i = m_i^T W m_i from exact int8 slices of W on v_mfma_i32_32x32x32_i8.
// (placeholder until the kernel lands; the fp64 MFMA path in eagle_kernels.hip is the default)
#include <hip/hip_runtime.h>
#include "../../include/eagle_hip.h"
#include "eagle_internal.h"
extern "C" int64_t eagle_vara_i8_workspace_bytes(long n_pad, long L_pad, int nslices) { return 16; }
extern "C" int eagle_dev_vara_i8(eagle_ctx* ctx, const int8_t* Mt8, long L_pad, long n_pad, long ld, const double* Wu,
                                 int nslices, void* ws, double* vara_out, double* err_bound_dev, void* stream) {
    return eagle_fail(ctx, EAGLE_ERR_ARG, "vara_i8 not built yet");
}
