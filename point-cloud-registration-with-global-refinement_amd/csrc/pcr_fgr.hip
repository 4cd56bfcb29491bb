// pcr_fgr.hip -- FPFH + Fast Global Registration (K6-K9).  PLACEHOLDER entry points until the kernels land:
// they fail loudly (no CPU fallback).
#include "pcr_device.h"

extern "C" int pcr_compute_fpfh_feature(pcr_context *ctx, const float *, const float *, int64_t, int, int, double, float *) {
    if (!ctx) return PCR_EINVAL;
    ctx->err = "compute_fpfh_feature: not implemented on the MI355X path yet";
    return PCR_EINVAL;
}
extern "C" int pcr_registration_fgr(pcr_context *ctx, const float *, const float *, int64_t, const float *, const float *, int64_t,
                                    const pcr_fgr_option *, pcr_result *, int32_t *) {
    if (!ctx) return PCR_EINVAL;
    ctx->err = "registration_fgr_based_on_feature_matching: not implemented on the MI355X path yet";
    return PCR_EINVAL;
}
