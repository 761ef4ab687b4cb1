// filters.hip -- FIR / IIR / Costas-loop / magnitude kernels (rows 2, 3, 5, 6, 7 of SURVEY.md §8a).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "launchers.hpp"

namespace urh {

int launch_costas(urhgpu_ctx *ctx, const void *d_iq, int64_t n, const urhgpu_params *p, float *d_qad) {
    (void)ctx; (void)d_iq; (void)n; (void)p; (void)d_qad;
    return URHGPU_ERR_UNSUPPORTED;
}

}  // namespace urh
