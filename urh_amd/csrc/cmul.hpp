// complex64 product exactly as GCC emits it for std::complex<float> / float _Complex (shared by the FIR and Costas kernels)
#pragma once
#include <hip/hip_runtime.h>

namespace urh {

// ---- complex64 product exactly as GCC emits it for std::complex<float> / float _Complex --------------------
__device__ __noinline__ inline float2 mulsc3_recover(float a, float b, float c, float d, float2 r) {
    // libgcc __mulsc3 (C99 G.5.1): only reached when both parts are NaN
    const float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    bool recalc = false;
    if (isinf(a) || isinf(b)) {
        a = copysignf(isinf(a) ? 1.f : 0.f, a); b = copysignf(isinf(b) ? 1.f : 0.f, b);
        if (isnan(c)) c = copysignf(0.f, c);
        if (isnan(d)) d = copysignf(0.f, d);
        recalc = true;
    }
    if (isinf(c) || isinf(d)) {
        c = copysignf(isinf(c) ? 1.f : 0.f, c); d = copysignf(isinf(d) ? 1.f : 0.f, d);
        if (isnan(a)) a = copysignf(0.f, a);
        if (isnan(b)) b = copysignf(0.f, b);
        recalc = true;
    }
    if (!recalc && (isinf(ac) || isinf(bd) || isinf(ad) || isinf(bc))) {
        if (isnan(a)) a = copysignf(0.f, a);
        if (isnan(b)) b = copysignf(0.f, b);
        if (isnan(c)) c = copysignf(0.f, c);
        if (isnan(d)) d = copysignf(0.f, d);
        recalc = true;
    }
    if (recalc) {
        r.x = __builtin_inff() * (a * c - b * d);
        r.y = __builtin_inff() * (a * d + b * c);
    }
    return r;
}
__device__ __forceinline__ float2 cmul(float2 x, float2 h) {
    float2 r;
    r.x = x.x * h.x - x.y * h.y;
    r.y = x.x * h.y + x.y * h.x;
    if (__builtin_expect((r.x != r.x) & (r.y != r.y), 0)) r = mulsc3_recover(x.x, x.y, h.x, h.y, r);
    return r;
}

}  // namespace urh
