// Finds arguments on which the FMA and the non-FMA evaluation of glibc's sinf / cosf round differently (about one float in 10^9):
// the discriminating inputs of urhgpu_host_libm_check (capi.hip).  Scans every float with |x| < 120.
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
float fma_sin(float), fma_cos(float), nof_sin(float), nof_cos(float);
int main(void) {
    uint32_t lim; float f = 120.0f; memcpy(&lim, &f, 4);
    long found = 0;
#pragma omp parallel for schedule(dynamic, 1 << 20)
    for (uint32_t u = 0x30000000u; u < lim; ++u) {
        for (int s = 0; s < 2; ++s) {
            uint32_t v = u | ((uint32_t)s << 31);
            float x; memcpy(&x, &v, 4);
            float a = fma_sin(x), b = nof_sin(x), c = fma_cos(x), d = nof_cos(x);
            uint32_t ua, ub, uc, ud; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4); memcpy(&uc, &c, 4); memcpy(&ud, &d, 4);
            if (ua != ub || uc != ud) {
                float hs = sinf(x), hc = cosf(x);
                uint32_t us, uh; memcpy(&us, &hs, 4); memcpy(&uh, &hc, 4);
#pragma omp critical
                { printf("0x%08x %s host_is_%s\n", v, ua != ub ? "sin" : "cos", (ua != ub ? us == ua : uh == uc) ? "fma" : "nofma"); found++; }
            }
        }
    }
    fprintf(stderr, "found %ld\n", found);
    return 0;
}
