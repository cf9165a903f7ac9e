/* Exhaustive check: orbx_sincosf_pinned() == libm sinf/cosf for every float in [0, 6.5].
 * gcc -O2 -ffp-contract=off -fopenmp tools/check_sincos.c -o /tmp/check_sincos -lm && /tmp/check_sincos [stride]
 * (stride > 1 samples every stride-th float; tests/test_sincos_pin.py uses a stride to stay fast) */
#include "../orb_slam2_detailed_comments_amd/csrc/orbx_sincos.h"
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char **argv) {
    uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
    float hi = 6.5f; uint32_t uhi; memcpy(&uhi, &hi, 4);
    long bad = 0, n = 0;
#pragma omp parallel for reduction(+:bad,n) schedule(static)
    for (uint32_t u = 0; u <= uhi; u += stride) {
        float f; memcpy(&f, &u, 4);
        struct OrbxSinCos r = orbx_sincosf_pinned(f);
        float a = sinf(f), b = cosf(f);
        if (memcmp(&a, &r.s, 4) || memcmp(&b, &r.c, 4)) bad++;
        n++;
    }
    printf("checked=%ld mismatches=%ld\n", n, bad);
    return bad != 0;
}
