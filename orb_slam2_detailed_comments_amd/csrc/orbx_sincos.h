// orbx_sincos.h -- sinf / cosf exactly as glibc >= 2.28 computes them for 0 <= x < 120 (double-precision
// polynomial after a pi/2 range reduction; algorithm of ARM optimized-routines sincosf, which is what
// glibc's sysdeps/ieee754/flt-32/s_sinf.c / s_cosf.c ship).  The reference calls cos()/sin() of the host
// libm on a float (src/ORBextractor.cc:186-187), so descriptor bits depend on these exact values.
//
// Verified bit-identical to this image's libm (glibc 2.35) for EVERY float in [0, 6.5]
// (1,087,373,313 values; tools/check_sincos.c).  The descriptor angle is always in [0, 2*pi].
// Only *, fma and conversions are used, so host (tests) and device evaluate identical IEEE operations.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__
#else
#define ORBX_HD
#endif

struct OrbxSinCos { float s, c; };

ORBX_HD static inline uint32_t orbx_abstop12(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return (u >> 20) & 0x7ff;
}

ORBX_HD static inline float orbx_sincos_poly(double x, double x2, int neg, int n) {
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                 C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double sg = neg ? -1.0 : 1.0;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = fma(x2, S3, S2);
        double x7 = x3 * x2;
        double s = fma(x3, S1, x);
        return (float)fma(x7, s1, s);
    } else {
        double x4 = x2 * x2;
        double c2 = fma(x2, sg * C4, sg * C3);
        double c1 = fma(x2, sg * C1, sg * C0);
        double x6 = x4 * x2;
        double c = fma(x4, sg * C2, c1);
        return (float)fma(x6, c2, c);
    }
}

// valid for 0 <= y < 120
ORBX_HD static inline struct OrbxSinCos orbx_sincosf_pinned(float y) {
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    struct OrbxSinCos r;
    double x = (double)y;
    if (orbx_abstop12(y) < orbx_abstop12(0x1.921FB6p-1f)) {
        if (orbx_abstop12(y) < orbx_abstop12(0x1p-12f)) { r.s = y; r.c = 1.0f; return r; }
        double x2 = x * x;
        r.s = orbx_sincos_poly(x, x2, 0, 0);
        r.c = orbx_sincos_poly(x, x2, 0, 1);
        return r;
    }
    double q = x * HPI_INV;
    int n = ((int32_t)q + 0x800000) >> 24;
    x = fma(-(double)n, HPI, x);
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    const int neg = (n & 2) != 0;
    r.s = orbx_sincos_poly(x * sgn, x * x, neg, n);
    r.c = orbx_sincos_poly(x * sgn, x * x, neg, n ^ 1);
    return r;
}
