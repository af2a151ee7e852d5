// srt_powf.h -- the project-wide definition of pow() on the render path (host + device).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SRT_HD __host__ __device__ inline
#else
#define SRT_HD inline
#endif

namespace srt {

// ------------------------------------------------------------------------------------------------
// srt_powf: the project-wide pow().  Reference call sites materials/material.cu:48
// (pow(1-cos, 5)) and color/color.cu:19 (pow(v, 0.416666)).  exp(y log x) in fp64 by a fixed
// operation sequence, rounded once to fp32.
// ------------------------------------------------------------------------------------------------
SRT_HD float srt_powf(float xf, float yf) {
    if (xf != xf || yf != yf) return xf + yf;
    if (yf == 0.0f || xf == 1.0f) return 1.0f;
    if (xf == 0.0f) return yf > 0.0f ? 0.0f : __builtin_inff();
    if (xf < 0.0f) return __builtin_nanf("");
    if (__builtin_isinf(xf)) return yf > 0.0f ? __builtin_inff() : 0.0f;
    double x = (double)xf, y = (double)yf;
    if (yf == 5.0f) {
        /* Schlick's (1-cos)^5 (materials/material.cu:48), the only integer exponent on the per-ray path: x^2 is exact in
         * fp64 (48 bits), the two further products round once each, so the result is within 2.2e-16 of x^5 before the
         * single rounding to fp32 -- the same accuracy class as the general branch at a fraction of the cost. */
        const double x2 = x * x;
        return (float)((x2 * x2) * x);
    }
    uint64_t bits;
    __builtin_memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ffu) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    __builtin_memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    double f = (m - 1.0) / (m + 1.0);
    double s = f * f;
    double p = 1.0 / 25.0;
    p = p * s + 1.0 / 23.0;
    p = p * s + 1.0 / 21.0;
    p = p * s + 1.0 / 19.0;
    p = p * s + 1.0 / 17.0;
    p = p * s + 1.0 / 15.0;
    p = p * s + 1.0 / 13.0;
    p = p * s + 1.0 / 11.0;
    p = p * s + 1.0 / 9.0;
    p = p * s + 1.0 / 7.0;
    p = p * s + 1.0 / 5.0;
    p = p * s + 1.0 / 3.0;
    p = p * s + 1.0;
    double lg = 2.0 * f * p;
    double A = y * (double)e;                 // exact: 24-bit * 11-bit
    double z = y * lg;
    double kd = __builtin_floor((A + z * 1.4426950408889634) + 0.5);
    if (kd > 1000.0) return __builtin_inff();
    if (kd < -1000.0) return 0.0f;
    double dd = A - kd;                        // exact
    double r = (dd * 0.693147180369123816490 + z) + dd * 1.90821492927058770002e-10;
    double q = 1.0 / 87178291200.0;
    q = q * r + 1.0 / 6227020800.0;
    q = q * r + 1.0 / 479001600.0;
    q = q * r + 1.0 / 39916800.0;
    q = q * r + 1.0 / 3628800.0;
    q = q * r + 1.0 / 362880.0;
    q = q * r + 1.0 / 40320.0;
    q = q * r + 1.0 / 5040.0;
    q = q * r + 1.0 / 720.0;
    q = q * r + 1.0 / 120.0;
    q = q * r + 1.0 / 24.0;
    q = q * r + 1.0 / 6.0;
    q = q * r + 0.5;
    q = q * r + 1.0;
    q = q * r + 1.0;
    uint64_t kb = (uint64_t)((long long)kd + 1023) << 52;
    double two_k;
    __builtin_memcpy(&two_k, &kb, 8);
    return (float)(q * two_k);
}

// srt_powf(x, 5.0f) for the Schlick call site (materials/material.cu:48), inlined: the same special cases in the same
// order and the same fp64 products as the y == 5 branch above, so the bits are srt_powf's (checked on the device by the
// op sweep, op 17) without the call and the general branch's code.
SRT_HD float srt_pow5f(float xf) {
    if (xf != xf) return xf + 5.0f;
    if (xf == 1.0f) return 1.0f;
    if (xf == 0.0f) return 0.0f;
    if (xf < 0.0f) return __builtin_nanf("");
    if (__builtin_isinf(xf)) return __builtin_inff();
    const double x = (double)xf, x2 = x * x;
    return (float)((x2 * x2) * x);
}

}  // namespace srt
