/*
 * rng_core.h -- the counter-based random stream of the renderer, per lane.
 *
 * The reference draws from a per-thread std::mt19937 seeded from std::random_device
 * (reference integrator.h:50-51) in depth-first order, which a wavefront renderer cannot
 * replay.  This stream replaces it: every random number is a pure function of
 * (seed, pixel, sample, stream, word), so any lane can compute any draw.
 *
 *   block  = Philox4x32-10(counter = {pixel, sample, stream, 0}, key = {seed lo, seed hi})
 *   stream 0                  camera: words 0,1 anti-aliasing jitter x,y (reference
 *                             integrator.cpp:248-249); words 2,3 aperture x,y (:254-255)
 *   stream 16 + 16*depth + l  path vertex `depth` (0 = first hit), light l: words 0..2 are the
 *                             three light-delocalisation normals (:121)
 *   uniform  = ((word >> 8) + 0.5) / 2^24          in (0,1), exact in binary32 and binary64
 *   normal   = inverse normal CDF (Acklam's rational approximation) of the uniform,
 *              evaluated in binary64 with +,-,*,/ only, plus one binary32 square root, so
 *              that the CPU checker and the GPU produce the same bits; narrowed to binary32.
 *
 * Compiled by hipcc for the kernels and by g++ for the CPU-side lane emulation used in tests.
 * Needs -ffp-contract=off.
 */
#ifndef PT_RNG_CORE_H
#define PT_RNG_CORE_H

#include <stdint.h>

#include "traverse_core.h"

namespace ptc {

/* Correctly rounded binary32 square root.  On the device this relies on hipcc's
 * -fhip-fp32-correctly-rounded-divide-sqrt (the default, and spelled out in the Makefile);
 * HIP's __fsqrt_rn is NOT used because without OCML_BASIC_ROUNDED_OPERATIONS it maps to the
 * approximate native square root. */
PT_HD float pt_sqrtf(float x) { return __builtin_sqrtf(x); }

struct Rng4 { uint32_t w[4]; };

PT_HD void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

PT_HD Rng4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, hi0, lo0);
        mulhilo(0xCD9E8D57u, c2, hi1, lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Rng4 o;
    o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

PT_HD Rng4 rng_block(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream) {
    return philox4x32_10(pixel, sample, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
}

PT_HD double uniform01(uint32_t w) { return ((double)(w >> 8) + 0.5) * (1.0 / 16777216.0); }

PT_HD double u2d(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }
PT_HD uint64_t d2u(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }

/* ln(x) for normal x in (0, 1]: x = m * 2^e with m in (sqrt(1/2), sqrt(2)],
 * ln m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^23. */
PT_HD double det_log(double x) {
    uint64_t bits = d2u(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = u2d((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

PT_HD double acklam_tail(double q) {
    const double c0 = -7.784894002430293e-03, c1 = -3.223964580411365e-01, c2 = -2.400758277161838e+00,
                 c3 = -2.549732539343734e+00, c4 = 4.374664141464968e+00, c5 = 2.938163982698783e+00;
    const double d0 = 7.784695709041462e-03, d1 = 3.224671290700398e-01, d2 = 2.445134137142996e+00,
                 d3 = 3.754408661907416e+00;
    return (((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) / ((((d0 * q + d1) * q + d2) * q + d3) * q + 1.0);
}

PT_HD float normal_from_uniform(double p) {
    const double a0 = -3.969683028665376e+01, a1 = 2.209460984245205e+02, a2 = -2.759285104469687e+02,
                 a3 = 1.383577518672690e+02, a4 = -3.066479806614716e+01, a5 = 2.506628277459239e+00;
    const double b0 = -5.447609879822406e+01, b1 = 1.615858368580409e+02, b2 = -1.556989798598866e+02,
                 b3 = 6.680131188771972e+01, b4 = -1.328068155288572e+01;
    const double p_low = 0.02425;
    double x;
    if (p < p_low) {
        double q = (double)pt_sqrtf((float)(-2.0 * det_log(p)));
        x = acklam_tail(q);
    } else if (p <= 1.0 - p_low) {
        double q = p - 0.5;
        double r = q * q;
        x = (((((a0 * r + a1) * r + a2) * r + a3) * r + a4) * r + a5) * q /
            (((((b0 * r + b1) * r + b2) * r + b3) * r + b4) * r + 1.0);
    } else {
        double q = (double)pt_sqrtf((float)(-2.0 * det_log(1.0 - p)));
        x = -acklam_tail(q);
    }
    return (float)x;
}

PT_HD float normal_from_word(uint32_t w) { return normal_from_uniform(uniform01(w)); }

} // namespace ptc

#endif
