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
 *   normal   = sqrt(2) * erfinv(+-(j + 0.5) / 2^23), sign and j from the word, erfinv by
 *              Giles' binary32 polynomials and a logarithm built from +,-,*,/ only, plus one
 *              binary32 square root, so that the CPU checker and the GPU produce the same bits.
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

/* ln(y) for a normal binary32 y in (0, 2): y = m * 2^e with m in (sqrt(1/2), sqrt(2)],
 * ln m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^9.  Only +, -, *, / on binary32, so every
 * IEEE machine agrees on the bits (absolute error below 1e-6). */
PT_HD float det_logf(float y) {
    uint32_t bits = f2u(y);
    int e = (int)((bits >> 23) & 0xffu) - 127;
    float m = u2f((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = 1.0f / 9.0f;
    p = p * z + 1.0f / 7.0f;
    p = p * z + 1.0f / 5.0f;
    p = p * z + 1.0f / 3.0f;
    p = p * z + 1.0f;
    return (float)e * 0.693147182f + (2.0f * s) * p;
}

/* Inverse error function in binary32 (M. Giles, "Approximating the erfinv function",
 * GPU Computing Gems, 2011): two polynomial branches, no division; absolute error < 5e-7. */
PT_HD float det_erfinvf(float x) {
    float w = -det_logf((1.0f - x) * (1.0f + x));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = 3.43273939e-07f + p * w;
        p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w;
        p = 0.246640727f + p * w;
        p = 1.50140941f + p * w;
    } else {
        w = pt_sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = 0.000100950558f + p * w;
        p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w;
        p = 1.00167406f + p * w;
        p = 2.83297682f + p * w;
    }
    return p * x;
}

/* Standard normal from one random word by inverse CDF: bit 31 is the sign, bits 30..8 pick
 * one of 2^23 cell midpoints x = (j + 0.5) / 2^23 in (0, 1) (exact in binary32, never 0 or 1),
 * and the variate is sqrt(2) * erfinv(+-x). */
PT_HD float normal_from_word(uint32_t w) {
    float x = ((float)((w >> 8) & 0x7fffffu) + 0.5f) * (1.0f / 8388608.0f);
    if (w & 0x80000000u) x = -x;
    return 1.41421354f * det_erfinvf(x);
}

} // namespace ptc

#endif
