/*
 * wave_scan.h -- inclusive scans over the 64 lanes of a wave (gfx950), built from DPP row shifts
 * inside the four rows of 16 lanes and two row broadcasts across them; no LDS, about eight vector
 * instructions each.  Every lane of the wave must execute the call.
 */
#ifndef PT_WAVE_SCAN_H
#define PT_WAVE_SCAN_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptw {

/* v of the lane `shift` to the left inside the lane's row of 16 (identity where there is none) */
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or(uint32_t identity, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v) {
    v += dpp_or<0x111, 0xf>(0u, v);   /* row_shr:1 */
    v += dpp_or<0x112, 0xf>(0u, v);   /* row_shr:2 */
    v += dpp_or<0x114, 0xf>(0u, v);   /* row_shr:4 */
    v += dpp_or<0x118, 0xf>(0u, v);   /* row_shr:8 */
    v += dpp_or<0x142, 0xa>(0u, v);   /* row_bcast:15 into rows 1 and 3 */
    v += dpp_or<0x143, 0xc>(0u, v);   /* row_bcast:31 into rows 2 and 3 */
    return v;
}

__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

__device__ __forceinline__ uint32_t wave_incl_scan_max(uint32_t v) {
    v = umax(v, dpp_or<0x111, 0xf>(0u, v));
    v = umax(v, dpp_or<0x112, 0xf>(0u, v));
    v = umax(v, dpp_or<0x114, 0xf>(0u, v));
    v = umax(v, dpp_or<0x118, 0xf>(0u, v));
    v = umax(v, dpp_or<0x142, 0xa>(0u, v));
    v = umax(v, dpp_or<0x143, 0xc>(0u, v));
    return v;
}

} // namespace ptw

#endif
