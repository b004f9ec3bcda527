// Micro-benchmark: one compare feeding eight independent v_cndmask_b32, mask in VCC vs in an SGPR pair.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = seed * 1.0000001f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            asm volatile("v_cmp_lt_f32 vcc, %8, %9\n\ts_nop 1\n\t"
                         "v_cndmask_b32 %0, %0, %9, vcc\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cndmask_b32 %2, %2, %9, vcc\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t"
                         "v_cndmask_b32 %4, %4, %9, vcc\n\tv_cndmask_b32 %5, %5, %9, vcc\n\tv_cndmask_b32 %6, %6, %9, vcc\n\tv_cndmask_b32 %7, %7, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(m) : "vcc");
        } else if (MODE == 1) {
            asm volatile("v_cmp_lt_f32 s[20:21], %8, %9\n\ts_nop 1\n\t"
                         "v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %9, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %9, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %9, s[20:21]\n\t"
                         "v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n\tv_cndmask_b32_e64 %5, %5, %9, s[20:21]\n\tv_cndmask_b32_e64 %6, %6, %9, s[20:21]\n\tv_cndmask_b32_e64 %7, %7, %9, s[20:21]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(m) : "s20", "s21");
        } else if (MODE == 2) {   /* e64 encoding, but naming vcc as the mask */
            asm volatile("v_cmp_lt_f32 vcc, %8, %9\n\ts_nop 1\n\t"
                         "v_cndmask_b32_e64 %0, %0, %9, vcc\n\tv_cndmask_b32_e64 %1, %1, %9, vcc\n\tv_cndmask_b32_e64 %2, %2, %9, vcc\n\tv_cndmask_b32_e64 %3, %3, %9, vcc\n\t"
                         "v_cndmask_b32_e64 %4, %4, %9, vcc\n\tv_cndmask_b32_e64 %5, %5, %9, vcc\n\tv_cndmask_b32_e64 %6, %6, %9, vcc\n\tv_cndmask_b32_e64 %7, %7, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(m) : "vcc");
        } else {                  /* reference: nine plain multiplies */
            asm volatile("v_mul_f32 %0, %0, %9\n\tv_mul_f32 %1, %1, %9\n\tv_mul_f32 %2, %2, %9\n\tv_mul_f32 %3, %3, %9\n\t"
                         "v_mul_f32 %4, %4, %9\n\tv_mul_f32 %5, %5, %9\n\tv_mul_f32 %6, %6, %9\n\tv_mul_f32 %7, %7, %9\n\tv_mul_f32 %0, %0, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(m));
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
void run(const char* name, float* d) {
    const int iters = 40000, blocks = 256 * 8, threads = 256;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    double groups = (double)iters * blocks * (threads / 64);
    printf("%-44s %8.3f ms  %6.2f ns per group of 9 per SIMD\n", name, ms, ms * 1e-3 / (groups / 1024.0) * 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    run<3>("9 x v_mul_f32", d);
    run<0>("v_cmp vcc + 8 x v_cndmask_b32 (e32, vcc)", d);
    run<2>("v_cmp vcc + 8 x v_cndmask_b32_e64 vcc", d);
    run<1>("v_cmp sgpr + 8 x v_cndmask_b32_e64 sgpr", d);
    return 0;
}
