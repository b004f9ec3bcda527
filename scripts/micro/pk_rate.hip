// Micro-benchmark: issue rate of v_pk_mul_f32 / v_pk_add_f32 against v_mul_f32 / v_add_f32 on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -o pk_rate pk_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 b0 = {a0, a1}, b1 = {a2, a3}, b2 = {a4, a5}, b3 = {a6, a7}, b4 = {a1, a0}, b5 = {a3, a2}, b6 = {a5, a4}, b7 = {a7, a6};
    const float m = 1.0000001f;
    const f2 mm = {m, m};
    for (int i = 0; i < iters; i++) {
        if (PK) {
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b0) : "v"(mm));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b1) : "v"(mm));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b2) : "v"(mm));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b3) : "v"(mm));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b4) : "v"(mm));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b5) : "v"(mm));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b6) : "v"(mm));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b7) : "v"(mm));
        } else {
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(m));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(m));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(m));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a3) : "v"(m));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a4) : "v"(m));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a5) : "v"(m));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a6) : "v"(m));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a7) : "v"(m));
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0.x + b0.y + b1.x + b1.y + b2.x + b2.y + b3.x + b3.y + b4.x + b4.y + b5.x + b5.y + b6.x + b6.y + b7.x + b7.y;
    if (r == 12345.678f) out[0] = r;
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    const int iters = 20000, blocks = 256 * 8, threads = 256;
    for (int pk = 0; pk < 2; pk++) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms;
        hipEventElapsedTime(&ms, a, b);
        double insts = (double)iters * 8 * blocks * (threads / 64);   // wave-level instructions
        printf("%s: %.3f ms, %.1f G wave-instr/s, %.2f cycles per wave-instr per SIMD at 2.4 GHz\n", pk ? "v_pk_mul_f32" : "v_mul_f32   ", ms,
               insts / ms / 1e6, 1024.0 * 2.4e9 / (insts / (ms * 1e-3)));
    }
    return 0;
}
