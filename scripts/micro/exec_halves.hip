// Micro-benchmark: does a gfx950 SIMD-32 skip the half of a wave64 instruction whose 32 lanes are all
// masked off?  Eight independent v_fma_f32 chains, 8 waves per SIMD, every CU, under four EXEC masks:
// all 64 lanes, the low 32 lanes only, the even lanes only (32 active, spread over both halves), one lane.
// Build: hipcc -O3 --offload-arch=gfx950 -o exec_halves exec_halves.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define S_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));

__global__ void k_masked(float* out, int iters, float seed, int mode) {
    const unsigned lane = threadIdx.x & 63u;
    const bool on = mode == 0 ? true : mode == 1 ? lane < 32u : mode == 2 ? (lane & 1u) == 0u : lane == 0u;
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = seed * 1.0000001f, c = seed * 0.5f;
    if (on) {
        for (int i = 0; i < iters; i++) { S_FMA(a0) S_FMA(a1) S_FMA(a2) S_FMA(a3) S_FMA(a4) S_FMA(a5) S_FMA(a6) S_FMA(a7) }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r == 12345.678f) out[0] = r;
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    const int iters = 40000, blocks = 256 * 8, threads = 256;
    const char* names[] = {"all 64 lanes", "low 32 lanes", "even lanes (32)", "one lane"};
    for (int mode = 0; mode < 4; mode++) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_masked, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, mode);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        const double insts = (double)iters * 8 * blocks * (threads / 64);
        printf("%-18s %8.3f ms  %.3f ns per wave-instruction per SIMD\n", names[mode], ms, ms * 1e6 / (insts / 1024.0));
    }
    return 0;
}
