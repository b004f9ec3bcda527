// Micro-benchmark: issue cost of the VALU instructions the trace kernels are made of (gfx950).
// Every kernel runs 8 independent chains of one instruction, 8 waves per SIMD, all CUs.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define KERNEL(name, stmt)                                                                        \
    __global__ void name(float* out, int iters, float seed) {                                     \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,        \
              a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                              \
        float m = seed * 1.0000001f, c = seed * 0.5f;                                             \
        (void)c;                                                                                  \
        for (int i = 0; i < iters; i++) {                                                         \
            stmt(a0) stmt(a1) stmt(a2) stmt(a3) stmt(a4) stmt(a5) stmt(a6) stmt(a7)               \
        }                                                                                         \
        float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                          \
        if (r == 12345.678f) out[0] = r;                                                          \
    }

#define S_MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(m));
#define S_ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(m));
#define S_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
#define S_MIN3(x) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
#define S_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(m));
#define S_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
#define S_CNDMASK_NOCLOB(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m));
#define S_CNDMASK_SGPR(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(m));
#define S_MUL_NOP(x) asm volatile("v_mul_f32 %0, %0, %1\n\ts_nop 0" : "+v"(x) : "v"(m));
#define S_CMP_CND(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
#define S_SMOV_CND(x) asm volatile("s_mov_b64 vcc, s[20:21]\n\ts_nop 3\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
#define S_SOR_CND64(x) asm volatile("s_or_b64 s[22:23], s[20:21], exec\n\ts_nop 3\n\tv_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(x) : "v"(m) : "s22", "s23", "scc");
#define S_CMP1_CND4(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
#define S_CMP(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(m) : "vcc");
#define S_CMP_E64(x) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(x), "v"(m) : "s20", "s21");
#define S_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define S_DIVSCALE(x) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(m) : "vcc");
#define S_DIVFMAS(x) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c) : "vcc");
#define S_DIVFIXUP(x) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
#define S_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(m));
#define S_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(m));
#define S_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(m));
#define S_BFE(x) asm volatile("v_bfe_u32 %0, %0, 3, 4" : "+v"(x));
#define S_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(m));
#define S_SALU(x) asm volatile("s_and_b64 s[20:21], s[20:21], exec" : : : "s20", "s21", "scc");

KERNEL(k_mul, S_MUL) KERNEL(k_add, S_ADD) KERNEL(k_fma, S_FMA) KERNEL(k_min3, S_MIN3) KERNEL(k_max, S_MAX)
KERNEL(k_cndmask, S_CNDMASK) KERNEL(k_cmp, S_CMP) KERNEL(k_cmp64, S_CMP_E64) KERNEL(k_rcp, S_RCP)
KERNEL(k_divscale, S_DIVSCALE) KERNEL(k_divfmas, S_DIVFMAS) KERNEL(k_divfixup, S_DIVFIXUP) KERNEL(k_mullo, S_MULLO)
KERNEL(k_cnd_noclob, S_CNDMASK_NOCLOB) KERNEL(k_cnd_sgpr, S_CNDMASK_SGPR) KERNEL(k_mul_nop, S_MUL_NOP) KERNEL(k_cmp_cnd, S_CMP_CND)
KERNEL(k_smov_cnd, S_SMOV_CND) KERNEL(k_sor_cnd64, S_SOR_CND64) KERNEL(k_cmp1_cnd4, S_CMP1_CND4)
KERNEL(k_lshladd, S_LSHLADD) KERNEL(k_and, S_AND) KERNEL(k_bfe, S_BFE) KERNEL(k_mov, S_MOV) KERNEL(k_salu, S_SALU)

typedef void (*kern_t)(float*, int, float);

int main() {
    float* d;
    hipMalloc(&d, 4);
    const int iters = 40000, blocks = 256 * 8, threads = 256;
    struct { const char* name; kern_t k; } list[] = {
        {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_min3_f32", k_min3}, {"v_max_f32", k_max},
        {"v_cndmask_b32", k_cndmask}, {"v_cndmask vcc (no nop)", k_cnd_noclob}, {"v_cndmask sgpr mask", k_cnd_sgpr},
        {"v_mul_f32 + s_nop 0", k_mul_nop}, {"s_mov vcc + v_cndmask vcc", k_smov_cnd}, {"s_or sgpr + v_cndmask e64", k_sor_cnd64},
        {"v_cmp + 4x v_cndmask vcc", k_cmp1_cnd4}, {"v_cmp+s_nop1+v_cndmask", k_cmp_cnd}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr", k_cmp64}, {"v_rcp_f32", k_rcp},
        {"v_div_scale_f32", k_divscale}, {"v_div_fmas_f32", k_divfmas}, {"v_div_fixup_f32", k_divfixup},
        {"v_mul_lo_u32", k_mullo}, {"v_lshl_add_u32", k_lshladd}, {"v_and_b32", k_and}, {"v_bfe_u32", k_bfe},
        {"v_mov_b32", k_mov}, {"s_and_b64 (SALU)", k_salu}};
    double base = 0;
    for (auto& e : list) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        double insts = (double)iters * 8 * blocks * (threads / 64);
        double per = ms * 1e-3 / (insts / 1024.0);   // seconds per wave-instruction per SIMD
        if (base == 0) base = per;
        printf("%-20s %8.3f ms  %6.2f ns/wave-instr/SIMD  %5.2fx v_mul_f32\n", e.name, ms, per * 1e9, per / base);
    }
    return 0;
}
