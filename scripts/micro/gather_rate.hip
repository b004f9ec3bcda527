// Micro-benchmark: how fast a CU's vector memory path serves 16-byte-per-lane loads whose 64
// lanes fall into 1, 16 or 64 different cache lines (gfx950), from tables that sit in L1, L2 and
// beyond.  The trace kernels fetch one 64-byte record per lane and round with four such loads.
// Build: hipcc -O3 --offload-arch=gfx950 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));
enum { OWN4 = 0, OWN3 = 1, QUAD = 2, SAME = 3, CONTIG = 4, OWN1 = 5, W1 = 6, W2 = 7, W3 = 8, PAIR = 9, ACT32 = 10, ACT16 = 11, DUP2 = 12, DUP4 = 13, ACT32S = 14, OWN4NT = 15, OWN4SC1 = 16, OWN4SC0SC1 = 17, OWN1NT = 18 };

template <int SHAPE>
__global__ __launch_bounds__(256) void k_gather(const float4* tab, uint32_t n_rec_mask, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        uint32_t rec = (x >> 8) & n_rec_mask;
        if (SHAPE == SAME) rec = __builtin_amdgcn_readfirstlane(rec);
        if (SHAPE == CONTIG) rec = (__builtin_amdgcn_readfirstlane(rec) & ~15u);
        const float4* p = tab + (size_t)rec * 4;
        float4 a, b, c, d;
        if (SHAPE == OWN4) { a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
        else if (SHAPE == OWN3) { a = p[0]; b = p[1]; c = p[2]; d = a; }
        else if (SHAPE == OWN1) { a = p[0]; b = a; c = a; d = a; }
        else if (SHAPE == ACT32 || SHAPE == ACT16 || SHAPE == ACT32S) { /* only some lanes load */
            const bool on = SHAPE == ACT32 ? lane < 32u : (SHAPE == ACT16 ? lane < 16u : (lane & 1u) == 0u);
            a = make_float4(0, 0, 0, 0);
            if (on) a = p[0];
            b = a; c = a; d = a; }
        else if (SHAPE == DUP2 || SHAPE == DUP4) { /* 2 or 4 neighbouring lanes load the same 16 bytes */
            uint32_t r = __shfl(rec, (int)(lane & (SHAPE == DUP2 ? ~1u : ~3u)));
            a = tab[(size_t)r * 4]; b = a; c = a; d = a; }
        else if (SHAPE == OWN4NT) { const f4v* q = (const f4v*)p;
            f4v v0 = __builtin_nontemporal_load(q), v1 = __builtin_nontemporal_load(q + 1), v2 = __builtin_nontemporal_load(q + 2), v3 = __builtin_nontemporal_load(q + 3);
            a = make_float4(v0.x, v0.y, v0.z, v0.w); b = make_float4(v1.x, v1.y, v1.z, v1.w); c = make_float4(v2.x, v2.y, v2.z, v2.w); d = make_float4(v3.x, v3.y, v3.z, v3.w); }
        else if (SHAPE == OWN1NT) { f4v v0 = __builtin_nontemporal_load((const f4v*)p); a = make_float4(v0.x, v0.y, v0.z, v0.w); b = a; c = a; d = a; }
        else if (SHAPE == OWN4SC1 || SHAPE == OWN4SC0SC1) {
            if (SHAPE == OWN4SC1)
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                             "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc0 sc1\n\t"
                             "global_load_dwordx4 %2, %4, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
        }
        else if (SHAPE == W1) { float v = *(const float*)p; a = make_float4(v, v, v, v); b = a; c = a; d = a; }
        else if (SHAPE == W2) { float2 v = *(const float2*)p; a = make_float4(v.x, v.y, v.x, v.y); b = a; c = a; d = a; }
        else if (SHAPE == W3) { const float* q = (const float*)p; a = make_float4(q[0], q[1], q[2], q[0]); b = a; c = a; d = a; }
        else if (SHAPE == PAIR) { /* two lanes per record: lane 2i takes chunks 0,1 and lane 2i+1 chunks 2,3 of record of lane 2i */
            uint32_t r = __shfl(rec, (int)(lane & ~1u));
            const float4* q = tab + (size_t)r * 4 + (lane & 1u) * 2u;
            a = q[0]; b = q[1]; c = a; d = b; }
        else if (SHAPE == SAME) { a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
        else if (SHAPE == CONTIG) { /* 64 lanes x 16 B = 1 KB contiguous per instruction, 4 instructions = 4 KB */
            a = p[lane]; b = p[64 + lane]; c = p[128 + lane]; d = p[192 + lane];
        } else { /* QUAD: instruction k serves the records of lanes 16k..16k+15, four lanes per record */
            uint32_t r0 = __shfl(rec, (int)(0 + (lane >> 2))), r1 = __shfl(rec, (int)(16 + (lane >> 2)));
            uint32_t r2 = __shfl(rec, (int)(32 + (lane >> 2))), r3 = __shfl(rec, (int)(48 + (lane >> 2)));
            a = tab[(size_t)r0 * 4 + (lane & 3u)]; b = tab[(size_t)r1 * 4 + (lane & 3u)];
            c = tab[(size_t)r2 * 4 + (lane & 3u)]; d = tab[(size_t)r3 * 4 + (lane & 3u)];
        }
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

typedef void (*kern_t)(const float4*, uint32_t, int, float*);

int main() {
    const size_t max_rec = (size_t)1 << 22; /* 256 MB of 64-byte records */
    float4* tab; float* out;
    hipMalloc(&tab, max_rec * 64 + 8192); hipMalloc(&out, 4);
    hipMemset(tab, 0, max_rec * 64 + 8192);
    int cus = 256; hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    const int blocks = cus * 8, iters = 2000;
    struct { const char* name; kern_t k; int loads; } shapes[] = {
        {"own record, 4 loads (64 lines each)", k_gather<OWN4>, 4}, {"own record, 3 loads", k_gather<OWN3>, 3},
        {"own record, 1 load", k_gather<OWN1>, 1},
        {"own record, 4 loads, nontemporal", k_gather<OWN4NT>, 4}, {"own record, 1 load, nontemporal", k_gather<OWN1NT>, 1},
        {"own record, 4 loads, sc1 (waits each iteration)", k_gather<OWN4SC1>, 4}, {"own record, 4 loads, sc0 sc1 (waits each iteration)", k_gather<OWN4SC0SC1>, 4},
        {"1 load, lanes 0-31 only", k_gather<ACT32>, 1}, {"1 load, lanes 0-15 only", k_gather<ACT16>, 1},
        {"1 load, even lanes only", k_gather<ACT32S>, 1},
        {"1 load, 2 lanes per address (32 addresses)", k_gather<DUP2>, 1}, {"1 load, 4 lanes per address (16 addresses)", k_gather<DUP4>, 1},
        {"own record, 1 load of 4 B", k_gather<W1>, 1}, {"own record, 1 load of 8 B", k_gather<W2>, 1},
        {"own record, 1 load of 12 B", k_gather<W3>, 1}, {"2 lanes per record, 2 loads (32 lines each)", k_gather<PAIR>, 2},
        {"4 lanes per record (16 lines each), 4 loads", k_gather<QUAD>, 4}, {"one record for the wave, 4 loads", k_gather<SAME>, 4},
        {"1 KB contiguous per load, 4 loads", k_gather<CONTIG>, 4}};
    struct { const char* name; uint32_t recs; } tables[] = {{"16 KB", 256}, {"2 MB", 1u << 15}, {"128 MB", 1u << 21}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& t : tables) {
        printf("table %s\n", t.name);
        for (auto& s : shapes) {
            hipLaunchKernelGGL(s.k, dim3(blocks), dim3(256), 0, 0, tab, t.recs - 1, 50, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(s.k, dim3(blocks), dim3(256), 0, 0, tab, t.recs - 1, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            double wave_loads = (double)blocks * 4 * iters * s.loads;
            double per_cu_per_us = wave_loads / cus / (ms * 1000.0);
            printf("  %-46s %8.3f ms  %7.2f wave-loads/us/CU  = %6.1f ns per wave-load per CU  (%.2f TB/s chip)\n", s.name, ms,
                   per_cu_per_us, 1000.0 / per_cu_per_us, wave_loads * 1024.0 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
