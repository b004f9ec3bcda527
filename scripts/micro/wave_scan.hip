// Checks the wave64 inclusive scans (sum and max) built from DPP row shifts and row broadcasts
// against a serial scan, on random and adversarial inputs (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 -I../../path_tracing_amd/csrc -o wave_scan wave_scan.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "wave_scan.h"

__global__ void k_scan(const uint32_t* in, uint32_t* out_sum, uint32_t* out_max, int n_waves) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = in[i];
    out_sum[i] = ptw::wave_incl_scan_add(v);
    out_max[i] = ptw::wave_incl_scan_max(v);
}

int main() {
    const int waves = 4096, n = waves * 64;
    std::vector<uint32_t> h(n), s(n), m(n);
    uint32_t x = 12345;
    for (int i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        int w = i / 64;
        h[i] = (w % 4 == 0) ? (x >> 28) : (w % 4 == 1) ? ((x >> 20) & 0xfff) : (w % 4 == 2) ? ((i % 64 == (w % 64)) ? 77u : 0u) : (x >> 8);
    }
    uint32_t *d_in, *d_s, *d_m;
    hipMalloc(&d_in, n * 4); hipMalloc(&d_s, n * 4); hipMalloc(&d_m, n * 4);
    hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_scan, dim3(n / 256), dim3(256), 0, 0, d_in, d_s, d_m, waves);
    hipMemcpy(s.data(), d_s, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(m.data(), d_m, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int w = 0; w < waves; w++) {
        uint32_t acc = 0, mx = 0;
        for (int l = 0; l < 64; l++) {
            int i = w * 64 + l;
            acc += h[i]; mx = h[i] > mx ? h[i] : mx;
            if (s[i] != acc || m[i] != mx) { if (bad < 5) printf("wave %d lane %d: sum %u want %u, max %u want %u\n", w, l, s[i], acc, m[i], mx); bad++; }
        }
    }
    printf("%s: %ld mismatches in %d lanes\n", bad ? "FAILED" : "ok", bad, n);
    return bad ? 1 : 0;
}
