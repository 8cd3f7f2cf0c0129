// Sustained rate of v_mfma_f32_32x32x16_f16 on the whole chip with NOTHING else in the kernel: no memory traffic, no LDS, no
// epilogue -- four independent accumulator tiles per wave (no dependent-issue stalls), W waves per SIMD on every CU.
// This is the rate the power-limited clock leaves of the guide's 2.5 PFLOP/s (= 256 CUs x 4 SIMDs x 1024 FLOP/cycle x 2.4 GHz):
// what the field kernels' roofline fraction can be compared with beside the nominal peak.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using h8 = _Float16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

template <int ACCS>
__global__ __launch_bounds__(256) void k_mfma(int iters, float* sink, long long* clocks) {
    // eight operand pairs of pseudo-random values in [-1, 1) (the data a field kernel multiplies toggles its multipliers; a
    // constant operand draws less power and holds a higher clock)
    h8 a[8], b[8];
    unsigned x = 1234567u + 7919u * (blockIdx.x * 256u + threadIdx.x);
    for (int u = 0; u < 8; ++u)
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u;
            a[u][i] = (_Float16)((float)(int)(x >> 8) * (1.f / 8388608.f) - 1.f);
            x = x * 1664525u + 1013904223u;
            b[u][i] = (_Float16)((float)(int)(x >> 8) * (1.f / 8388608.f) - 1.f);
        }
    f32x16 acc[ACCS];
    for (int t = 0; t < ACCS; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < ACCS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[(u + t) & 7], acc[t], 0, 0, 0);
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < ACCS; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = t1 - t0;
        clocks[1] = (long long)(m1 - m0);
    }
}

// the same for v_mfma_f32_32x32x2_f32 (the training backward's launch sequence runs on it): nominal 157.3 TFLOP/s
__global__ __launch_bounds__(256) void k_mfma_f32(int iters, float* sink) {
    float a[8], b[8];
    unsigned x = 7654321u + 7919u * (blockIdx.x * 256u + threadIdx.x);
    for (int u = 0; u < 8; ++u) {
        x = x * 1664525u + 1013904223u;
        a[u] = (float)(int)(x >> 8) * (1.f / 8388608.f) - 1.f;
        x = x * 1664525u + 1013904223u;
        b[u] = (float)(int)(x >> 8) * (1.f / 8388608.f) - 1.f;
    }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + t) & 7], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 123.456f) sink[0] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300000;   // ~150 - 300 ms per launch: long enough for the power controller to settle
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* sink;
    long long* clocks;
    hipMalloc(&sink, 64);
    hipMalloc(&clocks, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("{\"device\": \"%s\", \"cus\": %d, \"nominal_peak_tflops\": 2500.0, \"runs\": [", p.name, cus);
    bool first = true;
    for (int wps = 1; wps <= 2; ++wps) {          // waves per SIMD
        for (int rep = 0; rep < 4; ++rep) {       // (the first run of a configuration also warms the clocks up)
            const int blocks = cus * wps;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, iters, sink, clocks);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            long long h[2];
            hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost);
            const double mfmas = (double)blocks * 4 * iters * 8 * 4;
            const double tflops = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
            // every MFMA occupies its SIMD's matrix pipe for 32 cycles (8 passes): with the pipe never idle the clock it held is
            (void)h;
            const double ghz = (double)wps * iters * 8 * 4 * 32.0 / (ms * 1e-3) / 1e9;
            if (rep == 0) continue;
            printf("%s{\"waves_per_simd\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"frac_of_nominal\": %.3f, \"implied_clock_ghz\": %.3f}", first ? "" : ", ",
                   wps, ms, tflops, tflops / 2500.0, ghz);
            first = false;
        }
    }
    printf("], \"f32_nominal_peak_tflops\": 157.3, \"f32_runs\": [");
    first = true;
    for (int rep = 0; rep < 3; ++rep) {
        const int blocks = cus, it32 = iters / 2;   // 64 cycles per MFMA: half the iterations for the same duration
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma_f32, dim3(blocks), dim3(256), 0, 0, it32, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double tflops = (double)blocks * 4 * it32 * 32 * 4096.0 / (ms * 1e-3) / 1e12;   // 32x32x2 x 2 FLOP per MFMA
        if (rep == 0) continue;
        printf("%s{\"ms\": %.3f, \"tflops\": %.1f, \"frac_of_nominal\": %.3f}", first ? "" : ", ", ms, tflops, tflops / 157.3);
        first = false;
    }
    printf("]}\n");
    return 0;
}
