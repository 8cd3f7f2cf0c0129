#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = float __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void k(int iters, float* sink) {
    float a[8], b[8];
    unsigned x = 7654321u + 7919u * (blockIdx.x * 256u + threadIdx.x);
    for (int u = 0; u < 8; ++u) { x = x * 1664525u + 1013904223u; a[u] = (float)(int)(x >> 8) * (1.f / 8388608.f) - 1.f; x = x * 1664525u + 1013904223u; b[u] = (float)(int)(x >> 8) * (1.f / 8388608.f) - 1.f; }
    f32x16 acc[CH];
    for (int t = 0; t < CH; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < CH; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + t) & 7], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < CH; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 123.456f) sink[0] = s;
}
template <int CH> void run(int wps, int iters, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CH>, dim3(256 * wps), dim3(256), 0, 0, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("chains %d waves/SIMD %d: %.1f TFLOP/s (%.2f of 157.3)\n", CH, wps, 256.0 * wps * 4 * iters * 8 * CH * 4096.0 / (ms * 1e-3) / 1e12, 256.0 * wps * 4 * iters * 8 * CH * 4096.0 / (ms * 1e-3) / 1e12 / 157.3);
    }
}
int main() {
    float* sink; hipMalloc(&sink, 64);
    run<1>(1, 40000, sink); run<2>(1, 20000, sink); run<4>(1, 10000, sink); run<1>(3, 20000, sink); run<2>(3, 10000, sink); run<2>(4, 10000, sink);
    return 0;
}
