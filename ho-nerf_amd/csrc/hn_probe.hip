// Measurement aid, not a product path: the rate v_mfma_f32_32x32x16_f16 sustains on the whole chip with nothing else in the
// kernel (no memory traffic, no LDS, no epilogue; pseudo-random operands, four independent accumulator tiles per wave).
// The matrix pipe is power-limited: with random data this pool's MI355X holds ~1.56 GHz = 0.65 of the 2.5 PFLOP/s the guide
// quotes for 2.4 GHz (constant operands: 0.98).  bench.py times this launch next to the field kernel so that its roofline
// entry can state the fraction of the SUSTAINED rate beside the fraction of the nominal peak (tools/cpp/mfma_peak.hip is
// the stand-alone form; profiles/r03/mfma_sustained_rate.json).
#include "hn_common.h"

namespace hn {
using h8p = _Float16 __attribute__((ext_vector_type(8)));
using f32x16p = float __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_mfma_probe(int iters, float* sink) {
    h8p a[8], b[8];
    unsigned x = 1234567u + 7919u * (blockIdx.x * 256u + threadIdx.x);
    for (int u = 0; u < 8; ++u)
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u;
            a[u][i] = (_Float16)((float)(int)(x >> 8) * (1.f / 8388608.f) - 1.f);
            x = x * 1664525u + 1013904223u;
            b[u][i] = (_Float16)((float)(int)(x >> 8) * (1.f / 8388608.f) - 1.f);
        }
    f32x16p acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[(u + t) & 7], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 123.456f && sink != nullptr) sink[0] = s;   // (keeps the accumulators alive; never true in practice)
}
}  // namespace hn

extern "C" int hn_debug_mfma_probe(int waves_per_simd, int iters, double* flop, hn_stream_t stream) {
    using namespace hn;
    HN_REQUIRE(waves_per_simd >= 1 && waves_per_simd <= 2 && iters >= 1, "bad probe arguments");
    const int cus = device_cus();
    HN_REQUIRE(cus > 0, "no device");
    const int blocks = cus * waves_per_simd;
    hipLaunchKernelGGL(k_mfma_probe, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, (float*)nullptr);
    HN_LAUNCH_CHECK();
    if (flop != nullptr) *flop = (double)blocks * 4 /*waves*/ * (double)iters * 32 /*MFMAs per iteration*/ * 32768.0;
    return HN_OK;
}
