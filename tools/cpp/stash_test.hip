// standalone check of the Stash buffer helpers (round trip through every access form)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../ho-nerf_amd/csrc/hn_mlp2.h"
using namespace hn::v2;
__global__ void k(float4* scratch, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Stash sh;
    sh.init(scratch + (size_t)wave * 18 * SLOT_F4, 18, lane);
    f32x16 y;
    y = f32x16{0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15} + (lane * 100.f + wave * 10000.f);
    for (int t = 0; t < 8; ++t) sh.tile_store(3, t, y);
    h8 hi, lo;
    for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)(lane + j); lo[j] = (_Float16)(lane - j); }
    for (int b = 0; b < 21; ++b) for (int s = 0; s < 4; ++s) sh.frag_store(11 * SLOT_BYTES, 4 * b + s, hi, lo);
    sh.f32_store(17 * SLOT_BYTES + 5 * 256, (float)lane + 0.5f);
    float err = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
    for (int t = 0; t < 8; ++t) { f32x16 z = sh.tile_load(3, t); f32x16 dd = z - y; e1 += fabsf(dd[0]) + fabsf(dd[1]) + fabsf(dd[2]) + fabsf(dd[3]) + fabsf(dd[4]) + fabsf(dd[5]) + fabsf(dd[6]) + fabsf(dd[7]) + fabsf(dd[8]) + fabsf(dd[9]) + fabsf(dd[10]) + fabsf(dd[11]) + fabsf(dd[12]) + fabsf(dd[13]) + fabsf(dd[14]) + fabsf(dd[15]); }
    for (int b = 0; b < 21; ++b) for (int s = 0; s < 4; ++s) {
        h8 a, c; sh.frag_load(11 * SLOT_BYTES, 4 * b + s, a, c);
        for (int j = 0; j < 8; ++j) e2 += fabsf((float)a[j] - (float)hi[j]) + fabsf((float)c[j] - (float)lo[j]);
    }
    e3 = fabsf(sh.f32_load(17 * SLOT_BYTES + 5 * 256) - ((float)lane + 0.5f));
    out[threadIdx.x] = e1; out[256 + threadIdx.x] = e2; out[512 + threadIdx.x] = e3;
}
int main() {
    float4* s; float* o;
    hipMalloc(&s, 4 * 18 * 32768); hipMemset(s, 0xff, 4 * 18 * 32768); hipMalloc(&o, 768 * 4);
    k<<<1, 256>>>(s, o);
    std::vector<float> h(768); hipMemcpy(h.data(), o, 3072, hipMemcpyDeviceToHost);
    for (int p = 0; p < 3; ++p) { printf("part %d:", p); for (int i = 0; i < 256; i += 17) printf(" [%d]=%g", i, h[p * 256 + i]); printf("\n"); }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
