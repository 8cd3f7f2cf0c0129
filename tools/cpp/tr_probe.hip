// Probe of ds_read_b64_tr_b16 (gfx950) through the clang builtin: what each lane receives from a [row][col] image of 16-bit values.
//   hipcc --offload-arch=gfx950 -O2 tools/cpp/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 272;   // bytes per row: 128 columns x 2 B + 16
__global__ void k(int r0, int c0, short* out) {
    __shared__ __attribute__((aligned(16))) char lds[32 * PITCH];
    for (int i = threadIdx.x; i < 32 * 128; i += 64) *reinterpret_cast<short*>(lds + (i / 128) * PITCH + (i % 128) * 2) = (short)((i / 128) * 256 + (i % 128));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, q = l >> 2, p = l & 3;
    // group g: rows r0 + 4 * (g >> 1) .. + 3 ... no: keep it simple -- every group reads rows r0 + 4 g .. r0 + 4 g + 3, columns c0 .. c0 + 15
    const char* addr = lds + (r0 + 4 * g + q) * PITCH + (c0 + 4 * p) * 2;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * sizeof(short));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 0, 32, d);
    short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int g = lane >> 4, i = lane & 15;
            const int want = (4 * g + e) * 256 + (32 + i);   // lane i of the group: column c0 + i, row r0 + 4 g + e in element e
            if (h[lane * 4 + e] != want) {
                if (bad < 8) printf("lane %d e %d: got row %d col %d, expected row %d col %d\n", lane, e, h[lane * 4 + e] / 256, h[lane * 4 + e] % 256, want / 256, want % 256);
                ++bad;
            }
        }
    printf("ds_read_b64_tr_b16 probe: %d mismatches of 256\n", bad);
    return bad != 0;
}
