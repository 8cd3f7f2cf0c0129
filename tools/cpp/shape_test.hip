// Checks of the 16x16x32 helpers of hn_mlp2.h on the device: the lane-row exchanges against their definition, and a
// [32 x 64] x [64 x 32] product through mma_block in both layouts against a host reference.
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -DHN_MFMA16=1 -I ../../ho-nerf_amd/csrc -I ../../include shape_test.hip -o layer_bench_shape_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "hn_common.h"
#include "hn_mlp2.h"
using namespace hn::v2;

__global__ void k_rows(unsigned* out) {
    const int lane = threadIdx.x;
    unsigned x0 = lane, x1 = 256 + lane;
    rows_to16(x0, x1);
    out[lane] = x0;
    out[64 + lane] = x1;
    rows_to32(x0, x1);
    out[128 + lane] = x0;
    out[192 + lane] = x1;
}
__global__ void k_tile(int* out) {
    const int lane = threadIdx.x, c16 = lane & 15, g = lane >> 4;
    f32x16 t = zero16();
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = (float)((16 * (i >> 3) + 4 * g + (i & 3)) * 32 + 16 * ((i >> 2) & 1) + c16);   // row * 32 + col
    tile_out(t);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = (int)t[i];
}
// X[k][sample] (k < 64, sample < 32) given in the OLD fragment layout (4 k-steps), W as one tile of 4 blocks packed for
// the shape of this build; the product comes back as an OLD-layout C tile
__global__ void k_mm(const _Float16* wblk /* 4 blocks x (hi 1 KiB | lo 1 KiB) */, const float* X /* [64][32] */, float* C /* [32][32] */) {
    __shared__ __attribute__((aligned(16))) char lds[4 * KS_BYTES];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4 * KS_BYTES / 16; i += 64) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(wblk)[i];
    __syncthreads();
    const int c = lane & 31, hh = lane >> 5;
    h8 xh[4], xl[4];
    for (int s = 0; s < 4; ++s) {
        float f[8];
        for (int j = 0; j < 8; ++j) f[j] = X[(16 * s + 8 * hh + j) * 32 + c];
        split8(f, xh[s], xl[s]);
    }
#ifdef DIRECT_IN
    if (S16) {
        const int c16 = lane & 15, g = lane >> 4;
        for (int sp = 0; sp < 2; ++sp)
            for (int cb = 0; cb < 2; ++cb) {
                float f[8];
                for (int j = 0; j < 8; ++j) f[j] = X[(32 * sp + 8 * g + j) * 32 + 16 * cb + c16];
                split8(f, xh[2 * sp + cb], xl[2 * sp + cb]);
            }
    }
#else
    for (int s = 0; s < 4; s += 2) {
        frags_in(xh[s], xh[s + 1]);
        frags_in(xl[s], xl[s + 1]);
    }
#endif
    f32x16 c1 = zero16(), c2 = zero16();
    auto noslot = [](auto) {};
    static_for<4>([&](auto S) {
        constexpr int s = decltype(S)::value;
        const h8 ah = *reinterpret_cast<const h8*>(lds + s * KS_BYTES + lane * 16);
        const h8 al = *reinterpret_cast<const h8*>(lds + s * KS_BYTES + 1024 + lane * 16);
        mma_block<s, 0>(ah, al, xh, xl, c1, c2, noslot);
    });
    f32x16 z = combine(c1, c2);
#ifdef DIRECT_OUT
    if (S16) {
        const int c16 = lane & 15, g = lane >> 4;
        for (int i = 0; i < 16; ++i) C[(16 * (i >> 3) + 4 * g + (i & 3)) * 32 + 16 * ((i >> 2) & 1) + c16] = z[i];
        return;
    }
#endif
    tile_out(z);
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * hh) * 32 + c] = z[i];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k_rows, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int c16 = l & 15, q = (l >> 4) & 1, hh = l >> 5;   // new lane (c16, g = q + 2 hh): k-group a = hh, b = q
        const unsigned e0 = (hh ? 256 : 0) + c16 + 16 * 0 + 32 * q;   // N_0 <- F_{hh} of lane (q_old = 0, hh_old = q)
        const unsigned e1 = (hh ? 256 : 0) + c16 + 16 * 1 + 32 * q;   // N_1 <- F_{hh} of lane (q_old = 1, hh_old = q)
        if (h[l] != e0 || h[64 + l] != e1) { if (bad < 8) printf("lane %d: got %u %u expected %u %u\n", l, h[l], h[64 + l], e0, e1); ++bad; }
        if (h[128 + l] != (unsigned)l || h[192 + l] != 256u + l) { if (bad < 8) printf("lane %d: round trip %u %u\n", l, h[128 + l], h[192 + l]); ++bad; }
    }
    printf("row exchanges: %s\n", bad ? "MISMATCH" : "ok");
    {
        int* dt; hipMalloc(&dt, 1024 * 4);
        hipLaunchKernelGGL(k_tile, dim3(1), dim3(64), 0, 0, dt);
        int ht[1024]; hipMemcpy(ht, dt, sizeof(ht), hipMemcpyDeviceToHost);
        int badt = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 16; ++i) {
                const int e = ((i & 3) + 8 * (i >> 2) + 4 * (l >> 5)) * 32 + (l & 31);
                if (S16 && ht[l * 16 + i] != e) { if (badt < 12) printf("tile_out lane %d reg %d: got (r %d, c %d) expected (r %d, c %d)\n", l, i, ht[l * 16 + i] / 32, ht[l * 16 + i] % 32, e / 32, e % 32); ++badt; }
            }
        printf("tile_out: %s\n", badt ? "MISMATCH" : "ok");
    }
    // product
    std::vector<float> W(32 * 64), X(64 * 32), Cref(32 * 32, 0.f), C(32 * 32);
    for (size_t i = 0; i < W.size(); ++i) W[i] = 0.01f * (float)((int)((i * 2654435761u) >> 20) % 201 - 100);
    for (size_t i = 0; i < X.size(); ++i) X[i] = 0.013f * (float)((int)((i * 40503u + 7) >> 3) % 157 - 78);
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { double a = 0; for (int k = 0; k < 64; ++k) a += (double)W[r * 64 + k] * X[k * 32 + c]; Cref[r * 32 + c] = (float)a; }
    std::vector<_Float16> blk(4 * KS_BYTES / 2);
    for (int s = 0; s < 4; ++s)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int r = S16 ? 16 * (s & 1) + (l & 15) : (l & 31);
                const int k = S16 ? 32 * (s >> 1) + 8 * (l >> 4) + j : 16 * s + 8 * (l >> 5) + j;
                const float x = W[r * 64 + k];
                const _Float16 xh = (_Float16)x;
                blk[(size_t)s * 1024 + l * 8 + j] = xh;
                blk[(size_t)s * 1024 + 512 + l * 8 + j] = (_Float16)((x - (float)xh) * 2048.f);
            }
    _Float16* dw; float *dx, *dc;
    hipMalloc(&dw, blk.size() * 2); hipMalloc(&dx, X.size() * 4); hipMalloc(&dc, C.size() * 4);
    hipMemcpy(dw, blk.data(), blk.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mm, dim3(1), dim3(64), 0, 0, dw, dx, dc);
    hipMemcpy(C.data(), dc, C.size() * 4, hipMemcpyDeviceToHost);
    double e = 0, m = 0;
    for (size_t i = 0; i < C.size(); ++i) { e = fmax(e, fabs(C[i] - Cref[i])); m = fmax(m, fabs(Cref[i])); }
    printf("product (S16=%d): max abs err %.3e of max %.3e -> %s\n", (int)S16, e, m, e < 1e-5 * m ? "ok" : "MISMATCH");
    return hipGetLastError() != hipSuccess;
}
