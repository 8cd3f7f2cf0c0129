// GEMM micro-benchmark for the parameter-gradient backward's k_dense (a copy of the kernel of hn_field_bwd.hip): C[n,M] =
// A[n,K] * W[M,K]^T (+bias) on the fp32 MFMA, with diagnostic variants (gemm_variants.inc): no loads / no MFMA / no stores,
// register prefetch, four waves per SIMD, persistent workgroups.  Findings (round 3, n = 56 448, K = M = 256; DESIGN.md, note
// on the training backward): the kernel runs at 0.53 of the fp32-MFMA peak; its inner loop alone (MFMA + LDS operand reads)
// at 0.62 in this launch shape and 0.87 - 0.93 in long-running workgroups (mfma_lds_operands.hip); prefetch and higher
// occupancy change nothing, persistent workgroups +8 %.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 gemm_bench.hip -o gemm_bench && ./gemm_bench [n K M]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float BETA = 100.f;
__device__ __forceinline__ float softplus(float z) { return BETA * z > 20.f ? z : log1pf(expf(BETA * z)) / BETA; }
struct DenseArgs {
    const float* A;
    int lda;
    const float* W;
    int wsk, wsc;
    const float* bias;
    float* C;
    int ldc;
    int n, K, M;
    float alpha;
    int accumulate;
    int act;   // applied to the stored value: 0 none, 1 softplus(beta = 100), 2 ReLU (the tape's activations: no separate pass over C)
};
// Workgroup tile 128 x BN (BN = 128, or 64 for narrow outputs), K step 32; 4 waves as 2 x 2, each 64 x BN/2 outputs
// = 2 x (BN/64) MFMA tiles.  At 128 x 128 the operand traffic is 32 flop per byte of L2 read (the 64 x 64 tile of
// the first version, 16 flop/B, was bound by L2 -> LDS bandwidth at ~40 TFLOP/s).
template <int BN>
__global__ __launch_bounds__(256) void k_dense(const DenseArgs a) {
    constexpr int BM = 128, CT = BN / 64;       // CT column tiles per wave
    __shared__ float As[BM][33];
    __shared__ float Bs[32][BN + 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1, h = lane >> 5, j = lane & 31;
    const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;
    f32x16 acc[2][CT];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < CT; ++y)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[x][y][i] = 0.f;
    const bool a_vec = (a.lda & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
    const bool w_vec = ((a.wsk == 1 ? a.wsc : a.wsk) & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.W) & 15) == 0);
    for (int k0 = 0; k0 < a.K; k0 += 32) {
#pragma unroll
        for (int rep = 0; rep < BM / 64; ++rep) {   // A tile: thread = (row, 8 consecutive k)
            const int r = rep * 64 + (t >> 2), c8 = (t & 3) * 8;
            const int row = row0 + r;
            const float* src = a.A + (size_t)row * a.lda + k0 + c8;
            if (a_vec && row < a.n && k0 + c8 + 8 <= a.K) {
                const float4 v0 = reinterpret_cast<const float4*>(src)[0], v1 = reinterpret_cast<const float4*>(src)[1];
                As[r][c8 + 0] = v0.x; As[r][c8 + 1] = v0.y; As[r][c8 + 2] = v0.z; As[r][c8 + 3] = v0.w;
                As[r][c8 + 4] = v1.x; As[r][c8 + 5] = v1.y; As[r][c8 + 6] = v1.z; As[r][c8 + 7] = v1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) As[r][c8 + e] = (row < a.n && k0 + c8 + e < a.K) ? src[e] : 0.f;
            }
        }
        if (a.wsk == 1) {   // B(k, col) = W[col * wsc + k]: k contiguous -> thread = (col, 8 k's)
#pragma unroll
            for (int rep = 0; rep < BN / 64; ++rep) {
                const int c = rep * 64 + (t >> 2), k8 = (t & 3) * 8;
                const int col = col0 + c;
                const float* src = a.W + (size_t)col * a.wsc + k0 + k8;
                if (w_vec && col < a.M && k0 + k8 + 8 <= a.K) {
                    const float4 v0 = reinterpret_cast<const float4*>(src)[0], v1 = reinterpret_cast<const float4*>(src)[1];
                    Bs[k8 + 0][c] = v0.x; Bs[k8 + 1][c] = v0.y; Bs[k8 + 2][c] = v0.z; Bs[k8 + 3][c] = v0.w;
                    Bs[k8 + 4][c] = v1.x; Bs[k8 + 5][c] = v1.y; Bs[k8 + 6][c] = v1.z; Bs[k8 + 7][c] = v1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) Bs[k8 + e][c] = (col < a.M && k0 + k8 + e < a.K) ? src[e] : 0.f;
                }
            }
        } else {            // B(k, col) = W[k * wsk + col]: col contiguous -> thread = (k, 8 cols)
#pragma unroll
            for (int rep = 0; rep < BN / 64; ++rep) {
                const int kk = t >> 3, c8 = rep * 64 + (t & 7) * 8;
                const int k = k0 + kk;
                const float* src = a.W + (size_t)k * a.wsk + col0 + c8;
                if (w_vec && k < a.K && col0 + c8 + 8 <= a.M) {
                    const float4 v0 = reinterpret_cast<const float4*>(src)[0], v1 = reinterpret_cast<const float4*>(src)[1];
                    Bs[kk][c8 + 0] = v0.x; Bs[kk][c8 + 1] = v0.y; Bs[kk][c8 + 2] = v0.z; Bs[kk][c8 + 3] = v0.w;
                    Bs[kk][c8 + 4] = v1.x; Bs[kk][c8 + 5] = v1.y; Bs[kk][c8 + 6] = v1.z; Bs[kk][c8 + 7] = v1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) Bs[kk][c8 + e] = (k < a.K && col0 + c8 + e < a.M) ? src[e] : 0.f;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            float av[2], bv[CT];
#pragma unroll
            for (int x = 0; x < 2; ++x) av[x] = As[wr * 64 + x * 32 + j][2 * ks + h];
#pragma unroll
            for (int y = 0; y < CT; ++y) bv[y] = Bs[2 * ks + h][wc * (BN / 2) + y * 32 + j];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < CT; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int y = 0; y < CT; ++y) {
        const int col = col0 + wc * (BN / 2) + y * 32 + j;
        if (col < a.M) {
            const float b = a.bias != nullptr ? a.bias[col] : 0.f;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + wr * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < a.n) {
                        float v = a.alpha * acc[x][y][r] + b;
                        float* c = a.C + (size_t)row * a.ldc + col;
                        if (a.accumulate) v += *c;
                        if (a.act == 1) v = softplus(v);
                        if (a.act == 2) v = fmaxf(v, 0.f);
                        *c = v;
                    }
                }
        }
    }
}


#include "gemm_variants.inc"

static void fill(std::vector<float>& v, unsigned seed) {
    unsigned x = seed;
    for (auto& e : v) {
        x = x * 1664525u + 1013904223u;
        e = (float)(int)(x >> 8) * (1.f / 8388608.f) - 1.f;
    }
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 56448, K = argc > 2 ? atoi(argv[2]) : 256, M = argc > 3 ? atoi(argv[3]) : 256;
    const int LD = (K + 3) & ~3;   // row pitch of A and W (the product pads rows to 16 bytes)
    std::vector<float> hA((size_t)n * LD), hW((size_t)M * LD), hb(M);
    fill(hA, 1);
    fill(hW, 2);
    fill(hb, 3);
    float *A, *W, *b, *C, *Cref;
    hipMalloc(&A, hA.size() * 4);
    hipMalloc(&W, hW.size() * 4);
    hipMalloc(&b, hb.size() * 4);
    hipMalloc(&C, (size_t)n * M * 4);
    hipMalloc(&Cref, (size_t)n * M * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double flop = 2.0 * n * (double)K * M;
    auto time = [&](const char* name, auto launch, float* out) {
        hipMemset(out, 0, (size_t)n * M * 4);
        launch(out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) launch(out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 10;
        double err = 0.0;
        if (out != Cref) {
            std::vector<float> x((size_t)4096 * M), y((size_t)4096 * M);
            hipMemcpy(x.data(), out, x.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(y.data(), Cref, y.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0.0;
            for (size_t i = 0; i < x.size(); ++i) {
                err = fmax(err, fabs((double)x[i] - y[i]));
                mx = fmax(mx, fabs((double)y[i]));
            }
            err /= mx;
        }
        printf("%-28s %8.1f us  %6.1f TFLOP/s  (%.2f of 157.3)  max rel diff vs reference %.2e\n", name, ms * 1e3, flop / (ms * 1e-3) / 1e12,
               flop / (ms * 1e-3) / 1e12 / 157.3, err);
    };
    time("k_dense<64> (reference)", [&](float* out) {
        DenseArgs a{A, LD, W, 1, LD, b, out, M, n, K, M, 1.f, 0, 0};
        hipLaunchKernelGGL(k_dense<64>, dim3((M + 63) / 64, (n + 127) / 128), dim3(256), 0, 0, a);
    }, Cref);
    time("k_dense<128>", [&](float* out) {
        DenseArgs a{A, LD, W, 1, LD, b, out, M, n, K, M, 1.f, 0, 0};
        hipLaunchKernelGGL(k_dense<128>, dim3((M + 127) / 128, (n + 127) / 128), dim3(256), 0, 0, a);
    }, C);
    run_variants(A, W, b, C, Cref, n, K, M, LD, time);
    return 0;
}
