// Microbenchmark of run_layer (hn_mlp2.h): cycles per 16-k-step chunk of one wave, by variant.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DVARIANT=n [-DHN_PIECE_BRANCH] -I ../../ho-nerf_amd/csrc layer_bench.hip -o layer_bench_n
// VARIANT 0: softplus epilogue in the MFMA slots; 1: relu; 2: no epilogue work (identity, no fragments);
//         3: softplus + fp32 tile stash store; 4: dsig with stash pre-load (reverse sweep)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "hn_common.h"
#include "hn_mlp2.h"
using namespace hn::v2;
#ifdef CT
#define RL(OT, KS, TPC, TAIL, FR) run_layer_c<OT, KS, TPC, TAIL, FR, HB, HB>(ws,
#else
#define RL(OT, KS, TPC, TAIL, FR) run_layer<OT, KS, TPC, TAIL, FR>(ws, HB, HB,
#endif
#ifndef VARIANT
#define VARIANT 0
#endif
constexpr int HB = chunk_bytes(1, 16, true);
constexpr int HBB = chunk_bytes(4, 4, false);
#ifdef CT
constexpr int FCH = HBB;
#else
constexpr int FCH = 1;
#endif
struct Act { f32x16 v; };
__global__ __launch_bounds__(256) void k_bench(const char* blob, size_t bytes, int layers, float4* scratch, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    f16_flush_mode();
    WStream ws;
    ws.init(blob, bytes, lds, wave, lane);
#ifdef CT
    if (VARIANT == 5 || VARIANT == 6) ws.fetch_all_c<HBB>(); else ws.fetch_all_c<HB>();
#else
    ws.fetch_all((VARIANT == 5 || VARIANT == 6) ? HBB : HB);
#endif
    Stash sh;
    sh.init(scratch + ((size_t)blockIdx.x * WG_WAVES + wave) * 2 * SLOT_F4, 2, lane);
    h8 ah[16], al[16], bh[16], bl[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ah[s][j] = (_Float16)(0.01f * ((lane + s + j) % 7));
            al[s][j] = (_Float16)(0.5f * ((lane + 3 * s + j) % 5));
        }
    auto no_pre = [](auto, const char*) { return NoData{}; };
    auto no_store = [](auto, const auto&) {};
    auto to_regs = [&](h8(&oh)[16], h8(&ol)[16]) {
        return [&oh, &ol, &sh](auto T, EpiState& st, const auto&) {
            constexpr int t = decltype(T)::value;
            asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
            oh[2 * t] = st.hi[0]; ol[2 * t] = st.lo[0]; oh[2 * t + 1] = st.hi[1]; ol[2 * t + 1] = st.lo[1];
            if (VARIANT == 3) sh.tile_store(0, t, st.vec());
            return NoData{};
        };
    };
    float acc = 0.f;
    auto sink_fin = [&](auto T, EpiState& st, const auto&) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += st.v[i];
        return NoData{};
    };
    auto act_of = [&](auto T, const char*) { return Act{sh.tile_load(0, decltype(T)::value)}; };
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int l = 0; l < layers; l += 2) {
        if (VARIANT == 0 || VARIANT == 3) {
            RL(8, 16, 1, true, true) ah, al, lane, h, no_pre, PhSoftplus{}, to_regs(bh, bl), no_store);
            RL(8, 16, 1, true, true) bh, bl, lane, h, no_pre, PhSoftplus{}, to_regs(ah, al), no_store);
        } else if (VARIANT == 1) {
            RL(8, 16, 1, true, true) ah, al, lane, h, no_pre, PhRelu{}, to_regs(bh, bl), no_store);
            RL(8, 16, 1, true, true) bh, bl, lane, h, no_pre, PhRelu{}, to_regs(ah, al), no_store);
        } else if (VARIANT == 2) {
            RL(8, 16, 1, true, false) ah, al, lane, h, no_pre, PhIdentity{}, sink_fin, no_store);
            RL(8, 16, 1, true, false) ah, al, lane, h, no_pre, PhIdentity{}, sink_fin, no_store);
        } else if (VARIANT == 5 || VARIANT == 6) {
            // feature-pass shape: chunks of 4 tiles x 4 k-steps sharing 4 fragments, 8 live accumulator pairs
            f32x16 c1[8], c2[8];
            h8 fh[4], fl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { fh[i] = ah[i]; fl[i] = al[i]; }
#pragma unroll
            for (int i = 4; i < 16; ++i) asm volatile("" :: "v"(ah[i]), "v"(al[i]));   // the rest is dead from here
#pragma unroll
            for (int i = 0; i < 8; ++i) { c1[i] = zero16(); c2[i] = zero16(); }
#pragma unroll 1
            for (int cch = 0; cch < 8; ++cch) {
                static_for<2>([&](auto BLK) {
                    constexpr int blk = decltype(BLK)::value;
                    const char* buf = ws.template acquire<0>();
#ifdef CT
                    ws.begin_c<HBB>();
#else
                    ws.begin(HBB);
#endif
                    if constexpr (VARIANT == 5) {
                        static_for<4>([&](auto TI) {
                            constexpr int ti = decltype(TI)::value;
                            if constexpr (ti == 0)
                                mma_tile<4, 0, FCH>(ws, buf + ti * 4 * KS_BYTES, fh, fl, c1[4 * blk + ti], c2[4 * blk + ti], lane);
                            else
                                mma_tile<4, 0, 0>(ws, buf + ti * 4 * KS_BYTES, fh, fl, c1[4 * blk + ti], c2[4 * blk + ti], lane);
                        });
                    } else {
                        mma_chunk<4, 4, FCH>(ws, buf, fh, fl, &c1[4 * blk], &c2[4 * blk], lane);
                    }
                });
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc += c1[i][r] + c2[i][r];
        } else {
            RL(8, 16, 1, true, true) ah, al, lane, h, act_of, PhDsig{}, to_regs(bh, bl), no_store);
            RL(8, 16, 1, true, true) bh, bl, lane, h, act_of, PhDsig{}, to_regs(ah, al), no_store);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the last prefetch before the LDS goes away
    __syncthreads();
    float s = acc;
#pragma unroll
    for (int k = 0; k < ((VARIANT == 5 || VARIANT == 6) ? 4 : 16); ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)ah[k][j] + (float)al[k][j];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
__global__ void k_flush_check(const float* in, float* out, const unsigned short* hbits) {
    f16_flush_mode();
    const int i = threadIdx.x;
    h8 a, b;
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = in[(i + j) % 8];
    split8(x, a, b);
    out[i] = (float)a[0];                                  // hi of in[i]
    out[8 + i] = (float)b[0];                              // scaled lo of in[i]
    out[16 + i] = (float)__builtin_bit_cast(_Float16, hbits[i]);   // a stored fp16 bit pattern read back
}
int main(int argc, char** argv) {
    {
        const float hin[8] = {1.0f, 6.2e-5f, 6.0e-5f, 3.0e-6f, -3.0e-6f, 1.0e-7f, 2.9e-8f, 0.3337f};
        const unsigned short hb[8] = {0x3c00, 0x0400, 0x03ff, 0x0001, 0x8001, 0x0200, 0x7bff, 0x0000};
        float *din, *dout; unsigned short* dh; float hout[24];
        (void)hipMalloc(&din, 32); (void)hipMalloc(&dout, 96); (void)hipMalloc(&dh, 16);
        (void)hipMemcpy(din, hin, 32, hipMemcpyHostToDevice); (void)hipMemcpy(dh, hb, 16, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_flush_check, dim3(1), dim3(8), 0, 0, din, dout, dh);
        (void)hipMemcpy(hout, dout, 96, hipMemcpyDeviceToHost);
        for (int i = 0; i < 8; ++i) printf("  x=%.4e hi=%.6e lo/2048=%.6e sum-x=%.2e | half 0x%04x -> %.6e\n", hin[i], hout[i], hout[8 + i] / 2048.f, (hout[i] + hout[8 + i] / 2048.f) - hin[i], hb[i], hout[16 + i]);
    }
    const int layers = argc > 1 ? atoi(argv[1]) : 32;
    const int wgs = argc > 2 ? atoi(argv[2]) : 256;
    const size_t bytes = (VARIANT == 5 || VARIANT == 6) ? (size_t)HBB * 64 : (size_t)HB * 8 * 8;   // 8 layers of 8 chunks, cycled
    std::vector<_Float16> hostw(bytes / 2);
    for (size_t i = 0; i < hostw.size(); ++i) hostw[i] = (_Float16)(0.02f * (float)((int)(i * 2654435761u >> 24) % 13 - 6));
    char* blob; float4* scratch; long long* cyc; float* sink;
    hipMalloc(&blob, bytes); hipMemcpy(blob, hostw.data(), bytes, hipMemcpyHostToDevice);
    hipMalloc(&scratch, (size_t)wgs * 4 * 2 * SLOT_BYTES); hipMemset(scratch, 0, (size_t)wgs * 4 * 2 * SLOT_BYTES);
    hipMalloc(&cyc, wgs * 4 * 8); hipMalloc(&sink, wgs * 256 * 4);
    hipFuncSetAttribute((const void*)k_bench, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CHUNK_MAX);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_bench, dim3(wgs), dim3(256), 2 * CHUNK_MAX, 0, blob, bytes, layers, scratch, cyc, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> c(wgs * 4);
    hipMemcpy(c.data(), cyc, wgs * 4 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : c) mean += v; mean /= c.size();
    const int chunks = layers * 8;   // variants 5/6: 16 chunks per 2 'layers' as well
    printf("variant %d%s: %d wgs, %d chunks: %.3f ms -> %.3f us/chunk, %.0f ticks/chunk (%.0f ticks/us); MFMA floor 1536 cycles\n", VARIANT,
#ifdef HN_PIECE_BRANCH
           " [branchy pieces]",
#else
           "",
#endif
           wgs, chunks, ms, ms * 1e3 / chunks, mean / chunks, mean / (ms * 1e3));
    return hipGetLastError() != hipSuccess;
}
