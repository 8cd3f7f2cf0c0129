// Latency form of the object field's sdf-only kernel (see hn_field2_hand_q.hip for the design): the four waves of a workgroup
// share one block of 32 samples and split every layer's output tiles (wave w: tiles w and w + 4), activations exchanged
// through a double-buffered LDS buffer, A fragments straight from the packed sdf-only stream into a register ring.  Same
// MFMA sequence per accumulator, same epilogues, same summation order as k_field2_obj<0>: bit-identical results.
// The importance rounds of a fitting step evaluate 196 x 16 points per field (25 tiles of 128): 72 us per launch in the
// throughput form, on 25 CUs.  Reference: utils/fields.py:316-331 (SDFNetwork_OBJ.sdf), utils/renderer.py:465-496.
#define HN_OBJ_QUAD_TU 1
#include "hn_field2_obj.hip"

namespace hn {
namespace v2 {

constexpr int OQPRE = 8;
constexpr int OQ_L0 = 0;
constexpr int OQ_L1 = OQ_L0 + 2 * CB_L0;
constexpr int OQ_L2 = OQ_L1 + 8 * CB_HID;
constexpr int OQ_L3 = OQ_L2 + 8 * CB_HID;
constexpr int OQ_L4 = OQ_L3 + 7 * CB_HID;
constexpr int OQ_L5 = OQ_L4 + 8 * CB_HID;
constexpr int OQ_L6 = OQ_L5 + 8 * CB_HID;
constexpr int OQ_L7 = OQ_L6 + 8 * CB_HID;
constexpr int OQ_END = OQ_L7 + 8 * CB_HID;
constexpr int OQ_EXCH = 16 * KS_BYTES;
constexpr size_t OBJQ_LDS = 3 * OQ_EXCH + 256;

struct OStream {
    __amdgpu_buffer_rsrc_t rsrc;
    h8 ah[OQPRE], al[OQPRE];
    template <int SLOT>
    __device__ __forceinline__ void load(int l16, int off) {
        ah[SLOT] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, l16, off, 0));
        al[SLOT] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, l16, off + 1024, 0));
    }
};
// NB blocks (a multiple of the ring) with stream offsets off(q); the ring holds the next OQPRE blocks of the wave's sequence
// on entry and on exit (next(i): the i-th block behind this phase); the refill order is pinned (see hn_field2_hand_q.hip)
template <int NB, typename Off, typename Next, typename Body>
__device__ __forceinline__ void o_run_blocks(OStream& A, Off&& off, Next&& next, Body&& body) {
    static_assert(NB % OQPRE == 0, "phases are whole rings");
    const int l16 = lane_x16();
    static_for<NB>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        body(Q, A.ah[q % OQPRE], A.al[q % OQPRE]);
        if constexpr (q + OQPRE < NB)
            A.template load<q % OQPRE>(l16, off(q + OQPRE));
        else
            A.template load<q % OQPRE>(l16, next(q + OQPRE - NB));
        __builtin_amdgcn_sched_barrier(0);
    });
}
__device__ __forceinline__ f32x16 o_tail_tile_g(const __amdgpu_buffer_rsrc_t& rsrc, int off, int h) {
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, h * 64 + q * 16, off, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned wk = w[k];
            v[4 * q + k] = __builtin_bit_cast(float, wk);
        }
    }
    return v;
}
__device__ __forceinline__ void o_mma3(const h8& ah, const h8& al, const h8& xh, const h8& xl, f32x16& c1, f32x16& c2) {
    c1 = mfma16(ah, xh, c1);
    c2 = mfma16(ah, xl, c2);
    c2 = mfma16(al, xh, c2);
}

__global__ __launch_bounds__(256) void k_field2_obj_q(const Obj2Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f16_flush_mode();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    float* const a192x = reinterpret_cast<float*>(lds + 3 * OQ_EXCH);   // a4[192] of the block, one float per lane
    const int n_blocks = (a.n_pts + 31) / 32;
    OStream A;
    A.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.blob), 0, (int)a.blob_bytes, 0x00020000);
    // this wave's blocks: lin0 = tile `wave` of the layer's two chunks (4 tiles x 4 k-steps each); a hidden layer = its tiles
    // wave and wave + 4, 16 k-steps each
    auto l0_off = [&](int q) { return OQ_L0 + (q >> 2) * CB_L0 + (wave * 4 + (q & 3)) * KS_BYTES; };
    auto hid_off = [&](int lbase, int q) { return lbase + (wave + 4 * (q >> 4)) * CB_HID + (q & 15) * KS_BYTES; };
    if ((int)blockIdx.x < n_blocks) {
        const int l16 = lane_x16();
        static_for<OQPRE>([&](auto Q) { A.template load<decltype(Q)::value>(l16, l0_off(decltype(Q)::value)); });
    }
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int n = blk * 32 + j;
        const bool valid = n < a.n_pts;
        const int nn = valid ? n : a.n_pts - 1;
        const float p[3] = {a.pts[3 * nn], a.pts[3 * nn + 1], a.pts[3 * nn + 2]};
        h8 x0h[4], x0l[4];   // the X space fragments (every wave forms them: 27 sincos per lane)
        {
            float f[4][8];
            encode_x(p, h, f);
#pragma unroll
            for (int s = 0; s < 4; ++s) split8(f[s], x0h[s], x0l[s]);
        }
        h8 xh[16], xl[16];
        f32x16 c1[2], c2[2], nb[2];
        auto request_bias = [&](int lbase, int slot = 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) nb[k] = o_tail_tile_g(A.rsrc, lbase + (wave + 4 * k) * CB_HID + 16 * KS_BYTES + slot * 128, h);
        };
        // the two finished tiles -> softplus -> fragments of k-steps 2t, 2t + 1 (t = wave + 4 k) into exchange buffer xb;
        // n_tiles: the layer's tile count (lin3 has 7: tile 6 carries only neuron 192, tile 7 does not exist)
        auto publish = [&](int xb, int n_tiles) {
            char* const ex = lds + xb * OQ_EXCH;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = wave + 4 * k;
                EpiState st;
                arm(st);
                st.c1 = c1[k];
                st.c2 = c2[k];
                NoData nd;
                PhSoftplus ph;
                Epi<true, PhSoftplus, NoData> epi{st, ph, nd};
                epi.run_all();
                split_finish<true>(st);
                if (n_tiles == 7 && t == 6) {
                    a192x[lane] = st.v[0];                      // row 0 of the tile; lanes of half 0 hold neuron 192
                } else if (t < n_tiles) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        *reinterpret_cast<h8*>(ex + (2 * t + u) * KS_BYTES + lane * 16) = st.hi[u];
                        *reinterpret_cast<h8*>(ex + (2 * t + u) * KS_BYTES + 1024 + lane * 16) = st.lo[u];
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int ks_n = n_tiles == 7 ? 12 : 16;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s < ks_n) {
                    xh[s] = *reinterpret_cast<const h8*>(ex + s * KS_BYTES + lane * 16);
                    xl[s] = *reinterpret_cast<const h8*>(ex + s * KS_BYTES + 1024 + lane * 16);
                }
            }
        };
        auto hidden = [&](int lbase, auto&& next) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                c1[k] = nb[k];
                c2[k] = zero16();
            }
            o_run_blocks<32>(A, [&](int q) { return hid_off(lbase, q); }, next, [&](auto Q, const h8& ah, const h8& al) {
                constexpr int q = decltype(Q)::value;
                o_mma3(ah, al, xh[q & 15], xl[q & 15], c1[q >> 4], c2[q >> 4]);
            });
        };
        auto next_hidden = [&](int lbase) { return [lbase, &hid_off](int i) { return hid_off(lbase, i); }; };

        // ---- lin0: X -> a1; the biases are the tails of the two chunks (slot = tile within the chunk)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            c1[k] = o_tail_tile_g(A.rsrc, OQ_L0 + k * CB_L0 + 4 * 4 * KS_BYTES + wave * 128, h);
            c2[k] = zero16();
        }
        request_bias(OQ_L1);
        o_run_blocks<8>(A, [](int) { return 0; }, next_hidden(OQ_L1), [&](auto Q, const h8& ah, const h8& al) {
            constexpr int q = decltype(Q)::value;
            o_mma3(ah, al, x0h[q & 3], x0l[q & 3], c1[q >> 2], c2[q >> 2]);
        });
        publish(0, 8);
        hidden(OQ_L1, next_hidden(OQ_L2));
        request_bias(OQ_L2);
        publish(1, 8);
        hidden(OQ_L2, next_hidden(OQ_L3));
        request_bias(OQ_L3);
        publish(0, 8);
        // ---- lin3: 193 outputs = 7 tiles.  Wave 3 has no second tile: its blocks 16 .. 31 run on whatever follows in the
        //      stream (lin4's first tile) into accumulators nobody reads -- every wave keeps the same phase structure
        hidden(OQ_L3, next_hidden(OQ_L4));
        request_bias(OQ_L4);
        publish(1, 7);
        // ---- lin4 = [a4 (k-steps 0 .. 11) | X with a4[192] in its pad slot (k-step 15, half 1, element 7)] / sqrt2
        {
            const float v192 = other_half(a192x[lane], h);   // half 1 receives half 0's value
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                xh[12 + s] = x0h[s];
                xl[12 + s] = x0l[s];
            }
            const _Float16 vh = hi_part(v192);
            const _Float16 vl = (_Float16)((v192 - (float)vh) * LO_SCALE);
            xh[15][7] = h ? vh : xh[15][7];
            xl[15][7] = h ? vl : xl[15][7];
        }
        hidden(OQ_L4, next_hidden(OQ_L5));
        request_bias(OQ_L5);
        publish(0, 8);
        hidden(OQ_L5, next_hidden(OQ_L6));
        request_bias(OQ_L6);
        publish(1, 8);
        hidden(OQ_L6, next_hidden(OQ_L7));
        request_bias(OQ_L7);
        publish(0, 8);
        hidden(OQ_L7, [&](int i) { return l0_off(i); });   // (the next block's lin0; after the last block: unused)
        request_bias(OQ_L7, 1);                             // this wave's rows of W8[0, :]
        {
            float* const ex = reinterpret_cast<float*>(lds + OQ_EXCH);
            float* const ew = reinterpret_cast<float*>(lds + 2 * OQ_EXCH);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                EpiState st;
                arm(st);
                st.c1 = c1[k];
                st.c2 = c2[k];
                NoData nd;
                PhSoftplus ph;
                Epi<false, PhSoftplus, NoData> epi{st, ph, nd};
                epi.run_all();
                const int t = wave + 4 * k;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    ex[(t * 16 + i) * 64 + lane] = st.v[i];
                    ew[(t * 16 + i) * 64 + lane] = nb[k][i];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (wave == 0) {
                float sdf_acc = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sdf_acc = fmaf(ew[(t * 16 + i) * 64 + lane], ex[(t * 16 + i) * 64 + lane], sdf_acc);
                const float sdf = (half_sum(sdf_acc) + a.b8[0]) * a.inv_scale;
                if (valid && h == 0) a.sdf[n] = sdf;
            }
        }
    }
}

int launch_field2_obj_q(const Obj2Args& a, int n_blocks, int n_cus, hipStream_t stream) {
    if (a.blob_bytes != (size_t)OQ_END) {
        set_error("latency-form object kernel: stream of %zu bytes, expected %d", a.blob_bytes, OQ_END);
        return HN_EINVAL;
    }
    const int grid = n_blocks < n_cus ? n_blocks : n_cus;
    static std::atomic<uint64_t> lds_q{0};
    HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_obj_q), (int)OBJQ_LDS, &lds_q));
    hipLaunchKernelGGL(k_field2_obj_q, dim3(grid), dim3(256), OBJQ_LDS, stream, a);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace v2
}  // namespace hn
