// Adjoint of hn_field_eval: d loss / d (pts, rays_d, bt_inv, T_pose) from d loss / d (sdf, grad, rgb).
//
// The reference gets this from autograd through the two networks including the second-order path through
// `.gradient()` (utils/fields.py:165-177, 336-347 with create_graph=True; consumed by fitting_single.py:289-291,
// fitting_video.py:340-342).  Here the backward pass is the explicit sequence of sweeps specified and checked against
// autograd in oracle/field_bwd.py (steps 1-6 there), run as a sequence of launches over row-major [n, width] fp32
// arrays in the caller's workspace:
//   * k_dense: C (+)= alpha * A[n,K] * op(W) (+ bias) on v_mfma_f32_32x32x2_f32 (fp32 operands, fp32 accumulate),
//     64 x 64 output tile per workgroup, operands staged through LDS; W is addressed with two strides so that W, W^T
//     and column blocks of W (the skip / colour-input blocks) need no copies;
//   * element-wise kernels for the softplus tape and the products between the sweeps;
//   * the input maps (object: [p, enc10(p)]; hand: the 21-bone encoding) with their Jacobian, transposed Jacobian and
//     second-order terms.
// This first version favours being checkable step by step over speed: the fitting configurations evaluate ~4e4 samples
// per step, where the whole adjoint is a few milliseconds.
#include <functional>
#include <type_traits>

#include "hn_common.h"

namespace hn {
namespace v2 {
size_t field2_obj_adj_workspace_bytes(int n_pts, int n_cus);
int launch_field2_obj_adj(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int spr, const float* g_sdf,
                          const float* g_grad, const float* g_rgb, float* g_pts, float* g_rays_d, void* workspace,
                          size_t workspace_bytes, hipStream_t stream, const void* tape, const float* grad, const float* rgb,
                          float* sig = nullptr, size_t sig_pitch = 0, float* gb_out = nullptr);
int launch_field2_obj(const hn_field*, const float*, const float*, int, int, float*, float*, float*, float*, void*, size_t, bool, hipStream_t, void*, size_t);
size_t field2_obj_tape_bytes(int n_pts);
int field2_obj_signal_arrays();
size_t field2_hand_adj_workspace_bytes(int n_pts, int n_cus);
int launch_field2_hand_adj(const hn_field* f, const float* pts, int n_pts, const float* bt_inv, const float* T_pose, int n_frames,
                           int pts_per_frame, const float* g_sdf, const float* g_grad, const float* g_rgb, float* g_pts,
                           float* g_bt_inv, float* g_T_pose, void* workspace, size_t workspace_bytes, hipStream_t stream,
                           const void* tape, const float* grad, const float* rgb, float* sig = nullptr, size_t sig_pitch = 0,
                           float* gb_out = nullptr);
int launch_field2_hand(const hn_field*, const float*, int, const float*, const float*, int, int, float*, float*, float*, float*, void*, size_t, bool,
                       hipStream_t, void* tape, size_t tape_bytes);
size_t field2_hand_tape_bytes(int n_pts);
int field2_hand_signal_arrays();
size_t field2_hand_pose_rows_bytes(int n_pts);
}
namespace bwd {

// The fused adjoint kernels (hn_field2_*.hip, MODE 2) serve HN_PREC_F16X3 fields; the launch sequence below stays as
// the HN_PREC_FP32 implementation (an independent second opinion in the tests) and for the sdf-only adjoint.
static bool fused_adjoint(const hn_field* f, bool sdf_only) {
    return f->precision == HN_PREC_F16X3 && f->v2_adj != nullptr && !sdf_only;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float BETA = 100.f;

__device__ __forceinline__ float softplus(float z) {   // nn.Softplus(beta=100, threshold=20)
    return BETA * z > 20.f ? z : log1pf(expf(BETA * z)) / BETA;
}
// ---- C[n,M] (+)= alpha * A[n,K] * B (+ bias),  B(k, col) = W[k * wsk + col * wsc] --------------------------------
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

// ---- parameter gradients (`loss.backward()` into the networks, exp_runner.py:208-242) ---------------
// dW[m, k] += alpha * sum_i A[i, m] B[i, k]  (i < n): the outer products of a layer's adjoint signal with its input,
// reduced over the samples.  Workgroup = 128 x 128 tile of dW over one slice of the samples (grid z), 4 waves as 2 x 2
// quadrants of 2 x 2 MFMA tiles (v_mfma_f32_32x32x2_f32: the sample index is the MFMA's k; 4 MFMAs per 4 LDS reads),
// operands staged through LDS 32 samples at a time with row-contiguous (coalesced) global reads, the next step's reads
// issued before the barrier; partial tiles added with atomics (the caller zeroes dW).  With db != NULL the column K of B is taken as the
// constant 1: dW's column K is the bias gradient db[m] += alpha_b * sum_i A[i, m].
struct OuterArgs {
    const float* A;
    int lda, M;
    const float* B;
    int ldb, K;
    float* dW;
    int ldw;
    float* db;
    float alpha, alpha_b;
    int n, chunk;
    int dbg;
};
__global__ __launch_bounds__(256) void k_outer(const OuterArgs a) {
    constexpr int T = 128;                         // workgroup tile of dW: T x T, one 64 x 64 quadrant per wave
    __shared__ float As[32][T + 1];
    __shared__ float Bs[32][T + 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1, h = lane >> 5, j = lane & 31;
    const int m0 = blockIdx.y * T, k0 = blockIdx.x * T;
    const int KB = a.db != nullptr ? a.K + 1 : a.K;
    const int i_begin = blockIdx.z * a.chunk;
    const int i_end = min(a.n, i_begin + a.chunk);
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    // staging: thread = (column of the tile, samples s0, s0 + 2, ...): a wave reads 64 consecutive floats of a row
    const int col = t & (T - 1), s0 = t >> 7;
    const bool a_col = m0 + col < a.M;
    const int kc = k0 + col;
    const int b_kind = kc < a.K ? 0 : kc < KB ? 1 : 2;     // 0: column of B, 1: the constant 1 (bias gradient), 2: padding
    const float* pa = a.A + m0 + col;
    const float* pb = a.B + kc;
    float va[16], vb[16];
    auto fetch = [&](int i0) {                     // 32 samples x this thread's column of both operands -> registers
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + s0 + 2 * e;
            const bool live = i < i_end;
            va[e] = (live && a_col) ? pa[(size_t)i * a.lda] : 0.f;
            vb[e] = !live ? 0.f : b_kind == 0 ? pb[(size_t)i * a.ldb] : b_kind == 1 ? 1.f : 0.f;
        }
    };
    if (i_begin < i_end) fetch(i_begin);
    for (int i0 = i_begin; i0 < i_end; i0 += 32) {
        __syncthreads();                           // the previous step's MFMAs have read the tiles
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            As[s0 + 2 * e][col] = va[e];
            Bs[s0 + 2 * e][col] = vb[e];
        }
        __syncthreads();
        if (i0 + 32 < i_end) fetch(i0 + 32);       // the next step's reads are in flight under this step's MFMAs
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            float av[2], bv[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) av[x] = As[2 * ks + h][wr * 64 + x * 32 + j];
#pragma unroll
            for (int y = 0; y < 2; ++y) bv[y] = Bs[2 * ks + h][wc * 64 + y * 32 + j];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    if (!(a.dbg & 2) || ks == 0) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
    }
    if ((a.dbg & 1) && acc[0][0][0] != 12345.f) return;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int c = k0 + wc * 64 + y * 32 + j;
        if (c >= KB) continue;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < a.M) {
                    if (c < a.K)
                        atomicAdd(a.dW + (size_t)m * a.ldw + c, a.alpha * acc[x][y][r]);
                    else
                        atomicAdd(a.db + m, a.alpha_b * acc[x][y][r]);
                }
            }
    }
}
// ---- the same outer products, GROUPED: every product of one backward pass in ONE launch, on the bf16 MFMA -------------------------------
// A parameter-gradient pass forms ~20 .. 30 such products from arrays that all stand when the adjoint kernel has finished.  Launched one by
// one (k_outer) each needs ~128 slices of the samples to fill the chip and pays 128 x M x K atomics for it (~0.5 ms of a 2.9 ms total on the
// object field) plus a ramp and a tail per launch; together they are ~100 tiles of 128 x 128, so ~5 slices fill 512 workgroup slots and the
// atomics drop 25-fold.  Work item = (product, m tile, k tile, slice), consecutive items = the tiles of one (product, slice), which share their
// operand rows: the block index is mapped so that they run on ONE XCD (blocks b, b + 8, ... are neighbours there) and meet in its L2.
// Arithmetic: fp32 operands split three ways into bf16 (x = h + m + l EXACTLY: 8 + 8 + 8 significant bits, round-to-nearest each, fp32's exponent range,
// no scaling) and six v_mfma_f32_32x32x16_bf16 per tile pair (hh, hm, mh, mm, hl, lh: what is dropped is below 2^-23 of the product) -- 6 x 32
// cycles per 32 x 32 x 16 block against 8 x 64 on v_mfma_f32_32x32x2_f32.  A lane's MFMA operand is 8 consecutive SAMPLES of one column: the
// staging thread owns a column and 16 consecutive samples (row-coalesced dword loads, as k_outer), so the transposition costs nothing -- two
// ds_write_b128 per part; LDS image [part][column][32 samples] at an 80-byte column pitch (conflict-free for the 8-lane groups of the stores
// and the 16-lane groups of ds_read_b128).  Bias gradients: the staging threads of the first k tile sum their column on the way.
// A desc may carry a SECOND operand pair accumulated into the same tile (dW_l = zb_l^T a_{l-1} + dz_l^T v_{l-1}: one set of atomics for both).
// Measured and not kept (round 5): a second register set of loads in flight (two steps ahead) 1.3 -> 1.6 ms; eight-wave workgroups with four
// producer waves (loads two steps ahead, split, LDS images) and four consumer waves (MFMAs), two image buffers, one barrier per step, one
// workgroup per CU: 1.0 -> 1.15 ms (obj), 1.2 -> 2.0 ms (hand) -- the step's three parts (loads alone 0.5 ms, MFMAs +0.34, splits +0.17)
// overlap no better that way than across two independent workgroups per CU.
struct OuterDesc {
    const float *A, *B, *A2, *B2;
    float *dW, *db;
    int lda, ldb, lda2, ldb2, M, K, ldw, kt, item0, chunk;   // chunk: samples per slice of THIS product (a two-pair product gets twice the slices)
    float alpha, alpha_b;
};
constexpr int OUTER_MAX_DESCS = 32;
struct OuterGroupArgs {
    OuterDesc d[OUTER_MAX_DESCS];
    int n_desc, n, items, per_xcd, dbg;
};
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int OG_T = 128, OG_PITCH = 80;                       // tile edge; bytes per column of one part's image (32 samples x 2 B + 16)
constexpr int OG_PART = OG_T * OG_PITCH, OG_LDS = 6 * OG_PART;  // 61,440 B
__global__ __launch_bounds__(256, 2) void k_outer_group(const OuterGroupArgs g) {
    __shared__ __attribute__((aligned(16))) char lds[OG_LDS];
    // block -> item: blocks b, b + 8, b + 16, ... share an XCD; item ids run along them
    const int item = (blockIdx.x & 7) * g.per_xcd + (blockIdx.x >> 3);
    if (item >= g.items) return;
    int p = 0;
    while (p + 1 < g.n_desc && g.d[p + 1].item0 <= item) ++p;
    const OuterDesc& d = g.d[p];
    const int local = item - d.item0;
    const int mt_n = (d.M + OG_T - 1) / OG_T;
    const int tiles = mt_n * d.kt;
    const int slice = local / tiles, tile = local - slice * tiles;
    const int mt = tile / d.kt, kt = tile - mt * d.kt;
    const int m0 = mt * OG_T, k0 = kt * OG_T;
    const int i_begin = slice * d.chunk, i_end = min(g.n, i_begin + d.chunk);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1, h = lane >> 5, j = lane & 31;
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    // staging: thread = (column of the tile, 16 consecutive samples of the step's 32)
    const int col = t & (OG_T - 1), s0 = (t >> 7) * 16;
    const bool a_ok = m0 + col < d.M, b_ok = k0 + col < d.K;
    const bool want_bias = d.db != nullptr && kt == 0;
    float bsum = 0.f;
    char* const wbase = lds + col * OG_PITCH + s0 * 2;                       // this thread's 32 bytes of a part: + part * OG_PART
    const char* const ra = lds + (wr * 64 + j) * OG_PITCH + h * 16;          // A parts 0..2; + x * 32 columns, + ks * 32 bytes
    const char* const rb = lds + 3 * OG_PART + (wc * 64 + j) * OG_PITCH + h * 16;
    const int n_pairs = d.A2 != nullptr ? 2 : 1;
    const int rows = i_end - i_begin;
    for (int pair = 0; pair < n_pairs; ++pair) {
        // The slice of each operand as a raw buffer that ends with the slice's last row: rows past i_end and (offset pushed out of range)
        // columns past the matrix edge read as 0 without a predicate, and the row part of an address is wave-uniform arithmetic.
        const int lda = pair == 0 ? d.lda : d.lda2, ldb = pair == 0 ? d.ldb : d.ldb2;
        const float* A = pair == 0 ? d.A : d.A2;
        const float* B = pair == 0 ? d.B : d.B2;
        const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)i_begin * lda + m0), 0,
                                                                             rows > 0 ? ((rows - 1) * lda + min(OG_T, d.M - m0)) * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(B != nullptr ? B + (size_t)i_begin * ldb + k0 : A), 0,
            (rows > 0 && B != nullptr) ? ((rows - 1) * ldb + min(OG_T, d.K - k0)) * 4 : 0, 0x00020000);
        const unsigned OUT = 0x80000000u;                                   // beyond any slice (launch(): a slice is < 2^31 bytes)
        const unsigned voa = a_ok ? (unsigned)col * 4u : OUT, vob = b_ok ? (unsigned)col * 4u : OUT;
        float va[16], vb[16];
        auto fetch = [&](int i0) {                     // rows i0 + s0 .. + 15 of both operands, this thread's column
            const unsigned ra0 = (unsigned)(i0 - i_begin + s0) * (unsigned)lda * 4u, rb0 = (unsigned)(i0 - i_begin + s0) * (unsigned)ldb * 4u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                va[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra_, voa + ra0 + (unsigned)(e * lda * 4), 0, 0));
                vb[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb_, vob + rb0 + (unsigned)(e * ldb * 4), 0, 0));
            }
        };
        auto stage = [&](const float* v, char* w) {   // 16 floats -> h | m | l images (x = h + m + l exactly, each bf16 by RNE), 2 x 16 bytes each
            unsigned hh[8], mm[8], ll[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x2 x = {v[2 * q], v[2 * q + 1]};
                hh[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
                const f32x2 r = {x[0] - __uint_as_float(hh[q] << 16), x[1] - __uint_as_float(hh[q] & 0xffff0000u)};
                mm[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
                const f32x2 r2 = {r[0] - __uint_as_float(mm[q] << 16), r[1] - __uint_as_float(mm[q] & 0xffff0000u)};
                ll[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
            }
            uint4* w0 = reinterpret_cast<uint4*>(w);
            uint4* w1 = reinterpret_cast<uint4*>(w + OG_PART);
            uint4* w2 = reinterpret_cast<uint4*>(w + 2 * OG_PART);
            w0[0] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
            w0[1] = make_uint4(hh[4], hh[5], hh[6], hh[7]);
            w1[0] = make_uint4(mm[0], mm[1], mm[2], mm[3]);
            w1[1] = make_uint4(mm[4], mm[5], mm[6], mm[7]);
            w2[0] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
            w2[1] = make_uint4(ll[4], ll[5], ll[6], ll[7]);
        };
        if (i_begin < i_end) fetch(i_begin);
        for (int i0 = i_begin; i0 < i_end; i0 += 32) {
            __syncthreads();                           // the previous step's MFMAs have read the images
            if (!(g.dbg & 4) || i0 == i_begin) {
                stage(va, wbase);
                stage(vb, wbase + 3 * OG_PART);
            } else if (va[3] == 12345.f || vb[5] == 12345.f) {
                bsum += 1.f;
            }
            if (want_bias && pair == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) bsum += va[e];
            }
            __syncthreads();
            if (i0 + 32 < i_end) fetch(i0 + 32);       // the next step's reads are in flight under this step's MFMAs
            if (d.K > 0 && !(g.dbg & 2)) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 A_[2][3], B_[2][3];
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int part = 0; part < 3; ++part)
                            A_[x][part] = *reinterpret_cast<const bf16x8*>(ra + part * OG_PART + x * 32 * OG_PITCH + ks * 32);
#pragma unroll
                    for (int y = 0; y < 2; ++y)
#pragma unroll
                        for (int part = 0; part < 3; ++part)
                            B_[y][part] = *reinterpret_cast<const bf16x8*>(rb + part * OG_PART + y * 32 * OG_PITCH + ks * 32);
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int y = 0; y < 2; ++y) {   // small terms first
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[x][2], B_[y][0], acc[x][y], 0, 0, 0);
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[x][0], B_[y][2], acc[x][y], 0, 0, 0);
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[x][1], B_[y][1], acc[x][y], 0, 0, 0);
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[x][1], B_[y][0], acc[x][y], 0, 0, 0);
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[x][0], B_[y][1], acc[x][y], 0, 0, 0);
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[x][0], B_[y][0], acc[x][y], 0, 0, 0);
                        }
                }
            }
        }
    }
    if (want_bias && a_ok) atomicAdd(d.db + m0 + col, d.alpha_b * bsum);
    if (d.K <= 0) return;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int c = k0 + wc * 64 + y * 32 + j;
        if (c >= d.K) continue;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < d.M) atomicAdd(d.dW + (size_t)m * d.ldw + c, d.alpha * acc[x][y][r]);
            }
    }
}
// out[c] += scale * sum_i x[i, c]   (c < width <= 256; the gradient of the row W_8[0, :] that seeds the reverse sweep)
__global__ void k_colsum(const float* __restrict__ x, int n, int ld, int width, float scale, float* __restrict__ out) {
    const int c = threadIdx.x;
    const int i0 = blockIdx.x * 256, i1 = min(n, i0 + 256);
    if (c >= width) return;
    float s = 0.f;
    for (int i = i0; i < i1; ++i) s += x[(size_t)i * ld + c];
    atomicAdd(out + c, scale * s);
}

// ---- element-wise --------------------------------------------------------------------------------------------
__device__ __forceinline__ float sig_from_act(float a) { return 1.f - expf(-BETA * a); }   // sigma'(z) from a = softplus(z)

// The tape arrays are multiples of 4 floats only by luck of the widths (193-wide rows are not), so the element-wise
// kernels process 4 consecutive floats per thread with a scalar tail instead of assuming float4 alignment of rows; the
// arrays themselves start 256-byte aligned (Arena).
template <typename F>
__device__ __forceinline__ void for4(size_t n, F&& f) {
    const size_t i4 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4;
    if (i4 + 4 <= n) {
        f(i4, std::integral_constant<int, 4>{});
    } else {
        for (size_t i = i4; i < n; ++i) f(i, std::integral_constant<int, 1>{});
    }
}
template <int V>
struct Vec;
template <>
struct Vec<4> {
    float v[4];
    __device__ __forceinline__ static Vec ld(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        return {{t.x, t.y, t.z, t.w}};
    }
    __device__ __forceinline__ void st(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <>
struct Vec<1> {
    float v[1];
    __device__ __forceinline__ static Vec ld(const float* p) { return {{*p}}; }
    __device__ __forceinline__ void st(float* p) const { *p = v[0]; }
};
#define HN_EW(NAME, ARGS, ...)                                                       \
    __global__ void NAME ARGS {                                                      \
        for4(n, [&](size_t i, auto V_) {                                             \
            constexpr int V = decltype(V_)::value;                                   \
            __VA_ARGS__                                                              \
        });                                                                          \
    }
HN_EW(k_softplus, (float* __restrict__ z, size_t n), {
    Vec<V> x = Vec<V>::ld(z + i);
    for (int k = 0; k < V; ++k) x.v[k] = softplus(x.v[k]);
    x.st(z + i);
})
__global__ void k_bcast_row(const float* __restrict__ w, int width, float scale, float* __restrict__ out, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = w[i % width] * scale;
}
// dz = sigma'(z) * u
HN_EW(k_dz, (const float* __restrict__ act, const float* __restrict__ u, float* __restrict__ dz, size_t n), {
    const Vec<V> a = Vec<V>::ld(act + i), uu = Vec<V>::ld(u + i);
    Vec<V> o;
    for (int k = 0; k < V; ++k) o.v[k] = sig_from_act(a.v[k]) * uu.v[k];
    o.st(dz + i);
})
// forward-direction sweep: sb = u * dzb (kept), v = sigma' * dzb (input of the next product)
HN_EW(k_fwd_dir, (const float* __restrict__ act, const float* __restrict__ u, const float* __restrict__ dzb, float* __restrict__ sb,
                  float* __restrict__ v, size_t n), {
    const Vec<V> a = Vec<V>::ld(act + i), uu = Vec<V>::ld(u + i), d = Vec<V>::ld(dzb + i);
    Vec<V> o1, o2;
    for (int k = 0; k < V; ++k) {
        o1.v[k] = uu.v[k] * d.v[k];
        o2.v[k] = sig_from_act(a.v[k]) * d.v[k];
    }
    o1.st(sb + i);
    if (v != nullptr) o2.st(v + i);
})
// zb = sigma' * ab + sigma'' * sb
HN_EW(k_zb, (const float* __restrict__ act, const float* __restrict__ ab, const float* __restrict__ sb, float* __restrict__ zb, size_t n), {
    const Vec<V> a = Vec<V>::ld(act + i), x = Vec<V>::ld(ab + i), y = Vec<V>::ld(sb + i);
    Vec<V> o;
    for (int k = 0; k < V; ++k) {
        const float s = sig_from_act(a.v[k]);
        o.v[k] = s * x.v[k] + BETA * s * (1.f - s) * y.v[k];
    }
    o.st(zb + i);
})
HN_EW(k_relu, (float* __restrict__ x, size_t n), {
    Vec<V> t = Vec<V>::ld(x + i);
    for (int k = 0; k < V; ++k) t.v[k] = fmaxf(t.v[k], 0.f);
    t.st(x + i);
})
HN_EW(k_relu_mask, (const float* __restrict__ act, float* __restrict__ xb, size_t n), {
    const Vec<V> a = Vec<V>::ld(act + i);
    Vec<V> t = Vec<V>::ld(xb + i);
    for (int k = 0; k < V; ++k) t.v[k] = a.v[k] > 0.f ? t.v[k] : 0.f;
    t.st(xb + i);
})
#undef HN_EW
// rgb = sigmoid(zc); xb = g_rgb * rgb (1 - rgb)
__global__ void k_rgb_seed(const float* __restrict__ zc, const float* __restrict__ g_rgb, float* __restrict__ xb, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) {
        const float r = 1.f / (1.f + expf(-zc[i]));
        xb[i] = g_rgb[i] * r * (1.f - r);
    }
}
// z8 adjoint: [g_sdf / scale, fb]  (row width 257)
__global__ void k_z8_bar(const float* __restrict__ g_sdf, float inv_scale, float* __restrict__ z8b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z8b[(size_t)i * 257] = g_sdf[i] * inv_scale;
}

// ---- [x, enc_L(x)] of a 3-vector: features, J^T, J, second derivative ------------------------------------------
// layout: x (3), then per channel c: sin(2^k x_c) k<L, cos(2^k x_c) k<L
// (rows of W = 3 + 6 L <= 63 floats: a lane that wrote its own row made every store instruction touch 64 cache lines; each wave stages its 64
// rows in an LDS tile [W][65] and writes them one row per store instruction, W lanes wide -- blocks of 256 threads, a tile per wave)
template <int W>
__device__ __forceinline__ void store_row_tile(const float* tile, float* __restrict__ out, int ld, int i0, int n) {
    const int l = threadIdx.x & 63;
    const int rows = n - i0 < 64 ? n - i0 : 64;
    if (l < W)
        for (int sidx = 0; sidx < rows; ++sidx) out[(size_t)(i0 + sidx) * ld + l] = tile[l * 65 + sidx];
}
template <int L>
__global__ __launch_bounds__(256) void k_enc3(const float* __restrict__ x, int n, int rep, float* __restrict__ out, int ld) {
    constexpr int W = 3 + 6 * L;
    __shared__ float tiles[4][W * 65];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float* tile = tiles[threadIdx.x >> 6];
    const int ls = threadIdx.x & 63;
    if (i < n) {
        const float* xi = x + 3 * (size_t)(i / rep);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            tile[c * 65 + ls] = xi[c];
#pragma unroll
            for (int k = 0; k < L; ++k) {
                float s, co;
                sincosf(xi[c] * (float)(1 << k), &s, &co);
                tile[(3 + c * 2 * L + k) * 65 + ls] = s;
                tile[(3 + c * 2 * L + L + k) * 65 + ls] = co;
            }
        }
    }
    __syncthreads();
    store_row_tile<W>(tile, out, ld, blockIdx.x * blockDim.x + (threadIdx.x & ~63), n);
}
// out[i] (+)= J^T fbar [+ second-order term sum_f GX_f d2X_f gbar_c];  fbar row width ld
template <int L>
__global__ void k_enc3_pull(const float* __restrict__ x, int n, int rep, const float* __restrict__ fbar, int ld,
                            const float* __restrict__ GX, int ldg, const float* __restrict__ gbar, float* __restrict__ out,
                            int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* xi = x + 3 * (size_t)(i / rep);
    const float* fb = fbar + (size_t)i * ld;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc = fb[c];
        float sec = 0.f;
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const float f = (float)(1 << k);
            float s, co;
            sincosf(xi[c] * f, &s, &co);
            acc += f * co * fb[3 + c * 2 * L + k] - f * s * fb[3 + c * 2 * L + L + k];
            if (GX != nullptr) {
                const float* gx = GX + (size_t)i * ldg;
                sec += -f * f * (s * gx[3 + c * 2 * L + k] + co * gx[3 + c * 2 * L + L + k]);
            }
        }
        if (GX != nullptr) acc += sec * gbar[3 * (size_t)i + c];
        float* o = out + 3 * (size_t)i + c;
        *o = accumulate ? *o + acc : acc;
    }
}
// J gbar -> [n, 3 + 6L]
template <int L>
__global__ __launch_bounds__(256) void k_enc3_push(const float* __restrict__ x, int n, const float* __restrict__ gbar, float* __restrict__ out, int ld) {
    constexpr int W = 3 + 6 * L;
    __shared__ float tiles[4][W * 65];   // (staged stores: k_enc3)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float* tile = tiles[threadIdx.x >> 6];
    const int ls = threadIdx.x & 63;
    if (i < n) {
        const float* xi = x + 3 * (size_t)i;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gb = gbar[3 * (size_t)i + c];
            tile[c * 65 + ls] = gb;
#pragma unroll
            for (int k = 0; k < L; ++k) {
                const float f = (float)(1 << k);
                float s, co;
                sincosf(xi[c] * f, &s, &co);
                tile[(3 + c * 2 * L + k) * 65 + ls] = f * co * gb;
                tile[(3 + c * 2 * L + L + k) * 65 + ls] = -f * s * gb;
            }
        }
    }
    __syncthreads();
    store_row_tile<W>(tile, out, ld, blockIdx.x * blockDim.x + (threadIdx.x & ~63), n);
}
// rows of x [n, ld] scaled by s [n]
__global__ void k_scale_rows(float* __restrict__ x, const float* __restrict__ sc, int ld, size_t total) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < total) x[i] *= sc[i / ld];
}
__global__ void k_sigmoid(float* __restrict__ x, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) x[i] = 1.f / (1.f + expf(-x[i]));
}
__global__ void k_add3(const float* __restrict__ a, float* __restrict__ b, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) b[i] += a[i];
}
// per-ray sum of per-sample direction gradients
__global__ void k_sum_rays(const float* __restrict__ per_sample, int n_rays, int spr, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * 3) return;
    const int ray = i / 3, c = i % 3;
    float s = 0.f;
    for (int k = 0; k < spr; ++k) s += per_sample[3 * ((size_t)ray * spr + k) + c];
    out[i] = s;
}


// ---- hand input map (utils/fields.py:22-36, 134-147) ---------------------------------------------------------------
// Per bone b: q = R_b p + t_b - T_b, v = |q|, r = q / v, h = 1 - sigmoid(200 (v - cutoff_b)); 66 features
// [v, sin(2^k v) k<10, cos(2^k v) k<10, r (3), per c: sin(2^k r_c) k<7, cos(2^k r_c) k<7] * h.
// A feature is phi(y) h(v) with y one of (v, r_0, r_1, r_2); for a weight row G the bone's scalar function
// F(q) = sum_f G_f F_f(q) has  grad F = Sv r + sum_i Sr_i (e_i - r_i r) / v  with
//   S0_a = sum_{f on argument a} G phi,  S1_a = sum G phi',  S2_a = sum G phi'',  tot = sum_a S0_a,
//   Sv = h S1_v + h' tot,  Sr_i = h S1_{r_i}            (oracle/field_bwd.py, _HandInput)
__constant__ float c_cut[N_BONES] = {0.08f, 0.03f, 0.03f, 0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f, 0.03f, 0.02f,
                                     0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f};
constexpr float TAU = 200.f;
struct BoneQ {
    float v, r[3], h, h1, h2;
};
__device__ __forceinline__ BoneQ bone_q(const float p[3], const float* __restrict__ m, const float* __restrict__ T, int b) {
    BoneQ o;
    const float q0 = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3] - T[0];
    const float q1 = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7] - T[1];
    const float q2 = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11] - T[2];
    o.v = sqrtf(q0 * q0 + q1 * q1 + q2 * q2);
    o.r[0] = q0 / o.v;
    o.r[1] = q1 / o.v;
    o.r[2] = q2 / o.v;
    const float sg = 1.f / (1.f + expf(-TAU * (o.v - c_cut[b])));
    o.h = 1.f - sg;
    o.h1 = -TAU * sg * (1.f - sg);
    o.h2 = -TAU * TAU * sg * (1.f - sg) * (1.f - 2.f * sg);
    return o;
}
// visits the 66 features of a bone in the reference's order: fn(f, argument a in 0..3, phi, phi', phi'')
template <typename Fn>
__device__ __forceinline__ void bone_features(const BoneQ& q, Fn&& fn) {
    fn(0, 0, q.v, 1.f, 0.f);
    for (int k = 0; k < 10; ++k) {
        const float f = (float)(1 << k);
        float s, c;
        sincosf(q.v * f, &s, &c);
        fn(1 + k, 0, s, f * c, -f * f * s);
        fn(11 + k, 0, c, -f * s, -f * f * c);
    }
    for (int i = 0; i < 3; ++i) fn(21 + i, 1 + i, q.r[i], 1.f, 0.f);
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 7; ++k) {
            const float f = (float)(1 << k);
            float s, c;
            sincosf(q.r[i] * f, &s, &c);
            fn(24 + i * 14 + k, 1 + i, s, f * c, -f * f * s);
            fn(24 + i * 14 + 7 + k, 1 + i, c, -f * s, -f * f * c);
        }
}
struct FrameRef {
    const float* M;   // [21,4,4]
    const float* T;   // [21,3]
    int frame;
};
__device__ __forceinline__ FrameRef frame_of(int i, int ppf, int nf, const float* bt_inv, const float* T_pose) {
    int fr = i / ppf;
    fr = fr < nf ? fr : nf - 1;
    return {bt_inv + (size_t)fr * N_BONES * 16, T_pose + (size_t)fr * N_BONES * 3, fr};
}

constexpr int HAND_BONE_F = 66;   // values per (sample, bone): v, enc10(v), r, enc7(r)
// tile [66][65] (value f of the wave's sample s at f * 65 + s) -> rows i0 .. i0 + 63 of `out`, columns 66 b ..: per sample one 256-byte store
// of the wave and one of two lanes (reads: lane l takes f = l, stride 65 floats -- two lanes per bank)
__device__ __forceinline__ void store_bone_tile(const float* tile, float* __restrict__ out, int ld, int i0, int n, int b) {
    const int l = threadIdx.x;
    const int rows = n - i0 < 64 ? n - i0 : 64;
    for (int sidx = 0; sidx < rows; ++sidx) {
        float* o = out + (size_t)(i0 + sidx) * ld + b * HAND_BONE_F;
        o[l] = tile[l * 65 + sidx];
        if (l < HAND_BONE_F - 64) o[64 + l] = tile[(64 + l) * 65 + sidx];
    }
}
__global__ void k_hand_feat(const float* __restrict__ pts, int n, int ppf, int nf, const float* __restrict__ bt_inv,
                            const float* __restrict__ T_pose, float* __restrict__ X, int ld, float* __restrict__ r_out,
                            float* __restrict__ h_out) {
    // one wave = 64 samples x ONE bone (grid row: 21 x the parallelism of a loop).  The 66 values of a (sample, bone) are a 264-byte piece
    // of the sample's row: written by the lane that computed them they are 66 store instructions of 64 different cache lines each (the
    // kernel ran at 0.7 TB/s, store-bound); staged through LDS, the wave writes every sample's piece as one 256-byte + one 8-byte store.
    __shared__ float tile[HAND_BONE_F * 65];
    const int i0 = blockIdx.x * blockDim.x, i = i0 + threadIdx.x, b = blockIdx.y;
    if (i < n) {
        const float p[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
        const FrameRef fr = frame_of(i, ppf, nf, bt_inv, T_pose);
        const BoneQ q = bone_q(p, fr.M + 16 * b, fr.T + 3 * b, b);
        bone_features(q, [&](int f, int, float phi, float, float) { tile[f * 65 + threadIdx.x] = phi * q.h; });
        if (r_out != nullptr) {
            float* ro = r_out + ((size_t)i * N_BONES + b) * 3;
            ro[0] = q.r[0];
            ro[1] = q.r[1];
            ro[2] = q.r[2];
        }
        if (h_out != nullptr) h_out[(size_t)i * N_BONES + b] = q.h;
    }
    __syncthreads();
    store_bone_tile(tile, X, ld, i0, n, b);
}
// J gbar: directional derivative of every feature along dq = R_b gbar
__global__ void k_hand_push(const float* __restrict__ pts, int n, int ppf, int nf, const float* __restrict__ bt_inv,
                            const float* __restrict__ T_pose, const float* __restrict__ gbar, float* __restrict__ out, int ld) {
    __shared__ float tile[HAND_BONE_F * 65];   // (staged stores: k_hand_feat)
    const int i0 = blockIdx.x * blockDim.x, i = i0 + threadIdx.x, b = blockIdx.y;
    if (i < n) {
        const float p[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
        const float gb[3] = {gbar[3 * (size_t)i], gbar[3 * (size_t)i + 1], gbar[3 * (size_t)i + 2]};
        const FrameRef fr = frame_of(i, ppf, nf, bt_inv, T_pose);
        const float* m = fr.M + 16 * b;
        const BoneQ q = bone_q(p, m, fr.T + 3 * b, b);
        const float w[3] = {m[0] * gb[0] + m[1] * gb[1] + m[2] * gb[2], m[4] * gb[0] + m[5] * gb[1] + m[6] * gb[2],
                            m[8] * gb[0] + m[9] * gb[1] + m[10] * gb[2]};
        const float rw = q.r[0] * w[0] + q.r[1] * w[1] + q.r[2] * w[2];
        const float dy[4] = {rw, (w[0] - q.r[0] * rw) / q.v, (w[1] - q.r[1] * rw) / q.v, (w[2] - q.r[2] * rw) / q.v};
        bone_features(q, [&](int f, int a, float phi, float phi1, float) { tile[f * 65 + threadIdx.x] = phi1 * q.h * dy[a] + phi * q.h1 * rw; });
    }
    __syncthreads();
    store_bone_tile(tile, out, ld, i0, n, b);
}
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// one atomic per wave when all its samples belong to one frame (the usual case), else one per lane
__device__ __forceinline__ void pose_add(float* addr, float v, bool uniform, int lane) {
    if (uniform) {
        const float s = wave_sum64(v);
        if (lane == 0) atomicAdd(addr, s);
    } else if (v != 0.f) {
        atomicAdd(addr, v);
    }
}
// adjoint of q = R p + t - T for one (sample, bone): g_pts, and the pose gradients (qb must be 0 in inactive lanes)
__device__ __forceinline__ void spread(const float qb[3], const float p[3], const float* __restrict__ m, int frame, int b,
                                       float (&gp)[3], float* __restrict__ g_bt, float* __restrict__ g_T, bool uniform, int lane) {
    gp[0] += m[0] * qb[0] + m[4] * qb[1] + m[8] * qb[2];
    gp[1] += m[1] * qb[0] + m[5] * qb[1] + m[9] * qb[2];
    gp[2] += m[2] * qb[0] + m[6] * qb[1] + m[10] * qb[2];
    if (g_bt != nullptr) {
        float* gm = g_bt + ((size_t)frame * N_BONES + b) * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) pose_add(gm + 4 * r + c, qb[r] * p[c], uniform, lane);
            pose_add(gm + 4 * r + 3, qb[r], uniform, lane);
        }
    }
    if (g_T != nullptr) {
        float* gt = g_T + ((size_t)frame * N_BONES + b) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) pose_add(gt + r, -qb[r], uniform, lane);
    }
}
// MODE 0: out[n,3] = g = sum_b R_b^T grad_q F_b(G)                       (`.gradient()`)
// MODE 1: J^T G: out (+)= sum_b R_b^T grad_q F_b(G); pose gradients += spread
// MODE 2: second-order term of g . gbar with G fixed: explicit R_b^T (pose) and the Hessian-vector product along R_b gbar
template <int MODE>
__global__ void k_hand_pull(const float* __restrict__ pts, int n, int ppf, int nf, const float* __restrict__ bt_inv,
                            const float* __restrict__ T_pose, const float* __restrict__ G, int ld, const float* __restrict__ gbar,
                            float* __restrict__ out, int accumulate, float* __restrict__ g_bt, float* __restrict__ g_T) {
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i0 < n;
    const int i = valid ? i0 : n - 1;               // inactive lanes shadow the last sample and contribute zeros
    const int lane = threadIdx.x & 63;
    const float p[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
    const FrameRef fr = frame_of(i, ppf, nf, bt_inv, T_pose);
    const bool uniform = __ballot(fr.frame != __shfl(fr.frame, 0, 64)) == 0ull;
    float gp[3] = {0.f, 0.f, 0.f};
    float gb[3] = {0.f, 0.f, 0.f};
    if (MODE == 2) {
        gb[0] = gbar[3 * (size_t)i];
        gb[1] = gbar[3 * (size_t)i + 1];
        gb[2] = gbar[3 * (size_t)i + 2];
    }
    for (int b = blockIdx.y; b <= (int)blockIdx.y; ++b) {   // one bone per grid row: 21 x the parallelism of a loop
        const float* m = fr.M + 16 * b;
        const BoneQ q = bone_q(p, m, fr.T + 3 * b, b);
        const float* g = G + (size_t)i * ld + b * 66;
        float S0[4] = {0.f, 0.f, 0.f, 0.f}, S1[4] = {0.f, 0.f, 0.f, 0.f}, S2[4] = {0.f, 0.f, 0.f, 0.f};
        bone_features(q, [&](int f, int a, float phi, float phi1, float phi2) {
            const float gv = g[f];
            S0[a] += gv * phi;
            S1[a] += gv * phi1;
            S2[a] += gv * phi2;
        });
        const float tot = S0[0] + S0[1] + S0[2] + S0[3];
        const float Sv = q.h * S1[0] + q.h1 * tot;
        const float Sr[3] = {q.h * S1[1], q.h * S1[2], q.h * S1[3]};
        const float dot = Sr[0] * q.r[0] + Sr[1] * q.r[1] + Sr[2] * q.r[2];
        float dq[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) dq[c] = valid ? Sv * q.r[c] + (Sr[c] - dot * q.r[c]) / q.v : 0.f;
        if (MODE == 0) {
            gp[0] += m[0] * dq[0] + m[4] * dq[1] + m[8] * dq[2];
            gp[1] += m[1] * dq[0] + m[5] * dq[1] + m[9] * dq[2];
            gp[2] += m[2] * dq[0] + m[6] * dq[1] + m[10] * dq[2];
        } else if (MODE == 1) {
            spread(dq, p, m, fr.frame, b, gp, g_bt, g_T, uniform, lane);
        } else {
            if (g_bt != nullptr) {   // g = sum_b R_b^T dq_b depends on R_b explicitly
                float* gm = g_bt + ((size_t)fr.frame * N_BONES + b) * 16;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) pose_add(gm + 4 * r + c, dq[r] * gb[c], uniform, lane);
            }
            const float w[3] = {m[0] * gb[0] + m[1] * gb[1] + m[2] * gb[2], m[4] * gb[0] + m[5] * gb[1] + m[6] * gb[2],
                                m[8] * gb[0] + m[9] * gb[1] + m[10] * gb[2]};
            const float rw = q.r[0] * w[0] + q.r[1] * w[1] + q.r[2] * w[2];
            float wt[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) wt[c] = (w[c] - q.r[c] * rw) / q.v;
            const float B1[3] = {S1[1], S1[2], S1[3]}, B2[3] = {S2[1], S2[2], S2[3]};
            const float dSv_dv = 2.f * q.h1 * S1[0] + q.h * S2[0] + q.h2 * tot;
            const float hb_r = q.h1 * (B1[0] * q.r[0] + B1[1] * q.r[1] + B1[2] * q.r[2]);
            const float cc = q.h1 * (B1[0] * wt[0] + B1[1] * wt[1] + B1[2] * wt[2]);
            const float e[3] = {q.h * B2[0] * wt[0], q.h * B2[1] * wt[1], q.h * B2[2] * wt[2]};
            const float e_r = e[0] * q.r[0] + e[1] * q.r[1] + e[2] * q.r[2];
            const float sr_wt = Sr[0] * wt[0] + Sr[1] * wt[1] + Sr[2] * wt[2];
            float hv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float gSv = dSv_dv * q.r[c] + (q.h1 * B1[c] - hb_r * q.r[c]) / q.v;
                hv[c] = gSv * rw + Sv * wt[c] + cc * q.r[c] + (e[c] - e_r * q.r[c]) / q.v -
                        ((Sr[c] - dot * q.r[c]) / q.v * rw + dot * wt[c]) / q.v - sr_wt * q.r[c] / q.v;
                if (!valid) hv[c] = 0.f;
            }
            spread(hv, p, m, fr.frame, b, gp, g_bt, g_T, uniform, lane);
        }
    }
    if (valid) {
        if (MODE == 0 && accumulate == 2) {
            // the bone's share to its own plane of `out` [N_BONES][n][3]; k_sum_bones adds the planes in bone order.  This is the
            // TAPE's d sdf / d pts -- the normals the colour network sees in the backward pass's forward tape: summed with atomics
            // their last bits depended on the order the 21 grid rows arrived in, and one ReLU unit of one sample whose
            // pre-activation is within rounding of 0 then flipped from run to run (a whole row of a dW moving by 1e-4 of its
            // largest entry: found as an intermittent mismatch of two runs of the same training step, round 4)
            float* o = out + ((size_t)blockIdx.y * n + i) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = gp[c];
        } else {   // the 21 grid rows add their bone's share (the caller zeroes `out` unless it accumulates)
            float* o = out + 3 * (size_t)i;
#pragma unroll
            for (int c = 0; c < 3; ++c) atomicAdd(o + c, gp[c]);
        }
    }
}
// out[n,3] = sum over the 21 bone planes of part [N_BONES][n][3], in bone order
__global__ void k_sum_bones(const float* __restrict__ part, int n, float* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * 3) return;
    float acc = 0.f;
    for (int b = 0; b < N_BONES; ++b) acc += part[(size_t)b * n * 3 + t];
    out[t] = acc;
}

// ---- orchestration ---------------------------------------------------------------------------------------------
struct Arena {
    char* base;
    size_t used, cap;
    float* take(size_t floats) {
        const size_t bytes = (floats * sizeof(float) + 255) & ~size_t(255);
        float* p = base ? reinterpret_cast<float*>(base + used) : nullptr;
        used += bytes;
        return p;
    }
};
static inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

struct Ctx {
    hipStream_t s;
    int n;
    void dense(const float* A, int lda, int K, const float* W, int wsk, int wsc, int M, const float* bias, float alpha,
               float* C, int ldc, bool accumulate, int act = 0) const {
        DenseArgs a{A, lda, W, wsk, wsc, bias, C, ldc, n, K, M, alpha, accumulate ? 1 : 0, act};
        // 128 x 64 workgroup tiles measure ~5 % faster than 128 x 128 on the fitting sizes (more workgroups per CU)
        hipLaunchKernelGGL(k_dense<64>, dim3((M + 63) / 64, (n + 127) / 128), dim3(256), 0, s, a);
    }
    // C = A * W[:, c0:c0+K]^T  (W row-major [M, ldw])          "forward" use of a weight block
    void nt(const float* A, int lda, int K, const float* W, int ldw, int c0, int M, const float* bias, float alpha, float* C,
            int ldc, bool acc, int act = 0) const {
        dense(A, lda, K, W + c0, 1, ldw, M, bias, alpha, C, ldc, acc, act);
    }
    // C = A * W[:, c0:c0+M]      (A [n, rows of W])            "transposed" use of the same block
    void nn(const float* A, int lda, int K, const float* W, int ldw, int c0, int M, float alpha, float* C, int ldc, bool acc) const {
        dense(A, lda, K, W + c0, ldw, 1, M, nullptr, alpha, C, ldc, acc);
    }
    // dW[M, K] += alpha * A^T B over the samples (+ db[M] += sum A when db != NULL)
    void outer(const float* A, int lda, int M, const float* B, int ldb, int K, float alpha, float* dW, int ldw, float* db) const {
        const int KB = db != nullptr ? K + 1 : K;
        const int tiles = ((M + 127) / 128) * ((KB + 127) / 128);
        // ~512 workgroups (two per CU), a slice at least 256 samples long
        int slices = (512 + tiles - 1) / tiles;   // (1024: 5 % slower -- twice the atomics per output; 128: too few workgroups)
        const int max_slices = (n + 255) / 256;
        if (slices > max_slices) slices = max_slices;
        if (slices < 1) slices = 1;
        int chunk = (n + slices - 1) / slices;
        chunk = (chunk + 31) & ~31;
        slices = (n + chunk - 1) / chunk;
        static const int dbg = getenv("HN_DBG_OUTER") ? atoi(getenv("HN_DBG_OUTER")) : 0;
        OuterArgs a{A, lda, M, B, ldb, K, dW, ldw, db, alpha, 1.f, n, chunk, dbg};
        hipLaunchKernelGGL(k_outer, dim3((KB + 127) / 128, (M + 127) / 128, slices), dim3(256), 0, s, a);
    }
};

// The products of one backward pass, collected and launched together (k_outer_group)
struct OuterGroup {
    OuterGroupArgs g{};
    int tiles = 0;
    // dW[M, K] += alpha * A^T B over the samples (+ db[M] += alpha_b * sum A when db != NULL); K == 0: the column sums alone
    bool overflow = false;   // more products than the kernel's table holds: launch() refuses (the callers add ~20)
    OuterDesc spare{};
    OuterDesc& add(const float* A, int lda, int M, const float* B, int ldb, int K, float alpha, float* dW, int ldw, float* db, float alpha_b = 1.f) {
        if (g.n_desc >= OUTER_MAX_DESCS) {
            overflow = true;
            return spare;
        }
        OuterDesc& d = g.d[g.n_desc++];
        d = OuterDesc{A, B, nullptr, nullptr, dW, db, lda, ldb, 0, 0, M, K, ldw, K > 0 ? (K + OG_T - 1) / OG_T : 1, 0, 0, alpha, alpha_b};
        tiles += ((M + OG_T - 1) / OG_T) * d.kt;
        return d;
    }
    // ... + alpha * A2^T B2 into the same dW (same M, K)
    void add2(const float* A, int lda, const float* B, int ldb, const float* A2, int lda2, const float* B2, int ldb2, int M, int K, float alpha,
              float* dW, int ldw, float* db) {
        OuterDesc& d = add(A, lda, M, B, ldb, K, alpha, dW, ldw, db);
        d.A2 = A2;
        d.lda2 = lda2;
        d.B2 = B2;
        d.ldb2 = ldb2;
        tiles += ((M + OG_T - 1) / OG_T) * d.kt;   // twice the work per sample: counted twice, sliced twice as finely
    }
    bool full() const { return g.n_desc >= OUTER_MAX_DESCS; }
    int launch(hipStream_t s, int n) {
        HN_REQUIRE(!overflow, "more than %d outer products in one group", OUTER_MAX_DESCS);
        if (g.n_desc == 0 || n <= 0) return HN_OK;
        int cus = device_cus();
        if (cus <= 0) cus = 256;
        // two workgroups are resident per CU; items for ~two rounds of them: the object nets' ~124 work units measure the same with one
        // round (3.34 ms of a 56 448-sample pass either way), the hand nets' ~210 -- two slices per product at one round -- 7.00 -> 6.74 ms
        static const int wg_per_cu = getenv("HN_OUTER_WGS") ? atoi(getenv("HN_OUTER_WGS")) : 4;
        int slices = (wg_per_cu * cus) / tiles;
        const int max_slices = (n + 255) / 256;         // a slice at least 256 samples long
        if (slices > max_slices) slices = max_slices;
        if (slices < 1) slices = 1;
        int items = 0;
        for (int p = 0; p < g.n_desc; ++p) {
            OuterDesc& d = g.d[p];
            int sl = d.A2 != nullptr ? 2 * slices : slices;
            int chunk = (n + sl - 1) / sl;
            // (the kernel addresses a slice of an operand as one raw buffer: its rows x pitch stay below 2^31 bytes)
            const int max_ld = std::max(std::max(d.lda, d.ldb), std::max(d.lda2, d.ldb2));
            const int max_chunk = (int)(((1u << 31) - 4096u) / (4u * (unsigned)std::max(max_ld, 1))) & ~31;
            if (chunk > max_chunk) chunk = max_chunk;
            chunk = (chunk + 31) & ~31;
            sl = (n + chunk - 1) / chunk;
            d.chunk = chunk;
            d.item0 = items;
            items += ((d.M + OG_T - 1) / OG_T) * d.kt * sl;
        }
        g.n = n;
        g.items = items;
        static const int dbg = getenv("HN_DBG_OUTER") ? atoi(getenv("HN_DBG_OUTER")) : 0;
        g.dbg = dbg;
        g.per_xcd = (items + 7) / 8;
        hipLaunchKernelGGL(k_outer_group, dim3(g.per_xcd * 8), dim3(256), 0, s, g);
        HN_LAUNCH_CHECK();
        return HN_OK;
    }
};

size_t field_bwd_workspace(const hn_field* f, int n, Arena* out_layout);

// Buffers of one adjoint evaluation (all [n, width] row-major)
struct Bufs {
    float *X, *a[9], *u[8], *dz[8], *z8, *GX, *g, *din, *gin, *c[5], *xb, *cb[2], *Xb, *db, *gbin, *gb, *GXb, *dzb, *v, *sb[8], *ab,
        *zb, *gdir, *z8b;
};
static void layout(const hn_field* f, int n, Arena& ar, Bufs& b) {
    const bool obj = f->kind == HN_FIELD_OBJ;
    const size_t N = (size_t)n;
    const int Din = ((obj ? OBJ_IN : HAND_IN) + 3) & ~3;   // row pitch of the input-space arrays
    b.X = ar.take(N * Din);
    for (int l = 1; l <= 8; ++l) b.a[l] = ar.take(N * f->sdf_out[l - 1]);
    for (int l = 0; l < 8; ++l) {
        b.u[l] = ar.take(N * f->sdf_out[l]);
        b.dz[l] = ar.take(N * f->sdf_out[l]);
        b.sb[l] = ar.take(N * f->sdf_out[l]);
    }
    b.z8 = ar.take(N * 257);
    b.z8b = ar.take(N * 257);
    b.GX = ar.take(N * Din);
    b.GXb = ar.take(N * Din);
    b.Xb = ar.take(N * Din);
    b.g = ar.take(N * 3);
    b.gb = ar.take(N * 3);
    b.gdir = ar.take(N * 3);
    b.din = ar.take(N * 27);
    b.gin = ar.take(N * 27);
    b.db = ar.take(N * 27);
    b.gbin = ar.take(N * 27);
    for (int l = 1; l <= 4; ++l) b.c[l] = ar.take(N * H);
    b.xb = ar.take(N * 3);
    b.cb[0] = ar.take(N * H);
    b.cb[1] = ar.take(N * H);
    b.dzb = ar.take(N * H);
    b.v = ar.take(N * H);
    b.ab = ar.take(N * H);
    b.zb = ar.take(N * H);
}

// xb = g_rgb * rgb (1 - rgb) from the colour itself
__global__ void k_rgb_seed2(const float* __restrict__ rgb, const float* __restrict__ g_rgb, float* __restrict__ xb, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) xb[i] = g_rgb[i] * rgb[i] * (1.f - rgb[i]);
}
__global__ void k_scale1(const float* __restrict__ x, float scale, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] * scale;
}

// HN_TRAIN_FUSED=0: the parameter gradients of an f16x3 field through the generic launch sequence (A/B, cross-check)
static bool train_fused_enabled() {
    static const bool on = [] {
        const char* e = getenv("HN_TRAIN_FUSED");
        return !(e != nullptr && e[0] == '0');
    }();
    return on;
}
static bool fused_param_path(const hn_field* f);
// (hn_api.hip: whether a training render may keep its evaluation's tape for the backward pass)
bool param_path_is_fused(const hn_field* f) { return fused_param_path(f); }
static bool fused_param_path(const hn_field* f) {
    return train_fused_enabled() && f->precision == HN_PREC_F16X3 && f->v2_adjonly != nullptr &&
           f->v2_full != nullptr;   // (a field packed for training has no evaluation + adjoint program: v2_adj is not asked for)
}
// Buffers of the FUSED parameter-gradient path of an f16x3 field (round 5; SURVEY 8 f1, exp_runner.py:196-232): the taped f16x3
// evaluation (k_field2_obj<3>) and the adjoint from its tape in the form that also leaves the per-layer signals (k_field2_obj<5>:
// OSG_COUNT row-major [n, 256] arrays) replace the generic sequence's forward tape, its three sweeps and their element-wise launches
// (54 k_dense + ~40 small launches); the outer products over the samples (k_outer) and the encodings' small kernels stay.
struct FusedBufs {
    void* tape;
    size_t tape_bytes;
    float *sig, *sdf, *grad, *rgb, *feat, *gb, *X, *din, *gin, *xb, *GXb, *z8b0;
    size_t pitch;
    void* rows;          // hand: the adjoint kernel's per-tile pose-gradient rows (k_pose_part_reduce)
    size_t rows_bytes;
};
static void layout_fused(const hn_field* f, int n, Arena& ar, FusedBufs& b) {
    const size_t N = (size_t)n;
    const bool obj = f->kind == HN_FIELD_OBJ;
    const int DP = ((obj ? OBJ_IN : HAND_IN) + 3) & ~3;
    b.tape_bytes = obj ? v2::field2_obj_tape_bytes(n) : v2::field2_hand_tape_bytes(n);
    b.tape = ar.take((b.tape_bytes + 3) / 4);
    b.pitch = ((N * H + 63) / 64) * 64;
    b.sig = ar.take(b.pitch * (size_t)(obj ? v2::field2_obj_signal_arrays() : v2::field2_hand_signal_arrays()));
    b.sdf = ar.take(N);
    b.grad = ar.take(N * 3);
    b.rgb = ar.take(N * 3);
    b.feat = ar.take(N * H);
    b.gb = ar.take(N * 3);
    b.X = ar.take(N * DP);
    b.din = ar.take(N * 27);
    b.gin = ar.take(N * 27);
    b.xb = ar.take(N * 3);
    b.GXb = ar.take(N * DP);
    b.z8b0 = ar.take(N);
    b.rows_bytes = obj ? 0 : v2::field2_hand_pose_rows_bytes(n);
    b.rows = ar.take((b.rows_bytes + 3) / 4);
}

size_t field_bwd_workspace_bytes(const hn_field* f, int n) {
    Arena ar{nullptr, 0, 0};
    Bufs b;
    layout(f, n, ar, b);
    size_t need = ar.used;   // the generic sequence (also what the sdf-only adjoint uses)
    if (fused_adjoint(f, false)) {
        int cus = device_cus();
        if (cus <= 0) cus = 256;
        const size_t fused = f->kind == HN_FIELD_OBJ ? v2::field2_obj_adj_workspace_bytes(n, cus) : v2::field2_hand_adj_workspace_bytes(n, cus);
        need = fused > need ? fused : need;
    }
    if (fused_param_path(f)) {
        Arena af{nullptr, 0, 0};
        FusedBufs fb;
        layout_fused(f, n, af, fb);
        need = af.used > need ? af.used : need;
    }
    return need;
}

// mid (with g_params only): called once the forward tape stands (z8 [n,257]: column 0 = sdf * scale, then the feature
// vector; d sdf / d pts [n,3]; the colour network's pre-sigmoid output [n,3]) and before any upstream gradient is read:
// the caller derives g_sdf / g_grad / g_rgb from these values there (hn_render_single_bwd: alpha stage and compositing
// and their adjoints), so a training step's backward pass does not evaluate the field a second time.
typedef std::function<int(const float* z8, const float* grad, const float* rgb_pre)> MidHook;
// ... and for the fused form of the parameter-gradient path (below): the taped evaluation's outputs themselves (sdf [n], d sdf / d pts [n,3],
// rgb [n,3])
typedef std::function<int(const float* sdf, const float* grad, const float* rgb)> MidHook2;

// tape / grad / rgb (HN_PREC_F16X3 only, may be NULL): the tape a taped evaluation of the same points left and that
// evaluation's outputs -- the adjoint then runs alone instead of re-evaluating the field first
int field_eval_bwd(const hn_field* f, const float* pts, const float* rays_d, int n, int spr, const float* bt_inv,
                   const float* T_pose, int n_frames, int pts_per_frame, const float* g_sdf, const float* g_grad,
                   const float* g_rgb, float* g_pts, float* g_rays_d, float* g_bt_inv, float* g_T_pose, void* workspace,
                   size_t workspace_bytes, hipStream_t s, const void* tape, const float* grad, const float* rgb, float* g_params,
                   const MidHook* mid, const MidHook2* mid2, const float* sdf_t, const float* feat_t) {
    HN_REQUIRE(f != nullptr && f->raw != nullptr, "field has no folded weights");
    const bool obj = f->kind == HN_FIELD_OBJ;
    HN_REQUIRE(obj || (bt_inv != nullptr && T_pose != nullptr && n_frames >= 1 && pts_per_frame >= 1),
               "hand field needs bt_inv / T_pose and frame sizes");
    // g_grad == g_rgb == NULL: adjoint of `.sdf()` alone (get_stable_loss_cross, utils/renderer_batch.py:318-371):
    // tape + reverse sweep, then the input map with the rows of d sdf / d X scaled by g_sdf
    const bool sdf_only = g_grad == nullptr && g_rgb == nullptr;
    HN_REQUIRE(pts && g_sdf && g_pts && (sdf_only || (g_grad && g_rgb)) && spr >= 1 && n % spr == 0, "bad arguments");
    if (n == 0) return HN_OK;
    HN_REQUIRE(g_params == nullptr || !sdf_only, "parameter gradients need g_grad and g_rgb");
    HN_REQUIRE(mid == nullptr || g_params != nullptr, "the mid hook belongs to the parameter-gradient path");
    if (g_params != nullptr && fused_param_path(f) && (mid == nullptr || mid2 != nullptr)) {
        // ---- FUSED parameter-gradient path: taped evaluation -> [hook] -> adjoint from the tape that leaves the per-layer
        //      signals -> the outer products over the samples.  Same gradient slots, same pairs as the generic sequence below.
        Arena ar{reinterpret_cast<char*>(workspace), 0, workspace_bytes};
        FusedBufs b;
        layout_fused(f, n, ar, b);
        if (workspace == nullptr || ar.used > workspace_bytes) {
            set_error("adjoint workspace too small: %zu < %zu", workspace_bytes, ar.used);
            return HN_ENOMEM;
        }
        const Ctx cx{s, n};
        const size_t N = (size_t)n;
        const int Din = obj ? OBJ_IN : HAND_IN, DP = (Din + 3) & ~3;
        const int* LW = f->sdf_ld;
        const int LC0 = f->col_ld[0];
        const float rs2 = 0.70710678118654752f;
        const float inv_scale = 1.f / f->scale;
        const float* const* W = f->raw_sdf_w;
        const float* const* Bv = f->raw_sdf_b;
        const float* const* C = f->raw_col_w;
        const float* const* Cb = f->raw_col_b;
        const int H4 = f->sdf_in[4] - Din;
        auto width = [&](int l) { return f->sdf_out[l]; };
        float* const gp = g_params;
        auto G = [&](const float* w) { return gp + (w - reinterpret_cast<const float*>(f->raw)); };
        auto S = [&](int k) { return b.sig + (size_t)k * b.pitch; };
        enum { CB = 0, CC = 4, AA = 8, DZ = 16, VV = 24, ZB = 32, FB = 40 };   // OSG_* / HSG_* of hn_field2_obj.hip / hn_field2_hand.hip
        HN_REQUIRE(v2::field2_obj_signal_arrays() == 41 && v2::field2_hand_signal_arrays() == 41, "signal array layout changed");
        // 1. the taped evaluation (sdf, d sdf / d pts, rgb, the feature vector) -- or the one the caller kept from the forward pass
        //    (hn_render_single_taped: tape + its four outputs); 2. the caller's stages between the outputs and their adjoints
        if (tape != nullptr) {
            HN_REQUIRE(grad != nullptr && rgb != nullptr && sdf_t != nullptr && feat_t != nullptr, "a kept tape comes with the evaluation's sdf / grad / rgb / feature vector");
            b.tape = const_cast<void*>(tape);
            b.sdf = const_cast<float*>(sdf_t);
            b.grad = const_cast<float*>(grad);
            b.rgb = const_cast<float*>(rgb);
            b.feat = const_cast<float*>(feat_t);
        } else if (obj) {
            HN_TRY_RC(v2::launch_field2_obj(f, pts, rays_d, n, spr, b.sdf, b.grad, b.rgb, b.feat, nullptr, 0, true, s, b.tape, b.tape_bytes));
        } else {
            HN_TRY_RC(v2::launch_field2_hand(f, pts, n, bt_inv, T_pose, n_frames, pts_per_frame, b.sdf, b.grad, b.rgb, b.feat, nullptr, 0, true, s, b.tape,
                                             b.tape_bytes));
        }
        if (mid2 != nullptr) {
            HN_LAUNCH_CHECK();
            const int rc = (*mid2)(b.sdf, b.grad, b.rgb);
            if (rc != HN_OK) return rc;
        }
        // 3. the adjoint from the tape, leaving the signals (its sig arrays are written for every valid sample; columns 193 .. 255 of the
        //    object field's 193-wide layer-3 arrays are never read)
        if (obj) {
            HN_TRY_RC(v2::launch_field2_obj_adj(f, pts, rays_d, n, spr, g_sdf, g_grad, g_rgb, g_pts, g_rays_d, nullptr, 0, s, b.tape, b.grad, b.rgb, b.sig,
                                                b.pitch, b.gb));
        } else {
            // (the hand's colour network ignores the view direction, utils/fields.py:222-240: its gradient is exactly 0; the pose gradients
            //  accumulate into the caller's zeroed g_bt_inv / g_T_pose)
            if (g_rays_d != nullptr) HN_CHECK_HIP(hipMemsetAsync(g_rays_d, 0, (size_t)(n / spr) * 3 * sizeof(float), s));
            HN_TRY_RC(v2::launch_field2_hand_adj(f, pts, n, bt_inv, T_pose, n_frames, pts_per_frame, g_sdf, g_grad, g_rgb, g_pts, g_bt_inv, g_T_pose, b.rows,
                                                 b.rows_bytes, s, b.tape, b.grad, b.rgb, b.sig, b.pitch, b.gb));
        }
        // 4. what the outer products pair the signals with: the input features, the colour seed, J gb, g_sdf / scale
        if (obj) {
            hipLaunchKernelGGL(k_enc3<PTS_FREQS>, g1(n), dim3(256), 0, s, pts, n, 1, b.X, DP);
            hipLaunchKernelGGL(k_enc3<OBJ_DIR_FREQS>, g1(n), dim3(256), 0, s, rays_d, n, spr, b.din, 27);
            hipLaunchKernelGGL(k_enc3_push<PTS_FREQS>, g1(n), dim3(256), 0, s, pts, n, b.gb, b.GXb, DP);
        } else {
            hipLaunchKernelGGL(k_hand_feat, dim3((n + 63) / 64, N_BONES), dim3(64), 0, s, pts, n, pts_per_frame, n_frames, bt_inv, T_pose, b.X, DP,
                               (float*)nullptr, (float*)nullptr);
            hipLaunchKernelGGL(k_hand_push, dim3((n + 63) / 64, N_BONES), dim3(64), 0, s, pts, n, pts_per_frame, n_frames, bt_inv, T_pose, b.gb, b.GXb, DP);
        }
        hipLaunchKernelGGL(k_enc3<4>, g1(n), dim3(256), 0, s, b.grad, n, 1, b.gin, 27);
        hipLaunchKernelGGL(k_rgb_seed2, g1(N * 3), dim3(256), 0, s, b.rgb, g_rgb, b.xb, N * 3);
        hipLaunchKernelGGL(k_scale1, g1(n), dim3(256), 0, s, g_sdf, inv_scale, b.z8b0, n);
        const int o_d = Din, o_f = obj ? Din + 27 : Din, o_g = o_f + H;
        // 5.-7. the products, in one grouped launch (k_outer_group); HN_OUTER_GROUP=0: one k_outer launch each (A/B, cross-check)
        static const bool grouped = [] {
            const char* e = getenv("HN_OUTER_GROUP");
            return !(e != nullptr && e[0] == '0');
        }();
        OuterGroup og;
        auto prod = [&](const float* A, int lda, int M, const float* B, int ldb, int K, float alpha, float* dW, int ldw, float* db) {
            if (grouped)
                og.add(A, lda, M, B, ldb, K, alpha, dW, ldw, db);
            else
                cx.outer(A, lda, M, B, ldb, K, alpha, dW, ldw, db);
        };
        // dW += alpha (A^T B + A2^T B2): the first-order path's product (with the bias) and the reverse sweep's own use of the same matrix
        auto prod2 = [&](const float* A, int lda, const float* B, int ldb, const float* A2, int lda2, const float* B2, int ldb2, int M, int K,
                         float alpha, float* dW, int ldw, float* db) {
            if (grouped) {
                og.add2(A, lda, B, ldb, A2, lda2, B2, ldb2, M, K, alpha, dW, ldw, db);
            } else {
                cx.outer(A, lda, M, B, ldb, K, alpha, dW, ldw, db);
                cx.outer(A2, lda2, M, B2, ldb2, K, alpha, dW, ldw, nullptr);
            }
        };
        // colour network
        prod(b.xb, 3, 3, S(CC + 3), H, H, 1.f, G(C[4]), f->col_ld[4], G(Cb[4]));
        for (int l = 3; l >= 1; --l) prod(S(CB + (3 - l)), H, H, S(CC + l - 1), H, H, 1.f, G(C[l]), f->col_ld[l], G(Cb[l]));
        {
            const float* cb1 = S(CB + 3);
            prod(cb1, H, H, b.X, DP, Din, 1.f, G(C[0]), LC0, G(Cb[0]));
            if (obj) prod(cb1, H, H, b.din, 27, 27, 1.f, G(C[0]) + o_d, LC0, nullptr);
            prod(cb1, H, H, b.feat, H, H, 1.f, G(C[0]) + o_f, LC0, nullptr);
            prod(cb1, H, H, b.gin, 27, 27, 1.f, G(C[0]) + o_g, LC0, nullptr);
        }
        // W_8: row 0 <- g_sdf / scale (first-order) and the column sums of v_7 (the row seeds the reverse sweep); rows 1.. <- fb
        prod(b.z8b0, 1, 1, S(AA + 7), H, H, 1.f, G(W[8]), LW[8], G(Bv[8]));
        prod(S(FB), H, H, S(AA + 7), H, H, 1.f, G(W[8]) + LW[8], LW[8], G(Bv[8]) + 1);
        if (grouped)
            og.add(S(VV + 7), H, H, nullptr, 0, 0, 0.f, nullptr, 0, G(W[8]), inv_scale);
        else
            hipLaunchKernelGGL(k_colsum, dim3((n + 255) / 256), dim3(256), 0, s, S(VV + 7), n, H, H, inv_scale, G(W[8]));
        // W_l, l = 7 .. 0: zb_l^T [its input] (first-order path with the second-order sources; bias) + dz_l^T [the forward-direction sweep's
        // input of the layer] (the path through `.gradient()`)
        for (int l = 7; l >= 0; --l) {
            const float *zb = S(ZB + l), *dz = S(DZ + l);
            if (l == 4) {
                prod2(zb, H, S(AA + 3), H, dz, H, S(VV + 3), H, H, H4, rs2, G(W[4]), LW[4], G(Bv[4]));
                prod2(zb, H, b.X, DP, dz, H, b.GXb, DP, H, Din, rs2, G(W[4]) + H4, LW[4], nullptr);
            } else if (l == 0) {
                prod2(zb, H, b.X, DP, dz, H, b.GXb, DP, H, Din, 1.f, G(W[0]), LW[0], G(Bv[0]));
            } else {
                prod2(zb, H, S(AA + l - 1), H, dz, H, S(VV + l - 1), H, width(l), f->sdf_in[l], 1.f, G(W[l]), LW[l], G(Bv[l]));
            }
        }
        if (grouped) HN_TRY_RC(og.launch(s, n));
        HN_LAUNCH_CHECK();
        return HN_OK;
    }
    // g_params: gradients w.r.t. the folded weights / biases in the layout of f->raw (hn_field_param_offset), accumulated.
    // They are formed by the launch sequence below for either precision (the fused kernels keep no per-layer arrays).
    if (g_params == nullptr && fused_adjoint(f, sdf_only)) {
        if (obj)
            return v2::launch_field2_obj_adj(f, pts, rays_d, n, spr, g_sdf, g_grad, g_rgb, g_pts, g_rays_d, workspace, workspace_bytes, s,
                                             tape, grad, rgb);
        // the hand's colour network ignores the view direction (utils/fields.py:222-240): its gradient is exactly 0
        if (g_rays_d != nullptr) HN_CHECK_HIP(hipMemsetAsync(g_rays_d, 0, (size_t)(n / spr) * 3 * sizeof(float), s));
        return v2::launch_field2_hand_adj(f, pts, n, bt_inv, T_pose, n_frames, pts_per_frame, g_sdf, g_grad, g_rgb, g_pts, g_bt_inv,
                                          g_T_pose, workspace, workspace_bytes, s, tape, grad, rgb);
    }
    Arena ar{reinterpret_cast<char*>(workspace), 0, workspace_bytes};
    Bufs b;
    layout(f, n, ar, b);
    if (workspace == nullptr || ar.used > workspace_bytes) {
        set_error("adjoint workspace too small: %zu < %zu", workspace_bytes, ar.used);
        return HN_ENOMEM;
    }
    const Ctx cx{s, n};
    const size_t N = (size_t)n;
    const int Din = obj ? OBJ_IN : HAND_IN;
    const int DP = (Din + 3) & ~3;               // row pitch of the input-space arrays (X, GX, GXb, Xb)
    const int* LW = f->sdf_ld;                   // row pitch of the retained matrices
    const int LC0 = f->col_ld[0];
    const float rs2 = 0.70710678118654752f;
    const float inv_scale = 1.f / f->scale;
    const float* const* W = f->raw_sdf_w;
    const float* const* Bv = f->raw_sdf_b;
    const int H4 = f->sdf_in[4] - Din;   // hidden columns of lin4's input (193)
    auto width = [&](int l) { return f->sdf_out[l]; };
    // the gradient slot of a retained matrix / bias: same offset in g_params as in f->raw
    float* const gp = g_params;
    auto G = [&](const float* w) { return gp + (w - reinterpret_cast<const float*>(f->raw)); };

    // 1. forward tape -------------------------------------------------------------------------------------------
    if (obj)
        hipLaunchKernelGGL(k_enc3<PTS_FREQS>, g1(n), dim3(256), 0, s, pts, n, 1, b.X, DP);
    else
        hipLaunchKernelGGL(k_hand_feat, dim3((n + 63) / 64, N_BONES), dim3(64), 0, s, pts, n, pts_per_frame, n_frames, bt_inv, T_pose, b.X, DP,
                           (float*)nullptr, (float*)nullptr);
    b.a[0] = b.X;
    for (int l = 0; l < 8; ++l) {   // (the softplus is the last product's epilogue: no separate pass over the activations)
        if (l == 4) {
            cx.nt(b.a[4], H4, H4, W[4], LW[4], 0, width(4), Bv[4], rs2, b.a[5], width(4), false);
            cx.nt(b.X, DP, Din, W[4], LW[4], H4, width(4), nullptr, rs2, b.a[5], width(4), true, 1);
        } else {
            const int K = l == 0 ? Din : f->sdf_in[l];
            cx.nt(b.a[l], l == 0 ? DP : K, K, W[l], LW[l], 0, width(l), Bv[l], 1.f, b.a[l + 1], width(l), false, 1);
        }
    }
    cx.nt(b.a[8], H, H, W[8], LW[8], 0, 257, Bv[8], 1.f, b.z8, 257, false);
    // 2. reverse sweep ------------------------------------------------------------------------------------------
    hipLaunchKernelGGL(k_bcast_row, g1(N * H), dim3(256), 0, s, W[8], H, inv_scale, b.u[7], N * H);
    for (int l = 7; l >= 0; --l) {
        hipLaunchKernelGGL(k_dz, g1((N * width(l) + 3) / 4), dim3(256), 0, s, b.a[l + 1], b.u[l], b.dz[l], N * width(l));
        if (l > 0) {   // u_{l-1} = dz_l * Wh_l
            if (l == 4)
                cx.nn(b.dz[4], width(4), width(4), W[4], LW[4], 0, H4, rs2, b.u[3], H4, false);
            else
                cx.nn(b.dz[l], width(l), width(l), W[l], LW[l], 0, f->sdf_in[l], 1.f, b.u[l - 1], f->sdf_in[l], false);
        }
    }
    cx.nn(b.dz[0], H, H, W[0], LW[0], 0, Din, 1.f, b.GX, DP, false);
    cx.nn(b.dz[4], H, H, W[4], LW[4], H4, Din, rs2, b.GX, DP, true);
    if (sdf_only) {
        hipLaunchKernelGGL(k_scale_rows, g1(N * DP), dim3(256), 0, s, b.GX, g_sdf, DP, N * DP);
        if (obj) {
            hipLaunchKernelGGL(k_enc3_pull<PTS_FREQS>, g1(n), dim3(256), 0, s, pts, n, 1, b.GX, DP, nullptr, 0, nullptr, g_pts, 0);
        } else {
            HN_CHECK_HIP(hipMemsetAsync(g_pts, 0, N * 3 * sizeof(float), s));
            hipLaunchKernelGGL(k_hand_pull<1>, dim3((n + 63) / 64, N_BONES), dim3(64), 0, s, pts, n, pts_per_frame, n_frames, bt_inv, T_pose, b.GX,
                               DP, (const float*)nullptr, g_pts, 0, g_bt_inv, g_T_pose);
        }
        if (g_rays_d != nullptr) HN_CHECK_HIP(hipMemsetAsync(g_rays_d, 0, (size_t)(n / spr) * 3 * sizeof(float), s));
        HN_LAUNCH_CHECK();
        return HN_OK;
    }
    if (obj) {
        hipLaunchKernelGGL(k_enc3_pull<PTS_FREQS>, g1(n), dim3(256), 0, s, pts, n, 1, b.GX, DP, nullptr, 0, nullptr, b.g, 0);
    } else {
        // (per-bone planes in b.zb -- [n, 256] floats, free until step 5 -- then summed in bone order: a reproducible tape)
        static_assert(3 * N_BONES <= H, "the bone planes fit the [n, H] scratch array");
        hipLaunchKernelGGL(k_hand_pull<0>, dim3((n + 63) / 64, N_BONES), dim3(64), 0, s, pts, n, pts_per_frame, n_frames, bt_inv, T_pose, b.GX, DP,
                           (const float*)nullptr, b.zb, 2, (float*)nullptr, (float*)nullptr);
        hipLaunchKernelGGL(k_sum_bones, g1(N * 3), dim3(256), 0, s, b.zb, n, b.g);
    }
    // 3. colour network forward + backward ------------------------------------------------------------------------
    const float* const* C = f->raw_col_w;
    const float* const* Cb = f->raw_col_b;
    [[maybe_unused]] const int cin = f->col_in[0];                                  // obj 373 = 63 | 27 | 256 | 27; hand 1669 = 1386 | 256 | 27
    const int o_d = Din, o_f = obj ? Din + 27 : Din, o_g = o_f + H;
    hipLaunchKernelGGL(k_enc3<4>, g1(n), dim3(256), 0, s, b.g, n, 1, b.gin, 27);
    cx.nt(b.X, DP, Din, C[0], LC0, 0, H, Cb[0], 1.f, b.c[1], H, false);
    if (obj) {   // the hand's colour net ignores the view direction (utils/fields.py:222-240)
        hipLaunchKernelGGL(k_enc3<OBJ_DIR_FREQS>, g1(n), dim3(256), 0, s, rays_d, n, spr, b.din, 27);
        cx.nt(b.din, 27, 27, C[0], LC0, o_d, H, nullptr, 1.f, b.c[1], H, true);
    }
    cx.nt(b.z8 + 1, 257, H, C[0], LC0, o_f, H, nullptr, 1.f, b.c[1], H, true);
    cx.nt(b.gin, 27, 27, C[0], LC0, o_g, H, nullptr, 1.f, b.c[1], H, true, 2);   // (+ ReLU)
    for (int l = 1; l <= 3; ++l) cx.nt(b.c[l], H, H, C[l], H, 0, H, Cb[l], 1.f, b.c[l + 1], H, false, 2);
    cx.nt(b.c[4], H, H, C[4], H, 0, 3, Cb[4], 1.f, b.xb, 3, false);
    if (mid != nullptr) {
        HN_LAUNCH_CHECK();
        const int rc = (*mid)(b.z8, b.g, b.xb);
        if (rc != HN_OK) return rc;
    }
    hipLaunchKernelGGL(k_rgb_seed, g1(N * 3), dim3(256), 0, s, b.xb, g_rgb, b.xb, N * 3);
    if (gp) cx.outer(b.xb, 3, 3, b.c[4], H, H, 1.f, G(C[4]), f->col_ld[4], G(Cb[4]));
    cx.nn(b.xb, 3, 3, C[4], H, 0, H, 1.f, b.cb[0], H, false);
    hipLaunchKernelGGL(k_relu_mask, g1((N * H + 3) / 4), dim3(256), 0, s, b.c[4], b.cb[0], N * H);
    int cur = 0;
    for (int l = 3; l >= 1; --l) {   // cb[cur]: adjoint of layer l's pre-activation
        if (gp) cx.outer(b.cb[cur], H, H, b.c[l], H, H, 1.f, G(C[l]), f->col_ld[l], G(Cb[l]));
        cx.nn(b.cb[cur], H, H, C[l], H, 0, H, 1.f, b.cb[cur ^ 1], H, false);
        cur ^= 1;
        hipLaunchKernelGGL(k_relu_mask, g1((N * H + 3) / 4), dim3(256), 0, s, b.c[l], b.cb[cur], N * H);
    }
    const float* cb1 = b.cb[cur];
    if (gp) {   // lin0 by input block: [X | enc(dir) (obj) | feature vector | enc(gradient)]
        cx.outer(cb1, H, H, b.X, DP, Din, 1.f, G(C[0]), LC0, G(Cb[0]));
        if (obj) cx.outer(cb1, H, H, b.din, 27, 27, 1.f, G(C[0]) + o_d, LC0, nullptr);
        cx.outer(cb1, H, H, b.z8 + 1, 257, H, 1.f, G(C[0]) + o_f, LC0, nullptr);
        cx.outer(cb1, H, H, b.gin, 27, 27, 1.f, G(C[0]) + o_g, LC0, nullptr);
    }
    cx.nn(cb1, H, H, C[0], LC0, 0, Din, 1.f, b.Xb, DP, false);                    // Xb starts as the colour net's share
    hipLaunchKernelGGL(k_z8_bar, g1(n), dim3(256), 0, s, g_sdf, inv_scale, b.z8b, n);
    cx.nn(cb1, H, H, C[0], LC0, o_f, H, 1.f, b.z8b + 1, 257, false);               // fb
    cx.nn(cb1, H, H, C[0], LC0, o_g, 27, 1.f, b.gbin, 27, false);
    if (obj) {
        cx.nn(cb1, H, H, C[0], LC0, o_d, 27, 1.f, b.db, 27, false);
        hipLaunchKernelGGL(k_enc3_pull<OBJ_DIR_FREQS>, g1(n), dim3(256), 0, s, rays_d, n, spr, b.db, 27, nullptr, 0, nullptr, b.gdir, 0);
        if (g_rays_d != nullptr)
            hipLaunchKernelGGL(k_sum_rays, g1((size_t)(n / spr) * 3), dim3(256), 0, s, b.gdir, n / spr, spr, g_rays_d);
    } else if (g_rays_d != nullptr) {
        HN_CHECK_HIP(hipMemsetAsync(g_rays_d, 0, (size_t)(n / spr) * 3 * sizeof(float), s));
    }
    hipLaunchKernelGGL(k_enc3_pull<4>, g1(n), dim3(256), 0, s, b.g, n, 1, b.gbin, 27, nullptr, 0, nullptr, b.gb, 0);
    hipLaunchKernelGGL(k_add3, g1(N * 3), dim3(256), 0, s, g_grad, b.gb, N * 3);
    // 4. adjoint of the reverse sweep ---------------------------------------------------------------------------
    if (obj)
        hipLaunchKernelGGL(k_enc3_push<PTS_FREQS>, g1(n), dim3(256), 0, s, pts, n, b.gb, b.GXb, DP);
    else
        hipLaunchKernelGGL(k_hand_push, dim3((n + 63) / 64, N_BONES), dim3(64), 0, s, pts, n, pts_per_frame, n_frames, bt_inv, T_pose, b.gb, b.GXb, DP);
    cx.nt(b.GXb, DP, Din, W[0], LW[0], 0, H, nullptr, 1.f, b.dzb, H, false);
    if (gp) {   // the reverse sweep's own use of the matrices: GX = dz_0 W_0 + dz_4 W_4[:, skip] / sqrt2, u_{l-1} = dz_l W_l
        cx.outer(b.dz[0], H, H, b.GXb, DP, Din, 1.f, G(W[0]), LW[0], nullptr);
        cx.outer(b.dz[4], H, H, b.GXb, DP, Din, rs2, G(W[4]) + H4, LW[4], nullptr);
    }
    for (int l = 1; l <= 7; ++l) {
        const int wprev = width(l - 1);
        hipLaunchKernelGGL(k_fwd_dir, g1((N * wprev + 3) / 4), dim3(256), 0, s, b.a[l], b.u[l - 1], b.dzb, b.sb[l - 1], b.v, N * wprev);
        if (gp) cx.outer(b.dz[l], width(l), width(l), b.v, wprev, wprev, l == 4 ? rs2 : 1.f, G(W[l]), LW[l], nullptr);
        if (l == 4) {
            cx.nt(b.v, H4, H4, W[4], LW[4], 0, H, nullptr, rs2, b.dzb, H, false);
            cx.nt(b.GXb, DP, Din, W[4], LW[4], H4, H, nullptr, rs2, b.dzb, H, true);
        } else {
            cx.nt(b.v, wprev, wprev, W[l], LW[l], 0, width(l), nullptr, 1.f, b.dzb, width(l), false);
        }
    }
    hipLaunchKernelGGL(k_fwd_dir, g1((N * H + 3) / 4), dim3(256), 0, s, b.a[8], b.u[7], b.dzb, b.sb[7], gp ? b.v : (float*)nullptr, N * H);
    if (gp)     // u_7 = W_8[0, :] / scale
        hipLaunchKernelGGL(k_colsum, dim3((n + 255) / 256), dim3(256), 0, s, b.v, n, H, H, inv_scale, G(W[8]));
    // 5. first-order reverse sweep with the second-order sources ---------------------------------------------------
    cx.nn(b.z8b, 257, 257, W[8], LW[8], 0, H, 1.f, b.ab, H, false);
    if (gp) cx.outer(b.z8b, 257, 257, b.a[8], H, H, 1.f, G(W[8]), LW[8], G(Bv[8]));
    for (int l = 7; l >= 0; --l) {
        hipLaunchKernelGGL(k_zb, g1((N * width(l) + 3) / 4), dim3(256), 0, s, b.a[l + 1], b.ab, b.sb[l], b.zb, N * width(l));
        if (gp) {   // z_l = W_l [a_l (, X)] + b_l
            if (l == 4) {
                cx.outer(b.zb, H, H, b.a[4], H4, H4, rs2, G(W[4]), LW[4], G(Bv[4]));
                cx.outer(b.zb, H, H, b.X, DP, Din, rs2, G(W[4]) + H4, LW[4], nullptr);
            } else if (l == 0) {
                cx.outer(b.zb, H, H, b.X, DP, Din, 1.f, G(W[0]), LW[0], G(Bv[0]));
            } else {
                cx.outer(b.zb, width(l), width(l), b.a[l], f->sdf_in[l], f->sdf_in[l], 1.f, G(W[l]), LW[l], G(Bv[l]));
            }
        }
        if (l == 4) {
            cx.nn(b.zb, H, H, W[4], LW[4], H4, Din, rs2, b.Xb, DP, true);
            cx.nn(b.zb, H, H, W[4], LW[4], 0, H4, rs2, b.ab, H4, false);
        } else if (l == 0) {
            cx.nn(b.zb, H, H, W[0], LW[0], 0, Din, 1.f, b.Xb, DP, true);
        } else {
            cx.nn(b.zb, width(l), width(l), W[l], LW[l], 0, f->sdf_in[l], 1.f, b.ab, f->sdf_in[l], false);
        }
    }
    // 6. input map: g_pts = J^T Xb + second-order term ------------------------------------------------------------
    if (obj) {
        hipLaunchKernelGGL(k_enc3_pull<PTS_FREQS>, g1(n), dim3(256), 0, s, pts, n, 1, b.Xb, DP, b.GX, DP, b.gb, g_pts, 0);
    } else {   // the pose gradients accumulate into the caller's (zeroed) g_bt_inv / g_T_pose
        HN_CHECK_HIP(hipMemsetAsync(g_pts, 0, N * 3 * sizeof(float), s));
        hipLaunchKernelGGL(k_hand_pull<1>, dim3((n + 63) / 64, N_BONES), dim3(64), 0, s, pts, n, pts_per_frame, n_frames, bt_inv, T_pose, b.Xb, DP,
                           (const float*)nullptr, g_pts, 0, g_bt_inv, g_T_pose);
        hipLaunchKernelGGL(k_hand_pull<2>, dim3((n + 63) / 64, N_BONES), dim3(64), 0, s, pts, n, pts_per_frame, n_frames, bt_inv, T_pose, b.GX, DP,
                           b.gb, g_pts, 1, g_bt_inv, g_T_pose);
    }
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// ---- stand-alone module calls of the L1 surface (utils/fields.py) ------------------------------------------------------
// The fused field kernels evaluate sdf net + gradient + colour net together; the reference's classes can also be
// called one at a time (RenderingNetwork*.forward with caller-supplied features / normals; SDFNetwork.forward returns
// xyz_feature, r, h).  These entry points serve those calls with the generic kernels above.
int hand_features(const float* pts, int n, const float* bt_inv, const float* T_pose, int n_frames, int pts_per_frame,
                  float* xyz_feature, float* r, float* h, hipStream_t s) {
    HN_REQUIRE(pts && bt_inv && T_pose && xyz_feature && n_frames >= 1 && pts_per_frame >= 1, "bad arguments");
    if (n <= 0) return HN_OK;
    hipLaunchKernelGGL(k_hand_feat, dim3((n + 63) / 64, N_BONES), dim3(64), 0, s, pts, n, pts_per_frame, n_frames, bt_inv, T_pose,
                       xyz_feature, HAND_IN, r, h);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

size_t color_forward_workspace_bytes(const hn_field* f, int n) {
    Arena ar{nullptr, 0, 0};
    const size_t N = (size_t)n;
    ar.take(N * 64);
    ar.take(N * 27);
    ar.take(N * 27);
    ar.take(N * H);
    ar.take(N * H);
    (void)f;
    return ar.used;
}
// obj: x = points [n,3]; hand: x = xyz_feature [n,1386] (view_dirs unused, utils/fields.py:222-240)
int color_forward(const hn_field* f, const float* x, const float* view_dirs, const float* feature_vectors, const float* normals,
                  int n, float* rgb, void* workspace, size_t workspace_bytes, hipStream_t s) {
    HN_REQUIRE(f != nullptr && f->raw != nullptr, "field has no folded weights");
    const bool obj = f->kind == HN_FIELD_OBJ;
    HN_REQUIRE(x && feature_vectors && normals && rgb && (!obj || view_dirs), "bad arguments");
    if (n <= 0) return HN_OK;
    Arena ar{reinterpret_cast<char*>(workspace), 0, workspace_bytes};
    const size_t N = (size_t)n;
    float* X = ar.take(N * 64);
    float* din = ar.take(N * 27);
    float* gin = ar.take(N * 27);
    float* c0 = ar.take(N * H);
    float* c1 = ar.take(N * H);
    if (workspace == nullptr || ar.used > workspace_bytes) {
        set_error("colour workspace too small: %zu < %zu", workspace_bytes, ar.used);
        return HN_ENOMEM;
    }
    const Ctx cx{s, n};
    const float* const* C = f->raw_col_w;
    const float* const* Cb = f->raw_col_b;
    const int LC0 = f->col_ld[0];
    const int Din = obj ? OBJ_IN : HAND_IN;
    const int o_d = Din, o_f = obj ? Din + 27 : Din, o_g = o_f + H;
    hipLaunchKernelGGL(k_enc3<4>, g1(n), dim3(256), 0, s, normals, n, 1, gin, 27);
    if (obj) {
        hipLaunchKernelGGL(k_enc3<PTS_FREQS>, g1(n), dim3(256), 0, s, x, n, 1, X, 64);
        cx.nt(X, 64, Din, C[0], LC0, 0, H, Cb[0], 1.f, c0, H, false);
        hipLaunchKernelGGL(k_enc3<OBJ_DIR_FREQS>, g1(n), dim3(256), 0, s, view_dirs, n, 1, din, 27);
        cx.nt(din, 27, 27, C[0], LC0, o_d, H, nullptr, 1.f, c0, H, true);
    } else {
        cx.nt(x, HAND_IN, Din, C[0], LC0, 0, H, Cb[0], 1.f, c0, H, false);
    }
    cx.nt(feature_vectors, H, H, C[0], LC0, o_f, H, nullptr, 1.f, c0, H, true);
    cx.nt(gin, 27, 27, C[0], LC0, o_g, H, nullptr, 1.f, c0, H, true);
    hipLaunchKernelGGL(k_relu, g1((N * H + 3) / 4), dim3(256), 0, s, c0, N * H);
    float *cur = c0, *nxt = c1;
    for (int l = 1; l <= 3; ++l) {
        cx.nt(cur, H, H, C[l], H, 0, H, Cb[l], 1.f, nxt, H, false);
        hipLaunchKernelGGL(k_relu, g1((N * H + 3) / 4), dim3(256), 0, s, nxt, N * H);
        float* t = cur;
        cur = nxt;
        nxt = t;
    }
    cx.nt(cur, H, H, C[4], H, 0, 3, Cb[4], 1.f, rgb, 3, false);
    hipLaunchKernelGGL(k_sigmoid, g1(N * 3), dim3(256), 0, s, rgb, N * 3);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// ---- weight-norm chain rule (the last step of a training iteration's backward pass; utils/fields.py:113-121) -----------
// W[r, :] = g[r] v[r, :] / |v[r, :]|: from dW (row pitch ld, the layout of g_params) one workgroup per row forms
//   dg[r] = dW[r, :] . vhat,   dv[r, :] = (g[r] / |v|) (dW[r, :] - vhat (dW[r, :] . vhat)),   db[r] = dB[r]
// (what torch's `_weight_norm` backward computes).  g == NULL: a plain nn.Linear, dv = dW.
// all 14 layers in ONE launch (blockIdx.y = layer, blockIdx.x = matrix row: a launch of a dependent chain costs ~5 us whatever it does)
struct WnbLayer {
    const float *g, *v, *dW, *dB;
    float *dg, *dv, *db;
    int rows, cols, ld;
};
struct WnbTable {
    WnbLayer l[14];
};
__global__ __launch_bounds__(256) void k_weight_norm_bwd(const WnbTable tab) {
    __shared__ float red[2][4];
    const WnbLayer& L_ = tab.l[blockIdx.y];
    const int r = blockIdx.x, t = threadIdx.x;
    if (r >= L_.rows) return;
    const float *g = L_.g, *v = L_.v, *dW = L_.dW, *dB = L_.dB;
    float *dg = L_.dg, *dv = L_.dv, *db = L_.db;
    const int cols = L_.cols, ld = L_.ld;
    const float* vr = v + (size_t)r * cols;
    const float* wr = dW + (size_t)r * ld;
    if (t == 0 && db != nullptr) db[r] = dB[r];
    if (g == nullptr) {
        for (int c = t; c < cols; c += 256) dv[(size_t)r * cols + c] = wr[c];
        return;
    }
    float n2 = 0.f, dot = 0.f;
    for (int c = t; c < cols; c += 256) {
        const float x = vr[c];
        n2 = fmaf(x, x, n2);
        dot = fmaf(wr[c], x, dot);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        n2 += __shfl_xor(n2, o);
        dot += __shfl_xor(dot, o);
    }
    if ((t & 63) == 0) {
        red[0][t >> 6] = n2;
        red[1][t >> 6] = dot;
    }
    __syncthreads();
    n2 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    dot = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float nrm = sqrtf(n2), proj = dot / nrm, k = g[r] / nrm;
    if (t == 0) dg[r] = proj;
    for (int c = t; c < cols; c += 256) dv[(size_t)r * cols + c] = k * (wr[c] - (vr[c] / nrm) * proj);
}

int weight_norm_bwd(const hn_field* f, const hn_mlp_desc* sdf, const hn_mlp_desc* col, const float* g_params, const hn_mlp_desc* g_sdf,
                    const hn_mlp_desc* g_col, hipStream_t s) {
    HN_REQUIRE(f != nullptr && f->raw != nullptr && sdf && col && g_params && g_sdf && g_col, "null argument");
    const float* base = reinterpret_cast<const float*>(f->raw);
    WnbTable tab;
    int max_rows = 0, li = 0;
    for (int net = 0; net < 2; ++net) {
        const hn_mlp_desc* d = net == 0 ? sdf : col;
        const hn_mlp_desc* o = net == 0 ? g_sdf : g_col;
        const int nl = net == 0 ? 9 : 5;
        HN_REQUIRE(d->n_layers == nl && o->n_layers == nl, "layer count");
        for (int l = 0; l < nl; ++l) {
            const int rows = net == 0 ? f->sdf_out[l] : f->col_out[l], cols = net == 0 ? f->sdf_in[l] : f->col_in[l];
            const int ld = net == 0 ? f->sdf_ld[l] : f->col_ld[l];
            HN_REQUIRE(d->out_dim[l] == rows && d->in_dim[l] == cols, "shape of layer %d differs from the packed field", l);
            const float* dW = g_params + ((net == 0 ? f->raw_sdf_w[l] : f->raw_col_w[l]) - base);
            const float* dB = g_params + ((net == 0 ? f->raw_sdf_b[l] : f->raw_col_b[l]) - base);
            HN_REQUIRE(o->weight_v[l] != nullptr && (d->weight_g[l] == nullptr || o->weight_g[l] != nullptr), "missing output of layer %d", l);
            tab.l[li++] = WnbLayer{d->weight_g[l], d->weight_v[l], dW, dB, const_cast<float*>(o->weight_g[l]), const_cast<float*>(o->weight_v[l]),
                                   const_cast<float*>(o->bias[l]), rows, cols, ld};
            max_rows = rows > max_rows ? rows : max_rows;
        }
    }
    hipLaunchKernelGGL(k_weight_norm_bwd, dim3(max_rows, 14), dim3(256), 0, s, tab);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// ---- nearest candidate vertex (get_stable_loss_cross: scipy cKDTree.query(k=1), utils/renderer_batch.py:355-358) ------
// One wave per (set t, query vertex i) with query_mask[t,i] != 0: argmin over the vertices j with cand_mask[t,j] != 0
// of |p_i - p_j|^2 (ties: lowest j); selected[t, argmin] = 1 (the np.unique of the reference = a set).  A few thousand
// vertices: brute force, the vertex array stays in L2.
__global__ void k_nearest_masked(const float* __restrict__ pts, int n_verts, int n_sets, const unsigned char* __restrict__ query_mask,
                                 const unsigned char* __restrict__ cand_mask, unsigned char* __restrict__ selected,
                                 int* __restrict__ nearest) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= n_sets * n_verts) return;
    const int t = q / n_verts, i = q % n_verts;
    if (!query_mask[q]) {
        if (nearest != nullptr && lane == 0) nearest[q] = -1;
        return;
    }
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    float best = INFINITY;
    int arg = -1;
    const unsigned char* cm = cand_mask + (size_t)t * n_verts;
    for (int j = lane; j < n_verts; j += 64) {
        if (!cm[j]) continue;
        const float dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
        const float d = dx * dx + dy * dy + dz * dz;
        if (d < best) {
            best = d;
            arg = j;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oa = __shfl_xor(arg, off, 64);
        if (oa >= 0 && (arg < 0 || ob < best || (ob == best && oa < arg))) {
            best = ob;
            arg = oa;
        }
    }
    if (lane == 0) {
        if (arg >= 0) selected[(size_t)t * n_verts + arg] = 1;
        if (nearest != nullptr) nearest[q] = arg;
    }
}
int nearest_masked(const float* pts, int n_verts, int n_sets, const unsigned char* query_mask, const unsigned char* cand_mask,
                   unsigned char* selected, int* nearest, hipStream_t s) {
    HN_REQUIRE(pts && query_mask && cand_mask && selected && n_verts >= 0 && n_sets >= 0, "bad arguments");
    const size_t total = (size_t)n_verts * n_sets;
    if (total == 0) return HN_OK;
    HN_CHECK_HIP(hipMemsetAsync(selected, 0, total, s));
    hipLaunchKernelGGL(k_nearest_masked, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, pts, n_verts, n_sets, query_mask, cand_mask,
                       selected, nearest);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace bwd
}  // namespace hn

// Not part of the C ABI (no declaration in include/honerf.h; tests/test_training.py binds it by name): ONE outer product
// dW[M, K] += alpha A^T B, db[M] += sum A through k_outer_group, for checking the kernel against numpy on its own.
extern "C" int hn_debug_outer_product(const float* A, int lda, int M, const float* B, int ldb, int K, int n, float alpha, float* dW, int ldw, float* db,
                                      void* stream) {
    hn::bwd::OuterGroup og;
    og.add(A, lda, M, B, ldb, K, alpha, dW, ldw, db);
    return og.launch(reinterpret_cast<hipStream_t>(stream), n);
}
