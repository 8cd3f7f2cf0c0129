// v2 weight streams (HN_PREC_F16X3): every matrix of a field re-laid as the sequence of LDS
// chunks the v2 kernels consume, fp16 hi / scaled-lo MFMA A fragments (hn_mlp2.h).  The host
// lays a program out (which matrix element goes where: descriptors and index maps, plus the few
// hundred bias / row values of the chunk tails); the fragments themselves -- the fp16 hi / lo
// split of every matrix element, 5 - 27 MB per program -- are written on the device by
// k_fill_fragments from the folded matrices that are already there.  (Done on the host, as in
// round 1, a hand field took 230 ms to pack: fine for frozen networks, not for a training step
// that re-packs after every optimiser update; profiles/r02/README.md, "training step".)
//
// Reads the reference's state-dict layout through hn_mlp_desc (utils/fields.py:120-121,
// 216-217, 307-308, 382-383).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <map>
#include <mutex>
#include <vector>

#include "hn_mlp2.h"

namespace hn {
hipError_t pool_alloc_on(void** p, size_t bytes, hipStream_t s);   // hn_pack.hip: size-keyed cache in front of hipMalloc / hipFree
void pool_free(void* p);
namespace v2 {

struct HostMat {
    const float* w = nullptr;   // row-major [rows][cols], weight-norm folded (a slice of the pinned fetch buffer)
    const float* b = nullptr;   // [rows]
    const float* dev = nullptr;   // the same matrix on the device (what k_fill_fragments reads)
    int rows = 0, cols = 0;
    float at(int r, int c) const { return w[(size_t)r * cols + c]; }
};

// The stream being built is for the 16x16x32 MFMA shape (hn_mlp2.h, "MFMA shape").  k-slots are numbered
// kappa = 16 s + 8 h + j in both shapes (new: k-step pair s >> 1, lane group g = 2 (s & 1) + h); what differs is which
// tile row a slot's neuron is (the rule "a finished tile's registers are the next layer's fragments"), the lane order
// inside an A-fragment block, and the tail layout.
static thread_local bool g_s16 = false;
// neuron (row of a [256 x samples] activation) held in k-step s, lane half h, fragment element j
static inline int hid_k(int s, int h, int j) {
    if (g_s16) return 32 * (s >> 1) + 16 * (j >> 2) + 4 * (2 * (s & 1) + h) + (j & 3);
    return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
}
// inverse for rows of an accumulator whose rows ARE k-slots: natural row n -> (s, h, j)
// (the same on both shapes: such tiles are handed to the per-sample code in the old layout -- tile_out -- where
// register 8 s + j of lane half h is k-slot (s, h, j))
static inline void k_of_row(int n, int& s, int& h, int& j) {
    s = n >> 4;
    const int rem = n & 15;
    h = (rem >> 2) & 1;
    j = 4 * (rem >> 3) + (rem & 3);
}

// One 2 KiB A-fragment block (tile ti, k-step s of a chunk) for k_fill_fragments
struct FragBlock {
    const float* mat;          // folded matrix on the device, row-major [rows][cols]
    unsigned long long dst;    // byte offset of the block in the program
    int cols, transposed;
    float scale;
    int rowmap, colslot;       // offsets into the program's index-map array: 32 tile rows; the chunk's k-slot columns
    int s, s16;
};

struct Builder {
    std::vector<char> blob;    // the program with its fragment blocks left zero (tails and side data in place)
    std::vector<FragBlock> blocks;
    std::vector<int> maps;

    // One chunk: tiles x ks k-step blocks (+ tail).  rowmap[ti*32 + r] = matrix row of tile ti's row r
    // (-1 pad); colslot[s*16 + 8h + j] = matrix column of k-slot (s,h,j) (-1 pad).  transposed:
    // element (row, col) = M[col][row].
    void chunk(const HostMat& M, bool transposed, float scale, int tiles, int ks, const int* rowmap, const int* colslot,
               const float* tail /* 256 floats or nullptr */) {
        const size_t base = blob.size();
        if (g_s16 && (ks & 1)) abort();   // k-steps come in pairs on this shape: the caller pads
        blob.resize(base + (size_t)tiles * ks * KS_BYTES + (tail ? TAIL_BYTES : 0));
        const int rm = (int)maps.size();
        maps.insert(maps.end(), rowmap, rowmap + tiles * 32);
        const int cs = (int)maps.size();
        maps.insert(maps.end(), colslot, colslot + ks * 16);
        for (int ti = 0; ti < tiles; ++ti)
            for (int s = 0; s < ks; ++s)
                blocks.push_back(FragBlock{M.dev, (unsigned long long)(base + (size_t)(ti * ks + s) * KS_BYTES), M.cols, transposed ? 1 : 0,
                                           scale, rm + ti * 32, cs, s, g_s16 ? 1 : 0});
        if (tail) memcpy(blob.data() + base + (size_t)tiles * ks * KS_BYTES, tail, TAIL_BYTES);
    }
    // side data only (a multiple of 1 KiB)
    void raw(const void* data, size_t bytes) {
        const size_t base = blob.size();
        blob.resize(base + bytes);
        memcpy(blob.data() + base, data, bytes);
    }
};

// tail slot k <- 32 values given in natural tile-row order, stored [half][16]
static void tail_put(float* tail, int k, const float* v32) {
    if (g_s16) {   // [g][8]: rows 16 rb + 4 g + ii (hn_mlp2.h tail_tile)
        for (int g = 0; g < 4; ++g)
            for (int m = 0; m < 8; ++m) tail[k * 32 + g * 8 + m] = v32[16 * (m >> 2) + 4 * g + (m & 3)];
        return;
    }
    for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 16; ++i) tail[k * 32 + h * 16 + i] = v32[tile_row(i, h)];
}

// ---- k-slot descriptions of the encoded inputs -----------------------------------------------------
// [x(3), enc_L(x)] column of (channel c, frequency k, sin|cos) in the reference's layout
// (utils/fields.py:13-20): 3 + 2 L c + k (+ L for cos)
static inline int enc_col(int L, int c, int k, int is_cos) { return 3 + 2 * L * c + k + (is_cos ? L : 0); }

// obj sdf input (63 columns) over 4 k-steps: s = channel for frequencies 0..7; s = 3: frequencies 8, 9 of
// the three channels, then (p0 | p2), (p1 | pad).  Half 0 holds sines / first members, half 1 cosines.
static std::vector<int> obj_x_slots() {
    std::vector<int> c(64, -1);
    for (int s = 0; s < 3; ++s)
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 8; ++j) c[s * 16 + 8 * h + j] = enc_col(10, s, j, h);
    for (int h = 0; h < 2; ++h) {
        for (int j = 0; j < 6; ++j) c[48 + 8 * h + j] = enc_col(10, j >> 1, 8 + (j & 1), h);
        c[48 + 8 * h + 6] = h ? 2 : 0;
        c[48 + 8 * h + 7] = h ? -1 : 1;
    }
    return c;
}
// [v(3), enc_4(v)] (27 columns) over 2 k-steps
static std::vector<int> vec4_slots() {
    std::vector<int> c(32, -1);
    for (int h = 0; h < 2; ++h) {
        for (int j = 0; j < 8; ++j) c[8 * h + j] = enc_col(4, j >> 2, j & 3, h);
        for (int j = 0; j < 4; ++j) c[16 + 8 * h + j] = enc_col(4, 2, j, h);
        c[16 + 8 * h + 4] = h ? 2 : 0;
        c[16 + 8 * h + 5] = h ? -1 : 1;
    }
    return c;
}
static std::vector<int> hid_slots(int n_real = 256, int ks = 16) {
    std::vector<int> c(ks * 16, -1);
    for (int s = 0; s < ks; ++s)
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 8; ++j) {
                const int n = hid_k(s, h, j);
                c[s * 16 + 8 * h + j] = n < n_real ? n : -1;
            }
    return c;
}
static std::vector<int> offset(std::vector<int> v, int off) {
    for (int& x : v)
        if (x >= 0) x += off;
    return v;
}
static std::vector<int> rows_of_tile(int t, int n_real, int off = 0) {
    std::vector<int> r(32, -1);
    for (int i = 0; i < 32; ++i)
        if (32 * t + i < n_real) r[i] = off + 32 * t + i;
    return r;
}
static std::vector<int> cat(std::vector<int> a, const std::vector<int>& b) {
    a.insert(a.end(), b.begin(), b.end());
    return a;
}

// hidden layer, forward: out tile t, 16 k-steps, tail = bias (+ extra tail slots)
static void fwd_tiles(Builder& B, const HostMat& M, float scale, int out_tiles, int n_out, int row_off,
                      const std::vector<int>& slots, int ks, const float* const* extra_rows /* up to 3 */, int n_extra) {
    for (int t = 0; t < out_tiles; ++t) {
        const std::vector<int> rows = rows_of_tile(t, n_out, row_off);
        float tail[256] = {0.f};
        float v[32];
        for (int i = 0; i < 32; ++i) v[i] = rows[i] >= 0 ? M.b[rows[i]] : 0.f;
        tail_put(tail, 0, v);
        for (int e = 0; e < n_extra; ++e) {
            for (int i = 0; i < 32; ++i) v[i] = 32 * t + i < n_out ? extra_rows[e][32 * t + i] : 0.f;
            tail_put(tail, 1 + e, v);
        }
        B.chunk(M, false, scale, 1, ks, rows.data(), slots.data(), tail);
    }
}
// transposed hidden layer (reverse sweep): rows = the layer's inputs, K = its outputs
static void bwd_tiles(Builder& B, const HostMat& M, float scale, int in_tiles, int n_in, int col_off, int n_out, int ks) {
    const std::vector<int> slots = hid_slots(n_out, ks);
    for (int t = 0; t < in_tiles; ++t) {
        const std::vector<int> rows = rows_of_tile(t, n_in, col_off);
        B.chunk(M, true, scale, 1, ks, rows.data(), slots.data(), nullptr);
    }
}
// rows = k-slots of an encoded input (d sdf / d encoded input): tile u row r <-> slot of natural row 32u + r
static void slot_rows_T(Builder& B, const HostMat& M, float scale, const std::vector<int>& in_slots, int col_off) {
    const int tiles = (int)in_slots.size() / 32;
    const std::vector<int> slots = hid_slots();
    for (int u = 0; u < tiles; ++u) {
        std::vector<int> rows(32, -1);
        for (int r = 0; r < 32; ++r) {
            int s, h, j;
            k_of_row(32 * u + r, s, h, j);
            const int c = in_slots[s * 16 + 8 * h + j];
            rows[r] = c >= 0 ? col_off + c : -1;
        }
        B.chunk(M, true, scale, 1, 16, rows.data(), slots.data(), nullptr);
    }
}

// transposed rows of W8[1:, :] (the feature-vector rows) for the adjoint: out tile t = a8 neurons, K = feature index i
// (matrix row 1 + i); tail slot 0 = W8[0, tile rows] (the sdf row, multiplied by g_sdf in the kernel)
static void w8_T_tiles(Builder& B, const HostMat& M) {
    const std::vector<int> slots = offset(hid_slots(), 1);
    for (int t = 0; t < 8; ++t) {
        const std::vector<int> rows = rows_of_tile(t, 256);
        float tail[256] = {0.f};
        float v[32];
        for (int i = 0; i < 32; ++i) v[i] = M.at(0, rows[i]);
        tail_put(tail, 0, v);
        B.chunk(M, true, 1.f, 1, 16, rows.data(), slots.data(), tail);
    }
}

// obj sdf network, forward-oriented chunks lin0..lin7 (the forward pass and the adjoint's forward-direction sweep)
static void obj_sdf_forward_chunks(Builder& B, const HostMat* S) {
    const float rs2 = (float)(1.0 / sqrt(2.0));
    const std::vector<int> xs = obj_x_slots();
    const std::vector<int> hs = hid_slots();
    // lin0: 2 chunks of 4 tiles x 4 k-steps, tail = the 4 biases
    for (int c = 0; c < 2; ++c) {
        std::vector<int> rows;
        float tail[256] = {0.f};
        for (int ti = 0; ti < 4; ++ti) {
            const std::vector<int> r = rows_of_tile(4 * c + ti, 256);
            rows = cat(rows, r);
            float v[32];
            for (int i = 0; i < 32; ++i) v[i] = S[0].b[r[i]];
            tail_put(tail, ti, v);
        }
        B.chunk(S[0], false, 1.f, 4, 4, rows.data(), xs.data(), tail);
    }
    fwd_tiles(B, S[1], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    fwd_tiles(B, S[2], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    fwd_tiles(B, S[3], 1.f, 7, L3_OUT_OBJ, 0, hs, 16, nullptr, 0);
    {   // lin4: k-steps 0..11 = a4[0..191]; 12..15 = X slots (columns 193 + x) with a4[192] in the pad slot
        std::vector<int> slots = hid_slots(192, 12);
        std::vector<int> x = offset(xs, L3_OUT_OBJ);
        x[48 + 8 + 7] = 192;
        slots = cat(slots, x);
        fwd_tiles(B, S[4], rs2, 8, 256, 0, slots, 16, nullptr, 0);
    }
    fwd_tiles(B, S[5], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    fwd_tiles(B, S[6], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    {   // lin7, tail slot 1 = W8[0, :]
        std::vector<float> w8(256);
        for (int i = 0; i < 256; ++i) w8[i] = S[8].at(0, i);
        const float* extra[1] = {w8.data()};
        fwd_tiles(B, S[7], 1.f, 8, 256, 0, hs, 16, extra, 1);
    }
}
// obj sdf network, transposed chunks of the reverse sweep: W7^T .. W1^T, then W0^T and W4[:, 193:]^T over the X slots
static void obj_sdf_reverse_chunks(Builder& B, const HostMat* S) {
    const float rs2 = (float)(1.0 / sqrt(2.0));
    const std::vector<int> xs = obj_x_slots();
    bwd_tiles(B, S[7], 1.f, 8, 256, 0, 256, 16);
    bwd_tiles(B, S[6], 1.f, 8, 256, 0, 256, 16);
    bwd_tiles(B, S[5], 1.f, 8, 256, 0, 256, 16);
    bwd_tiles(B, S[4], rs2, 7, L3_OUT_OBJ, 0, 256, 16);      // columns 0..192 of W4
    bwd_tiles(B, S[3], 1.f, 8, 256, 0, L3_OUT_OBJ, 13);
    bwd_tiles(B, S[2], 1.f, 8, 256, 0, 256, 16);
    bwd_tiles(B, S[1], 1.f, 8, 256, 0, 256, 16);
    slot_rows_T(B, S[0], 1.f, xs, 0);                 // W0^T over the X slots (2 tiles)
    slot_rows_T(B, S[4], rs2, xs, L3_OUT_OBJ);       // W4[:, 193:]^T over the X slots
}

static void obj_adjoint_chunks(Builder& B, const HostMat* S, const HostMat* C);

// The obj program (contract with k_field2_obj): sdf forward [+ feature rows + reverse sweep + colour [+ adjoint]]
// mode 0: sdf only; 1: full evaluation; 2: full evaluation followed by its adjoint (hn_field_eval_bwd); 3: the adjoint
// alone (from a kept tape)
static void build_obj_stream(Builder& B, const HostMat* S, const HostMat* C, int mode) {
    if (mode == 3) {
        obj_adjoint_chunks(B, S, C);
        return;
    }
    const std::vector<int> xs = obj_x_slots();
    const std::vector<int> hs = hid_slots();
    obj_sdf_forward_chunks(B, S);
    if (mode == 0) return;
    fwd_tiles(B, S[8], 1.f, 8, 256, 1, hs, 16, nullptr, 0);   // feature rows 1..256
    obj_sdf_reverse_chunks(B, S);
    // colour lin0: [enc(p) 63 | enc(d) 27 | feature 256 | enc(g) 27] (utils/fields.py:389-396)
    const std::vector<int> v4 = vec4_slots();
    const std::vector<int> misc = cat(cat(xs, offset(v4, OBJ_IN)), offset(v4, OBJ_IN + 27 + H));
    {
        const std::vector<int> fv = offset(hs, OBJ_IN + 27);
        for (int t = 0; t < 8; ++t) {
            const std::vector<int> rows = rows_of_tile(t, 256);
            B.chunk(C[0], false, 1.f, 1, 16, rows.data(), fv.data(), nullptr);
            float tail[256] = {0.f};
            float v[32];
            for (int i = 0; i < 32; ++i) v[i] = C[0].b[rows[i]];
            tail_put(tail, 0, v);
            B.chunk(C[0], false, 1.f, 1, 8, rows.data(), misc.data(), tail);
        }
    }
    fwd_tiles(B, C[1], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    fwd_tiles(B, C[2], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    {   // lin3, tail slots 1..3 = the three rows of lin4
        std::vector<float> w0(256), w1(256), w2(256);
        for (int i = 0; i < 256; ++i) {
            w0[i] = C[4].at(0, i);
            w1[i] = C[4].at(1, i);
            w2[i] = C[4].at(2, i);
        }
        const float* extra[3] = {w0.data(), w1.data(), w2.data()};
        fwd_tiles(B, C[3], 1.f, 8, 256, 0, hs, 16, extra, 3);
    }
    if (mode < 2) return;
    obj_adjoint_chunks(B, S, C);
}

// ---- adjoint (oracle/field_bwd.py steps 3b-6; contract with obj_adjoint in hn_field2_obj.hip) -------------------
static void obj_adjoint_chunks(Builder& B, const HostMat* S, const HostMat* C) {
    const std::vector<int> xs = obj_x_slots();
    const std::vector<int> v4 = vec4_slots();
    const std::vector<int> misc = cat(cat(xs, offset(v4, OBJ_IN)), offset(v4, OBJ_IN + 27 + H));
    // colour network backward: the three rows of lin4 (one tail-format KiB each: slot t = the row's 32 values of
    // tile t), C3^T, C2^T, C1^T, then C0^T over the feature-vector rows and over the [enc(p) | enc(d) | enc(g)]
    // slots (4 tiles: X, X, d, g)
    for (int c = 0; c < 3; ++c) {
        float tail[256];
        for (int t = 0; t < 8; ++t) {
            float v[32];
            for (int i = 0; i < 32; ++i) v[i] = C[4].at(c, 32 * t + i);
            tail_put(tail, t, v);
        }
        B.raw(tail, TAIL_BYTES);
    }
    bwd_tiles(B, C[3], 1.f, 8, 256, 0, 256, 16);
    bwd_tiles(B, C[2], 1.f, 8, 256, 0, 256, 16);
    bwd_tiles(B, C[1], 1.f, 8, 256, 0, 256, 16);
    bwd_tiles(B, C[0], 1.f, 8, 256, OBJ_IN + 27, 256, 16);
    slot_rows_T(B, C[0], 1.f, misc, 0);
    obj_sdf_forward_chunks(B, S);         // forward-direction sweep dzb_l = W_l (sigma'_{l-1} dzb_{l-1}) (biases unused)
    w8_T_tiles(B, S[8]);                  // adjoint of a8 = W8[1:, :]^T fb (+ g_sdf W8[0, :] from the tail)
    obj_sdf_reverse_chunks(B, S);         // second reverse sweep
}

// One wave per block: lane l writes the 8 hi and 8 lo halves of its fragment (16 B each).
// 32x32x16: block s = k-step s, lane = (row l & 31, half l >> 5).  16x16x32: block s = row block s & 1 of k-step pair
// s >> 1, lane = (row l & 15 of the block, k-group g = l >> 4).  hi = fp16(x) (IEEE, subnormals kept: the MFMA reads
// them as such), lo = fp16((x - hi) * LO_SCALE).
__global__ __launch_bounds__(256) void k_fill_fragments(const FragBlock* __restrict__ blocks, int n_blocks, const int* __restrict__ maps,
                                                        char* __restrict__ prog) {
    const int bi = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (bi >= n_blocks) return;
    const FragBlock d = blocks[bi];
    const int r = d.s16 ? 16 * (d.s & 1) + (l & 15) : (l & 31);
    const int kbase = d.s16 ? 32 * (d.s >> 1) + 8 * (l >> 4) : d.s * 16 + 8 * (l >> 5);
    const int row = maps[d.rowmap + r];
    typedef _Float16 h8v __attribute__((ext_vector_type(8)));
    h8v hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = maps[d.colslot + kbase + j];
        float x = 0.f;
        if (row >= 0 && col >= 0) x = (d.transposed ? d.mat[(size_t)col * d.cols + row] : d.mat[(size_t)row * d.cols + col]) * d.scale;
        const _Float16 xh = (_Float16)x;
        hi[j] = xh;
        lo[j] = (_Float16)((x - (float)xh) * LO_SCALE);
    }
    _Float16* out = reinterpret_cast<_Float16*>(prog + d.dst);
    *reinterpret_cast<h8v*>(out + l * 8) = hi;
    *reinterpret_cast<h8v*>(out + 512 + l * 8) = lo;
}

// ---- pack plans: a re-pack without the host in the loop ------------------------------------------------------------------
// A program's LAYOUT -- fragment descriptors, index maps, and which bias / matrix element every float of the chunk tails is
// -- depends on the field kind and the mode only.  The first pack of a (kind, mode) lays the program out on the host as
// above and, beside it, once more over SENTINEL matrices (element i of tensor t = the float 2^23 + its global index,
// `dev` = 1 + the tensor's number): the descriptors of that second layout carry tensor numbers instead of device pointers
// and its tails spell out their sources.  The plan is checked against the real layout (every tail float reproduced bit
// for bit from the fetched weights) and kept on the device; later packs of the same kind and mode are one memset and two
// launches -- no weights fetched to the host, no layout pass, no upload, no wait (a training step re-packs after every
// optimiser update: hand 2.0 -> 0.1 ms).  A layout that computes a tail value instead of copying it fails the check and is
// never cached.
constexpr int N_SRC = 28;   // w_sdf[0..8], w_col[0..4], bias_sdf[0..8], bias_col[0..4]
struct SrcTable {
    const float* p[N_SRC];
};
struct TailEntry {
    unsigned dst;     // float index in the program
    int src;          // tensor number
    unsigned idx;     // element of that tensor
};
struct PackPlan {
    size_t bytes = 0;
    FragBlock* blocks = nullptr;   // device; `mat` = (const float*)(1 + tensor number)
    int n_blocks = 0;
    int* maps = nullptr;           // device
    TailEntry* tails = nullptr;    // device
    int n_tails = 0;
    bool usable = false, tried = false;
};
static std::map<long long, PackPlan> g_plans;   // key: device, kind, mode, MFMA shape
__global__ __launch_bounds__(256) void k_fill_fragments_plan(const FragBlock* __restrict__ blocks, int n_blocks, const int* __restrict__ maps,
                                                             const SrcTable tab, char* __restrict__ prog) {
    const int bi = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (bi >= n_blocks) return;
    const FragBlock d = blocks[bi];
    const float* mat = tab.p[(int)(reinterpret_cast<uintptr_t>(d.mat) - 1)];
    const int r = d.s16 ? 16 * (d.s & 1) + (l & 15) : (l & 31);
    const int kbase = d.s16 ? 32 * (d.s >> 1) + 8 * (l >> 4) : d.s * 16 + 8 * (l >> 5);
    const int row = maps[d.rowmap + r];
    typedef _Float16 h8v __attribute__((ext_vector_type(8)));
    h8v hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = maps[d.colslot + kbase + j];
        float x = 0.f;
        if (row >= 0 && col >= 0) x = (d.transposed ? mat[(size_t)col * d.cols + row] : mat[(size_t)row * d.cols + col]) * d.scale;
        const _Float16 xh = (_Float16)x;
        hi[j] = xh;
        lo[j] = (_Float16)((x - (float)xh) * LO_SCALE);
    }
    _Float16* out = reinterpret_cast<_Float16*>(prog + d.dst);
    *reinterpret_cast<h8v*>(out + l * 8) = hi;
    *reinterpret_cast<h8v*>(out + 512 + l * 8) = lo;
}
__global__ void k_fill_tails(const TailEntry* __restrict__ tails, int n, const SrcTable tab, float* __restrict__ prog) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TailEntry e = tails[i];
    prog[e.dst] = tab.p[e.src][e.idx];
}

// Host staging of a pack: two pinned buffers kept for the life of the process (grow-only).  Everything that crosses the
// bus here goes through them.  A copy to or from PAGEABLE memory makes the driver register those pages for DMA, and when
// the pages are released afterwards (a std::vector of a few MB is an mmap that free() unmaps) the MMU notifier evicts
// and restores the process's GPU queues: a 25 - 35 ms stall of the next launch, seen as a hand field's re-pack "taking"
// 3, 9 or 35 ms from one process to the next (profiles/r02/README.md, "training step").
struct Pinned {
    void* p = nullptr;
    size_t cap = 0;
    void* get(size_t n) {
        if (n <= cap) return p;
        if (p != nullptr) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = (n + n / 4 + 4095) & ~size_t(4095);
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
        cap = want;
        return p;
    }
};
static std::mutex g_pack_mu;          // packs of one process take turns (they share the staging buffers)
static Pinned g_pin_fetch, g_pin_up;

static int upload(const Builder& B, void** dev, size_t* bytes, hipStream_t stream) {
    const std::vector<char>& blob = B.blob;
    const size_t nbk = B.blocks.size() * sizeof(FragBlock), nm = B.maps.size() * sizeof(int);
    const size_t blob_pad = (blob.size() + 255) & ~size_t(255), nb_pad = (nbk + 255) & ~size_t(255);
    char* stage = reinterpret_cast<char*>(g_pin_up.get(blob_pad + nb_pad + nm + 256));
    if (stage == nullptr) {
        set_error("hipHostMalloc of %zu bytes for the pack staging buffer failed", blob_pad + nb_pad + nm);
        return HN_ENOMEM;
    }
    memcpy(stage, blob.data(), blob.size());
    memcpy(stage + blob_pad, B.blocks.data(), nbk);
    memcpy(stage + blob_pad + nb_pad, B.maps.data(), nm);
    HN_CHECK_HIP(pool_alloc_on(dev, blob.size(), stream));
    HN_CHECK_HIP(hipMemcpyAsync(*dev, stage, blob.size(), hipMemcpyHostToDevice, stream));
    void* tmp = nullptr;
    if (!B.blocks.empty()) {
        HN_CHECK_HIP(pool_alloc_on(&tmp, nb_pad + nm, stream));
        hipError_t e = hipMemcpyAsync(tmp, stage + blob_pad, nb_pad + nm, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) {
            const int n_blocks = (int)B.blocks.size();
            hipLaunchKernelGGL(k_fill_fragments, dim3((n_blocks + 3) / 4), dim3(256), 0, stream, reinterpret_cast<const FragBlock*>(tmp),
                               n_blocks, reinterpret_cast<const int*>(reinterpret_cast<char*>(tmp) + nb_pad), reinterpret_cast<char*>(*dev));
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            pool_free(tmp);
            set_error("packing a weight stream on the device failed: %s", hipGetErrorString(e));
            return HN_EHIP;
        }
    }
    const hipError_t e = hipStreamSynchronize(stream);     // the staging buffer is free again after this
    if (tmp != nullptr) pool_free(tmp);
    HN_CHECK_HIP(e);
    *bytes = blob.size();
    return HN_OK;
}

void build_hand_stream(Builder& B, const HostMat* S, const HostMat* C, int mode);

// w_sdf / w_col: device pointers to the folded matrices (row-major [out][in])
int build_v2_streams(hn_field* f, const hn_mlp_desc* sdf, const hn_mlp_desc* col, float* const* w_sdf,
                     float* const* w_col, hipStream_t stream, bool eval_only) {
    HostMat S[9], C[5];
    std::lock_guard<std::mutex> pack_lock(g_pack_mu);
    const auto t_begin = std::chrono::steady_clock::now();
    const bool eval16_ = f->kind == HN_FIELD_OBJ ? (HN_OBJ_EVAL_MFMA16 != 0) : (HN_HAND_EVAL_MFMA16 != 0);
    const int n_modes = eval16_ ? 5 : 4;
    // HN_PACK_EVAL_ONLY (a training step's per-iteration re-pack): the sdf-only and evaluation programs and the adjoint-from-a-tape program
    // (+ the taped evaluation's copy where it is a separate one) -- the parameter gradients come from the taped evaluation and that
    // adjoint (hn_field_bwd.hip, fused parameter-gradient path).  Never the evaluation + adjoint program (mode 2).
    auto wanted = [&](int mode) { return !eval_only || mode != 2; };
    const int plan_dev = current_device();   // (a plan's arrays live on the device it was derived on)
    auto plan_key = [&](int mode) { return (long long)(plan_dev + 1) * 100000 + (long long)f->kind * 100 + mode * 10 + (eval16_ ? 1 : 0); };
    auto slot_of = [&](int mode, void*** dst, size_t** nb) {
        *dst = mode == 0 ? &f->v2_sdf : (mode == 1 ? &f->v2_full : (mode == 2 ? &f->v2_adj : (mode == 3 ? &f->v2_adjonly : &f->v2_tape)));
        *nb = mode == 0 ? &f->v2_sdf_bytes
                        : (mode == 1 ? &f->v2_full_bytes : (mode == 2 ? &f->v2_adj_bytes : (mode == 3 ? &f->v2_adjonly_bytes : &f->v2_tape_bytes)));
    };
    {   // every program of this pack has a checked plan: the device does it all, nothing waits
        bool all = getenv("HN_PACK_NO_PLAN") == nullptr;
        for (int mode = 0; mode < n_modes && all; ++mode) {
            if (!wanted(mode)) continue;
            auto it = g_plans.find(plan_key(mode));
            all = it != g_plans.end() && it->second.usable;
        }
        if (all) {
            SrcTable tab;
            for (int l = 0; l < 9; ++l) {
                tab.p[l] = w_sdf[l];
                tab.p[14 + l] = reinterpret_cast<const float*>(sdf->bias[l]);
            }
            for (int l = 0; l < 5; ++l) {
                tab.p[9 + l] = w_col[l];
                tab.p[23 + l] = reinterpret_cast<const float*>(col->bias[l]);
            }
            int n_built = 0;
            for (int mode = 0; mode < n_modes; ++mode) {
                if (!wanted(mode)) continue;
                ++n_built;
                const PackPlan& P = g_plans[plan_key(mode)];
                void** dst;
                size_t* nb;
                slot_of(mode, &dst, &nb);
                HN_CHECK_HIP(pool_alloc_on(dst, P.bytes, stream));
                HN_CHECK_HIP(hipMemsetAsync(*dst, 0, P.bytes, stream));
                hipLaunchKernelGGL(k_fill_fragments_plan, dim3((P.n_blocks + 3) / 4), dim3(256), 0, stream, P.blocks, P.n_blocks, P.maps, tab,
                                   reinterpret_cast<char*>(*dst));
                if (P.n_tails > 0)
                    hipLaunchKernelGGL(k_fill_tails, dim3((P.n_tails + 255) / 256), dim3(256), 0, stream, P.tails, P.n_tails, tab,
                                       reinterpret_cast<float*>(*dst));
                HN_LAUNCH_CHECK();
                *nb = P.bytes;
            }
            if (getenv("HN_PACK_TIMING") != nullptr)
                fprintf(stderr, "[hn pack] %d programs from their plans %.2f ms\n", n_built,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
            return HN_OK;
        }
    }
    size_t total = 0;
    for (int l = 0; l < 9; ++l) total += (size_t)sdf->out_dim[l] * sdf->in_dim[l] + sdf->out_dim[l];
    for (int l = 0; l < 5; ++l) total += (size_t)col->out_dim[l] * col->in_dim[l] + col->out_dim[l];
    float* host = reinterpret_cast<float*>(g_pin_fetch.get(total * sizeof(float)));
    if (host == nullptr) {
        set_error("hipHostMalloc of %zu bytes for the weight fetch buffer failed", total * sizeof(float));
        return HN_ENOMEM;
    }
    auto fetch = [&](const hn_mlp_desc* d, int l, float* dev_w, HostMat& M) -> int {
        M.rows = d->out_dim[l];
        M.cols = d->in_dim[l];
        M.dev = dev_w;
        const size_t nw = (size_t)M.rows * M.cols;
        M.w = host;
        M.b = host + nw;
        HN_CHECK_HIP(hipMemcpyAsync(host, dev_w, nw * sizeof(float), hipMemcpyDeviceToHost, stream));
        HN_CHECK_HIP(hipMemcpyAsync(host + nw, d->bias[l], (size_t)M.rows * sizeof(float), hipMemcpyDeviceToHost, stream));
        host += nw + M.rows;
        return HN_OK;
    };
    for (int l = 0; l < 9; ++l) {
        const int rc = fetch(sdf, l, w_sdf[l], S[l]);
        if (rc != HN_OK) return rc;
    }
    for (int l = 0; l < 5; ++l) {
        const int rc = fetch(col, l, w_col[l], C[l]);
        if (rc != HN_OK) return rc;
    }
    HN_CHECK_HIP(hipStreamSynchronize(stream));
    const bool timing = getenv("HN_PACK_TIMING") != nullptr;     // stage times of a pack on stderr (not a launch path)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_fetch = now();
    if (timing) fprintf(stderr, "[hn pack] weights to host %.2f ms\n", ms(t_begin, t_fetch));
    // modes 0 .. 3 as the kernels' MODE 0, 1, 2, 4; "mode 4": the taped evaluation's copy of the mode-1 program on the
    // adjoint kernels' MFMA shape, built only where the evaluation kernels use the other one
    const bool eval16 = f->kind == HN_FIELD_OBJ ? (HN_OBJ_EVAL_MFMA16 != 0) : (HN_HAND_EVAL_MFMA16 != 0);
    for (int mode = 0; mode < (eval16 ? 5 : 4); ++mode) {
        if (!wanted(mode)) continue;
        const auto t_mode = now();
        Builder B;
        g_s16 = eval16 && mode < 2;
        const int prog = mode == 4 ? 1 : mode;
        if (f->kind == HN_FIELD_OBJ) {
            build_obj_stream(B, S, C, prog);
        } else {
            build_hand_stream(B, S, C, prog);
        }
        g_s16 = false;
        void** dst = mode == 0 ? &f->v2_sdf : (mode == 1 ? &f->v2_full : (mode == 2 ? &f->v2_adj : (mode == 3 ? &f->v2_adjonly : &f->v2_tape)));
        size_t* nb = mode == 0 ? &f->v2_sdf_bytes
                               : (mode == 1 ? &f->v2_full_bytes : (mode == 2 ? &f->v2_adj_bytes : (mode == 3 ? &f->v2_adjonly_bytes : &f->v2_tape_bytes)));
        const auto t_laid = now();
        const int rc = upload(B, dst, nb, stream);
        if (rc != HN_OK) return rc;
        {   // the plan of this (kind, mode), derived once: the same layout over sentinel matrices
            PackPlan& P = g_plans[plan_key(mode)];
            if (!P.tried) {
                P.tried = true;
                HostMat SS[9], CC[5];
                std::vector<float> sent;
                size_t base[N_SRC + 1];
                {
                    size_t off = 0;
                    auto put = [&](int t, size_t n) {
                        base[t] = off;
                        off += n;
                    };
                    for (int l = 0; l < 9; ++l) put(l, (size_t)S[l].rows * S[l].cols);
                    for (int l = 0; l < 5; ++l) put(9 + l, (size_t)C[l].rows * C[l].cols);
                    for (int l = 0; l < 9; ++l) put(14 + l, (size_t)S[l].rows);
                    for (int l = 0; l < 5; ++l) put(23 + l, (size_t)C[l].rows);
                    base[N_SRC] = off;
                    if (off < (1u << 23)) {
                        sent.resize(off);
                        for (size_t i = 0; i < off; ++i) {
                            const unsigned bits = 0x4B000000u + (unsigned)i;   // 2^23 + i: exact, survives every copy
                            memcpy(&sent[i], &bits, 4);
                        }
                    }
                }
                bool ok = !sent.empty();
                if (ok) {
                    for (int l = 0; l < 9; ++l) {
                        SS[l] = S[l];
                        SS[l].w = sent.data() + base[l];
                        SS[l].b = sent.data() + base[14 + l];
                        SS[l].dev = reinterpret_cast<const float*>((uintptr_t)(1 + l));
                    }
                    for (int l = 0; l < 5; ++l) {
                        CC[l] = C[l];
                        CC[l].w = sent.data() + base[9 + l];
                        CC[l].b = sent.data() + base[23 + l];
                        CC[l].dev = reinterpret_cast<const float*>((uintptr_t)(1 + 9 + l));
                    }
                    Builder B2;
                    g_s16 = eval16 && mode < 2;
                    if (f->kind == HN_FIELD_OBJ)
                        build_obj_stream(B2, SS, CC, prog);
                    else
                        build_hand_stream(B2, SS, CC, prog);
                    g_s16 = false;
                    ok = B2.blob.size() == B.blob.size() && B2.blocks.size() == B.blocks.size() && B2.maps == B.maps;
                    std::vector<TailEntry> tails;
                    const float* real[N_SRC];
                    for (int l = 0; l < 9; ++l) {
                        real[l] = S[l].w;
                        real[14 + l] = S[l].b;
                    }
                    for (int l = 0; l < 5; ++l) {
                        real[9 + l] = C[l].w;
                        real[23 + l] = C[l].b;
                    }
                    const size_t nf = B.blob.size() / 4;
                    for (size_t i = 0; i < nf && ok; ++i) {
                        unsigned b2, b1;
                        memcpy(&b2, B2.blob.data() + 4 * i, 4);
                        memcpy(&b1, B.blob.data() + 4 * i, 4);
                        if (b2 == 0u) {
                            ok = b1 == 0u || b1 == 0x80000000u;   // a tail float that is zero in the sentinel layout is zero in the real one
                            continue;
                        }
                        const size_t g = (size_t)b2 - 0x4B000000u;
                        if (b2 < 0x4B000000u || g >= base[N_SRC]) {   // a computed value, not a copy: no plan for this layout
                            ok = false;
                            break;
                        }
                        int t = 0;
                        while (g >= base[t + 1]) ++t;
                        const unsigned idx = (unsigned)(g - base[t]);
                        unsigned want;
                        memcpy(&want, real[t] + idx, 4);
                        if (want != b1) {
                            ok = false;
                            break;
                        }
                        tails.push_back(TailEntry{(unsigned)i, t, idx});
                    }
                    for (size_t k = 0; k < B.blocks.size() && ok; ++k) {   // same descriptors up to the matrix reference
                        const FragBlock &x = B.blocks[k], &y = B2.blocks[k];
                        const uintptr_t tnum = reinterpret_cast<uintptr_t>(y.mat);
                        ok = x.dst == y.dst && x.cols == y.cols && x.transposed == y.transposed && x.scale == y.scale && x.rowmap == y.rowmap &&
                             x.colslot == y.colslot && x.s == y.s && x.s16 == y.s16 && tnum >= 1 && tnum <= 14 &&
                             x.mat == (tnum <= 9 ? S[tnum - 1].dev : C[tnum - 10].dev);
                    }
                    if (ok) {
                        const size_t nbk = B2.blocks.size() * sizeof(FragBlock), nm = B2.maps.size() * sizeof(int), nt = tails.size() * sizeof(TailEntry);
                        const size_t o1 = (nbk + 255) & ~size_t(255), o2 = o1 + ((nm + 255) & ~size_t(255));
                        char* stage2 = reinterpret_cast<char*>(g_pin_up.get(o2 + nt + 256));
                        void* devp = nullptr;
                        if (stage2 != nullptr && hipMalloc(&devp, o2 + nt + 256) == hipSuccess) {
                            memcpy(stage2, B2.blocks.data(), nbk);
                            memcpy(stage2 + o1, B2.maps.data(), nm);
                            memcpy(stage2 + o2, tails.data(), nt);
                            if (hipMemcpyAsync(devp, stage2, o2 + nt, hipMemcpyHostToDevice, stream) == hipSuccess &&
                                hipStreamSynchronize(stream) == hipSuccess) {
                                P.bytes = B.blob.size();
                                P.blocks = reinterpret_cast<FragBlock*>(devp);
                                P.n_blocks = (int)B2.blocks.size();
                                P.maps = reinterpret_cast<int*>(reinterpret_cast<char*>(devp) + o1);
                                P.tails = reinterpret_cast<TailEntry*>(reinterpret_cast<char*>(devp) + o2);
                                P.n_tails = (int)tails.size();
                                P.usable = true;
                            } else {
                                (void)hipFree(devp);
                            }
                        }
                    }
                }
                if (timing) fprintf(stderr, "[hn pack] program %d: plan %s\n", mode, P.usable ? "kept" : "not usable");
            }
        }
        if (timing)
            fprintf(stderr, "[hn pack] program %d: %zu bytes, %zu fragment blocks, layout %.2f ms, upload + fill %.2f ms\n", mode, B.blob.size(),
                    B.blocks.size(), ms(t_mode, t_laid), ms(t_laid, now()));
    }
    return HN_OK;
}

// ---- hand field ---------------------------------------------------------------------------------------
// One bone's 66 features (utils/fields.py:142-147): [v, sin(2^k v) k<10, cos(2^k v) k<10, r(3), per channel
// c: sin(2^k r_c) k<7, cos(2^k r_c) k<7], all times the bone mask h.  33 (first | second) pairs; 32 of them
// fill 4 k-steps per bone (half 0 holds sines / first members, half 1 cosines / second members), the 33rd,
// (r_1 | r_2), goes to the LEFTOVER block: 3 k-steps whose element j of k-step u belongs to bone 8u + j.
//   k-step 0: v, frequencies 0..7          k-step 1: v, 8..9; r_0, 0..5
//   k-step 2: r_0, 6; r_1, 0..6            k-step 3: r_2, 0..6; (v | r_0)
static int bone_col(int idx) { return idx; }
static std::vector<int> bone_slots(int bone) {
    std::vector<int> c(64, -1);
    const int b0 = BONE_FEAT * bone;
    auto vcol = [&](int k, int h) { return b0 + 1 + k + (h ? PTS_FREQS : 0); };
    auto rcol = [&](int ch, int k, int h) { return b0 + 24 + 2 * HAND_DIR_FREQS * ch + k + (h ? HAND_DIR_FREQS : 0); };
    for (int h = 0; h < 2; ++h) {
        for (int j = 0; j < 8; ++j) c[0 * 16 + 8 * h + j] = vcol(j, h);
        for (int j = 0; j < 2; ++j) c[1 * 16 + 8 * h + j] = vcol(8 + j, h);
        for (int j = 2; j < 8; ++j) c[1 * 16 + 8 * h + j] = rcol(0, j - 2, h);
        c[2 * 16 + 8 * h + 0] = rcol(0, 6, h);
        for (int j = 1; j < 8; ++j) c[2 * 16 + 8 * h + j] = rcol(1, j - 1, h);
        for (int j = 0; j < 7; ++j) c[3 * 16 + 8 * h + j] = rcol(2, j, h);
        c[3 * 16 + 8 * h + 7] = h ? b0 + 21 : b0 + 0;   // (v | r_0)
    }
    (void)bone_col;
    return c;
}
static std::vector<int> left_slots() {
    std::vector<int> c(48, -1);
    for (int u = 0; u < 3; ++u)
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 8; ++j) {
                const int bone = 8 * u + j;
                if (bone < N_BONES) c[u * 16 + 8 * h + j] = BONE_FEAT * bone + (h ? 23 : 22);   // (r_1 | r_2)
            }
    return c;
}
// W[all 8 output tiles][feature space] in the order feature_pass consumes it: per bone two chunks (tiles 0..3,
// tiles 4..7; 4 k-steps), then the two leftover chunks (3 k-steps, + tail when given)
static void feature_block(Builder& B, const HostMat& M, float scale, int col_off, const float* tail0, const float* tail1) {
    std::vector<int> rows[2];
    for (int blk = 0; blk < 2; ++blk)
        for (int ti = 0; ti < 4; ++ti) rows[blk] = cat(rows[blk], rows_of_tile(4 * blk + ti, 256));
    for (int b = 0; b < N_BONES; ++b) {
        const std::vector<int> slots = offset(bone_slots(b), col_off);
        for (int blk = 0; blk < 2; ++blk) B.chunk(M, false, scale, 4, 4, rows[blk].data(), slots.data(), nullptr);
    }
    std::vector<int> ls = offset(left_slots(), col_off);
    const int lks = g_s16 ? 4 : 3;   // the 16x16x32 shape takes k-steps in pairs: a fourth, empty one
    ls.resize(16 * lks, -1);
    B.chunk(M, false, scale, 4, lks, rows[0].data(), ls.data(), tail0);
    B.chunk(M, false, scale, 4, lks, rows[1].data(), ls.data(), tail1);
}
static void tail_biases4(float* tail, const HostMat& M, int pass) {
    for (int ti = 0; ti < 4; ++ti) {
        float v[32];
        for (int i = 0; i < 32; ++i) v[i] = M.b[32 * (4 * pass + ti) + i];
        tail_put(tail, ti, v);
    }
}
// d sdf / d features = W0^T dz0 + W4[:, 256:]^T dz4: rows = the feature slots of every bone (2 tiles) and of
// the leftover block (2 tiles); per tile first the W0 chunk, then the W4 chunk (same accumulator)
static void one_slot_tile_T(Builder& B, const HostMat& M, float scale, const std::vector<int>& in_slots, int col_off, int u) {
    const std::vector<int> slots = hid_slots();
    std::vector<int> rows(32, -1);
    for (int r = 0; r < 32; ++r) {
        int s, h, j;
        k_of_row(32 * u + r, s, h, j);
        const int c = in_slots[s * 16 + 8 * h + j];
        rows[r] = c >= 0 ? col_off + c : -1;
    }
    B.chunk(M, true, scale, 1, 16, rows.data(), slots.data(), nullptr);
}
// d sdf / d features rows in the order every Jacobian pass consumes them: the leftover block first (its per-bone values
// are parked while the bones are visited), then the bones; one matrix (M4 == nullptr) or two
static void feature_rows_T_adj(Builder& B, const HostMat& M0, const HostMat* M4, float scale4, int col_off4) {
    for (int k = 0; k <= N_BONES; ++k) {
        const int b = k == 0 ? N_BONES : k - 1;
        std::vector<int> sl = b < N_BONES ? bone_slots(b) : left_slots();
        sl.resize(64, -1);
        for (int u = 0; u < 2; ++u) {
            one_slot_tile_T(B, M0, 1.f, sl, 0, u);
            if (M4 != nullptr) one_slot_tile_T(B, *M4, scale4, sl, col_off4, u);
        }
    }
}

// hand sdf network, forward-oriented chunks lin0..lin7 (the forward pass; without the bias tails of lin0: the adjoint's
// forward-direction sweep)
static void hand_sdf_forward_chunks(Builder& B, const HostMat* S, bool lin0_bias_tails) {
    const float rs2 = (float)(1.0 / sqrt(2.0));
    const std::vector<int> hs = hid_slots();
    // lin0: two passes of 4 output tiles over the feature space; the leftover chunk's tail holds the 4 biases
    if (lin0_bias_tails) {
        float tail0[256] = {0.f}, tail1[256] = {0.f};
        tail_biases4(tail0, S[0], 0);
        tail_biases4(tail1, S[0], 1);
        feature_block(B, S[0], 1.f, 0, tail0, tail1);
    } else {
        feature_block(B, S[0], 1.f, 0, nullptr, nullptr);
    }
    fwd_tiles(B, S[1], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    fwd_tiles(B, S[2], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    fwd_tiles(B, S[3], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    // lin4 = [a4 (256) | features (1386)] / sqrt2: 8 hidden tiles (+bias), then the feature block
    fwd_tiles(B, S[4], rs2, 8, 256, 0, hs, 16, nullptr, 0);
    feature_block(B, S[4], rs2, H, nullptr, nullptr);
    fwd_tiles(B, S[5], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    fwd_tiles(B, S[6], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    {
        std::vector<float> w8(256);
        for (int i = 0; i < 256; ++i) w8[i] = S[8].at(0, i);
        const float* extra[1] = {w8.data()};
        fwd_tiles(B, S[7], 1.f, 8, 256, 0, hs, 16, extra, 1);
    }
}

static void hand_adjoint_chunks(Builder& B, const HostMat* S, const HostMat* C);

// The hand program (contract with k_field2_hand).  mode 0: sdf only; 1: full evaluation; 2: full evaluation followed by
// its adjoint (hn_field2_hand_adj.inl); 3: the adjoint alone (from a kept tape)
void build_hand_stream(Builder& B, const HostMat* S, const HostMat* C, int mode) {
    if (mode == 3) {
        hand_adjoint_chunks(B, S, C);
        return;
    }
    const float rs2 = (float)(1.0 / sqrt(2.0));
    const std::vector<int> hs = hid_slots();
    hand_sdf_forward_chunks(B, S, true);
    if (mode == 0) return;
    fwd_tiles(B, S[8], 1.f, 8, 256, 1, hs, 16, nullptr, 0);
    for (int l = 7; l >= 1; --l) bwd_tiles(B, S[l], l == 4 ? rs2 : 1.f, 8, 256, 0, 256, 16);
    feature_rows_T_adj(B, S[0], &S[4], rs2, H);   // leftover block first: its rows join the bones' sums (hn_field2_hand.hip)
    // colour lin0 = [features 1386 | feature vector 256 | enc4(g) 27] (utils/fields.py:224-229)
    {
        const std::vector<int> v4 = vec4_slots();
        const std::vector<int> fv = offset(hs, HAND_IN);
        const std::vector<int> gs = offset(v4, HAND_IN + H);
        for (int t = 0; t < 8; ++t) {
            const std::vector<int> rows = rows_of_tile(t, 256);
            B.chunk(C[0], false, 1.f, 1, 16, rows.data(), fv.data(), nullptr);
        }
        feature_block(B, C[0], 1.f, 0, nullptr, nullptr);
        for (int p = 0; p < 2; ++p) {
            std::vector<int> rows;
            for (int ti = 0; ti < 4; ++ti) rows = cat(rows, rows_of_tile(4 * p + ti, 256));
            float tail[256] = {0.f};
            tail_biases4(tail, C[0], p);
            B.chunk(C[0], false, 1.f, 4, 2, rows.data(), gs.data(), tail);
        }
    }
    fwd_tiles(B, C[1], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    fwd_tiles(B, C[2], 1.f, 8, 256, 0, hs, 16, nullptr, 0);
    {
        std::vector<float> w0(256), w1(256), w2(256);
        for (int i = 0; i < 256; ++i) {
            w0[i] = C[4].at(0, i);
            w1[i] = C[4].at(1, i);
            w2[i] = C[4].at(2, i);
        }
        const float* extra[3] = {w0.data(), w1.data(), w2.data()};
        fwd_tiles(B, C[3], 1.f, 8, 256, 0, hs, 16, extra, 3);
    }
    if (mode < 2) return;
    hand_adjoint_chunks(B, S, C);
}

// ---- adjoint (contract with hn_field2_hand_adj.inl) ------------------------------------------------------------------
static void hand_adjoint_chunks(Builder& B, const HostMat* S, const HostMat* C) {
    const float rs2 = (float)(1.0 / sqrt(2.0));
    for (int c = 0; c < 3; ++c) {   // the three rows of colour lin4, one tail-format KiB each
        float tail[256];
        for (int t = 0; t < 8; ++t) {
            float v[32];
            for (int i = 0; i < 32; ++i) v[i] = C[4].at(c, 32 * t + i);
            tail_put(tail, t, v);
        }
        B.raw(tail, TAIL_BYTES);
    }
    bwd_tiles(B, C[3], 1.f, 8, 256, 0, 256, 16);
    bwd_tiles(B, C[2], 1.f, 8, 256, 0, 256, 16);
    bwd_tiles(B, C[1], 1.f, 8, 256, 0, 256, 16);
    bwd_tiles(B, C[0], 1.f, 8, 256, HAND_IN, 256, 16);          // C0^T, feature-vector rows -> fb
    slot_rows_T(B, C[0], 1.f, vec4_slots(), HAND_IN + H);      // C0^T, enc(g) rows (one tile) -> gb
    feature_rows_T_adj(B, C[0], nullptr, 1.f, 0);              // C0^T, feature rows (leftover first)
    hand_sdf_forward_chunks(B, S, false);                      // forward-direction sweep (no biases)
    w8_T_tiles(B, S[8]);
    for (int l = 7; l >= 1; --l) bwd_tiles(B, S[l], l == 4 ? rs2 : 1.f, 8, 256, 0, 256, 16);
    feature_rows_T_adj(B, S[0], &S[4], rs2, H);                // W0^T zb0 + W4x^T zb4 (leftover first)
}

}  // namespace v2
}  // namespace hn
