// Weight packer: reads the reference's state-dict layout (lin{l}.weight_g / weight_v / bias,
// old-style weight-norm, utils/fields.py:120-121, 216-217, 307-308, 382-383), folds
// W = g * v / ||v||_row once and re-lays every matrix the kernels multiply by into MFMA
// A-fragment order (hn_mlp.h): float4 [out_tile][step/4][lane], where the fragment of
// lane l for k-step s is  scale * W[rowmap[32 t + (l&31)]][colmap[2 s + (l>>5)]].
// Row maps and column maps describe our own orderings of neurons / input columns; -1 = pad.
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "hn_common.h"

namespace hn {

// ---- device memory of packed fields: a size-keyed cache in front of hipMalloc / hipFree -------------------------------
// A training loop re-packs its field after every optimiser step (honerf_amd/training.py): ~70 allocations and as many
// frees per pack.  hipFree synchronises the device and both calls cost from tens of microseconds to several
// milliseconds depending on the state of the driver's pools (measured: 3 - 36 ms per re-pack of a hand field with the
// same code, profiles/r02/README.md "training step").  Freed blocks are therefore kept, keyed by (device, size), and
// handed out again: after the first two packs a re-pack allocates nothing.  The cache only ever holds what fields of
// this process have released; hn_field_destroy's contract is unchanged (no work that uses the field may be in flight).
// A pack's own temporaries (the folded matrices) go back while the pack's kernels may still be reading them: those blocks
// carry a fence (an event recorded on the pack's stream at release time).  Taking such a block on the same stream is
// ordered by the stream; on another stream the taker's stream waits for the event; a taker that names no stream waits on
// the host.
namespace {
struct PoolKey {
    int dev;
    size_t bytes;
    bool operator<(const PoolKey& o) const { return dev != o.dev ? dev < o.dev : bytes < o.bytes; }
};
struct PoolFence {
    hipEvent_t ev = nullptr;
    hipStream_t s = nullptr;
    ~PoolFence() {
        if (ev != nullptr) (void)hipEventDestroy(ev);
    }
};
std::mutex g_pool_mu;
std::multimap<PoolKey, void*> g_pool_free;
std::map<void*, PoolKey> g_pool_live;
std::map<void*, std::shared_ptr<PoolFence>> g_pool_fence;   // released blocks whose last reader may still be in flight
hipError_t pool_take(void** p, size_t bytes, bool ordered, hipStream_t s) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bytes = (bytes + 255) & ~size_t(255);
    std::shared_ptr<PoolFence> fence;
    bool hit = false;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_free.find(PoolKey{dev, bytes});
        if (it != g_pool_free.end()) {
            *p = it->second;
            g_pool_free.erase(it);
            g_pool_live[*p] = PoolKey{dev, bytes};
            auto fi = g_pool_fence.find(*p);
            if (fi != g_pool_fence.end()) {
                fence = fi->second;
                g_pool_fence.erase(fi);
            }
            hit = true;
        }
    }
    if (hit) {
        if (fence && fence->ev != nullptr) {
            if (!ordered) return hipEventSynchronize(fence->ev);
            if (fence->s != s) return hipStreamWaitEvent(s, fence->ev, 0);
        }
        return hipSuccess;
    }
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_pool_live[*p] = PoolKey{dev, bytes};
    }
    return e;
}
void pool_give(void* p, const std::shared_ptr<PoolFence>& fence) {
    if (p == nullptr) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_live.find(p);
    if (it == g_pool_live.end()) {   // not ours: straight to the driver
        (void)hipFree(p);
        return;
    }
    g_pool_free.insert({it->second, p});
    g_pool_live.erase(it);
    if (fence) g_pool_fence[p] = fence;
}
}  // namespace
hipError_t pool_alloc(void** p, size_t bytes) { return pool_take(p, bytes, false, nullptr); }
// the block's first use is enqueued on `s`
hipError_t pool_alloc_on(void** p, size_t bytes, hipStream_t s) { return pool_take(p, bytes, true, s); }
size_t pool_trim() {   // hn_release_cached_memory: every cached (released) block back to the driver
    std::vector<void*> blocks;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (auto& kv : g_pool_free) {
            blocks.push_back(kv.second);
            bytes += kv.first.bytes;
        }
        g_pool_free.clear();
        g_pool_fence.clear();
    }
    for (void* p : blocks) (void)hipFree(p);   // (hipFree waits for the device: the fences are moot)
    return bytes;
}
void pool_free(void* p) { pool_give(p, nullptr); }
// release blocks that work already enqueued on `s` may still read
void pool_free_after(void* const* blocks, size_t n, hipStream_t s) {
    if (n == 0) return;
    auto fence = std::make_shared<PoolFence>();
    fence->s = s;
    if (hipEventCreateWithFlags(&fence->ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(fence->ev, s) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s);   // no event to be had: wait here, release unfenced
        fence.reset();
    }
    for (size_t i = 0; i < n; ++i) pool_give(blocks[i], fence);
}

// the per-layer launches of a pack batched into one each (blockIdx.y = layer): a re-pack is a chain of dependent launches
struct PackLayer {
    const float *g, *v, *b;   // weight_g (or NULL), weight_v, bias of the layer
    float *w, *raw_w, *raw_b; // folded matrix; its retained row-pitched copy; the retained bias
    int out, in, ld;
};
struct PackTable {
    PackLayer l[14];
};
__global__ void k_fold_weight_norm_all(const PackTable tab) {
    const PackLayer& L_ = tab.l[blockIdx.y];
    const int row = blockIdx.x, in = L_.in;
    if (row >= L_.out) return;
    const float* vr = L_.v + (size_t)row * in;
    float scale = 1.f;
    if (L_.g != nullptr) {
        float ss = 0.f;
        for (int i = threadIdx.x; i < in; i += blockDim.x) ss = fmaf(vr[i], vr[i], ss);
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        __shared__ float part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
        __syncthreads();
        ss = part[0] + part[1] + part[2] + part[3];
        scale = L_.g[row] / sqrtf(ss);   // torch._weight_norm: v * (g / ||v||)
    }
    // the folded row, its copy in the retained row-pitched block (pad columns stay zero: the block is cleared first) and the bias
    for (int i = threadIdx.x; i < in; i += blockDim.x) {
        const float x = vr[i] * scale;
        L_.w[(size_t)row * in + i] = x;
        if (L_.raw_w != nullptr) L_.raw_w[(size_t)row * L_.ld + i] = x;
    }
    if (threadIdx.x == 0 && L_.raw_b != nullptr) L_.raw_b[row] = L_.b[row];
}

// dst fragment order; src row-major [src_rows][src_cols]; transposed: element (row, col) = src[col][row]
__global__ void k_pack(const float* __restrict__ src, int src_cols, int transposed, const int* __restrict__ rowmap,
                       const int* __restrict__ colmap, int out_tiles, int steps, float scale,
                       float4* __restrict__ dst) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [tile][q][lane]
    const size_t total = (size_t)out_tiles * (steps / 4) * 64;
    if (idx >= total) return;
    const int lane = idx & 63;
    const int q = (idx >> 6) % (steps / 4);
    const int t = (idx >> 6) / (steps / 4);
    const int row = rowmap[32 * t + (lane & 31)];
    float v[4];
    for (int jj = 0; jj < 4; ++jj) {
        const int col = colmap[2 * (4 * q + jj) + (lane >> 5)];
        float x = 0.f;
        if (row >= 0 && col >= 0) x = transposed ? src[(size_t)col * src_cols + row] : src[(size_t)row * src_cols + col];
        v[jj] = x * scale;
    }
    dst[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void k_gather_vec(const float* __restrict__ src, const int* __restrict__ map, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = map[i] >= 0 ? src[map[i]] : 0.f;
}

// ---- index-space descriptions ----------------------------------------------------------------
// "tile-row space": position 32 t + i  -> neuron (or -1).
static std::vector<int> identity_rows(int n, int tiles, int offset = 0) {
    std::vector<int> m(32 * tiles, -1);
    for (int i = 0; i < n; ++i) m[i] = offset + i;
    return m;
}
// B-operand columns when the input is an accumulator tile array with the given tile-row map:
// k-step s = 16 u + r contracts tile rows tile_row(r, 0) and tile_row(r, 1) of tile u.
static std::vector<int> cols_from_tiles(const std::vector<int>& tile_rows) {
    const int tiles = (int)tile_rows.size() / 32;
    std::vector<int> c(2 * 16 * tiles);
    for (int u = 0; u < tiles; ++u)
        for (int r = 0; r < 16; ++r)
            for (int h = 0; h < 2; ++h) c[2 * (16 * u + r) + h] = tile_rows[32 * u + tile_row(r, h)];
    return c;
}
// Tile-row map of an accumulator whose rows are the members of k-step pairs:
// tile u, row i  <->  pair 16 u + r, member h  with i = tile_row(r, h).
static std::vector<int> rows_from_pairs(const std::vector<int>& pair_cols) {
    const int pairs = (int)pair_cols.size() / 2;
    const int tiles = (pairs + 15) / 16;
    std::vector<int> m(32 * tiles, -1);
    for (int p = 0; p < pairs; ++p)
        for (int h = 0; h < 2; ++h) m[32 * (p / 16) + tile_row(p % 16, h)] = pair_cols[2 * p + h];
    return m;
}

// pair list (2 columns per k-step) of [x, enc_L(x)] for a 3-vector, reference column order
// [x(3), per channel c: sin 2^0..2^(L-1), cos 2^0..2^(L-1)] (utils/fields.py:13-20), padded to `steps`.
static std::vector<int> vec_pairs(int L, int base, int steps) {
    std::vector<int> c(2 * steps, -1);
    int s = 0;
    for (int ch = 0; ch < 3; ++ch)
        for (int k = 0; k < L; ++k, ++s) {
            c[2 * s] = base + 3 + 2 * L * ch + k;
            c[2 * s + 1] = base + 3 + 2 * L * ch + L + k;
        }
    c[2 * s] = base + 0;
    c[2 * s + 1] = base + 1;
    ++s;
    c[2 * s] = base + 2;
    return c;
}

// pair list of one hand bone's 66 features (utils/fields.py:142-147): [v, sin(2^k v) k<10,
// cos(2^k v) k<10, r(3), per channel c: sin(2^k r_c) k<7, cos(2^k r_c) k<7], 36 steps.
static void bone_pairs(int bone, int base, std::vector<int>& c) {
    const int b0 = base + BONE_FEAT * bone;
    std::vector<int> p(2 * BONE_STEPS, -1);
    p[0] = b0 + 0;
    p[1] = b0 + 21;
    p[2] = b0 + 22;
    p[3] = b0 + 23;
    for (int k = 0; k < PTS_FREQS; ++k) {
        p[2 * (2 + k)] = b0 + 1 + k;
        p[2 * (2 + k) + 1] = b0 + 11 + k;
    }
    for (int ch = 0; ch < 3; ++ch)
        for (int k = 0; k < HAND_DIR_FREQS; ++k) {
            const int s = 12 + HAND_DIR_FREQS * ch + k;
            p[2 * s] = b0 + 24 + 2 * HAND_DIR_FREQS * ch + k;
            p[2 * s + 1] = b0 + 24 + 2 * HAND_DIR_FREQS * ch + HAND_DIR_FREQS + k;
        }
    c.insert(c.end(), p.begin(), p.end());
}
static std::vector<int> hand_pairs(int base, int n_bones_padded) {
    std::vector<int> c;
    for (int b = 0; b < n_bones_padded; ++b) {
        if (b < N_BONES)
            bone_pairs(b, base, c);
        else
            c.insert(c.end(), 2 * BONE_STEPS, -1);
    }
    return c;
}

// ---- the packer -----------------------------------------------------------------------------
struct Packer {
    hipStream_t stream;
    bool dry = true;          // first pass: only count bytes
    char* base = nullptr;
    size_t used = 0;
    std::vector<void*> temps; // device temporaries to free at the end
    int status = HN_OK;

    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        void* p = dry ? nullptr : base + used;
        used += bytes;
        return p;
    }
    // Row / column maps of a packed matrix depend on the field kind only, not on the weights: uploaded once per
    // (device, content) and kept for the life of the process (a few hundred KiB), so that a re-pack issues no
    // host-to-device copies of its own here (~60 small pageable copies per pack, each a blocking staged transfer).
    int* upload(const std::vector<int>& v) {
        static std::mutex mu;
        static std::map<std::pair<int, std::vector<int>>, int*> cache;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find({dev, v});
        if (it != cache.end()) return it->second;
        int* d = nullptr;
        if (hipMalloc(&d, v.size() * sizeof(int)) != hipSuccess) {
            status = HN_ENOMEM;
            return nullptr;
        }
        if (hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            status = HN_EHIP;
            (void)hipFree(d);
            return nullptr;
        }
        cache[{dev, v}] = d;
        return d;
    }
    PackedMat mat(const float* w_eff, int src_cols, bool transposed, const std::vector<int>& rowmap,
                  const std::vector<int>& colmap, float scale) {
        PackedMat m;
        m.out_tiles = (int)rowmap.size() / 32;
        m.steps = (int)colmap.size() / 2;
        const size_t n4 = (size_t)m.out_tiles * (m.steps / 4) * 64;
        float4* dst = reinterpret_cast<float4*>(take(n4 * sizeof(float4)));
        m.w = dst;
        if (dry || status != HN_OK) return m;
        int* dr = upload(rowmap);
        int* dc = upload(colmap);
        if (status != HN_OK) return m;
        hipLaunchKernelGGL(k_pack, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, w_eff, src_cols,
                           transposed ? 1 : 0, dr, dc, m.out_tiles, m.steps, scale, dst);
        return m;
    }
    const float* vec(const float* src, const std::vector<int>& map) {
        float* dst = reinterpret_cast<float*>(take(map.size() * sizeof(float)));
        if (dry || status != HN_OK) return dst;
        int* dm = upload(map);
        if (status != HN_OK) return dst;
        hipLaunchKernelGGL(k_gather_vec, dim3((unsigned)((map.size() + 255) / 256)), dim3(256), 0, stream, src, dm,
                           (int)map.size(), dst);
        return dst;
    }
    void free_temps() {   // the pack's kernels on `stream` may still be reading them: released behind a fence
        pool_free_after(temps.data(), temps.size(), stream);
        temps.clear();
    }
};

static int check_shapes(int kind, const hn_mlp_desc* sdf, const hn_mlp_desc* col) {
    const int in0 = kind == HN_FIELD_OBJ ? OBJ_IN : HAND_IN;
    HN_REQUIRE(sdf->n_layers == 9 && col->n_layers == 5, "expected 9 sdf and 5 colour layers, got %d / %d",
               sdf->n_layers, col->n_layers);
    for (int l = 0; l < 9; ++l) {
        int out = H, in = H;
        if (l == 0) in = in0;
        if (l == 8) out = H + 1;
        if (kind == HN_FIELD_OBJ && l == 3) out = L3_OUT_OBJ;
        if (kind == HN_FIELD_HAND && l == 4) in = H + HAND_IN;
        HN_REQUIRE(sdf->out_dim[l] == out && sdf->in_dim[l] == in, "sdf lin%d: expected [%d,%d], got [%d,%d]", l, out,
                   in, sdf->out_dim[l], sdf->in_dim[l]);
    }
    const int cin = kind == HN_FIELD_OBJ ? OBJ_IN + 27 + H + 27 : HAND_IN + H + 27;
    for (int l = 0; l < 5; ++l) {
        const int out = l == 4 ? 3 : H, in = l == 0 ? cin : H;
        HN_REQUIRE(col->out_dim[l] == out && col->in_dim[l] == in, "colour lin%d: expected [%d,%d], got [%d,%d]", l,
                   out, in, col->out_dim[l], col->in_dim[l]);
    }
    return HN_OK;
}

static int build(hn_field* f, Packer& pk, const hn_mlp_desc* sdf, const hn_mlp_desc* col, float* const* w_sdf,
                 float* const* w_col, const float* w8_host_row_bias /* [b8, cb0, cb1, cb2] */) {
    const bool obj = f->kind == HN_FIELD_OBJ;
    const float rs2 = (float)(1.0 / sqrt(2.0));
    const std::vector<int> hid = identity_rows(H, NT);
    const std::vector<int> hid_cols = cols_from_tiles(hid);
    // input ("X") space of the sdf net
    const std::vector<int> x_pairs = obj ? vec_pairs(PTS_FREQS, 0, OBJ_X_STEPS) : hand_pairs(0, N_BONES);
    // rows of d sdf / d X: obj 2 tiles; hand 6 groups x 9 tiles over 24 (padded) bones
    const std::vector<int> x_rows = rows_from_pairs(obj ? x_pairs : hand_pairs(0, BONE_GROUP * N_GROUPS));

    // ---- sdf forward -------------------------------------------------------------------
    f->sdf_fwd[0] = pk.mat(w_sdf[0], sdf->in_dim[0], false, hid, x_pairs, 1.f);
    f->sdf_bias[0] = pk.vec(sdf->bias[0], hid);
    for (int l = 1; l <= 7; ++l) {
        std::vector<int> rows = hid, cols = hid_cols;
        float scale = 1.f;
        if (obj && l == 3) rows = identity_rows(L3_OUT_OBJ, 7);
        if (l == 4) {
            scale = rs2;
            if (obj) cols = cols_from_tiles(identity_rows(L3_OUT_OBJ, 7));
        }
        f->sdf_fwd[l] = pk.mat(w_sdf[l], sdf->in_dim[l], false, rows, cols, scale);
        f->sdf_bias[l] = pk.vec(sdf->bias[l], rows);
    }
    {   // skip columns of lin4 over the X space
        std::vector<int> cols = x_pairs;
        const int off = obj ? L3_OUT_OBJ : H;
        for (int& c : cols)
            if (c >= 0) c += off;
        f->sdf_skip = pk.mat(w_sdf[4], sdf->in_dim[4], false, hid, cols, rs2);
        // and its transpose: rows = X space
        std::vector<int> xr = x_rows;
        for (int& c : xr)
            if (c >= 0) c += off;
        f->sdf_bwd_in4 = pk.mat(w_sdf[4], sdf->in_dim[4], true, xr, hid_cols, rs2);
    }
    f->sdf_bwd_in0 = pk.mat(w_sdf[0], sdf->in_dim[0], true, x_rows, hid_cols, 1.f);
    {   // lin8: row 0 = sdf (plain vector), rows 1..256 = feature vector
        f->sdf_fwd[8] = pk.mat(w_sdf[8], sdf->in_dim[8], false, identity_rows(H, NT, 1), hid_cols, 1.f);
        f->sdf_bias[8] = pk.vec(sdf->bias[8], identity_rows(H, NT, 1));
        f->sdf_w8row = pk.vec(w_sdf[8], identity_rows(H, NT));
        f->sdf_b8 = w8_host_row_bias[0];
    }
    // ---- sdf reverse sweep: W_l^T, rows = layer input space, contraction over layer outputs
    for (int l = 1; l <= 7; ++l) {
        std::vector<int> rows = hid, cols = hid_cols;
        float scale = 1.f;
        if (l == 4) {
            scale = rs2;
            if (obj) rows = identity_rows(L3_OUT_OBJ, 7);
        }
        if (obj && l == 3) cols = cols_from_tiles(identity_rows(L3_OUT_OBJ, 7));
        f->sdf_bwd[l] = pk.mat(w_sdf[l], sdf->in_dim[l], true, rows, cols, scale);
    }
    // ---- colour net ----------------------------------------------------------------------
    const int cin = col->in_dim[0];
    if (obj) {
        // [enc10(p) 63 | enc4(d) 27 | feature 256 | enc4(g) 27]  (utils/fields.py:389-396)
        f->col_in_x = pk.mat(w_col[0], cin, false, hid, x_pairs, 1.f);
        f->col_in_d = pk.mat(w_col[0], cin, false, hid, vec_pairs(OBJ_DIR_FREQS, OBJ_IN, VEC_STEPS), 1.f);
        std::vector<int> fc = hid_cols;
        for (int& c : fc) c += OBJ_IN + 27;
        f->col_in_f = pk.mat(w_col[0], cin, false, hid, fc, 1.f);
        f->col_in_g = pk.mat(w_col[0], cin, false, hid, vec_pairs(GRAD_FREQS, OBJ_IN + 27 + H, VEC_STEPS), 1.f);
    } else {
        // [xyz_feature 1386 | feature 256 | enc4(g) 27]  (utils/fields.py:224-229)
        f->col_in_x = pk.mat(w_col[0], cin, false, hid, x_pairs, 1.f);
        std::vector<int> fc = hid_cols;
        for (int& c : fc) c += HAND_IN;
        f->col_in_f = pk.mat(w_col[0], cin, false, hid, fc, 1.f);
        f->col_in_g = pk.mat(w_col[0], cin, false, hid, vec_pairs(GRAD_FREQS, HAND_IN + H, VEC_STEPS), 1.f);
    }
    f->col_bias[0] = pk.vec(col->bias[0], hid);
    for (int l = 1; l <= 3; ++l) {
        f->col_fwd[l] = pk.mat(w_col[l], H, false, hid, hid_cols, 1.f);
        f->col_bias[l] = pk.vec(col->bias[l], hid);
    }
    {
        std::vector<int> m(3 * H);
        for (int i = 0; i < 3 * H; ++i) m[i] = i;
        f->col_wlast = pk.vec(w_col[4], m);
        for (int c = 0; c < 3; ++c) f->col_blast[c] = w8_host_row_bias[1 + c];
    }
    return pk.status;
}

namespace v2 {
int build_v2_streams(hn_field* f, const hn_mlp_desc* sdf, const hn_mlp_desc* col, float* const* w_sdf,
                     float* const* w_col, hipStream_t stream, bool eval_only);
}

int field_create(int kind, const hn_mlp_desc* sdf, const hn_mlp_desc* col, float variance, float scale, int precision,
                 hn_field** out, hipStream_t stream) {
    HN_REQUIRE(out != nullptr && sdf != nullptr && col != nullptr, "null argument");
    HN_REQUIRE(kind == HN_FIELD_OBJ || kind == HN_FIELD_HAND, "unknown field kind %d", kind);
    // HN_PACK_EVAL_ONLY: no adjoint weight streams (training re-packs every step and differentiates through
    // hn_field_param_bwd, which works on the retained row-major matrices)
    const bool eval_only = (precision & HN_PACK_EVAL_ONLY) != 0;
    precision &= ~HN_PACK_EVAL_ONLY;
    // HN_PREC_F16 is HN_PREC_F16X3 (same packed streams, same adjoint / tape kernels) whose evaluation kernels take one
    // MFMA pass per product in the hidden layers
    const bool single_pass = precision == HN_PREC_F16;
    if (single_pass) precision = HN_PREC_F16X3;
    HN_REQUIRE(precision == HN_PREC_FP32 || precision == HN_PREC_F16X3, "unsupported precision %d", precision);
    int rc = check_shapes(kind, sdf, col);
    if (rc != HN_OK) return rc;

    const bool timing = getenv("HN_PACK_TIMING") != nullptr;     // stage times of a pack on stderr (not a launch path)
    auto t_prev = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) {
        if (!timing) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[hn pack] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    hn_field* f = new hn_field();
    f->kind = kind;
    f->precision = precision;
    f->single_pass = single_pass ? 1 : 0;
    f->variance = variance;
    f->scale = scale;
    {
        float s = expf(variance * 10.f);
        f->inv_s = fminf(fmaxf(s, 1e-6f), 1e6f);
    }
    // fold weight norm into temporaries
    Packer pk;
    pk.stream = stream;
    float* w_sdf[9] = {};
    float* w_col[5] = {};
    // The retained row-major block (folded matrices with 16-byte aligned rows, biases: what the adjoint path reads) is laid
    // out first, so that ONE launch folds the weight norm of all 14 layers, writes the folded matrices the packers read and
    // fills the retained block (it was 14 + 14 launches and 14 copies).
    {
        size_t total = 0;
        auto pad = [](size_t n) { return (n + 63) & ~size_t(63); };
        auto pitch = [](int in) { return (in + 3) & ~3; };   // rows start 16-byte aligned: vector loads in k_dense
        for (int l = 0; l < 9; ++l) total += pad((size_t)sdf->out_dim[l] * pitch(sdf->in_dim[l])) + pad(sdf->out_dim[l]);
        for (int l = 0; l < 5; ++l) total += pad((size_t)col->out_dim[l] * pitch(col->in_dim[l])) + pad(col->out_dim[l]);
        if (pool_alloc_on(&f->raw, total * sizeof(float), stream) != hipSuccess) {
            set_error("hipMalloc of %zu bytes for the folded weights failed", total * sizeof(float));
            rc = HN_ENOMEM;
        } else {
            (void)hipMemsetAsync(f->raw, 0, total * sizeof(float), stream);
            f->raw_floats = total;
            PackTable tab;
            float* q = reinterpret_cast<float*>(f->raw);
            int max_out = 0;
            for (int li = 0; li < 14 && rc == HN_OK; ++li) {
                const bool is_sdf = li < 9;
                const hn_mlp_desc* d = is_sdf ? sdf : col;
                const int l = is_sdf ? li : li - 9;
                const int out = d->out_dim[l], in = d->in_dim[l], ld = pitch(in);
                float** dst = is_sdf ? &w_sdf[l] : &w_col[l];
                if (pool_alloc_on(reinterpret_cast<void**>(dst), (size_t)out * in * sizeof(float), stream) != hipSuccess) {
                    set_error("hipMalloc of a folded matrix failed");
                    rc = HN_ENOMEM;
                    break;
                }
                pk.temps.push_back(*dst);
                float* raw_w = q;
                q += pad((size_t)out * ld);
                float* raw_b = q;
                q += pad((size_t)out);
                tab.l[li] = PackLayer{reinterpret_cast<const float*>(d->weight_g[l]), reinterpret_cast<const float*>(d->weight_v[l]),
                                      reinterpret_cast<const float*>(d->bias[l]), *dst, raw_w, raw_b, out, in, ld};
                max_out = out > max_out ? out : max_out;
                if (is_sdf) {
                    f->sdf_out[l] = out;
                    f->sdf_in[l] = in;
                    f->sdf_ld[l] = ld;
                    f->raw_sdf_w[l] = raw_w;
                    f->raw_sdf_b[l] = raw_b;
                } else {
                    f->col_out[l] = out;
                    f->col_in[l] = in;
                    f->col_ld[l] = ld;
                    f->raw_col_w[l] = raw_w;
                    f->raw_col_b[l] = raw_b;
                }
            }
            if (rc == HN_OK) {
                hipLaunchKernelGGL(k_fold_weight_norm_all, dim3(max_out, 14), dim3(256), 0, stream, tab);
                if (hipGetLastError() != hipSuccess) {
                    set_error("folding the weight norm failed to launch");
                    rc = HN_EHIP;
                }
            }
        }
    }
    // The fp32 kernel family's fragment programs (and the two last-layer biases it keeps on the host) are built for HN_PREC_FP32
    // fields only: an f16x3 field never launches those kernels, and its re-pack -- the per-iteration cost of a training loop --
    // then waits for nothing on the host (the f16x3 kernels read lin8's bias from the retained copy on the device).
    const bool fp32_programs = precision == HN_PREC_FP32;
    float hostb[4] = {0.f, 0.f, 0.f, 0.f};
    if (rc == HN_OK && fp32_programs) {
        if (hipMemcpyAsync(&hostb[0], sdf->bias[8], sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipMemcpyAsync(&hostb[1], col->bias[4], 3 * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) {
            set_error("reading last-layer biases failed");
            rc = HN_EHIP;
        }
    }
    stage("fold + last-layer biases to host");
    if (rc == HN_OK && fp32_programs) {
        pk.dry = true;
        pk.used = 0;
        build(f, pk, sdf, col, w_sdf, w_col, hostb);
        f->blob_bytes = pk.used;
        if (pool_alloc_on(&f->blob, f->blob_bytes, stream) != hipSuccess) {
            set_error("hipMalloc of %zu bytes for packed weights failed", f->blob_bytes);
            rc = HN_ENOMEM;
        }
    }
    if (rc == HN_OK && fp32_programs) {
        pk.dry = false;
        pk.base = reinterpret_cast<char*>(f->blob);
        pk.used = 0;
        rc = build(f, pk, sdf, col, w_sdf, w_col, hostb);
        if (rc == HN_OK && hipGetLastError() != hipSuccess) {
            set_error("a pack kernel failed to launch");
            rc = HN_EHIP;
        }
        if (hipStreamSynchronize(stream) != hipSuccess && rc == HN_OK) {
            set_error("pack kernels failed");
            rc = HN_EHIP;
        }
    }
    stage("fp32 fragments (sizing pass, allocation, k_pack launches, sync)");
    if (rc == HN_OK && fp32_programs && hipStreamSynchronize(stream) != hipSuccess) {   // (f16x3: stream order is all the later launches need)
        set_error("copying the folded weights failed");
        rc = HN_EHIP;
    }
    stage("retained row-major matrices");
    if (rc == HN_OK && precision == HN_PREC_F16X3) rc = v2::build_v2_streams(f, sdf, col, w_sdf, w_col, stream, eval_only);
    pk.free_temps();
    stage("f16x3 programs");
    if (rc != HN_OK) {
        if (f->v2_full) pool_free(f->v2_full);
        if (f->v2_sdf) pool_free(f->v2_sdf);
        if (f->v2_adj) pool_free(f->v2_adj);
        if (f->v2_adjonly) pool_free(f->v2_adjonly);
        if (f->v2_tape) pool_free(f->v2_tape);
        if (f->blob) pool_free(f->blob);
        if (f->raw) pool_free(f->raw);
        delete f;
        return rc;
    }
    *out = f;
    return HN_OK;
}

}  // namespace hn
