// extern "C" surface of libhonerf (include/honerf.h) and the launch sequences of the two
// whole renders.  Nothing here allocates or synchronises (except field create/destroy).
#include <stdarg.h>
#include <atomic>
#include <functional>
#include <mutex>
#include <unordered_map>
#include <stdlib.h>
#include <string.h>

#include "hn_common.h"

namespace hn {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// implemented in the other translation units
int field_create(int kind, const hn_mlp_desc* sdf, const hn_mlp_desc* col, float variance, float scale, int precision,
                 hn_field** out, hipStream_t stream);
int ray_gen(const float*, const float*, const float*, const float*, const float*, int, int, float*, float*, hipStream_t);
int obj_local_fwd(const float*, const float*, const float*, const float*, int, int, float*, float*, hipStream_t, bool transposed = false);
int obj_local_bwd(const float*, const float*, const float*, const float*, const float*, const float*, int, int, float*,
                  float*, float*, float*, hipStream_t);
int coarse_z(const float*, int, int, float, float, float, float*, hipStream_t);
int sample_points(const float*, const float*, const float*, int, int, int, float, float*, float*, hipStream_t);
int sample_points_bwd(const float*, const float*, int, int, int, float, float*, float*, hipStream_t);
int upsample(const float*, const float*, int, int, int, float, float*, int64_t*, hipStream_t);
bool upsample_fused(const float*, const float*, int, int, int, float, float*, float*, int, int, const float*, const float*, float*, hipStream_t,
                    const UpsPre* pre = nullptr);
bool upsample_fused_ok(int n_rays, int k, int n_new);
int merge(const float*, const float*, const float*, const float*, int, int, int, int, float*, float*, int64_t*,
          hipStream_t);
int sort_rows(const float*, int, int, float*, hipStream_t);
int alpha(const float*, const float*, const float*, const float*, int, int, float, float*, float*, hipStream_t, const float* inv_s_dev = nullptr);
int composite1(const float*, const float*, const float*, const float*, int, int, float*, float*, float*, float*, float*,
               hipStream_t);
int composite2(const float*, const float*, const float*, const float*, const float*, const float*, int, int, float*,
               float*, float*, float*, float*, hipStream_t, float eik_scale = 1.f);
int alpha_bwd(const float*, const float*, const float*, const float*, const float*, const float*, int, int, float, float*,
              float*, float*, hipStream_t, bool g_rays_d_zeroed = false, const float* inv_s_dev = nullptr);
int alpha_inv_s_bwd(const float*, const float*, const float*, const float*, const float*, const float*, int, int, float, float*,
                    hipStream_t, const float* inv_s_dev = nullptr);
int composite1_bwd(const float*, const float*, const float*, const float*, const float*, int, int, float*, float*, float*,
                   hipStream_t);
int composite2_bwd(const float*, const float*, const float*, const float*, const float*, const float*, int, int, float*,
                   float*, float*, float*, hipStream_t);
int alpha_bwd_up(const float*, const float*, const float*, const float*, const float*, int, int, float, float, const float*, const float*, const float*,
                 float*, float*, float*, const int*, const int*, const float*, float*, float*, float*, float* const*, const size_t*, int, hipStream_t, float* g_rays_d_samples = nullptr,
                 const int* seg = nullptr);
int obj_rays_bwd(const float*, const float*, int, int, int, float, const float*, const float*, const float*, const float*, const float*, const float*,
                 float*, float*, float*, float*, hipStream_t, bool transposed = false, float* part = nullptr, unsigned* counter = nullptr,
                 const float* gd_alpha_samples = nullptr, const float* gd_colour_samples = nullptr);
int dual_prologue(const float*, const float*, const float*, const float*, int, int, float*, float*, const float*, int, float, float, float, float*, float*,
                  float*, int, hipStream_t, bool transposed = false);
int mat3_inverse(const float*, int, float*, hipStream_t);
int mat3_inverse_bwd(const float*, const float*, int, float*, hipStream_t);
int stable_pts(const float*, int, int, int, const float*, const float*, float*, float*, hipStream_t);
int stable_pts_bwd(const float*, int, int, int, const float*, float*, float*, hipStream_t);
size_t stable_value_scratch_bytes(int, int);
int stable_value(const float*, const float*, int, int, int, float*, float*, void*, size_t, hipStream_t);
size_t window_loss_scratch_bytes(int, int);
int window_loss(const float*, const float*, const float*, const float*, int, const float*, const float*, int, const float*, const float*, int, const float*,
                const float*, const float*, const float*, const float*, int, const float*, int, const float*, void*, size_t, float*, float*, float*, float*,
                float*, hipStream_t);
int window_loss_bwd(const float*, const float*, const float*, const float*, int, const float*, const float*, int, const float*, const float*, const float*,
                    const float*, const float*, const float*, int, float*, float*, float*, float*, float*, float*, float*, float*, hipStream_t);
size_t fit_step_loss_scratch_bytes(int, int);
int fit_step_loss(const float*, const float*, const float*, const float*, int, const float*, const float*, int, const float*, const float*, int, const float*,
                  const float*, const float*, const float*, const float*, int, const float*, void*, size_t, float*, float*, float*, float*, float*, hipStream_t);
int fit_step_loss_bwd(const float*, const float*, const float*, const float*, int, const float*, const float*, int, const float*, const float*, const float*,
                      const float*, const float*, const float*, int, float*, float*, float*, float*, float*, float*, float*, hipStream_t);
int fit_step_loss_frames(int, const float*, const float*, const float*, const float*, int, const float*, const float*, int, const float*, const float*, int,
                         const float*, const float*, const float*, const float*, const float* const*, const int*, const float*, void*, size_t, float*, float*,
                         float*, float*, float*, hipStream_t);
int fit_step_loss_bwd_frames(int, const float*, const float*, const float*, const float*, int, const float*, const float*, int, const float*, const float*,
                             const float*, const float*, const float*, const float*, int, float*, float*, float*, float*, float*, float*, float*, hipStream_t);
int train_loss(const float*, const float*, const float*, const float*, const float*, int, float, float, float*, hipStream_t);
int variance_to_inv_s(const float*, float*, hipStream_t);
int variance_chain(const float*, const float*, float*, hipStream_t);
int train_loss_bwd(const float*, const float*, const float*, const float*, int, const float*, const float*, float, float, float*, float*, float*, hipStream_t);
int fit_loss_sums(const float*, const float*, const float*, const float*, int, const float*, const float*, int, float*, hipStream_t);
int fit_total(const float*, const float*, const float*, const float*, int, const float*, float*, float*, hipStream_t);
int adam_step(int, float* const*, const float* const*, float* const*, float* const*, const int*, const float*, float, float, float, const int*, hipStream_t);
int fit_total_bwd(const float*, const float*, const float*, const float*, const float*, int, float*, float*, float*, float*, hipStream_t);
int fit_loss_grads(const float*, const float*, const float*, const float*, int, const float*, const float*, int, const float*,
                   const float*, float*, float*, float*, float*, hipStream_t);
size_t field_obj_workspace_bytes(int n_pts, int n_cus);
size_t field_hand_workspace_bytes(int n_pts, int n_cus);
int launch_field_obj(const hn_field*, const float*, const float*, int, int, float*, float*, float*, float*, void*, size_t,
                     bool, hipStream_t);
int launch_field_hand(const hn_field*, const float*, int, const float*, const float*, int, int, float*, float*, float*,
                      float*, void*, size_t, bool, hipStream_t);

namespace v2 {
size_t field2_obj_workspace_bytes(int n_pts, int n_cus);
int launch_field2_obj(const hn_field*, const float*, const float*, int, int, float*, float*, float*, float*, void*, size_t,
                      bool, hipStream_t, void* tape, size_t tape_bytes);
size_t field2_hand_workspace_bytes(int n_pts, int n_cus);
int launch_field2_hand(const hn_field*, const float*, int, const float*, const float*, int, int, float*, float*, float*,
                       float*, void*, size_t, bool, hipStream_t, void* tape, size_t tape_bytes);
size_t field2_obj_tape_bytes(int n_pts);
size_t field2_hand_tape_bytes(int n_pts);
int hand_dropped_samples(unsigned long long* n, bool reset);
}

constexpr int MAX_DEVICES = 64;

__global__ void k_scale(float* v, int n, float s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] *= s;
}
__global__ void k_copy_cols(const float* __restrict__ src, int n_rows, int n_src, float* __restrict__ dst, int n_dst,
                            int dst_off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * n_src) return;
    const int r = i / n_src, c = i % n_src;
    dst[(size_t)r * n_dst + dst_off + c] = src[i];
}

// g_sdf / g_grad of one field entering its adjoint: what alpha_bwd produced, plus the caller's direct gradients on the
// per-sample sdf / gradient outputs (the contact / penetration losses read sdf_*; fitting_single.py:268-281), plus the
// eikonal term gradient_error = mean((|g| - 1)^2) (d|g|/dg = 0 at g = 0, as torch's norm backward)
__global__ void k_upstream(float* __restrict__ gs, float* __restrict__ gg, const float* __restrict__ g_sdf_out,
                           const float* __restrict__ g_grad_out, const float* __restrict__ grad, const float* __restrict__ g_eik,
                           int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (g_sdf_out != nullptr) gs[i] += g_sdf_out[i];
    float a[3] = {0.f, 0.f, 0.f};
    if (g_grad_out != nullptr) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = g_grad_out[3 * (size_t)i + c];
    }
    if (g_eik != nullptr) {
        const float gx = grad[3 * (size_t)i], gy = grad[3 * (size_t)i + 1], gz = grad[3 * (size_t)i + 2];
        const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
        const float k = (2.f / (float)n) * g_eik[0] * (nrm - 1.f) / fmaxf(nrm, 1e-30f);
        a[0] += k * gx;
        a[1] += k * gy;
        a[2] += k * gz;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) gg[3 * (size_t)i + c] += a[c];
}
// out = a + b (+ c) (+ d)
__global__ void k_add4(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                       const float* __restrict__ d, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = a[i] + b[i];
    if (c != nullptr) v += c[i];
    if (d != nullptr) v += d[i];
    out[i] = v;
}

// two sums in one launch: out0 = a0 + b0, out1 = a1 + b1 + c1 + d1
__global__ void k_add4x2(const float* __restrict__ a0, const float* __restrict__ b0, float* __restrict__ out0, const float* __restrict__ a1,
                         const float* __restrict__ b1, const float* __restrict__ c1, const float* __restrict__ d1, float* __restrict__ out1,
                         int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out0[i] = a0[i] + b0[i];
    out1[i] = a1[i] + b1[i] + c1[i] + d1[i];
}
// up to four small buffers zeroed by one launch (a launch of a dependent chain costs ~5 us whatever it does)
struct ZeroList {
    float* p[4];
    int n[4];
};
__global__ void k_zero_many(ZeroList z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (z.p[b] != nullptr && i < z.n[b]) z.p[b][i] = 0.f;
}
static int zero_many(std::initializer_list<std::pair<float*, size_t>> bufs, hipStream_t s) {
    ZeroList z{};
    int k = 0, most = 0;
    for (const auto& b : bufs) {
        if (b.first == nullptr || b.second == 0) continue;
        z.p[k] = b.first;
        z.n[k] = (int)b.second;
        most = most > (int)b.second ? most : (int)b.second;
        ++k;
    }
    if (k == 0) return HN_OK;
    hipLaunchKernelGGL(k_zero_many, dim3((most + 255) / 256), dim3(256), 0, s, z);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// ---- a second stream per device: the hand and the object track of the two-field renders are independent until
// their results meet (the sorted depths; the compositing), and each of them alone leaves most CUs idle at the fitting
// sizes (294 sample tiles on 256 CUs = two rounds, the second 15 % full).  Fork / join are event waits on the device:
// nothing here blocks the host.  Created once per device on first use (the only allocation outside field_create).
struct SideStream {
    hipStream_t s2 = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, gate = nullptr;
};
static std::atomic<int> g_quad_max_blocks{[] {
    const char* e = getenv("HN_QUAD_MAX_BLOCKS");
    return e ? atoi(e) : -1;
}()};
int quad_max_blocks_override() { return g_quad_max_blocks.load(std::memory_order_relaxed); }
static std::atomic<int> g_fused_rounds{1};   // hn_debug_fused_rounds
std::atomic<int> g_pace_phantom{0};
int pace_phantom_members() { return g_pace_phantom.load(std::memory_order_relaxed); }
static SideStream g_side[MAX_DEVICES];
// The side stream and its two events are one set per device, shared by every caller stream: the host-side enqueue of a
// fork ... join section must not interleave with another host thread's (its join could otherwise wait on the other
// thread's record).  The lock covers the ENQUEUE only; the device-side order is the streams' own.
static std::mutex g_side_mutex[MAX_DEVICES];
struct SideLock {
    std::unique_lock<std::mutex> lk;
    explicit SideLock(SideStream* x) {
        if (x != nullptr) lk = std::unique_lock<std::mutex>(g_side_mutex[x - g_side]);
    }
};
static std::atomic<int> g_side_state[MAX_DEVICES];   // 0 unknown, 1 being created, 2 ready, 3 unavailable
static SideStream* side_stream() {
    const int dev = current_device();
    if (dev < 0 || dev >= MAX_DEVICES) return nullptr;
    int st = g_side_state[dev].load(std::memory_order_acquire);
    if (st == 0) {
        int expect = 0;
        if (g_side_state[dev].compare_exchange_strong(expect, 1)) {
            SideStream& x = g_side[dev];
            // the LOWEST priority the device offers: where workgroups of both streams are waiting for a CU, the caller's stream goes
            // first -- it carries the hand's branch, the longer one of a fitting step (its small launches must not queue behind the
            // object's 392-block coarse pass)
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
            const bool ok = hipStreamCreateWithPriority(&x.s2, hipStreamNonBlocking, least) == hipSuccess &&
                            hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess &&
                            hipEventCreateWithFlags(&x.join, hipEventDisableTiming) == hipSuccess &&
                            hipEventCreateWithFlags(&x.gate, hipEventDisableTiming) == hipSuccess;
            g_side_state[dev].store(ok ? 2 : 3, std::memory_order_release);
        }
        while ((st = g_side_state[dev].load(std::memory_order_acquire)) == 1) {
        }
    }
    return st == 2 ? &g_side[dev] : nullptr;
}
// s2 continues from where s is now / s continues once s2 has caught up
static int fork_to(SideStream* x, hipStream_t s) {
    HN_CHECK_HIP(hipEventRecord(x->fork, s));
    HN_CHECK_HIP(hipStreamWaitEvent(x->s2, x->fork, 0));
    return HN_OK;
}
// s2 goes on only when s has reached this point (a second meeting inside a fork ... join section)
static int gate_to(SideStream* x, hipStream_t s) {
    HN_CHECK_HIP(hipEventRecord(x->gate, s));
    HN_CHECK_HIP(hipStreamWaitEvent(x->s2, x->gate, 0));
    return HN_OK;
}
static int join_from(SideStream* x, hipStream_t s) {
    HN_CHECK_HIP(hipEventRecord(x->join, x->s2));
    HN_CHECK_HIP(hipStreamWaitEvent(s, x->join, 0));
    return HN_OK;
}

static thread_local const int* g_launch_n_pts_dev = nullptr;
const int* launch_n_pts_dev() { return g_launch_n_pts_dev; }
void set_launch_n_pts_dev(const int* p) { g_launch_n_pts_dev = p; }
// (set by the caller around an object adjoint launch: g_rays_d is [n, 3], one row per SAMPLE, stored -- the caller sums the rows of a
// ray in a fixed order -- instead of [n / spr, 3] accumulated with atomics)
static thread_local bool g_launch_dir_per_sample = false;
bool launch_dir_per_sample() { return g_launch_dir_per_sample; }
void set_launch_dir_per_sample(bool on) { g_launch_dir_per_sample = on; }
static thread_local const int* g_launch_orig_idx = nullptr;
const int* launch_orig_idx() { return g_launch_orig_idx; }
void set_launch_orig_idx(const int* p) { g_launch_orig_idx = p; }
static thread_local const int* g_launch_frame_seg = nullptr;
const int* launch_frame_seg() { return g_launch_frame_seg; }
void set_launch_frame_seg(const int* p) { g_launch_frame_seg = p; }

// ---- exact far-field skip of the hand field (SURVEY B-11, hn_field_set_compaction) ---------------------------------------
// A sample whose bone masks h_b = 1 - sigmoid(200 (v_b - cutoff_b)) (utils/fields.py:33-35) are ALL exactly 0 in fp32 sees
// an all-zero 1386-wide input: its sdf and colour are the same constants for every such sample, its gradient is exactly 0,
// and it contributes exactly 0 to every adjoint output.  Along the rays of a fitting step 40 - 60 % of the samples are of
// that kind, scattered through every 128-sample tile.  The two-field render therefore evaluates the hand field on the
// COMPACTED list of the other samples plus ONE far sample (whose outputs are the constants) and scatters the results back:
// 110 - 172 tiles instead of 294, i.e. one round of tiles on 256 CUs instead of two.  Results are bit-identical to the
// dense evaluation (no result depends on how samples are grouped into tiles: tested).  The classification is conservative:
// dead only if 200 (v_b - cutoff_b) > 20 for every bone, where the kernel's own sigmoid has been exactly 1 since ~16.7.
__constant__ float c_cutoff_api[21] = {0.08f, 0.03f, 0.03f, 0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f, 0.03f, 0.02f,
                                       0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f};
// Order-preserving compaction in two launches (classify + count per 256 samples; slots per COMPACT_SPB-sample block): the compact list keeps the dense order, so the
// samples of one frame stay together and a wave's 32 samples share their pose except at the frame boundaries).
// idx[k]: dense index of compact slot k; pos[i]: compact slot of sample i or -1.
constexpr int COMPACT_SPB = 2048;
// the classification of one sample: live unless every bone mask is certainly exactly 0
__device__ __forceinline__ bool hand_sample_live(const float* __restrict__ pts, int i, const float* __restrict__ bt_inv,
                                                 const float* __restrict__ T_pose, int n_frames, int pts_per_frame) {
    const float p0 = pts[3 * (size_t)i], p1 = pts[3 * (size_t)i + 1], p2 = pts[3 * (size_t)i + 2];
    int frame = i / pts_per_frame;
    frame = frame < n_frames ? frame : n_frames - 1;
    const float* M = bt_inv + (size_t)frame * 21 * 16;
    const float* T = T_pose + (size_t)frame * 21 * 3;
    bool live = false;
    for (int b = 0; b < 21; ++b) {
        const float* m = M + 16 * b;
        const float q0 = m[0] * p0 + m[1] * p1 + m[2] * p2 + m[3] - T[3 * b];
        const float q1 = m[4] * p0 + m[5] * p1 + m[6] * p2 + m[7] - T[3 * b + 1];
        const float q2 = m[8] * p0 + m[9] * p1 + m[10] * p2 + m[11] - T[3 * b + 2];
        const float v = sqrtf(q0 * q0 + q1 * q1 + q2 * q2);
        live = live || !(200.f * (v - c_cutoff_api[b]) > 20.f);   // (a NaN coordinate counts as live)
    }
    return live;
}
// pass 1 (one sample per thread): pos[i] = 1 (live) / 0, counts[u] = live samples of the 128-sample UNIT u (two per block)
__global__ __launch_bounds__(256) void k_hand_live_count(const float* __restrict__ pts, int n, const float* __restrict__ bt_inv,
                                                         const float* __restrict__ T_pose, int n_frames, int pts_per_frame,
                                                         int* __restrict__ pos, int* __restrict__ counts) {
    __shared__ int wave_cnt[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool live = false;
    if (i < n) {
        live = hand_sample_live(pts, i, bt_inv, T_pose, n_frames, pts_per_frame);
        pos[i] = live ? 1 : 0;
    }
    const int c = __popcll(__ballot(live));
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x < 2 && (int)(2 * blockIdx.x + threadIdx.x) * 128 < n) counts[2 * blockIdx.x + threadIdx.x] = wave_cnt[2 * threadIdx.x] + wave_cnt[2 * threadIdx.x + 1];
}
// pass 2: slots in dense order; the last block appends the far sample behind the M live ones and writes n_dev = M + 1, the
// sample count the field kernels read.  The far sample -- the stand-in whose outputs every skipped sample receives -- is
// the FIRST DEAD SAMPLE of the launch itself (classified by the predicate above, under its own frame's pose: idx[M] is its
// dense index, which is where the field kernels take a compact sample's frame from), so that it is dead by construction
// whatever the scene's scale or frame of reference; a launch without a dead sample has nobody to stand in for and gets a
// placeholder that nothing reads.
//
// FRAME-ALIGNED layout (`af` = the number of frames, > 1; pts_per_frame a multiple of 128; `seg` = the record's table): the live samples
// of frame f start at slot seg[f], a multiple of 128 -- every 128-sample tile and every wave of the field kernels then holds samples
// of ONE frame, and the tiles of a frame hold exactly the samples, in the lanes, that a launch of that frame ALONE would give them:
// what a frame's fit computes does not depend on which other frames share its launches (fitting.fit_frames_batched; the sums over a
// frame's samples are formed per tile and added per frame, k_pose_part_reduce).  The slots between a frame's last live sample and the
// next frame's start (at most 127) are PADS: copies of that last live sample's point with idx = -1 - (its dense index) -- evaluated
// like any sample (same frame: the wave stays uniform), never scattered back (no pos[] points at them), their upstream gradients
// zero (k_alpha_bwd_up / k_hand_gather_up), so that they add exact zeros.  The last frame has no pads: the far sample follows its live
// samples directly, as in a one-frame launch.  seg[0 .. af]: first slot per frame (seg[af]: the far sample's slot);
// seg[af + 1 + f]: live samples of frame f; n_dev[2] = af.  pts_per_frame a multiple of 128 bounds the slots by n + 1, as before.
constexpr int COMPACT_MAX_FRAMES = 64;
constexpr int COMPACT_UPB = COMPACT_SPB / 128;   // units per block of pass 2
__global__ __launch_bounds__(256) void k_hand_compact_write(const float* __restrict__ pts, int n, const int* __restrict__ counts,
                                                            int* __restrict__ idx, int* __restrict__ pos, float* __restrict__ pts_c,
                                                            int* __restrict__ n_dev, const float* __restrict__ bt_inv,
                                                            const float* __restrict__ T_pose, int n_frames, int pts_per_frame, int af,
                                                            int* __restrict__ seg) {
    __shared__ int red[4];
    __shared__ int wcnt[2][4];
    __shared__ int far_unit[4], far_lane[4];
    __shared__ int tot[COMPACT_MAX_FRAMES + 1], segs[COMPACT_MAX_FRAMES + 2];
    __shared__ int ubase[COMPACT_UPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the block behind the COMPACT_SPB-sample blocks appends the far sample (beside them, not after them: the launch is on the
    // critical path of a fitting step twice)
    const bool tail = blockIdx.x == gridDim.x - 1;
    const int n_units = (n + 127) / 128;
    const int upf = af > 0 ? pts_per_frame / 128 : 1;   // units per frame
    if (af > 0) {
        for (int f = threadIdx.x; f <= af; f += 256) tot[f] = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < n_units; t += 256) atomicAdd(&tot[t / upf], counts[t]);   // (integers: any order, the same sums)
        __syncthreads();
        if (threadIdx.x == 0) {
            int s = 0;
            for (int f = 0; f < af; ++f) {
                segs[f] = s;
                s += f + 1 < af ? ((tot[f] + 127) & ~127) : tot[f];
            }
            segs[af] = s;
        }
        __syncthreads();
    }
    // live samples in front of this block's first unit (frame-aligned: in front of it IN ITS FRAME); the tail block: in all units
    const int u0 = tail ? n_units : (int)blockIdx.x * COMPACT_UPB;
    const int t_first = (af > 0 && !tail) ? (u0 / upf) * upf : 0;
    int part = 0;
    int first_dead_unit = 0x7fffffff;   // (tail block) the first unit that holds a dead sample
    for (int t = t_first + threadIdx.x; t < u0; t += 256) {
        const int c = counts[t];
        part += c;
        if (tail) {
            const int valid = n - t * 128 < 128 ? n - t * 128 : 128;
            if (c < valid) first_dead_unit = first_dead_unit < t ? first_dead_unit : t;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        part += __shfl_xor(part, o, 64);
        const int other = __shfl_xor(first_dead_unit, o, 64);
        first_dead_unit = first_dead_unit < other ? first_dead_unit : other;
    }
    if (lane == 0) {
        red[wave] = part;
        far_unit[wave] = first_dead_unit;
    }
    __syncthreads();
    int run = red[0] + red[1] + red[2] + red[3];
    if (tail) {
        int far_i = -1;
        int b = far_unit[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) b = b < far_unit[w] ? b : far_unit[w];
        if (b != 0x7fffffff) {   // re-classify that unit's samples (pos[] is being rewritten by the other blocks): its first dead one
            const int i = b * 128 + threadIdx.x;
            const bool dead = threadIdx.x < 128 && i < n && !hand_sample_live(pts, i, bt_inv, T_pose, n_frames, pts_per_frame);
            const unsigned long long m = __ballot(dead);
            if (lane == 0) far_lane[wave] = m != 0ull ? __ffsll((long long)m) - 1 : -1;
            __syncthreads();
#pragma unroll
            for (int w = 1; w >= 0; --w)
                if (far_lane[w] >= 0) far_i = b * 128 + w * 64 + far_lane[w];
        }
        if (af > 0) {
            run = segs[af];
            for (int f = threadIdx.x; f <= af; f += 256) seg[f] = segs[f];
            for (int f = threadIdx.x; f < af; f += 256) seg[af + 1 + f] = tot[f];
        }
        if (threadIdx.x == 0) {
            idx[run] = far_i >= 0 ? far_i : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) pts_c[3 * (size_t)run + c] = far_i >= 0 ? pts[3 * (size_t)far_i + c] : 10.f;
            n_dev[0] = run + 1;
            n_dev[1] = far_i;   // dense index of the stand-in (-1: the launch has no dead sample)
            n_dev[2] = af;
        }
        return;
    }
    // first slot of every unit of this block
    if (threadIdx.x == 0) {
        int r = run + (af > 0 ? segs[u0 / upf] : 0);
        for (int k = 0; k < COMPACT_UPB; ++k) {
            const int u = u0 + k;
            if (u >= n_units) {
                ubase[k] = r;
                continue;
            }
            if (af > 0 && k > 0 && u % upf == 0) r = segs[u / upf];   // a new frame starts with this unit
            ubase[k] = r;
            r += counts[u];
        }
    }
    __syncthreads();
    for (int it = 0; it < COMPACT_SPB / 256; ++it) {
        const int i = blockIdx.x * COMPACT_SPB + it * 256 + threadIdx.x;
        const bool live = i < n && pos[i] != 0;
        const unsigned long long m = __ballot(live);
        if (lane == 0) wcnt[it & 1][wave] = __popcll(m);
        __syncthreads();   // (double-buffered: the next iteration's writes cannot pass this iteration's reads)
        if (i < n) {
            if (live) {
                const int unit = 2 * it + (wave >> 1);
                const int k = ubase[unit] + ((wave & 1) ? wcnt[it & 1][wave - 1] : 0) + __popcll(m & ((1ull << lane) - 1ull));
                idx[k] = i;
                pos[i] = k;
                const float p0 = pts[3 * (size_t)i], p1 = pts[3 * (size_t)i + 1], p2 = pts[3 * (size_t)i + 2];
                pts_c[3 * (size_t)k] = p0;
                pts_c[3 * (size_t)k + 1] = p1;
                pts_c[3 * (size_t)k + 2] = p2;
                if (af > 0) {   // the frame's last live sample fills the pads up to the next frame's first slot
                    const int f = (u0 + unit) / upf;
                    if (f + 1 < af && k == segs[f] + tot[f] - 1) {
                        for (int q = k + 1; q < segs[f + 1]; ++q) {
                            idx[q] = -1 - i;
                            pts_c[3 * (size_t)q] = p0;
                            pts_c[3 * (size_t)q + 1] = p1;
                            pts_c[3 * (size_t)q + 2] = p2;
                        }
                    }
                }
            } else {
                pos[i] = -1;
            }
        }
    }
}
// compact results -> the dense per-sample arrays (dead samples: the far sample's values)
__global__ void k_hand_scatter(const int* __restrict__ pos, int n, const int* __restrict__ n_dev, const float* __restrict__ sdf_c,
                               const float* __restrict__ grad_c, const float* __restrict__ rgb_c, float* __restrict__ sdf,
                               float* __restrict__ grad, float* __restrict__ rgb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = pos[i] >= 0 ? pos[i] : n_dev[0] - 1;
    sdf[i] = sdf_c[k];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        grad[3 * (size_t)i + c] = grad_c[3 * (size_t)k + c];
        rgb[3 * (size_t)i + c] = rgb_c[3 * (size_t)k + c];
    }
}
__global__ void k_hand_scatter_sdf(const int* __restrict__ pos, int n, const int* __restrict__ n_dev, const float* __restrict__ sdf_c,
                                   float* __restrict__ sdf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sdf[i] = sdf_c[pos[i] >= 0 ? pos[i] : n_dev[0] - 1];
}
// upstream gradients of the dense arrays -> compact (the far sample's are 0: it stands for samples that contribute nothing)
__global__ void k_hand_gather_up(const int* __restrict__ idx, int n_max, const int* __restrict__ n_dev, const float* __restrict__ gs,
                                 const float* __restrict__ gg, const float* __restrict__ gr, float* __restrict__ gs_c,
                                 float* __restrict__ gg_c, float* __restrict__ gr_c) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_max || k >= n_dev[0]) return;
    const int src = idx[k];
    const bool far = k == n_dev[0] - 1 || src < 0;   // (the stand-in; a pad of the frame-aligned layout: zero upstream gradients)
    const int i = far ? 0 : src;
    gs_c[k] = far ? 0.f : gs[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gg_c[3 * (size_t)k + c] = far ? 0.f : gg[3 * (size_t)i + c];
        gr_c[3 * (size_t)k + c] = far ? 0.f : gr[3 * (size_t)i + c];
    }
}
// Training (parameter gradients): upstream gradients of the dense arrays -> compact.  The far sample stands for ALL dead
// samples: they share its input (all-zero features), so their parameter-gradient contributions are (sum of their upstream
// d loss / d sdf) x d f(0) / d theta and likewise for the colour -- its upstream values are the SUMS over the dead samples.
// Their d loss / d gradient multiplies d gradient / d theta = 0 (the encoding's Jacobian is exactly 0 there): left 0.
__global__ void k_hand_gather_up_sum(const int* __restrict__ idx, const int* __restrict__ pos, int n, const int* __restrict__ n_dev,
                                     const float* __restrict__ gs, const float* __restrict__ gg, const float* __restrict__ gr,
                                     float* __restrict__ gs_c, float* __restrict__ gg_c, float* __restrict__ gr_c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = n_dev[0] - 1;
    // live slots k < M: a straight gather (thread i serves slot i)
    if (i < M) {
        const int src = idx[i];
        gs_c[i] = gs[src];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gg_c[3 * (size_t)i + c] = gg[3 * (size_t)src + c];
            gr_c[3 * (size_t)i + c] = gr[3 * (size_t)src + c];
        }
    }
    // dead samples: wave-reduced sums into slot M (zeroed by the caller)
    const bool dead = i < n && pos[i] < 0;
    float v[4] = {dead ? gs[i] : 0.f, dead ? gr[3 * (size_t)i] : 0.f, dead ? gr[3 * (size_t)i + 1] : 0.f, dead ? gr[3 * (size_t)i + 2] : 0.f};
    if (__ballot(dead) != 0ull) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_xor(v[c], off, 64);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(gs_c + M, v[0]);
            atomicAdd(gr_c + 3 * (size_t)M, v[1]);
            atomicAdd(gr_c + 3 * (size_t)M + 1, v[2]);
            atomicAdd(gr_c + 3 * (size_t)M + 2, v[3]);
        }
    }
}
__global__ void k_zero_slot(const int* __restrict__ n_dev, float* __restrict__ gs_c, float* __restrict__ gg_c, float* __restrict__ gr_c) {
    const int M = n_dev[0] - 1;
    if (threadIdx.x == 0) gs_c[M] = 0.f;
    if (threadIdx.x < 3) {
        gg_c[3 * (size_t)M + threadIdx.x] = 0.f;
        gr_c[3 * (size_t)M + threadIdx.x] = 0.f;
    }
}
// d loss / d pts of the compact list -> dense (dead samples: exactly 0)
__global__ void k_hand_scatter3(const int* __restrict__ pos, int n, const float* __restrict__ v_c, float* __restrict__ v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = pos[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[3 * (size_t)i + c] = k >= 0 ? v_c[3 * (size_t)k + c] : 0.f;
}
// samples the hand field's tape / workspace are sized for in the two-field renders (one more with compaction: the far sample)
static int hand_cap(const hn_field* hand, size_t N) { return (int)N + ((hand != nullptr && hand->compact_far_field) ? 1 : 0); }
// whether the two-field render of these sizes compacts the hand field's samples (the same answer in the forward pass, the
// backward pass and the size queries)
static bool hand_compaction(const hn_field* hand, int n_frames, size_t N) {
    return hand != nullptr && hand->compact_far_field && hand->kind == HN_FIELD_HAND && hand->precision == HN_PREC_F16X3 && n_frames >= 1 &&
           N >= 4096;
}
// the compaction record kept with the tape (the backward pass needs it): [n_dev, pad x3 | counts | idx N + 1 | pos N | pts_c | grad_c |
// rgb_c | sdf_c], the per-sample float arrays N + 1 long
struct CompactRec {
    int *n_dev, *counts, *idx, *pos, *seg;
    float *pts_c, *grad_c, *rgb_c, *sdf_c;
    static size_t n_counts(size_t N) { return ((N + 127) / 128 + 3) & ~size_t(3); }
    static size_t bytes(size_t N) {
        return 16 + (n_counts(N) + 2 * N + 4) * sizeof(int) + (N + 1) * 10 * sizeof(float) + 64 + (2 * COMPACT_MAX_FRAMES + 2) * sizeof(int);
    }
    void at(void* base, size_t N) {
        char* p = reinterpret_cast<char*>(base);
        n_dev = reinterpret_cast<int*>(p);
        counts = reinterpret_cast<int*>(p + 16);
        idx = counts + n_counts(N);
        pos = idx + N + 4;
        pts_c = reinterpret_cast<float*>(pos + N);
        grad_c = pts_c + 3 * (N + 1);
        rgb_c = grad_c + 3 * (N + 1);
        sdf_c = rgb_c + 3 * (N + 1);
        seg = reinterpret_cast<int*>(sdf_c + (N + 1)) + 16;   // the frame table of the frame-aligned layout (k_hand_compact_write)
    }
    // frames of the frame-aligned layout for these sizes (0: the plain layout)
    static int aligned_frames(int n, int n_frames, int pts_per_frame) {
        return (n_frames > 1 && n_frames <= COMPACT_MAX_FRAMES && pts_per_frame % 128 == 0 && (size_t)n_frames * pts_per_frame == (size_t)n) ? n_frames : 0;
    }
};
// the compact list of the hand's live samples of `pts` (dense order) + the far sample
static int compact_hand(CompactRec& cr, const float* pts, int n, const float* bt_inv, const float* T_pose, int n_frames, int pts_per_frame,
                        hipStream_t s) {
    const int nb = (n + COMPACT_SPB - 1) / COMPACT_SPB;
    hipLaunchKernelGGL(k_hand_live_count, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, bt_inv, T_pose, n_frames, pts_per_frame, cr.pos, cr.counts);
    hipLaunchKernelGGL(k_hand_compact_write, dim3(nb + 1), dim3(256), 0, s, pts, n, cr.counts, cr.idx, cr.pos, cr.pts_c, cr.n_dev, bt_inv, T_pose, n_frames,
                       pts_per_frame, CompactRec::aligned_frames(n, n_frames, pts_per_frame), cr.seg);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// bump allocator over the caller's workspace
struct Arena {
    char* base;
    size_t cap;
    size_t used = 0;
    bool ok = true;
    Arena(void* p, size_t n) : base(reinterpret_cast<char*>(p)), cap(n) {}
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        if (base == nullptr) {   // sizing pass
            used += bytes;
            return nullptr;
        }
        if (used + bytes > cap) {
            ok = false;
            return nullptr;
        }
        void* p = base + used;
        used += bytes;
        return p;
    }
    float* f(size_t n) { return reinterpret_cast<float*>(take(n * sizeof(float))); }
};

static std::atomic<int> g_cus[MAX_DEVICES];   // 0 = not queried yet
int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}
int device_cus() {
    const int dev = current_device();
    if (dev < 0) return 0;
    const int slot = dev % MAX_DEVICES;
    int v = g_cus[slot].load(std::memory_order_relaxed);
    if (v == 0 || dev >= MAX_DEVICES) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        v = n;
        if (dev < MAX_DEVICES) g_cus[slot].store(v, std::memory_order_relaxed);
    }
    return v;
}
int ensure_dynamic_lds(const void* kernel, int bytes, void* mask_atomic_u64) {
    auto* mask = reinterpret_cast<std::atomic<uint64_t>*>(mask_atomic_u64);
    const int dev = current_device();
    const uint64_t bit = (dev >= 0 && dev < MAX_DEVICES) ? (uint64_t(1) << dev) : 0;
    if (bit != 0 && (mask->load(std::memory_order_acquire) & bit)) return HN_OK;
    HN_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (bit != 0) mask->fetch_or(bit, std::memory_order_release);
    return HN_OK;
}

static size_t field_ws(const hn_field* f, int n_pts) {
    int cus = device_cus();
    if (cus <= 0) cus = 256;
    if (f->precision == HN_PREC_F16X3)
        return f->kind == HN_FIELD_OBJ ? v2::field2_obj_workspace_bytes(n_pts, cus) : v2::field2_hand_workspace_bytes(n_pts, cus);
    return f->kind == HN_FIELD_OBJ ? field_obj_workspace_bytes(n_pts, cus) : field_hand_workspace_bytes(n_pts, cus);
}

static int field_sdf(const hn_field* f, const float* pts, int n, const float* bt, const float* Tp, int n_frames, int ppf,
                     float* sdf, void* ws, size_t ws_bytes, hipStream_t s) {
    if (f->precision == HN_PREC_F16X3 && f->kind == HN_FIELD_OBJ)
        return v2::launch_field2_obj(f, pts, nullptr, n, 1, sdf, nullptr, nullptr, nullptr, ws, ws_bytes, false, s, nullptr, 0);
    if (f->precision == HN_PREC_F16X3)
        return v2::launch_field2_hand(f, pts, n, bt, Tp, n_frames, ppf, sdf, nullptr, nullptr, nullptr, ws, ws_bytes, false, s, nullptr, 0);
    if (f->kind == HN_FIELD_OBJ) return launch_field_obj(f, pts, nullptr, n, 1, sdf, nullptr, nullptr, nullptr, ws, ws_bytes, false, s);
    return launch_field_hand(f, pts, n, bt, Tp, n_frames, ppf, sdf, nullptr, nullptr, nullptr, ws, ws_bytes, false, s);
}
// bytes of the tape a taped evaluation of n_pts points keeps for its adjoint (0: this field keeps none)
static size_t field_tape(const hn_field* f, int n_pts) {
    if (f->precision != HN_PREC_F16X3 || f->v2_adjonly == nullptr) return 0;
    return f->kind == HN_FIELD_OBJ ? v2::field2_obj_tape_bytes(n_pts) : v2::field2_hand_tape_bytes(n_pts);
}
// hn_debug_field_timer: event pairs around the evaluation launches (a measurement aid; see include/honerf.h)
static std::atomic<int> g_field_timer{0};
static std::mutex g_field_timer_mutex;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_field_timer_pairs;
static int field_eval_launch(const hn_field* f, const float* pts, const float* rays_d, int n, int spr, const float* bt,
                             const float* Tp, int n_frames, int ppf, float* sdf, float* grad, float* rgb, float* feat, void* ws,
                             size_t ws_bytes, hipStream_t s, void* tape, size_t tape_bytes);
static int field_eval(const hn_field* f, const float* pts, const float* rays_d, int n, int spr, const float* bt,
                      const float* Tp, int n_frames, int ppf, float* sdf, float* grad, float* rgb, float* feat, void* ws,
                      size_t ws_bytes, hipStream_t s, void* tape = nullptr, size_t tape_bytes = 0) {
    if (g_field_timer.load(std::memory_order_relaxed) == 0 || n <= 0)
        return field_eval_launch(f, pts, rays_d, n, spr, bt, Tp, n_frames, ppf, sdf, grad, rgb, feat, ws, ws_bytes, s, tape, tape_bytes);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HN_CHECK_HIP(hipEventCreate(&e0));
    HN_CHECK_HIP(hipEventCreate(&e1));
    HN_CHECK_HIP(hipEventRecord(e0, s));
    const int rc = field_eval_launch(f, pts, rays_d, n, spr, bt, Tp, n_frames, ppf, sdf, grad, rgb, feat, ws, ws_bytes, s, tape, tape_bytes);
    HN_CHECK_HIP(hipEventRecord(e1, s));
    std::lock_guard<std::mutex> lk(g_field_timer_mutex);
    g_field_timer_pairs.emplace_back(e0, e1);
    return rc;
}
static int field_eval_launch(const hn_field* f, const float* pts, const float* rays_d, int n, int spr, const float* bt,
                             const float* Tp, int n_frames, int ppf, float* sdf, float* grad, float* rgb, float* feat, void* ws,
                             size_t ws_bytes, hipStream_t s, void* tape, size_t tape_bytes) {
    if (f->precision == HN_PREC_F16X3 && f->kind == HN_FIELD_OBJ)
        return v2::launch_field2_obj(f, pts, rays_d, n, spr, sdf, grad, rgb, feat, ws, ws_bytes, true, s, tape, tape_bytes);
    if (f->precision == HN_PREC_F16X3)
        return v2::launch_field2_hand(f, pts, n, bt, Tp, n_frames, ppf, sdf, grad, rgb, feat, ws, ws_bytes, true, s, tape, tape_bytes);
    if (f->kind == HN_FIELD_OBJ) return launch_field_obj(f, pts, rays_d, n, spr, sdf, grad, rgb, feat, ws, ws_bytes, true, s);
    return launch_field_hand(f, pts, n, bt, Tp, n_frames, ppf, sdf, grad, rgb, feat, ws, ws_bytes, true, s);
}

#define HN_TRY(expr)             \
    do {                         \
        int _rc = (expr);        \
        if (_rc != HN_OK) return _rc; \
    } while (0)

// One importance-sampling track (utils/renderer.py:214-234): coarse sdf, then up_sample_steps x
// {up_sample, cat_z_vals}.  On return z_cur [n_rays, n_samples + n_importance] is sorted.
// new_z_all (optional) collects every step's new depths, [n_rays, cat_stride] at column cat_off + 2*i*n_new
struct Track {
    float *z_a, *z_b, *sdf_a, *sdf_b, *z_new, *z_new2, *sdf_new, *pts;
};
static void track_alloc(Arena& ar, Track& t, size_t n_rays, int S, int n_new) {
    t.z_a = ar.f(n_rays * S);
    t.z_b = ar.f(n_rays * S);
    t.sdf_a = ar.f(n_rays * S);
    t.sdf_b = ar.f(n_rays * S);
    t.z_new = ar.f(n_rays * n_new);
    t.z_new2 = ar.f(n_rays * n_new);
    t.sdf_new = ar.f(n_rays * n_new);
    t.pts = ar.f(n_rays * S * 3);
}

namespace bwd {
bool param_path_is_fused(const hn_field* f);   // hn_field_bwd.hip
}
// What a training render keeps for its backward pass (hn_render_single_taped -> hn_render_single_bwd_taped): the tape of
// the final evaluation of the n_rays x S samples (the compacted list's rows for a hand field with the far-field aggregation) and that
// evaluation's outputs.  One caller-owned block: [tape | sdf | grad | rgb | feature vector], `rows` rows.
struct SingleKeep {
    void* tape;
    size_t tape_bytes;
    float *sdf, *grad, *rgb, *feat;
    size_t total;
    SingleKeep(const hn_field* f, size_t N, void* base) {
        const size_t rows = (size_t)hand_cap(f->kind == HN_FIELD_HAND ? f : nullptr, N);
        tape_bytes = (field_tape(f, (int)rows) + 255) & ~size_t(255);
        char* p = reinterpret_cast<char*>(base);
        size_t off = 0;
        auto take = [&](size_t bytes) {
            void* q = p ? p + off : nullptr;
            off += (bytes + 255) & ~size_t(255);
            return q;
        };
        tape = take(tape_bytes);
        sdf = reinterpret_cast<float*>(take(rows * sizeof(float)));
        grad = reinterpret_cast<float*>(take(rows * 3 * sizeof(float)));
        rgb = reinterpret_cast<float*>(take(rows * 3 * sizeof(float)));
        feat = reinterpret_cast<float*>(take(rows * H * sizeof(float)));
        total = off;
    }
};
// The live count of a kept block's compacted list, on its way to the host while the render is still running: hn_render_single_taped
// starts the copy right behind its compaction and records an event; hn_render_single_bwd_taped -- whose launches are sized by that
// count -- waits for THAT event only (long past when the backward pass is called) instead of for the whole stream.  Keyed by the block.
struct KeptCount {
    int* host = nullptr;      // pinned
    hipEvent_t ev = nullptr;
};
static std::mutex g_kept_mu;
static std::unordered_map<const void*, KeptCount> g_kept;
static int kept_count_send(const void* block, const int* n_dev, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_kept_mu);
    // (blocks come and go with the caller's allocator: beyond 256 entries the ones whose copy has landed long ago are dropped --
    // a backward pass that still wanted one reads its count back itself)
    if (g_kept.size() >= 256 && g_kept.find(block) == g_kept.end()) {
        for (auto it = g_kept.begin(); it != g_kept.end();) {
            if (hipEventQuery(it->second.ev) == hipSuccess) {
                (void)hipEventDestroy(it->second.ev);
                (void)hipHostFree(it->second.host);
                it = g_kept.erase(it);
            } else
                ++it;
        }
    }
    KeptCount& e = g_kept[block];
    if (e.host == nullptr) {
        HN_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&e.host), sizeof(int), hipHostMallocDefault));
        HN_CHECK_HIP(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));
    }
    *e.host = -1;
    HN_CHECK_HIP(hipMemcpyAsync(e.host, n_dev, sizeof(int), hipMemcpyDeviceToHost, s));
    HN_CHECK_HIP(hipEventRecord(e.ev, s));
    return HN_OK;
}
static void kept_counts_release() {
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_kept_mu);
    for (auto& kv : g_kept) {
        if (kv.second.ev != nullptr) (void)hipEventDestroy(kv.second.ev);
        if (kv.second.host != nullptr) (void)hipHostFree(kv.second.host);
    }
    g_kept.clear();
}
static bool kept_count_take(const void* block, int* n_c) {   // false: no count was sent for this block
    std::lock_guard<std::mutex> lk(g_kept_mu);   // (held over the wait: the entry must not be dropped under it; the copy landed long ago)
    auto it = g_kept.find(block);
    if (it == g_kept.end()) return false;
    const KeptCount& e = it->second;
    if (hipEventSynchronize(e.ev) != hipSuccess || *e.host < 0) return false;
    *n_c = *e.host;
    return true;
}
// a render whose backward pass takes the fused parameter-gradient path (f16x3 field packed with its tape programs) can keep its tape
static bool single_keep_applies(const hn_field* f) { return f != nullptr && bwd::param_path_is_fused(f) && field_tape(f, 128) != 0; }

static int render_single_impl(const hn_field* f, const float* rays_o, const float* rays_d, const float* t_rand,
                              int n_rays, double near, double far, int n_samples, int n_importance, int steps,
                              const float* bt_inv, const float* T_pose, float* color, float* cdf, float* weight_sum,
                              float* weight_max, float* gradient_error, float* z_vals, void* workspace,
                              size_t workspace_bytes, hipStream_t s, size_t* need, void* keep_block = nullptr, size_t keep_bytes = 0) {
    HN_REQUIRE(n_samples >= 2 && n_importance >= 0 && n_rays >= 0, "bad sample counts");
    HN_REQUIRE(n_importance == 0 || (steps >= 1 && n_importance % steps == 0), "n_importance must divide into steps");
    const int S = n_samples + n_importance;
    HN_REQUIRE(S <= 640, "at most 640 samples per ray (hn_upsample's cdf rows)");
    const int n_new = n_importance > 0 ? n_importance / steps : 0;
    const size_t N = (size_t)n_rays * S;
    Arena ar(workspace, workspace_bytes);
    Track t;
    track_alloc(ar, t, n_rays, S, n_new > 0 ? n_new : 1);
    float* dists = ar.f(N);
    float* sdf = ar.f(N);
    float* grad = ar.f(N * 3);
    float* rgb = ar.f(N * 3);
    float* al = ar.f(N);
    // hand field with hn_field_set_compaction: every evaluation runs on the compacted list of the samples with a live bone mask
    // (+ one far sample), bit-identical results (the two-field renders do the same: see the kernels above)
    const bool may_compact = f->kind == HN_FIELD_HAND && f->compact_far_field;
    const size_t fws_bytes = field_ws(f, may_compact ? hand_cap(f, N) : (int)N);
    void* fws = ar.take(fws_bytes);
    void* crec_ws = may_compact ? ar.take(CompactRec::bytes(N)) : nullptr;
    if (need != nullptr) {
        *need = ar.used;
        return HN_OK;
    }
    if (!ar.ok) {
        set_error("render_single workspace too small: %zu bytes given", workspace_bytes);
        return HN_ENOMEM;
    }
    if (n_rays == 0) return HN_OK;
    const float sample_dist = (float)((far - near) / (double)n_samples);
    const int hand_ppf = n_rays * S;   // single field: one frame
    // sdf at n points: dense, or through the compacted list
    // (a pass of at most one 32-sample block per CU runs in the latency form, whose launch takes the same ~85 us for 1 block as for 256:
    //  compacting it -- three more launches, ~22 us -- cannot shorten it; the results are the same bits either way)
    const int cus_now = device_cus() > 0 ? device_cus() : 256;
    auto sdf_pass = [&](const float* p, int n, float* out) -> int {
        if (!(may_compact && hand_compaction(f, 1, (size_t)n) && (size_t)n > (size_t)32 * (size_t)cus_now))
            return field_sdf(f, p, n, bt_inv, T_pose, 1, n, out, fws, fws_bytes, s);
        CompactRec cr;
        cr.at(crec_ws, (size_t)n);
        HN_TRY(compact_hand(cr, p, n, bt_inv, T_pose, 1, n, s));
        set_launch_n_pts_dev(cr.n_dev);
        set_launch_orig_idx(cr.idx);
        const int rc = field_sdf(f, cr.pts_c, n + 1, bt_inv, T_pose, 1, n, cr.sdf_c, fws, fws_bytes, s);
        set_launch_n_pts_dev(nullptr);
        set_launch_orig_idx(nullptr);
        HN_TRY(rc);
        hipLaunchKernelGGL(k_hand_scatter_sdf, dim3((n + 255) / 256), dim3(256), 0, s, cr.pos, n, cr.n_dev, cr.sdf_c, out);
        HN_LAUNCH_CHECK();
        return HN_OK;
    };
    HN_TRY(coarse_z(t_rand, n_rays, n_samples, (float)near, (float)(far - near), sample_dist, t.z_a, s));
    float* z_cur = t.z_a;
    if (n_importance > 0) {
        int k = n_samples;
        HN_TRY(sample_points(rays_o, rays_d, t.z_a, n_rays, k, 0, 0.f, t.pts, nullptr, s));
        HN_TRY(sdf_pass(t.pts, n_rays * k, t.sdf_a));
        float *za = t.z_a, *zb = t.z_b, *sa = t.sdf_a, *sb = t.sdf_b;
        // A round is up_sample -> sdf of the new depths -> cat_z_vals.  For the batch sizes of a training iteration the wave form of up_sample
        // takes the round's other small launches along, as in the two-field render (hn_debug_fused_rounds: off): the previous round's
        // cat_z_vals runs at the head of this round's launch (`pending`), the new depths' sample positions at its end -- one launch per round
        // beside the field's instead of three; the same operations in the same order.
        const bool fused_rounds = g_fused_rounds.load(std::memory_order_relaxed) != 0;
        float *zn = t.z_new, *zn2 = t.z_new2;
        bool pending = false;
        for (int i = 0; i < steps; ++i) {
            const bool more = i + 1 < steps;
            UpsPre pre{};
            const UpsPre* pp = nullptr;
            float* z_out = zn;
            if (pending) {   // (k counts the merged row already; this round's depths go to the other small buffer)
                pre = UpsPre{zn, t.sdf_new, n_new, 0, zb, sb, nullptr, nullptr, nullptr};
                pp = &pre;
                z_out = zn2;
            }
            if (!fused_rounds || !upsample_fused(za, sa, n_rays, k, n_new, (float)(64 << i), z_out, nullptr, 0, 0, rays_o, rays_d, more ? t.pts : nullptr, s, pp)) {
                HN_REQUIRE(pp == nullptr, "render_single: the fused up_sample launch failed");
                HN_TRY(upsample(za, sa, n_rays, k, n_new, (float)(64 << i), zn, nullptr, s));
                if (more) HN_TRY(sample_points(rays_o, rays_d, zn, n_rays, n_new, 0, 0.f, t.pts, nullptr, s));
            }
            if (pending) {
                float* tmp = za; za = zb; zb = tmp;
                tmp = sa; sa = sb; sb = tmp;
                tmp = zn; zn = zn2; zn2 = tmp;
                pending = false;
            }
            if (more) {
                HN_TRY(sdf_pass(t.pts, n_rays * n_new, t.sdf_new));
                if (fused_rounds && upsample_fused_ok(n_rays, k + n_new, n_new)) {
                    pending = true;
                } else {
                    HN_TRY(merge(za, zn, sa, t.sdf_new, n_rays, k, n_new, 0, zb, sb, nullptr, s));
                    float* tmp = za; za = zb; zb = tmp;
                    tmp = sa; sa = sb; sb = tmp;
                }
            } else {
                HN_TRY(merge(za, zn, nullptr, nullptr, n_rays, k, n_new, 0, zb, nullptr, nullptr, s));
                float* tmp = za; za = zb; zb = tmp;
            }
            k += n_new;
        }
        z_cur = za;
    }
    HN_TRY(sample_points(rays_o, rays_d, z_cur, n_rays, S, 1, sample_dist, t.pts, dists, s));
    // keep_block (hn_render_single_taped): the final evaluation keeps its tape and leaves its outputs there, row for row what the
    // backward pass's own taped evaluation of the same points would produce (the same kernel on the same list)
    SingleKeep kp(f, N, keep_block);
    if (keep_block != nullptr) {
        HN_REQUIRE(single_keep_applies(f), "hn_render_single_taped: the field's backward pass does not take a tape (hn_render_single_tape_bytes is 0)");
        HN_REQUIRE(keep_bytes >= kp.total, "hn_render_single_taped: tape block too small (%zu < %zu)", keep_bytes, kp.total);
    }
    if (may_compact && hand_compaction(f, 1, N)) {
        CompactRec cr;
        cr.at(crec_ws, N);
        HN_TRY(compact_hand(cr, t.pts, (int)N, bt_inv, T_pose, 1, hand_ppf, s));
        if (keep_block != nullptr) HN_TRY(kept_count_send(keep_block, cr.n_dev, s));
        float *sdf_c = keep_block ? kp.sdf : cr.sdf_c, *grad_c = keep_block ? kp.grad : cr.grad_c, *rgb_c = keep_block ? kp.rgb : cr.rgb_c;
        set_launch_n_pts_dev(cr.n_dev);
        set_launch_orig_idx(cr.idx);
        const int rc = field_eval(f, cr.pts_c, rays_d, (int)N + 1, S, bt_inv, T_pose, 1, hand_ppf, sdf_c, grad_c, rgb_c, keep_block ? kp.feat : nullptr, fws,
                                  fws_bytes, s, keep_block ? kp.tape : nullptr, keep_block ? kp.tape_bytes : 0);
        set_launch_n_pts_dev(nullptr);
        set_launch_orig_idx(nullptr);
        HN_TRY(rc);
        hipLaunchKernelGGL(k_hand_scatter, dim3(((int)N + 255) / 256), dim3(256), 0, s, cr.pos, (int)N, cr.n_dev, sdf_c, grad_c, rgb_c, sdf, grad, rgb);
        HN_LAUNCH_CHECK();
    } else if (keep_block != nullptr) {
        HN_TRY(field_eval(f, t.pts, rays_d, (int)N, S, bt_inv, T_pose, 1, hand_ppf, kp.sdf, kp.grad, kp.rgb, kp.feat, fws, fws_bytes, s, kp.tape, kp.tape_bytes));
        sdf = kp.sdf;
        grad = kp.grad;
        rgb = kp.rgb;
    } else {
        HN_TRY(field_eval(f, t.pts, rays_d, (int)N, S, bt_inv, T_pose, 1, hand_ppf, sdf, grad, rgb, nullptr, fws, fws_bytes, s));
    }
    HN_TRY(alpha(sdf, grad, rays_d, dists, (int)N, S, f->inv_s, al, cdf, s, f->inv_s_dev));
    HN_CHECK_HIP(hipMemsetAsync(gradient_error, 0, sizeof(float), s));
    HN_TRY(composite1(al, cdf, rgb, grad, n_rays, S, color, nullptr, weight_sum, weight_max, gradient_error, s));
    hipLaunchKernelGGL(k_scale, dim3(1), dim3(64), 0, s, gradient_error, 1, 1.f / (float)N);
    HN_LAUNCH_CHECK();
    if (z_vals != nullptr) HN_CHECK_HIP(hipMemcpyAsync(z_vals, z_cur, N * sizeof(float), hipMemcpyDeviceToDevice, s));
    return HN_OK;
}

static int render_dual_impl(const hn_field* hand, const hn_field* obj, const float* rays_o, const float* rays_d,
                            const float* t_rand, int n_frames, int rpf, double near, double far, int n_samples,
                            int n_importance, int steps, const float* bt_inv, const float* T_pose, const float* Ro,
                            const float* To, int batch_quirk, float* color, float* weight_sum, float* sdf_hand,
                            float* sdf_obj, float* grad_hand, float* grad_obj, float* gradient_error, float* z_vals,
                            void* workspace, size_t workspace_bytes, hipStream_t s, size_t* need, size_t* aux_offsets = nullptr,
                            void* tape = nullptr, size_t tape_bytes = 0, size_t* compact_offsets = nullptr, int flags = 0) {
    const bool ro_t = (flags & HN_DUAL_RO_TRANSPOSED) != 0;        // Ro holds obj_r: its transpose is the rotation to apply
    const bool obj_side = (flags & HN_DUAL_OBJ_POSE_ON_SIDE) != 0;  // Ro / To are produced on the side stream (pipelined fitting step)
    HN_REQUIRE(n_samples >= 2 && n_importance >= 0 && n_frames >= 1 && rpf >= 0, "bad sizes");
    HN_REQUIRE(n_importance == 0 || (steps >= 1 && n_importance % steps == 0), "n_importance must divide into steps");
    HN_REQUIRE(hand->kind == HN_FIELD_HAND && obj->kind == HN_FIELD_OBJ, "field kinds (hand, obj) expected");
    HN_REQUIRE(hand->inv_s_dev == nullptr && obj->inv_s_dev == nullptr, "a field whose inv_s lives on the device (hn_field_set_inv_s_device) serves the single-field renders only");
    const int n_rays = n_frames * rpf;
    const int S = n_samples + 2 * n_importance;
    const int St = n_samples + n_importance;   // per-track length
    HN_REQUIRE(St <= 640 && S <= 1024, "at most 640 depths per track and 1 024 per ray (hn_upsample's cdf rows, hn_sort_rows)");
    const int n_new = n_importance > 0 ? n_importance / steps : 0;
    const size_t N = (size_t)n_rays * S;
    Arena ar(workspace, workspace_bytes);
    float* o_obj = ar.f((size_t)n_rays * 3);
    float* d_obj = ar.f((size_t)n_rays * 3);
    Track th, to;
    track_alloc(ar, th, n_rays, St, n_new > 0 ? n_new : 1);
    track_alloc(ar, to, n_rays, St, n_new > 0 ? n_new : 1);
    float* zcat = ar.f(N);
    float* z = ar.f(N);
    float* pts = ar.f(N * 3);
    float* dists = ar.f(N);
    float* pts_o = ar.f(N * 3);      // the object track runs on its own stream: its own point / dist / field buffers
    float* dists_o = ar.f(N);
    const size_t off_rgb_h = ar.used;
    float* rgb_h = ar.f(N * 3);
    const size_t off_rgb_o = ar.used;
    float* rgb_o = ar.f(N * 3);
    const size_t off_al_h = ar.used;
    float* al_h = ar.f(N);
    const size_t off_al_o = ar.used;
    float* al_o = ar.f(N);
    const size_t fws_h = field_ws(hand, hand_cap(hand, N)), fws_o = field_ws(obj, (int)N);
    void* fwsh = ar.take(fws_h);
    void* fwso = ar.take(fws_o);
    const size_t crec_off = ar.used;
    void* crec_ws = hand->compact_far_field ? ar.take(CompactRec::bytes(N)) : nullptr;   // (without a tape the record lives here)
    // With a tape the four arrays live in the TAPE (behind the two fields' tapes): the caller keeps that buffer until the
    // backward pass, so nothing has to be copied out of the workspace (4 copy launches per fitting step).
    const size_t tapes_bytes = field_tape(hand, hand_cap(hand, N)) + field_tape(obj, (int)N);
    if (tape != nullptr && tapes_bytes != 0 && tape_bytes >= tapes_bytes + 8 * N * sizeof(float)) {
        float* aux = reinterpret_cast<float*>(reinterpret_cast<char*>(tape) + tapes_bytes);
        rgb_h = aux;
        rgb_o = aux + 3 * N;
        al_h = aux + 6 * N;
        al_o = aux + 7 * N;
    }
    if (compact_offsets != nullptr) {   // where the compaction records' device-side counts live (hn_render_dual_compact_offsets)
        const bool fin = hand_compaction(hand, n_frames, N) && tapes_bytes != 0;
        const bool coarse = n_importance > 0 && hand_compaction(hand, n_frames, (size_t)n_rays * n_samples);
        compact_offsets[0] = fin ? tapes_bytes + 8 * N * sizeof(float) : (size_t)-1;
        compact_offsets[1] = coarse ? crec_off : (size_t)-1;
    }
    if (aux_offsets != nullptr) {   // where the final evaluation leaves rgb / alpha of both fields (bytes into the workspace)
        aux_offsets[0] = off_rgb_h;
        aux_offsets[1] = off_rgb_o;
        aux_offsets[2] = off_al_h;
        aux_offsets[3] = off_al_o;
    }
    if (need != nullptr) {
        *need = ar.used;
        return HN_OK;
    }
    if (!ar.ok) {
        set_error("render_dual workspace too small: %zu bytes given", workspace_bytes);
        return HN_ENOMEM;
    }
    if (n_rays == 0) return HN_OK;
    const float sample_dist = (float)((far - near) / (double)n_samples);
    const int quirk = (batch_quirk && n_frames > 1) ? rpf : 0;
    SideStream* side = side_stream();
    SideLock side_lock(side);
    const hipStream_t so = side != nullptr ? side->s2 : s;   // the object track's stream (s itself if no second stream)
    // object-local rays, the shared coarse depths of both tracks, the first columns of the concatenated depth list: one launch
    HN_REQUIRE(!obj_side || side != nullptr, "HN_DUAL_OBJ_POSE_ON_SIDE needs the device's second stream");
    HN_TRY(dual_prologue(rays_o, rays_d, obj_side ? nullptr : Ro, To, n_frames, rpf, o_obj, d_obj, t_rand, n_samples, (float)near, (float)(far - near),
                         sample_dist, th.z_a, n_importance > 0 ? to.z_a : nullptr, zcat, S, s, ro_t));
    const float* z_final = zcat;
    bool obj_rays_made = !obj_side;
    if (n_importance > 0) {
        if (side != nullptr) HN_TRY(fork_to(side, s));
        if (obj_side) {   // the object's pose comes from the side stream: its local rays are made there, behind whatever produced Ro / To
            HN_TRY(obj_local_fwd(rays_o, rays_d, Ro, To, n_frames, rpf, o_obj, d_obj, so, ro_t));
            obj_rays_made = true;
        }
        // the two importance-sampling tracks (utils/renderer.py:463-496) are independent: hand on s, object on so
        // The two tracks are queued STAGE BY STAGE (coarse pass, then each importance round), the object's launches of a
        // stage first: queued track by track, the second track's first kernel reached its stream only after the ~25
        // launches of the first had been issued by the host (~120 us of a 650 us phase); and the object's launches must not
        // wait behind the hand's coarse launch, which takes every CU it can get for its 392 blocks.
        struct TrackRun {
            Track* t;
            const hn_field* f;
            const float *ro, *rd;
            hipStream_t st;
            void* fws;
            size_t fwb;
            int nf, which, k;
        };
        TrackRun runs[2] = {{&to, obj, o_obj, d_obj, so, fwso, fws_o, 1, 1, n_samples}, {&th, hand, rays_o, rays_d, s, fwsh, fws_h, n_frames, 0, n_samples}};
        // Coarse pass.  With the far-field skip the hand's coarse launch is ~120 live blocks of the latency form behind three small
        // kernels (points, classify, compact); the object's 392 blocks, released first, take every CU (one 512-register workgroup
        // each) and the hand's small kernels wait a whole block time for a wave slot -- and the hand's track is the longer one (its
        // fine rounds cost twice the object's).  So: the hand's small kernels first, the object's track held back (gate) until they
        // are done, then the hand's field launch queued ahead of the object's.  Without the skip (392 dense hand blocks): the
        // object's launches first, as before.
        const int nc = n_rays * n_samples;
        const bool coarse_compact = crec_ws != nullptr && hand_compaction(hand, n_frames, (size_t)nc);
        auto obj_coarse = [&]() -> int {
            HN_TRY(sample_points(o_obj, d_obj, to.z_a, n_rays, n_samples, 0, 0.f, to.pts, nullptr, so));
            return field_sdf(obj, to.pts, nc, bt_inv, T_pose, 1, nc, to.sdf_a, fwso, fws_o, so);
        };
        HN_TRY(sample_points(rays_o, rays_d, th.z_a, n_rays, n_samples, 0, 0.f, th.pts, nullptr, s));
        UpsPre coarse_gather{};
        const bool fused_rounds = g_fused_rounds.load(std::memory_order_relaxed) != 0;
        if (coarse_compact) {
            // the hand's coarse pass on the samples with a live bone (the record of the final evaluation is written later)
            CompactRec cr;
            cr.at(crec_ws, (size_t)nc);
            HN_TRY(compact_hand(cr, th.pts, nc, bt_inv, T_pose, n_frames, rpf * n_samples, s));
            if (side != nullptr) HN_TRY(gate_to(side, s));
            set_launch_n_pts_dev(cr.n_dev);
            set_launch_orig_idx(cr.idx);
            const int rc = field_sdf(hand, cr.pts_c, nc + 1, bt_inv, T_pose, n_frames, rpf * n_samples, cr.sdf_c, fwsh, fws_h, s);
            set_launch_n_pts_dev(nullptr);
            set_launch_orig_idx(nullptr);
            HN_TRY(rc);
            HN_TRY(obj_coarse());
            if (steps >= 1 && fused_rounds && upsample_fused_ok(n_rays, n_samples, n_new)) {
                coarse_gather = UpsPre{nullptr, nullptr, 0, 0, nullptr, th.sdf_a, cr.pos, cr.n_dev, cr.sdf_c};   // the first up_sample launch reads through the record
            } else {
                hipLaunchKernelGGL(k_hand_scatter_sdf, dim3((nc + 255) / 256), dim3(256), 0, s, cr.pos, nc, cr.n_dev, cr.sdf_c, th.sdf_a);
                HN_LAUNCH_CHECK();
            }
        } else {
            HN_TRY(obj_coarse());
            HN_TRY(field_sdf(hand, th.pts, nc, bt_inv, T_pose, n_frames, rpf * n_samples, th.sdf_a, fwsh, fws_h, s));
        }
        // A round is up_sample -> sdf of the new depths -> cat_z_vals; in the wave form of up_sample (the fitting loops' batch sizes)
        // a round's cat_z_vals runs at the head of the NEXT round's up_sample launch (UpsPre: `pending`), and the hand's coarse sdf row is
        // read through the compaction record by the first one: one launch per round and track beside the field's instead of two / three.
        bool pending[2] = {false, false};
        for (int i = 0; i < steps; ++i) {
            for (TrackRun& r : runs) {
                Track& t = *r.t;
                // up_sample, the new depths' columns of the concatenated list and (unless this is the last round) the new sample
                // positions: one launch for the fitting loops' batch sizes, three otherwise
                const int col = n_samples + (2 * i + r.which) * n_new;
                const bool more = i + 1 < steps;
                UpsPre pre{};
                const UpsPre* pp = nullptr;
                float* z_out = t.z_new;
                if (pending[r.which]) {
                    // (r.k counts the merged row already; the new depths of this round go to the other small buffer)
                    pre = UpsPre{t.z_new, t.sdf_new, n_new, quirk, t.z_b, t.sdf_b, nullptr, nullptr, nullptr};
                    pp = &pre;
                    z_out = t.z_new2;
                } else if (i == 0 && r.which == 0 && coarse_gather.pos != nullptr) {
                    pp = &coarse_gather;
                }
                if (!fused_rounds ||
                    !upsample_fused(t.z_a, t.sdf_a, n_rays, r.k, n_new, (float)(64 << i), z_out, zcat, S, col, r.ro, r.rd, more ? t.pts : nullptr, r.st, pp)) {
                    HN_REQUIRE(pp == nullptr, "render_dual: the fused up_sample launch failed");
                    HN_TRY(upsample(t.z_a, t.sdf_a, n_rays, r.k, n_new, (float)(64 << i), t.z_new, nullptr, r.st));
                    hipLaunchKernelGGL(k_copy_cols, dim3((n_rays * n_new + 255) / 256), dim3(256), 0, r.st, t.z_new, n_rays, n_new, zcat, S, col);
                    HN_LAUNCH_CHECK();
                    if (more) HN_TRY(sample_points(r.ro, r.rd, t.z_new, n_rays, n_new, 0, 0.f, t.pts, nullptr, r.st));
                }
                if (pending[r.which]) {
                    float* tmp = t.z_a; t.z_a = t.z_b; t.z_b = tmp;
                    tmp = t.sdf_a; t.sdf_a = t.sdf_b; t.sdf_b = tmp;
                    tmp = t.z_new; t.z_new = t.z_new2; t.z_new2 = tmp;
                    pending[r.which] = false;
                }
                if (i + 1 < steps) {
                    HN_TRY(field_sdf(r.f, t.pts, n_rays * n_new, bt_inv, T_pose, r.nf, r.which == 0 ? rpf * n_new : n_rays * n_new, t.sdf_new,
                                     r.fws, r.fwb, r.st));
                    if (fused_rounds && upsample_fused_ok(n_rays, r.k + n_new, n_new)) {
                        pending[r.which] = true;
                    } else {
                        HN_TRY(merge(t.z_a, t.z_new, t.sdf_a, t.sdf_new, n_rays, r.k, n_new, quirk, t.z_b, t.sdf_b, nullptr, r.st));
                        float* tmp = t.z_a; t.z_a = t.z_b; t.z_b = tmp;
                        tmp = t.sdf_a; t.sdf_a = t.sdf_b; t.sdf_b = tmp;
                    }
                }
                r.k += n_new;
            }
        }
        if (side != nullptr) HN_TRY(join_from(side, s));
        HN_TRY(sort_rows(zcat, n_rays, S, z, s));
        z_final = z;
    }
    // both fields at the shared sorted depths (utils/renderer.py:500-510), side by side.  With a tape buffer the
    // evaluations keep their tapes ([hand | object]) for hn_render_dual_bwd.
    const size_t tape_h = field_tape(hand, hand_cap(hand, N)), tape_o = field_tape(obj, (int)N);
    void *tp_h = nullptr, *tp_o = nullptr;
    if (tape != nullptr) {
        HN_REQUIRE(tape_bytes >= tape_h + tape_o, "render_dual tape too small: %zu < %zu", tape_bytes, tape_h + tape_o);
        tp_h = tape_h ? tape : nullptr;
        tp_o = tape_o ? reinterpret_cast<char*>(tape) + tape_h : nullptr;
    }
    // The object branch is released (fork) only when the hand's evaluation kernel is next in line on s: released earlier, its
    // 294-tile kernel takes every CU (one 512-register workgroup per CU) and the small launches in front of the hand's
    // evaluation -- its sample points, the compaction -- wait a whole tile time for a wave slot (measured: 245 us).
    bool forked = false;
    // CROWDED (several frames side by side: more than two tiles per CU): the hand's evaluation alone fills every CU for a tile time or
    // more, and a small launch of the object branch queued beside it -- its sample points -- waits that long for a wave slot, the
    // object's evaluation behind it (measured, 4 frames: k_sample_points_t 1.25 ms in front of k_field2_obj<3>).  So the object's
    // small launches go FIRST, beside the hand's small ones, the hand's stream waits for them, and the two evaluations are queued
    // back to back.
    const bool crowded = side != nullptr && (N + 127) / 128 >= (size_t)2 * (size_t)(device_cus() > 0 ? device_cus() : 256);
    bool obj_pts_made = false;
    if (crowded) {
        HN_TRY(fork_to(side, s));
        forked = true;
        if (!obj_rays_made) {
            HN_TRY(obj_local_fwd(rays_o, rays_d, Ro, To, n_frames, rpf, o_obj, d_obj, so, ro_t));
            obj_rays_made = true;
        }
        HN_TRY(sample_points(o_obj, d_obj, z_final, n_rays, S, 1, sample_dist, pts_o, dists_o, so));
        obj_pts_made = true;
    }
    HN_TRY(sample_points(rays_o, rays_d, z_final, n_rays, S, 1, sample_dist, pts, dists, s));
    if (hand_compaction(hand, n_frames, N)) {
        // the hand field on the samples with a live bone + one far sample; the record stays with the tape for the backward pass
        const size_t rec_off = tape_h + tape_o + 8 * N * sizeof(float);
        const bool in_tape = tape != nullptr && tape_h + tape_o != 0 && tape_bytes >= rec_off + CompactRec::bytes(N);
        CompactRec cr;
        cr.at(in_tape ? reinterpret_cast<char*>(tape) + rec_off : crec_ws, N);
        HN_TRY(compact_hand(cr, pts, (int)N, bt_inv, T_pose, n_frames, rpf * S, s));
        if (crowded)
            HN_TRY(join_from(side, s));
        else if (side != nullptr)
            HN_TRY(fork_to(side, s));
        forked = true;
        set_launch_n_pts_dev(cr.n_dev);
        set_launch_orig_idx(cr.idx);
        const int rc = field_eval(hand, cr.pts_c, rays_d, (int)N + 1, S, bt_inv, T_pose, n_frames, rpf * S, cr.sdf_c, cr.grad_c, cr.rgb_c, nullptr, fwsh,
                                  fws_h, s, in_tape ? tp_h : nullptr, in_tape ? tape_h : 0);
        set_launch_n_pts_dev(nullptr);
        set_launch_orig_idx(nullptr);
        HN_TRY(rc);
        hipLaunchKernelGGL(k_hand_scatter, dim3(((int)N + 255) / 256), dim3(256), 0, s, cr.pos, (int)N, cr.n_dev, cr.sdf_c, cr.grad_c, cr.rgb_c, sdf_hand,
                           grad_hand, rgb_h);
        HN_LAUNCH_CHECK();
    } else {
        if (crowded)
            HN_TRY(join_from(side, s));
        else if (side != nullptr)
            HN_TRY(fork_to(side, s));
        forked = true;
        HN_TRY(field_eval(hand, pts, rays_d, (int)N, S, bt_inv, T_pose, n_frames, rpf * S, sdf_hand, grad_hand, rgb_h, nullptr,
                          fwsh, fws_h, s, tp_h, tape_h));
    }
    (void)forked;
    HN_TRY(alpha(sdf_hand, grad_hand, rays_d, dists, (int)N, S, hand->inv_s, al_h, nullptr, s));
    if (!obj_rays_made) HN_TRY(obj_local_fwd(rays_o, rays_d, Ro, To, n_frames, rpf, o_obj, d_obj, so, ro_t));
    if (!obj_pts_made) HN_TRY(sample_points(o_obj, d_obj, z_final, n_rays, S, 1, sample_dist, pts_o, dists_o, so));
    HN_TRY(field_eval(obj, pts_o, d_obj, (int)N, S, nullptr, nullptr, 1, (int)N, sdf_obj, grad_obj, rgb_o, nullptr, fwso,
                      fws_o, so, tp_o, tape_o));
    HN_TRY(alpha(sdf_obj, grad_obj, d_obj, dists_o, (int)N, S, obj->inv_s, al_o, nullptr, so));
    // (on s while it waits for the object branch: off the critical path)
    HN_CHECK_HIP(hipMemsetAsync(gradient_error, 0, 2 * sizeof(float), s));
    if (z_vals != nullptr) HN_CHECK_HIP(hipMemcpyAsync(z_vals, z_final, N * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (side != nullptr) HN_TRY(join_from(side, s));
    HN_TRY(composite2(al_h, rgb_h, grad_hand, al_o, rgb_o, grad_obj, n_rays, S, color, weight_sum, nullptr, nullptr,
                      gradient_error, s, 1.f / (float)N));
    return HN_OK;
}

hipError_t pool_alloc(void** p, size_t bytes);   // hn_pack.hip
void pool_free(void* p);
size_t pool_trim();
namespace bwd {
int weight_norm_bwd(const hn_field*, const hn_mlp_desc*, const hn_mlp_desc*, const float*, const hn_mlp_desc*, const hn_mlp_desc*, hipStream_t);
typedef std::function<int(const float* z8, const float* grad, const float* rgb_pre)> MidHook;
typedef std::function<int(const float* sdf, const float* grad, const float* rgb)> MidHook2;
size_t field_bwd_workspace_bytes(const hn_field* f, int n);
int field_eval_bwd(const hn_field* f, const float* pts, const float* rays_d, int n, int spr, const float* bt_inv,
                   const float* T_pose, int n_frames, int pts_per_frame, const float* g_sdf, const float* g_grad,
                   const float* g_rgb, float* g_pts, float* g_rays_d, float* g_bt_inv, float* g_T_pose, void* workspace,
                   size_t workspace_bytes, hipStream_t s, const void* tape, const float* grad, const float* rgb,
                   float* g_params = nullptr, const MidHook* mid = nullptr, const MidHook2* mid2 = nullptr, const float* sdf_t = nullptr,
                   const float* feat_t = nullptr);
}

// Backward pass of the two-field render (what loss.backward() runs through NeuSRenderer_fitting.render in the fitting
// loops: fitting_single.py:289-291, fitting_video.py:340-342).  Depths carry no gradient (sampled under no_grad), so it
// runs through the compositing, the two alpha stages, the two field evaluations (fused adjoint kernels, hand on s and
// object on the second stream) and the ray transforms.
static int render_dual_bwd_impl(const hn_field* hand, const hn_field* obj, const float* rays_o, const float* rays_d, int n_frames,
                                int rpf, int S, float sample_dist, const float* bt_inv, const float* T_pose, const float* Ro,
                                const float* To, const float* z, const float* sdf_h, const float* grad_h, const float* rgb_h,
                                const float* alpha_h, const float* sdf_o, const float* grad_o, const float* rgb_o,
                                const float* alpha_o, const float* g_color, const float* g_wsum, const float* g_sdf_h,
                                const float* g_sdf_o, const float* g_grad_h, const float* g_grad_o, const float* g_eik,
                                float* g_rays_o, float* g_rays_d, float* g_bt_inv, float* g_T_pose, float* g_Ro, float* g_To,
                                void* workspace, size_t workspace_bytes, hipStream_t s, size_t* need, const void* tape = nullptr, int flags = 0) {
    const bool ro_t = (flags & HN_DUAL_RO_TRANSPOSED) != 0;
    const bool no_join = (flags & HN_DUAL_BWD_NO_JOIN) != 0;   // the object branch's results stay on the side stream (pipelined fitting step)
    HN_REQUIRE(hand->kind == HN_FIELD_HAND && obj->kind == HN_FIELD_OBJ, "field kinds (hand, obj) expected");
    HN_REQUIRE(n_frames >= 1 && rpf >= 0 && S >= 1, "bad sizes");
    const int n_rays = n_frames * rpf;
    const size_t N = (size_t)n_rays * S, R3 = (size_t)n_rays * 3;
    Arena ar(workspace, workspace_bytes);
    float *o_l = ar.f(R3), *d_l = ar.f(R3);
    float *g_ah = ar.f(N), *g_ao = ar.f(N), *g_rgbh = ar.f(N * 3), *g_rgbo = ar.f(N * 3);
    float *pts_h = ar.f(N * 3), *dists_h = ar.f(N), *pts_o = ar.f(N * 3), *dists_o = ar.f(N);
    float *gs_h = ar.f(N), *gg_h = ar.f(N * 3), *gd_h = ar.f(R3), *gs_o = ar.f(N), *gg_o = ar.f(N * 3), *gd_o = ar.f(R3);
    float* rays_part = ar.f((size_t)n_rays * 12);                                   // k_obj_rays_bwd: the rays' addends of g_Ro / g_To
    unsigned* rays_counter = reinterpret_cast<unsigned*>(ar.f(64));
    float* gd_os = ar.f(N * 3);                                                     // the object's alpha stage: d loss / d rays_d per sample
    float* gdir_os = ar.f(N * 3);                                                   // ... and its colour network's
    bool dir_per_sample = false;
    float *gp_h = ar.f(N * 3), *gp_o = ar.f(N * 3), *gdir_h = ar.f(R3), *gdir_o = ar.f(R3);
    float *go_h = ar.f(R3), *gdd_h = ar.f(R3), *g_ro2 = ar.f(R3), *g_rd2 = ar.f(R3);
    const size_t bws_h = bwd::field_bwd_workspace_bytes(hand, hand_cap(hand, N)), bws_o = bwd::field_bwd_workspace_bytes(obj, (int)N);
    void* bwh = ar.take(bws_h);
    void* bwo = ar.take(bws_o);
    // compacted upstream gradients / d loss / d pts of the hand field (hn_field_set_compaction)
    float *gs_c = nullptr, *gg_c = nullptr, *gr_c = nullptr, *gp_c = nullptr;
    if (hand->compact_far_field) {
        gs_c = ar.f(N + 1);
        gg_c = ar.f((N + 1) * 3);
        gr_c = ar.f((N + 1) * 3);
        gp_c = ar.f((N + 1) * 3);
    }
    if (need != nullptr) {
        *need = ar.used;
        return HN_OK;
    }
    if (!ar.ok) {
        set_error("render_dual_bwd workspace too small: %zu bytes given", workspace_bytes);
        return HN_ENOMEM;
    }
    if (n_rays == 0) return HN_OK;
    HN_REQUIRE(g_color && g_bt_inv && g_T_pose && g_Ro && g_To, "null output / upstream gradient");
    HN_REQUIRE((g_rays_o == nullptr) == (g_rays_d == nullptr), "g_rays_o and g_rays_d: both or neither");
    const int n = (int)N;
    // the tapes the forward pass kept ([hand | object]): the adjoints then run alone, nothing is evaluated again
    const size_t tape_h = field_tape(hand, hand_cap(hand, N)), tape_o = field_tape(obj, n);
    const void* tp_h = (tape != nullptr && tape_h) ? tape : nullptr;
    const void* tp_o = (tape != nullptr && tape_o) ? reinterpret_cast<const char*>(tape) + tape_h : nullptr;
    // the forward pass compacted the hand's samples (same predicate; its record sits behind the tapes and the rgb / alpha block)
    const bool compact = tape != nullptr && tape_h + tape_o != 0 && hand_compaction(hand, n_frames, N);
    CompactRec cr{};
    if (compact) cr.at(const_cast<char*>(reinterpret_cast<const char*>(tape)) + tape_h + tape_o + 8 * N * sizeof(float), N);
    SideStream* side = side_stream();
    SideLock side_lock(side);
    const hipStream_t so = side != nullptr ? side->s2 : s;
    // d loss / d the WORLD rays is optional (both NULL: the fitting loops' rays come from fixed cameras and carry no gradient):
    // without it the hand branch ends with its adjoint kernel
    const bool want_rays = g_rays_o != nullptr;
    HN_TRY(composite2_bwd(alpha_h, rgb_h, alpha_o, rgb_o, g_color, g_wsum, n_rays, S, g_ah, g_rgbh, g_ao, g_rgbo, s));
    // The object branch's small launches are released here (fork) and run beside the hand's; its adjoint kernel itself waits at
    // a second meeting point (gate) until the hand's adjoint kernel is next in line on s: that kernel's tiles are the long ones
    // and must get their CUs first, the object's 294 shorter tiles then take what is left at once.
    if (side != nullptr) HN_TRY(fork_to(side, s));
    // hand branch (s): ONE launch in front of the adjoint kernel (k_alpha_bwd_up: alpha stage, the caller's direct gradients, the
    // eikonal term, the gather onto the compact list, the zero fills of what the adjoint accumulates into)
    if (want_rays) HN_TRY(zero_many({{gd_h, R3}, {compact ? gdir_h : nullptr, R3}}, s));
    if (!compact) HN_TRY(sample_points(rays_o, rays_d, z, n_rays, S, 1, sample_dist, pts_h, dists_h, s));   // (the compact list carries its points)
    {
        float* zb[2] = {g_bt_inv, g_T_pose};
        const size_t zn[2] = {(size_t)n_frames * 21 * 16, (size_t)n_frames * 21 * 3};
        HN_TRY(alpha_bwd_up(sdf_h, grad_h, rays_d, z, g_ah, n, S, sample_dist, hand->inv_s, g_sdf_h, g_grad_h, g_eik, gs_h, gg_h, want_rays ? gd_h : nullptr,
                            compact ? cr.pos : nullptr, compact ? cr.n_dev : nullptr, g_rgbh, gs_c, gg_c, gr_c, zb, zn, 2, s, nullptr,
                            (compact && CompactRec::aligned_frames(n, n_frames, rpf * S) > 0) ? cr.seg : nullptr));
    }
    // object branch (so) up to its adjoint kernel
    HN_TRY(obj_local_fwd(rays_o, rays_d, Ro, To, n_frames, rpf, o_l, d_l, so, ro_t));
    HN_TRY(sample_points(o_l, d_l, z, n_rays, S, 1, sample_dist, pts_o, dists_o, so));
    {
        // (g_Ro / g_To are written, not accumulated into, by k_obj_rays_bwd's last block; its counter must start at zero)
        float* zb[4] = {gd_o, g_Ro, g_To, reinterpret_cast<float*>(rays_counter)};
        const size_t zn[4] = {R3, (size_t)n_frames * 9, (size_t)n_frames * 3, 1};
        // (the alpha stage's d loss / d rays_d goes out per SAMPLE, gd_os, and is summed per ray by k_obj_rays_bwd in sample order:
        // summed here with atomics per ray, d loss / d Ro differed in its last bits from run to run)
        HN_TRY(alpha_bwd_up(sdf_o, grad_o, d_l, z, g_ao, n, S, sample_dist, obj->inv_s, g_sdf_o, g_grad_o, g_eik != nullptr ? g_eik + 1 : nullptr, gs_o, gg_o, nullptr,
                            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, zb + 1, zn + 1, 3, so, gd_os));
    }
    // (crowded -- several frames side by side, see render_dual_impl --: the hand's adjoint kernel waits for the object's small launches
    // instead of the other way round: queued beside a kernel that fills every CU they waited a tile time, 1.65 ms, for a wave slot)
    const bool crowded = side != nullptr && (N + 127) / 128 >= (size_t)2 * (size_t)(device_cus() > 0 ? device_cus() : 256);
    if (crowded)
        HN_TRY(join_from(side, s));
    else if (side != nullptr)
        HN_TRY(gate_to(side, s));
    if (compact) {
        set_launch_n_pts_dev(cr.n_dev);
        set_launch_orig_idx(cr.idx);
        set_launch_frame_seg(CompactRec::aligned_frames(n, n_frames, rpf * S) > 0 ? cr.seg : nullptr);
        // (the hand's colour network ignores the view direction, utils/fields.py:222-240: no d loss / d rays_d through it)
        const int rc = bwd::field_eval_bwd(hand, cr.pts_c, rays_d, n + 1, 1, bt_inv, T_pose, n_frames, rpf * S, gs_c, gg_c, gr_c, gp_c, nullptr, g_bt_inv,
                                           g_T_pose, bwh, bws_h, s, tp_h, cr.grad_c, cr.rgb_c);
        set_launch_n_pts_dev(nullptr);
        set_launch_orig_idx(nullptr);
        set_launch_frame_seg(nullptr);
        HN_TRY(rc);
        if (want_rays) {
            hipLaunchKernelGGL(k_hand_scatter3, dim3((n + 255) / 256), dim3(256), 0, s, cr.pos, n, gp_c, gp_h);
            HN_LAUNCH_CHECK();
        }
    } else {
        HN_TRY(bwd::field_eval_bwd(hand, pts_h, rays_d, n, S, bt_inv, T_pose, n_frames, rpf * S, gs_h, gg_h, g_rgbh, gp_h, gdir_h, g_bt_inv,
                                   g_T_pose, bwh, bws_h, s, tp_h, grad_h, rgb_h));
    }
    if (want_rays) HN_TRY(sample_points_bwd(z, gp_h, n_rays, S, 1, sample_dist, go_h, gdd_h, s));
    {
        // (the colour network's d loss / d rays_d per SAMPLE for f16x3 fields, summed per ray by k_obj_rays_bwd: no atomics)
        const bool per_sample = obj->v2_adj != nullptr;
        set_launch_dir_per_sample(per_sample);
        const int rc = bwd::field_eval_bwd(obj, pts_o, d_l, n, S, nullptr, nullptr, 1, n, gs_o, gg_o, g_rgbo, gp_o, per_sample ? gdir_os : gdir_o, nullptr,
                                           nullptr, bwo, bws_o, so, tp_o, grad_o, rgb_o);
        set_launch_dir_per_sample(false);
        HN_TRY(rc);
        dir_per_sample = per_sample;
    }
    // behind the object's adjoint: the ray map and the adjoint of convert_obj_to_local in one launch
    HN_TRY(obj_rays_bwd(z, gp_o, n_frames, rpf, S, sample_dist, nullptr, dir_per_sample ? nullptr : gdir_o, rays_o, rays_d, Ro, To, want_rays ? g_ro2 : nullptr,
                        want_rays ? g_rd2 : nullptr, g_Ro, g_To, so, ro_t, rays_part, rays_counter, gd_os, dir_per_sample ? gdir_os : nullptr));
    HN_REQUIRE(!no_join || (!want_rays && side != nullptr), "HN_DUAL_BWD_NO_JOIN: no ray gradients, and the device's second stream");
    if (side != nullptr && !no_join) HN_TRY(join_from(side, s));
    if (want_rays) {
        hipLaunchKernelGGL(k_add4x2, dim3(((int)R3 + 255) / 256), dim3(256), 0, s, go_h, g_ro2, g_rays_o, gdd_h, gd_h, gdir_h, g_rd2, g_rays_d, (int)R3);
        HN_LAUNCH_CHECK();
    }
    return HN_OK;
}

// sdf = z8[:, 0] / scale, rgb = sigmoid(pre) from the adjoint's forward tape
__global__ void k_tape_outputs(const float* __restrict__ z8, const float* __restrict__ pre, float inv_scale, int n,
                               float* __restrict__ sdf, float* __restrict__ rgb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sdf[i] = z8[(size_t)i * 257] * inv_scale;
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[3 * (size_t)i + c] = 1.f / (1.f + expf(-pre[3 * (size_t)i + c]));
}

// Backward pass of the single-field render into the field's PARAMETERS (what loss.backward() runs through
// NeuSRenderer.render in exp_runner.train, exp_runner.py:208-242) and into the rays / pose inputs.  The
// depths carry no gradient (utils/renderer.py:215 no_grad), so it runs through render_core at the depths z the
// forward pass produced (utils/renderer.py:107-177): the field is evaluated there again, then
// composite1_bwd -> alpha_bwd (+ d/d inv_s) -> eikonal term -> field adjoint with parameter gradients -> ray map.
static int render_single_bwd_impl(const hn_field* f, const float* rays_o, const float* rays_d, int n_rays, int S, float sample_dist,
                                  const float* bt_inv, const float* T_pose, const float* z, const float* g_color,
                                  const float* g_wsum, const float* g_eik, float* g_params, float* g_inv_s, float* g_rays_o,
                                  float* g_rays_d, float* g_bt_inv, float* g_T_pose, void* workspace, size_t workspace_bytes,
                                  hipStream_t s, size_t* need, const void* keep_block = nullptr, size_t keep_bytes = 0) {
    HN_REQUIRE(n_rays >= 0 && S >= 1, "bad sizes");
    const size_t N = (size_t)n_rays * S, R3 = (size_t)n_rays * 3;
    Arena ar(workspace, workspace_bytes);
    float *pts = ar.f(N * 3), *dists = ar.f(N), *sdf = ar.f(N), *rgb = ar.f(N * 3), *al = ar.f(N), *c = ar.f(N);
    float *g_al = ar.f(N), *g_c = ar.f(N), *g_rgb = ar.f(N * 3), *gs = ar.f(N), *gg = ar.f(N * 3), *gd = ar.f(R3);
    float *gp = ar.f(N * 3), *gdir = ar.f(R3), *gdd = ar.f(R3), *go = ar.f(R3), *pose_scratch = ar.f(21 * 16 + 21 * 3);
    const size_t bws_bytes = bwd::field_bwd_workspace_bytes(f, f->kind == HN_FIELD_HAND ? hand_cap(f, N) : (int)N);
    void* bws = ar.take(bws_bytes);
    // hn_field_set_compaction (hand): the adjoint runs on the live samples + ONE far sample that carries the summed upstream
    // gradients of all dead ones (k_hand_gather_up_sum)
    const bool may_compact = f->kind == HN_FIELD_HAND && f->compact_far_field;
    void* crec_ws = nullptr;
    float *grad_d = nullptr, *gs_c = nullptr, *gg_c = nullptr, *gr_c = nullptr, *gp_c = nullptr;
    if (may_compact) {
        crec_ws = ar.take(CompactRec::bytes(N));
        grad_d = ar.f(N * 3);
        gs_c = ar.f(N + 1);
        gg_c = ar.f((N + 1) * 3);
        gr_c = ar.f((N + 1) * 3);
        gp_c = ar.f((N + 1) * 3);
    }
    if (need != nullptr) {
        *need = ar.used;
        return HN_OK;
    }
    if (n_rays == 0) return HN_OK;      // an empty batch: nothing to add to g_params, nothing to launch
    if (!ar.ok) {
        set_error("render_single_bwd workspace too small: %zu bytes given", workspace_bytes);
        return HN_ENOMEM;
    }
    const bool hand = f->kind == HN_FIELD_HAND;
    HN_REQUIRE(rays_o && rays_d && z && g_color && g_params, "null argument");
    HN_REQUIRE(!hand || (bt_inv && T_pose), "hand field needs bt_inv / T_pose");
    const int n = (int)N;
    HN_TRY(sample_points(rays_o, rays_d, z, n_rays, S, 1, sample_dist, pts, dists, s));
    const bool compact = may_compact && hand_compaction(f, 1, N);
    CompactRec cr{};
    int n_c = n;   // rows the adjoint runs on
    if (compact) {
        cr.at(crec_ws, N);
        HN_TRY(compact_hand(cr, pts, n, bt_inv, T_pose, 1, n, s));
        // The launch sequence of the parameter-gradient adjoint is sized on the host by the live count.  With a kept block the forward pass
        // has sent it already (the same points give the same list); otherwise it is read back here (the one place this library waits for
        // a stream).
        if (!(keep_block != nullptr && kept_count_take(keep_block, &n_c))) {
            HN_CHECK_HIP(hipMemcpyAsync(&n_c, cr.n_dev, sizeof(int), hipMemcpyDeviceToHost, s));
            HN_CHECK_HIP(hipStreamSynchronize(s));
        }
        HN_REQUIRE(n_c >= 1 && n_c <= n + 1, "compaction count out of range: %d", n_c);
    }
    // The field is NOT evaluated again: the adjoint's own forward tape (exact fp32) supplies sdf / gradient / colour;
    // the alpha stage, the compositing and their adjoints run in the hook, between the tape and the sweeps.
    const bwd::MidHook mid = [&](const float* z8, const float* g_field_in, const float* rgb_pre) -> int {
        const float* g_field = g_field_in;
        if (compact) {   // tape rows -> dense per-sample arrays (dead samples: the far sample's values)
            hipLaunchKernelGGL(k_tape_outputs, dim3((n_c + 255) / 256), dim3(256), 0, s, z8, rgb_pre, 1.f / f->scale, n_c, cr.sdf_c, cr.rgb_c);
            hipLaunchKernelGGL(k_hand_scatter, dim3((n + 255) / 256), dim3(256), 0, s, cr.pos, n, cr.n_dev, cr.sdf_c, g_field_in, cr.rgb_c, sdf, grad_d, rgb);
            HN_LAUNCH_CHECK();
            g_field = grad_d;
        } else
        hipLaunchKernelGGL(k_tape_outputs, dim3((n + 255) / 256), dim3(256), 0, s, z8, rgb_pre, 1.f / f->scale, n, sdf, rgb);
        HN_TRY(alpha(sdf, g_field, rays_d, dists, n, S, f->inv_s, al, c, s, f->inv_s_dev));
        HN_TRY(composite1_bwd(al, c, rgb, g_color, g_wsum, n_rays, S, g_al, g_c, g_rgb, s));
        HN_TRY(alpha_bwd(sdf, g_field, rays_d, dists, g_al, g_c, n, S, f->inv_s, gs, gg, gd, s, false, f->inv_s_dev));
        if (g_inv_s != nullptr) {
            HN_CHECK_HIP(hipMemsetAsync(g_inv_s, 0, sizeof(float), s));
            HN_TRY(alpha_inv_s_bwd(sdf, g_field, rays_d, dists, g_al, g_c, n, S, f->inv_s, g_inv_s, s, f->inv_s_dev));
        }
        hipLaunchKernelGGL(k_upstream, dim3((n + 255) / 256), dim3(256), 0, s, gs, gg, (const float*)nullptr, (const float*)nullptr, g_field,
                           g_eik, n);
        if (compact) {   // dense upstream gradients -> the compact rows; the far sample's = the sums over the dead samples
            hipLaunchKernelGGL(k_zero_slot, dim3(1), dim3(64), 0, s, cr.n_dev, gs_c, gg_c, gr_c);
            hipLaunchKernelGGL(k_hand_gather_up_sum, dim3((n + 255) / 256), dim3(256), 0, s, cr.idx, cr.pos, n, cr.n_dev, gs, gg, g_rgb, gs_c, gg_c, gr_c);
        }
        HN_LAUNCH_CHECK();
        return HN_OK;
    };
    // ... the same stages for the FUSED parameter-gradient path of an f16x3 field (hn_field_bwd.hip): the taped evaluation's own
    // outputs instead of the generic tape's last-layer rows
    const bwd::MidHook2 mid2 = [&](const float* sdf_t, const float* grad_t, const float* rgb_t) -> int {
        if (compact) {   // the compact rows' outputs -> dense per-sample arrays (dead samples: the far sample's values)
            hipLaunchKernelGGL(k_hand_scatter, dim3((n + 255) / 256), dim3(256), 0, s, cr.pos, n, cr.n_dev, sdf_t, grad_t, rgb_t, sdf, grad_d, rgb);
            HN_LAUNCH_CHECK();
            sdf_t = sdf;
            grad_t = grad_d;
            rgb_t = rgb;
        }
        HN_TRY(alpha(sdf_t, grad_t, rays_d, dists, n, S, f->inv_s, al, c, s, f->inv_s_dev));
        HN_TRY(composite1_bwd(al, c, rgb_t, g_color, g_wsum, n_rays, S, g_al, g_c, g_rgb, s));
        HN_TRY(alpha_bwd(sdf_t, grad_t, rays_d, dists, g_al, g_c, n, S, f->inv_s, gs, gg, gd, s, false, f->inv_s_dev));
        if (g_inv_s != nullptr) {
            HN_CHECK_HIP(hipMemsetAsync(g_inv_s, 0, sizeof(float), s));
            HN_TRY(alpha_inv_s_bwd(sdf_t, grad_t, rays_d, dists, g_al, g_c, n, S, f->inv_s, g_inv_s, s, f->inv_s_dev));
        }
        hipLaunchKernelGGL(k_upstream, dim3((n + 255) / 256), dim3(256), 0, s, gs, gg, (const float*)nullptr, (const float*)nullptr, grad_t, g_eik, n);
        if (compact) {   // dense upstream gradients -> the compact rows; the far sample's = the sums over the dead samples
            hipLaunchKernelGGL(k_zero_slot, dim3(1), dim3(64), 0, s, cr.n_dev, gs_c, gg_c, gr_c);
            hipLaunchKernelGGL(k_hand_gather_up_sum, dim3((n + 255) / 256), dim3(256), 0, s, cr.idx, cr.pos, n, cr.n_dev, gs, gg, g_rgb, gs_c, gg_c, gr_c);
        }
        HN_LAUNCH_CHECK();
        return HN_OK;
    };
    float *gbt = g_bt_inv, *gtp = g_T_pose;
    // the adjoint accumulates the pose gradients: into the caller's arrays when given.  The generic sequence's bone-map kernels always write
    // them (scratch when nobody asks); the fused adjoint kernel skips its pose-gradient block when neither is asked for (exp_runner trains the
    // networks on fixed poses)
    if (hand && !(bwd::param_path_is_fused(f) && gbt == nullptr && gtp == nullptr)) {
        if (gbt == nullptr) gbt = pose_scratch;
        if (gtp == nullptr) gtp = pose_scratch + 21 * 16;
        HN_CHECK_HIP(hipMemsetAsync(gbt, 0, 21 * 16 * sizeof(float), s));
        HN_CHECK_HIP(hipMemsetAsync(gtp, 0, 21 * 3 * sizeof(float), s));
    }
    // keep_block (hn_render_single_bwd_taped): the forward pass's tape and outputs stand in for the taped evaluation
    const SingleKeep kp(f, N, const_cast<void*>(keep_block));
    if (keep_block != nullptr) {
        HN_REQUIRE(single_keep_applies(f), "hn_render_single_bwd_taped: the field's backward pass does not take a tape");
        HN_REQUIRE(keep_bytes >= kp.total, "hn_render_single_bwd_taped: tape block too small (%zu < %zu)", keep_bytes, kp.total);
    }
    const void* k_tape = keep_block ? kp.tape : nullptr;
    const float *k_sdf = keep_block ? kp.sdf : nullptr, *k_grad = keep_block ? kp.grad : nullptr, *k_rgb = keep_block ? kp.rgb : nullptr,
                *k_feat = keep_block ? kp.feat : nullptr;
    if (compact) {
        // (the hand's colour network ignores the view direction: d loss / d rays_d through it is exactly 0)
        HN_CHECK_HIP(hipMemsetAsync(gdir, 0, R3 * sizeof(float), s));
        HN_TRY(bwd::field_eval_bwd(f, cr.pts_c, rays_d, n_c, 1, bt_inv, T_pose, 1, n_c, gs_c, gg_c, gr_c, gp_c, nullptr, gbt, gtp, bws, bws_bytes, s,
                                   k_tape, k_grad, k_rgb, g_params, &mid, &mid2, k_sdf, k_feat));
        hipLaunchKernelGGL(k_hand_scatter3, dim3((n + 255) / 256), dim3(256), 0, s, cr.pos, n, gp_c, gp);
        HN_LAUNCH_CHECK();
    } else {
        HN_TRY(bwd::field_eval_bwd(f, pts, rays_d, n, S, bt_inv, T_pose, 1, n, gs, gg, g_rgb, gp, gdir, gbt, gtp, bws, bws_bytes, s, k_tape,
                                   k_grad, k_rgb, g_params, &mid, &mid2, k_sdf, k_feat));
    }
    HN_TRY(sample_points_bwd(z, gp, n_rays, S, 1, sample_dist, go, gdd, s));
    if (g_rays_o != nullptr) HN_CHECK_HIP(hipMemcpyAsync(g_rays_o, go, R3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (g_rays_d != nullptr)
        hipLaunchKernelGGL(k_add4, dim3(((int)R3 + 255) / 256), dim3(256), 0, s, gdd, gd, gdir, (const float*)nullptr, g_rays_d, (int)R3);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace hn

using namespace hn;

namespace hn {
namespace bwd {
int hand_features(const float*, int, const float*, const float*, int, int, float*, float*, float*, hipStream_t);
size_t color_forward_workspace_bytes(const hn_field* f, int n);
int color_forward(const hn_field*, const float*, const float*, const float*, const float*, int, float*, void*, size_t, hipStream_t);
int nearest_masked(const float*, int, int, const unsigned char*, const unsigned char*, unsigned char*, int*, hipStream_t);
}
}
extern "C" {

int hn_hand_features(const float* pts, int n_pts, const float* bt_inv, const float* T_pose, int n_frames, int pts_per_frame,
                     float* xyz_feature, float* r, float* h, hn_stream_t stream) {
    return hn::bwd::hand_features(pts, n_pts, bt_inv, T_pose, n_frames, pts_per_frame, xyz_feature, r, h, (hipStream_t)stream);
}
size_t hn_color_forward_workspace_bytes(const hn_field* f, int n_pts) {
    return (f == nullptr || n_pts <= 0) ? 0 : hn::bwd::color_forward_workspace_bytes(f, n_pts);
}
int hn_color_forward(const hn_field* f, const float* x, const float* view_dirs, const float* feature_vectors,
                     const float* normals, int n_pts, float* rgb, void* workspace, size_t workspace_bytes, hn_stream_t stream) {
    HN_REQUIRE(f != nullptr, "null field");
    return hn::bwd::color_forward(f, x, view_dirs, feature_vectors, normals, n_pts, rgb, workspace, workspace_bytes,
                                  (hipStream_t)stream);
}
int hn_pose_chain(const float* ori_pose, const float* bone_len, const unsigned char* is_right, const float* params, int n_frames,
                  float* bt_inv, float* joint_3d, float* jac, hn_stream_t stream) {
    return hn::pose_chain(ori_pose, bone_len, is_right, params, n_frames, bt_inv, joint_3d, jac, (hipStream_t)stream);
}
int hn_pose_chain_bwd(const float* jac, const float* g_bt_inv, const float* g_joint_3d, int n_frames, float* g_params, hn_stream_t stream) {
    return hn::pose_chain_bwd(jac, g_bt_inv, g_joint_3d, n_frames, g_params, (hipStream_t)stream);
}

int hn_rigid_pose(const float* bt_inv0, const float* joints0, const float* Ro_pred, const float* To_pred, const float* params, int n_frames,
                  int with_palm, float* out, float* jac, hn_stream_t stream) {
    return hn::rigid_pose(bt_inv0, joints0, Ro_pred, To_pred, params, n_frames, with_palm, out, jac, (hipStream_t)stream);
}
int hn_verts_loss(const float* Ra, const float* ta, const float* Rb, const float* tb, const float* verts, int n_verts, int n_pairs, float* loss,
                  float* gR, float* gt, hn_stream_t stream) {
    return hn::verts_loss(Ra, ta, Rb, tb, verts, n_verts, n_pairs, loss, gR, gt, (hipStream_t)stream);
}
int hn_leaf_rows_gather(const float* const* leaves6, const long long* rows, int n_rows, int n_frames, float* prm_hand, float* prm_obj,
                        hn_stream_t stream) {
    return hn::leaf_rows_gather(leaves6, rows, n_rows, n_frames, prm_hand, prm_obj, (hipStream_t)stream);
}
int hn_leaf_rows_scatter(const float* g, const long long* rows, int n_rows, int n_frames, float* out, hn_stream_t stream) {
    return hn::leaf_rows_scatter(g, rows, n_rows, n_frames, out, (hipStream_t)stream);
}
int hn_pose_side_vjp(const float* jac_h, const float* jac_o, const float* g_bt_inv, const float* g_joint_3d, const float* g_obj_r, const float* g_obj_t,
                     const float* g_obj_r2, const float* g_obj_t2, int n_frames, int which, float* out, hn_stream_t stream) {
    return hn::pose_side_vjp(jac_h, jac_o, g_bt_inv, g_joint_3d, g_obj_r, g_obj_t, g_obj_r2, g_obj_t2, n_frames, which, out, (hipStream_t)stream);
}
int hn_jacobian_vjp(const float* jac, const float* g, int n_frames, int n_out, int n_in, float* out, hn_stream_t stream) {
    return hn::jacobian_vjp(jac, g, n_frames, n_out, n_in, out, (hipStream_t)stream);
}

int hn_nearest_masked(const float* pts, int n_verts, int n_sets, const unsigned char* query_mask, const unsigned char* cand_mask,
                      unsigned char* selected, int32_t* nearest, hn_stream_t stream) {
    return hn::bwd::nearest_masked(pts, n_verts, n_sets, query_mask, cand_mask, selected, nearest, (hipStream_t)stream);
}


int hn_version(void) { return HN_VERSION; }
int hn_side_stream(hn_stream_t* out) {
    HN_REQUIRE(out != nullptr, "null argument");
    SideStream* x = side_stream();
    HN_REQUIRE(x != nullptr, "the device's second stream could not be created");
    *out = (hn_stream_t)x->s2;
    return HN_OK;
}
int hn_stream_wait(hn_stream_t waiter, hn_stream_t on) {
    // `waiter` goes on only when everything queued on `on` so far has finished (an event of the library's, recorded and waited for at
    // once: the record is consumed by the wait before the next call re-records it)
    static std::mutex mu;
    static hipEvent_t ev[MAX_DEVICES] = {};
    const int dev = current_device();
    HN_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "bad device");
    std::lock_guard<std::mutex> lk(mu);
    if (ev[dev] == nullptr) HN_CHECK_HIP(hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming));
    HN_CHECK_HIP(hipEventRecord(ev[dev], (hipStream_t)on));
    HN_CHECK_HIP(hipStreamWaitEvent((hipStream_t)waiter, ev[dev], 0));
    return HN_OK;
}
int hn_debug_quad_max_blocks(int max_blocks) {
    hn::g_quad_max_blocks.store(max_blocks);
    return HN_OK;
}
int hn_debug_fused_rounds(int on) {
    hn::g_fused_rounds.store(on != 0 ? 1 : 0);
    return HN_OK;
}
int hn_debug_field_timer(int on) {
    hn::g_field_timer.store(on != 0 ? 1 : 0);
    return HN_OK;
}
int hn_debug_field_timer_read(double* total_ms, int* launches) {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;
    {
        std::lock_guard<std::mutex> lk(hn::g_field_timer_mutex);
        pairs.swap(hn::g_field_timer_pairs);
    }
    double sum = 0.0;
    int rc = HN_OK;
    for (auto& pr : pairs) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) != hipSuccess || hipEventElapsedTime(&ms, pr.first, pr.second) != hipSuccess) rc = HN_EHIP;
        sum += ms;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (total_ms != nullptr) *total_ms = sum;
    if (launches != nullptr) *launches = (int)pairs.size();
    if (rc != HN_OK) hn::set_error("hn_debug_field_timer_read: an event of the timer could not be read");
    return rc;
}
int hn_debug_pace_phantom(int members) {
    hn::g_pace_phantom.store(members < 0 ? 0 : members);
    return HN_OK;
}
int hn_dropped_samples(unsigned long long* count, int reset) { return hn::v2::hand_dropped_samples(count, reset != 0); }
const char* hn_last_error(void) { return g_err; }
int hn_device_cus(void) { return device_cus(); }

int hn_field_create(int kind, const hn_mlp_desc* sdf, const hn_mlp_desc* color, float variance, float scale,
                    int precision, hn_field** out, hn_stream_t stream) {
    return field_create(kind, sdf, color, variance, scale, precision, out, (hipStream_t)stream);
}
int hn_field_destroy(hn_field* f) {
    if (f == nullptr) return HN_OK;
    if (f->blob != nullptr) pool_free(f->blob);
    if (f->v2_full != nullptr) pool_free(f->v2_full);
    if (f->v2_sdf != nullptr) pool_free(f->v2_sdf);
    if (f->v2_adj != nullptr) pool_free(f->v2_adj);
    if (f->v2_adjonly != nullptr) pool_free(f->v2_adjonly);
    if (f->v2_tape != nullptr) pool_free(f->v2_tape);
    if (f->raw != nullptr) pool_free(f->raw);
    delete f;
    return HN_OK;
}
size_t hn_field_bwd_workspace_bytes(const hn_field* f, int n_pts) {
    if (f == nullptr || n_pts <= 0) return 0;
    return hn::bwd::field_bwd_workspace_bytes(f, n_pts);
}
int hn_field_eval_bwd(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int samples_per_ray,
                      const float* bt_inv, const float* T_pose, int n_frames, int pts_per_frame, const float* g_sdf,
                      const float* g_grad, const float* g_rgb, float* g_pts, float* g_rays_d, float* g_bt_inv,
                      float* g_T_pose, void* workspace, size_t workspace_bytes, hn_stream_t stream) {
    return hn::bwd::field_eval_bwd(f, pts, rays_d, n_pts, samples_per_ray, bt_inv, T_pose, n_frames, pts_per_frame, g_sdf, g_grad,
                                   g_rgb, g_pts, g_rays_d, g_bt_inv, g_T_pose, workspace, workspace_bytes,
                                   reinterpret_cast<hipStream_t>(stream), nullptr, nullptr, nullptr);
}
size_t hn_release_cached_memory(void) {
    hn::kept_counts_release();   // (the pinned words and events of hn_render_single_taped's counts: pool_trim synchronises the device first)
    return hn::pool_trim();
}
int hn_weight_norm_bwd(const hn_field* f, const hn_mlp_desc* sdf, const hn_mlp_desc* color, const float* g_params,
                       const hn_mlp_desc* g_sdf, const hn_mlp_desc* g_color, hn_stream_t stream) {
    return hn::bwd::weight_norm_bwd(f, sdf, color, g_params, g_sdf, g_color, reinterpret_cast<hipStream_t>(stream));
}
size_t hn_field_param_floats(const hn_field* f) { return (f == nullptr || f->raw == nullptr) ? 0 : f->raw_floats; }
int hn_field_param_offset(const hn_field* f, int net, int layer, size_t* w_off, size_t* b_off, int* out_dim, int* in_dim, int* ld) {
    HN_REQUIRE(f != nullptr && f->raw != nullptr, "field has no folded weights");
    HN_REQUIRE((net == 0 && layer >= 0 && layer < 9) || (net == 1 && layer >= 0 && layer < 5), "no such layer: net %d layer %d", net, layer);
    const float* base = reinterpret_cast<const float*>(f->raw);
    const float* w = net == 0 ? f->raw_sdf_w[layer] : f->raw_col_w[layer];
    const float* b = net == 0 ? f->raw_sdf_b[layer] : f->raw_col_b[layer];
    if (w_off) *w_off = (size_t)(w - base);
    if (b_off) *b_off = (size_t)(b - base);
    if (out_dim) *out_dim = net == 0 ? f->sdf_out[layer] : f->col_out[layer];
    if (in_dim) *in_dim = net == 0 ? f->sdf_in[layer] : f->col_in[layer];
    if (ld) *ld = net == 0 ? f->sdf_ld[layer] : f->col_ld[layer];
    return HN_OK;
}
int hn_field_param_bwd(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int samples_per_ray,
                       const float* bt_inv, const float* T_pose, int n_frames, int pts_per_frame, const float* g_sdf,
                       const float* g_grad, const float* g_rgb, float* g_params, float* g_pts, float* g_rays_d, float* g_bt_inv,
                       float* g_T_pose, void* workspace, size_t workspace_bytes, hn_stream_t stream) {
    HN_REQUIRE(g_params != nullptr, "g_params is NULL");
    return hn::bwd::field_eval_bwd(f, pts, rays_d, n_pts, samples_per_ray, bt_inv, T_pose, n_frames, pts_per_frame, g_sdf, g_grad,
                                   g_rgb, g_pts, g_rays_d, g_bt_inv, g_T_pose, workspace, workspace_bytes,
                                   reinterpret_cast<hipStream_t>(stream), nullptr, nullptr, nullptr, g_params);
}
size_t hn_render_single_bwd_workspace_bytes(const hn_field* f, int n_rays, int samples_per_ray) {
    if (f == nullptr || n_rays <= 0 || samples_per_ray <= 0) return 0;
    size_t need = 0;
    (void)render_single_bwd_impl(f, nullptr, nullptr, n_rays, samples_per_ray, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &need);
    return need;
}
int hn_render_single_bwd(const hn_field* f, const float* rays_o, const float* rays_d, int n_rays, int samples_per_ray,
                         float sample_dist, const float* bt_inv, const float* T_pose, const float* z_vals, const float* g_color,
                         const float* g_weight_sum, const float* g_gradient_error, float* g_params, float* g_inv_s, float* g_rays_o,
                         float* g_rays_d, float* g_bt_inv, float* g_T_pose, void* workspace, size_t workspace_bytes,
                         hn_stream_t stream) {
    HN_REQUIRE(f != nullptr && f->raw != nullptr, "field has no folded weights");
    return render_single_bwd_impl(f, rays_o, rays_d, n_rays, samples_per_ray, sample_dist, bt_inv, T_pose, z_vals, g_color,
                                  g_weight_sum, g_gradient_error, g_params, g_inv_s, g_rays_o, g_rays_d, g_bt_inv, g_T_pose, workspace,
                                  workspace_bytes, reinterpret_cast<hipStream_t>(stream), nullptr);
}
size_t hn_render_single_tape_bytes(const hn_field* f, int n_rays, int samples_per_ray) {
    if (f == nullptr || n_rays <= 0 || samples_per_ray <= 0 || !single_keep_applies(f)) return 0;
    return SingleKeep(f, (size_t)n_rays * samples_per_ray, nullptr).total;
}
int hn_render_single_bwd_taped(const hn_field* f, const float* rays_o, const float* rays_d, int n_rays, int samples_per_ray,
                               float sample_dist, const float* bt_inv, const float* T_pose, const float* z_vals, const float* g_color,
                               const float* g_weight_sum, const float* g_gradient_error, float* g_params, float* g_inv_s, float* g_rays_o,
                               float* g_rays_d, float* g_bt_inv, float* g_T_pose, const void* tape, size_t tape_bytes, void* workspace,
                               size_t workspace_bytes, hn_stream_t stream) {
    HN_REQUIRE(f != nullptr && f->raw != nullptr, "field has no folded weights");
    HN_REQUIRE(tape != nullptr, "hn_render_single_bwd_taped: tape is NULL");
    return render_single_bwd_impl(f, rays_o, rays_d, n_rays, samples_per_ray, sample_dist, bt_inv, T_pose, z_vals, g_color,
                                  g_weight_sum, g_gradient_error, g_params, g_inv_s, g_rays_o, g_rays_d, g_bt_inv, g_T_pose, workspace,
                                  workspace_bytes, reinterpret_cast<hipStream_t>(stream), nullptr, tape, tape_bytes);
}
float hn_field_inv_s(const hn_field* f) { return f ? f->inv_s : 0.f; }
int hn_field_set_inv_s_device(hn_field* f, const float* inv_s_dev) {
    if (f == nullptr) {
        hn::set_error("hn_field_set_inv_s_device: null field");
        return HN_EINVAL;
    }
    f->inv_s_dev = inv_s_dev;
    return HN_OK;
}
int hn_field_set_compaction(hn_field* f, int enabled) {
    if (f == nullptr) {
        hn::set_error("hn_field_set_compaction: null field");
        return HN_EINVAL;
    }
    f->compact_far_field = enabled ? 1 : 0;
    return HN_OK;
}
int hn_field_set_culling(hn_field* f, int enabled) {
    if (f == nullptr) {
        hn::set_error("hn_field_set_culling: null field");
        return HN_EINVAL;
    }
    f->cull_far_field = enabled ? 1 : 0;
    return HN_OK;
}

int hn_ray_gen(const float* xy, const float* R, const float* T, const float* focal, const float* principal, int n_cams,
               int rays_per_cam, float* rays_o, float* rays_d, hn_stream_t stream) {
    return ray_gen(xy, R, T, focal, principal, n_cams, rays_per_cam, rays_o, rays_d, (hipStream_t)stream);
}
int hn_obj_local_fwd(const float* rays_o, const float* rays_d, const float* Ro, const float* To, int n_frames,
                     int rays_per_frame, float* o_out, float* d_out, hn_stream_t stream) {
    return obj_local_fwd(rays_o, rays_d, Ro, To, n_frames, rays_per_frame, o_out, d_out, (hipStream_t)stream);
}
int hn_obj_local_bwd(const float* rays_o, const float* rays_d, const float* Ro, const float* To, const float* g_o_out,
                     const float* g_d_out, int n_frames, int rays_per_frame, float* g_rays_o, float* g_rays_d,
                     float* g_Ro, float* g_To, hn_stream_t stream) {
    return obj_local_bwd(rays_o, rays_d, Ro, To, g_o_out, g_d_out, n_frames, rays_per_frame, g_rays_o, g_rays_d, g_Ro,
                         g_To, (hipStream_t)stream);
}
int hn_coarse_z(const float* t_rand, int n_rays, int n_samples, double near, double far, float* z, hn_stream_t stream) {
    return coarse_z(t_rand, n_rays, n_samples, (float)near, (float)(far - near),
                    (float)((far - near) / (double)n_samples), z, (hipStream_t)stream);
}
int hn_sample_points(const float* rays_o, const float* rays_d, const float* z, int n_rays, int n, int mid,
                     float sample_dist, float* pts, float* dists, hn_stream_t stream) {
    return sample_points(rays_o, rays_d, z, n_rays, n, mid, sample_dist, pts, dists, (hipStream_t)stream);
}
int hn_sample_points_bwd(const float* z, const float* g_pts, int n_rays, int n, int mid, float sample_dist, float* g_rays_o,
                         float* g_rays_d, hn_stream_t stream) {
    return sample_points_bwd(z, g_pts, n_rays, n, mid, sample_dist, g_rays_o, g_rays_d, (hipStream_t)stream);
}
int hn_upsample(const float* z, const float* sdf, int n_rays, int k, int n_new, float inv_s, float* z_new,
                int64_t* inds, hn_stream_t stream) {
    return upsample(z, sdf, n_rays, k, n_new, inv_s, z_new, inds, (hipStream_t)stream);
}
int hn_merge(const float* z, const float* z_new, const float* sdf, const float* sdf_new, int n_rays, int k, int m,
             int quirk_rays_per_frame, float* z_out, float* sdf_out, int64_t* index, hn_stream_t stream) {
    return merge(z, z_new, sdf, sdf_new, n_rays, k, m, quirk_rays_per_frame, z_out, sdf_out, index, (hipStream_t)stream);
}
int hn_sort_rows(const float* v, int n_rays, int n, float* out, hn_stream_t stream) {
    return sort_rows(v, n_rays, n, out, (hipStream_t)stream);
}

size_t hn_field_workspace_bytes(const hn_field* f, int n_pts) { return f ? field_ws(f, n_pts) : 0; }

int hn_field_sdf(const hn_field* f, const float* pts, int n_pts, const float* bt_inv, const float* T_pose, int n_frames,
                 int pts_per_frame, float* sdf, void* workspace, size_t workspace_bytes, hn_stream_t stream) {
    HN_REQUIRE(f != nullptr, "null field");
    return field_sdf(f, pts, n_pts, bt_inv, T_pose, n_frames, pts_per_frame, sdf, workspace, workspace_bytes,
                     (hipStream_t)stream);
}
int hn_field_eval(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int samples_per_ray,
                  const float* bt_inv, const float* T_pose, int n_frames, int pts_per_frame, float* sdf, float* grad,
                  float* rgb, float* feat, void* workspace, size_t workspace_bytes, hn_stream_t stream) {
    HN_REQUIRE(f != nullptr, "null field");
    return field_eval(f, pts, rays_d, n_pts, samples_per_ray, bt_inv, T_pose, n_frames, pts_per_frame, sdf, grad, rgb,
                      feat, workspace, workspace_bytes, (hipStream_t)stream);
}

int hn_alpha(const float* sdf, const float* grad, const float* rays_d, const float* dists, int n_pts,
             int samples_per_ray, float inv_s, float* alpha_out, float* c, hn_stream_t stream) {
    return alpha(sdf, grad, rays_d, dists, n_pts, samples_per_ray, inv_s, alpha_out, c, (hipStream_t)stream);
}
int hn_composite1(const float* alpha_in, const float* c, const float* rgb, const float* grad, int n_rays, int S,
                  float* color, float* weights, float* weight_sum, float* weight_max, float* eik_sum,
                  hn_stream_t stream) {
    return composite1(alpha_in, c, rgb, grad, n_rays, S, color, weights, weight_sum, weight_max, eik_sum,
                      (hipStream_t)stream);
}
int hn_composite2(const float* alpha_h, const float* rgb_h, const float* grad_h, const float* alpha_o,
                  const float* rgb_o, const float* grad_o, int n_rays, int S, float* color, float* weight_sum,
                  float* w_hand, float* w_obj, float* eik_sum, hn_stream_t stream) {
    return composite2(alpha_h, rgb_h, grad_h, alpha_o, rgb_o, grad_o, n_rays, S, color, weight_sum, w_hand, w_obj,
                      eik_sum, (hipStream_t)stream);
}

int hn_alpha_bwd(const float* sdf, const float* grad, const float* rays_d, const float* dists, const float* g_alpha,
                 const float* g_c, int n_pts, int samples_per_ray, float inv_s, float* g_sdf, float* g_grad,
                 float* g_rays_d, hn_stream_t stream) {
    return alpha_bwd(sdf, grad, rays_d, dists, g_alpha, g_c, n_pts, samples_per_ray, inv_s, g_sdf, g_grad, g_rays_d,
                     (hipStream_t)stream);
}
int hn_composite1_bwd(const float* alpha_in, const float* c, const float* rgb, const float* g_color, const float* g_weight_sum,
                      int n_rays, int S, float* g_alpha, float* g_c, float* g_rgb, hn_stream_t stream) {
    return composite1_bwd(alpha_in, c, rgb, g_color, g_weight_sum, n_rays, S, g_alpha, g_c, g_rgb, (hipStream_t)stream);
}
int hn_composite2_bwd(const float* alpha_h, const float* rgb_h, const float* alpha_o, const float* rgb_o,
                      const float* g_color, const float* g_weight_sum, int n_rays, int S, float* g_alpha_h, float* g_rgb_h,
                      float* g_alpha_o, float* g_rgb_o, hn_stream_t stream) {
    return composite2_bwd(alpha_h, rgb_h, alpha_o, rgb_o, g_color, g_weight_sum, n_rays, S, g_alpha_h, g_rgb_h, g_alpha_o,
                          g_rgb_o, (hipStream_t)stream);
}

int hn_fit_loss_sums(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays,
                     const float* sdf_hand, const float* sdf_obj, int n_samples, float* sums6, hn_stream_t stream) {
    return fit_loss_sums(color, weight_sum, true_rgb, true_mask, n_rays, sdf_hand, sdf_obj, n_samples, sums6, (hipStream_t)stream);
}
int hn_fit_loss_grads(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays,
                      const float* sdf_hand, const float* sdf_obj, int n_samples, const float* sums6, const float* g4,
                      float* g_color, float* g_weight_sum, float* g_sdf_hand, float* g_sdf_obj, hn_stream_t stream) {
    return fit_loss_grads(color, weight_sum, true_rgb, true_mask, n_rays, sdf_hand, sdf_obj, n_samples, sums6, g4, g_color,
                          g_weight_sum, g_sdf_hand, g_sdf_obj, (hipStream_t)stream);
}

int hn_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const int* sizes, const float* lr, float beta1, float beta2, float eps, const int* steps, hn_stream_t stream) {
    return adam_step(n_tensors, params, grads, exp_avg, exp_avg_sq, sizes, lr, beta1, beta2, eps, steps, (hipStream_t)stream);
}
int hn_fit_total(const float* sums6, const float* verts_loss, const float* joint_3d, const float* joint3d_pred, int n_joints,
                 const float* weights5, float* terms8, float* g_joint, hn_stream_t stream) {
    return fit_total(sums6, verts_loss, joint_3d, joint3d_pred, n_joints, weights5, terms8, g_joint, (hipStream_t)stream);
}
int hn_mat3_inverse(const float* R, int n, float* out, hn_stream_t stream) { return hn::mat3_inverse(R, n, out, (hipStream_t)stream); }
int hn_mat3_inverse_bwd(const float* R_inv, const float* g_out, int n, float* g_R, hn_stream_t stream) {
    return hn::mat3_inverse_bwd(R_inv, g_out, n, g_R, (hipStream_t)stream);
}
int hn_stable_pts(const float* pts, int n_frames, int n_verts, int stride, const float* obj_r, const float* obj_t, float* pts_world, float* p0,
                  hn_stream_t stream) {
    return hn::stable_pts(pts, n_frames, n_verts, stride, obj_r, obj_t, pts_world, p0, (hipStream_t)stream);
}
int hn_stable_pts_bwd(const float* pts, int n_frames, int n_verts, int stride, const float* g_pts_world, float* g_obj_r, float* g_obj_t,
                      hn_stream_t stream) {
    return hn::stable_pts_bwd(pts, n_frames, n_verts, stride, g_pts_world, g_obj_r, g_obj_t, (hipStream_t)stream);
}
size_t hn_stable_value_scratch_bytes(int n_frames, int n_sel) { return hn::stable_value_scratch_bytes(n_frames, n_sel); }
int hn_stable_value(const float* sdf, const float* p0, int n_frames, int n_sel, int strict_reference, float* value, float* d_sdf, void* scratch,
                    size_t scratch_bytes, hn_stream_t stream) {
    return hn::stable_value(sdf, p0, n_frames, n_sel, strict_reference, value, d_sdf, scratch, scratch_bytes, (hipStream_t)stream);
}
size_t hn_window_loss_scratch_bytes(int n_rays, int n_samples) { return hn::window_loss_scratch_bytes(n_rays, n_samples); }
int hn_window_loss(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_hand,
                   const float* sdf_obj, int n_samples, const float* joint_3d, const float* joint3d_pred, int n_frames, const float* obj_r,
                   const float* obj_t, const float* Ro_pred, const float* To_pred, const float* verts, int n_verts, const float* stable, int anchor,
                   const float* weights7, void* scratch, size_t scratch_bytes, float* sums6, float* terms10, float* g_joint, float* gR, float* gt,
                   hn_stream_t stream) {
    return hn::window_loss(color, weight_sum, true_rgb, true_mask, n_rays, sdf_hand, sdf_obj, n_samples, joint_3d, joint3d_pred, n_frames, obj_r, obj_t,
                           Ro_pred, To_pred, verts, n_verts, stable, anchor, weights7, scratch, scratch_bytes, sums6, terms10, g_joint, gR, gt,
                           (hipStream_t)stream);
}
int hn_window_loss_bwd(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_hand,
                       const float* sdf_obj, int n_samples, const float* sums6, const float* g_loss, const float* weights7, const float* g_joint,
                       const float* gR, const float* gt, int n_frames, float* g_color, float* g_weight_sum, float* g_sdf_hand, float* g_sdf_obj,
                       float* g_joint_out, float* gR_out, float* gt_out, float* g_stable, hn_stream_t stream) {
    return hn::window_loss_bwd(color, weight_sum, true_rgb, true_mask, n_rays, sdf_hand, sdf_obj, n_samples, sums6, g_loss, weights7, g_joint, gR, gt,
                               n_frames, g_color, g_weight_sum, g_sdf_hand, g_sdf_obj, g_joint_out, gR_out, gt_out, g_stable, (hipStream_t)stream);
}
size_t hn_field_tape_bytes(const hn_field* f, int n_pts) { return (f == nullptr || n_pts <= 0) ? 0 : field_tape(f, n_pts); }
int hn_field_eval_taped(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int samples_per_ray, const float* bt_inv,
                        const float* T_pose, int n_frames, int pts_per_frame, float* sdf, float* grad, float* rgb, void* workspace,
                        size_t workspace_bytes, void* tape, size_t tape_bytes, hn_stream_t stream) {
    HN_REQUIRE(f != nullptr && tape != nullptr && tape_bytes >= field_tape(f, n_pts) && field_tape(f, n_pts) != 0,
               "hn_field_eval_taped: an HN_PREC_F16X3 field with its adjoint programs and a tape of hn_field_tape_bytes");
    return field_eval(f, pts, rays_d, n_pts, samples_per_ray, bt_inv, T_pose, n_frames, pts_per_frame, sdf, grad, rgb, nullptr, workspace,
                      workspace_bytes, (hipStream_t)stream, tape, tape_bytes);
}
int hn_field_eval_bwd_taped(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int samples_per_ray, const float* bt_inv,
                            const float* T_pose, int n_frames, int pts_per_frame, const float* g_sdf, const float* g_grad, const float* g_rgb,
                            const float* grad, const float* rgb, const void* tape, float* g_pts, float* g_rays_d, float* g_bt_inv, float* g_T_pose,
                            void* workspace, size_t workspace_bytes, hn_stream_t stream) {
    HN_REQUIRE(f != nullptr && tape != nullptr && g_grad != nullptr && g_rgb != nullptr && grad != nullptr && rgb != nullptr,
               "hn_field_eval_bwd_taped: tape, upstream gradients of all three outputs, and the evaluation's grad / rgb");
    return bwd::field_eval_bwd(f, pts, rays_d, n_pts, samples_per_ray, bt_inv, T_pose, n_frames, pts_per_frame, g_sdf, g_grad, g_rgb, g_pts, g_rays_d,
                               g_bt_inv, g_T_pose, workspace, workspace_bytes, (hipStream_t)stream, tape, grad, rgb);
}
size_t hn_fit_step_loss_scratch_bytes(int n_rays, int n_samples) { return hn::fit_step_loss_scratch_bytes(n_rays, n_samples); }
int hn_fit_step_loss(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_hand,
                     const float* sdf_obj, int n_samples, const float* joint_3d, const float* joint3d_pred, int n_joints, const float* Ra, const float* ta,
                     const float* Rb, const float* tb, const float* verts, int n_verts, const float* weights5, void* scratch, size_t scratch_bytes,
                     float* sums6, float* terms8, float* g_joint, float* gR, float* gt, hn_stream_t stream) {
    return hn::fit_step_loss(color, weight_sum, true_rgb, true_mask, n_rays, sdf_hand, sdf_obj, n_samples, joint_3d, joint3d_pred, n_joints, Ra, ta, Rb, tb,
                             verts, n_verts, weights5, scratch, scratch_bytes, sums6, terms8, g_joint, gR, gt, (hipStream_t)stream);
}
int hn_fit_step_loss_bwd(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_hand,
                         const float* sdf_obj, int n_samples, const float* sums6, const float* g_loss, const float* weights5, const float* g_joint,
                         const float* gR, const float* gt, int n_joints, float* g_color, float* g_weight_sum, float* g_sdf_hand, float* g_sdf_obj,
                         float* g_joint_out, float* gR_out, float* gt_out, hn_stream_t stream) {
    return hn::fit_step_loss_bwd(color, weight_sum, true_rgb, true_mask, n_rays, sdf_hand, sdf_obj, n_samples, sums6, g_loss, weights5, g_joint, gR, gt,
                                 n_joints, g_color, g_weight_sum, g_sdf_hand, g_sdf_obj, g_joint_out, gR_out, gt_out, (hipStream_t)stream);
}
int hn_fit_step_loss_frames(int n_frames, const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays,
                            const float* sdf_hand, const float* sdf_obj, int n_samples, const float* joint_3d, const float* joint3d_pred, int n_joints,
                            const float* Ra, const float* ta, const float* Rb, const float* tb, const float* const* verts, const int* n_verts,
                            const float* weights5, void* scratch, size_t scratch_bytes, float* sums6, float* terms8, float* g_joint, float* gR, float* gt,
                            hn_stream_t stream) {
    return hn::fit_step_loss_frames(n_frames, color, weight_sum, true_rgb, true_mask, n_rays, sdf_hand, sdf_obj, n_samples, joint_3d, joint3d_pred, n_joints,
                                    Ra, ta, Rb, tb, verts, n_verts, weights5, scratch, scratch_bytes, sums6, terms8, g_joint, gR, gt, (hipStream_t)stream);
}
int hn_fit_step_loss_bwd_frames(int n_frames, const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays,
                                const float* sdf_hand, const float* sdf_obj, int n_samples, const float* sums6, const float* g_loss,
                                const float* weights5, const float* g_joint, const float* gR, const float* gt, int n_joints, float* g_color,
                                float* g_weight_sum, float* g_sdf_hand, float* g_sdf_obj, float* g_joint_out, float* gR_out, float* gt_out,
                                hn_stream_t stream) {
    return hn::fit_step_loss_bwd_frames(n_frames, color, weight_sum, true_rgb, true_mask, n_rays, sdf_hand, sdf_obj, n_samples, sums6, g_loss, weights5,
                                        g_joint, gR, gt, n_joints, g_color, g_weight_sum, g_sdf_hand, g_sdf_obj, g_joint_out, gR_out, gt_out,
                                        (hipStream_t)stream);
}
int hn_variance_to_inv_s(const float* variance, float* inv_s, hn_stream_t stream) { return hn::variance_to_inv_s(variance, inv_s, (hipStream_t)stream); }
int hn_variance_chain(const float* g_inv_s, const float* inv_s, float* g_variance, hn_stream_t stream) {
    return hn::variance_chain(g_inv_s, inv_s, g_variance, (hipStream_t)stream);
}
int hn_train_loss(const float* color, const float* weight_sum, const float* gradient_error, const float* true_rgb, const float* true_mask, int n_rays,
                  float igr_weight, float mask_weight, float* terms6, hn_stream_t stream) {
    return hn::train_loss(color, weight_sum, gradient_error, true_rgb, true_mask, n_rays, igr_weight, mask_weight, terms6, (hipStream_t)stream);
}
int hn_train_loss_bwd(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays, const float* terms6,
                      const float* g_loss, float igr_weight, float mask_weight, float* g_color, float* g_weight_sum, float* g_gradient_error,
                      hn_stream_t stream) {
    return hn::train_loss_bwd(color, weight_sum, true_rgb, true_mask, n_rays, terms6, g_loss, igr_weight, mask_weight, g_color, g_weight_sum,
                              g_gradient_error, (hipStream_t)stream);
}
int hn_fit_total_bwd(const float* g_loss, const float* weights5, const float* g_joint, const float* gR, const float* gt, int n_joints,
                     float* g4, float* g_joint_out, float* gR_out, float* gt_out, hn_stream_t stream) {
    return fit_total_bwd(g_loss, weights5, g_joint, gR, gt, n_joints, g4, g_joint_out, gR_out, gt_out, (hipStream_t)stream);
}

size_t hn_render_single_workspace_bytes(const hn_field* f, int n_rays, int n_samples, int n_importance) {
    size_t need = 0;
    if (f == nullptr) return 0;
    if (render_single_impl(f, nullptr, nullptr, nullptr, n_rays, 0.0, 1.0, n_samples, n_importance,
                           n_importance > 0 ? 1 : 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, 0, nullptr, &need) != HN_OK)
        return 0;
    return need;
}
int hn_render_single(const hn_field* f, const float* rays_o, const float* rays_d, const float* t_rand, int n_rays,
                     double near, double far, int n_samples, int n_importance, int up_sample_steps, const float* bt_inv,
                     const float* T_pose, float* color, float* cdf, float* weight_sum, float* weight_max,
                     float* gradient_error, float* z_vals, void* workspace, size_t workspace_bytes, hn_stream_t stream) {
    HN_REQUIRE(f != nullptr, "null field");
    return render_single_impl(f, rays_o, rays_d, t_rand, n_rays, near, far, n_samples, n_importance, up_sample_steps,
                              bt_inv, T_pose, color, cdf, weight_sum, weight_max, gradient_error, z_vals, workspace,
                              workspace_bytes, (hipStream_t)stream, nullptr);
}

int hn_render_single_taped(const hn_field* f, const float* rays_o, const float* rays_d, const float* t_rand, int n_rays,
                           double near, double far, int n_samples, int n_importance, int up_sample_steps, const float* bt_inv,
                           const float* T_pose, float* color, float* cdf, float* weight_sum, float* weight_max,
                           float* gradient_error, float* z_vals, void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                           hn_stream_t stream) {
    HN_REQUIRE(f != nullptr, "null field");
    HN_REQUIRE(tape != nullptr, "hn_render_single_taped: tape is NULL");
    return render_single_impl(f, rays_o, rays_d, t_rand, n_rays, near, far, n_samples, n_importance, up_sample_steps,
                              bt_inv, T_pose, color, cdf, weight_sum, weight_max, gradient_error, z_vals, workspace,
                              workspace_bytes, (hipStream_t)stream, nullptr, tape, tape_bytes);
}

size_t hn_render_dual_workspace_bytes(const hn_field* hand, const hn_field* obj, int n_rays, int n_samples,
                                      int n_importance) {
    size_t need = 0;
    if (hand == nullptr || obj == nullptr) return 0;
    if (render_dual_impl(hand, obj, nullptr, nullptr, nullptr, 1, n_rays, 0.0, 1.0, n_samples, n_importance,
                         n_importance > 0 ? 1 : 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &need) != HN_OK)
        return 0;
    return need;
}
size_t hn_render_dual_bwd_workspace_bytes(const hn_field* hand, const hn_field* obj, int n_rays, int samples_per_ray) {
    size_t need = 0;
    if (hand == nullptr || obj == nullptr) return 0;
    if (render_dual_bwd_impl(hand, obj, nullptr, nullptr, 1, n_rays, samples_per_ray, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                             &need) != HN_OK)
        return 0;
    return need;
}
int hn_render_dual_bwd(const hn_field* hand, const hn_field* obj, const float* rays_o, const float* rays_d, int n_frames,
                       int rays_per_frame, int samples_per_ray, float sample_dist, const float* bt_inv, const float* T_pose,
                       const float* Ro, const float* To, const float* z_vals, const float* sdf_hand, const float* grad_hand,
                       const float* rgb_hand, const float* alpha_hand, const float* sdf_obj, const float* grad_obj,
                       const float* rgb_obj, const float* alpha_obj, const float* g_color, const float* g_weight_sum,
                       const float* g_sdf_hand, const float* g_sdf_obj, const float* g_grad_hand, const float* g_grad_obj,
                       const float* g_gradient_error, float* g_rays_o, float* g_rays_d, float* g_bt_inv, float* g_T_pose,
                       float* g_Ro, float* g_To, void* workspace, size_t workspace_bytes, const void* tape, int flags, hn_stream_t stream) {
    HN_REQUIRE(hand != nullptr && obj != nullptr, "null field");
    return render_dual_bwd_impl(hand, obj, rays_o, rays_d, n_frames, rays_per_frame, samples_per_ray, sample_dist, bt_inv, T_pose, Ro,
                                To, z_vals, sdf_hand, grad_hand, rgb_hand, alpha_hand, sdf_obj, grad_obj, rgb_obj, alpha_obj,
                                g_color, g_weight_sum, g_sdf_hand, g_sdf_obj, g_grad_hand, g_grad_obj, g_gradient_error, g_rays_o,
                                g_rays_d, g_bt_inv, g_T_pose, g_Ro, g_To, workspace, workspace_bytes, (hipStream_t)stream, nullptr, tape, flags);
}
int hn_render_dual_aux_offsets(const hn_field* hand, const hn_field* obj, int n_rays, int n_samples, int n_importance,
                               int up_sample_steps, size_t* offsets4) {
    HN_REQUIRE(hand != nullptr && obj != nullptr && offsets4 != nullptr, "null argument");
    size_t need = 0;
    return render_dual_impl(hand, obj, nullptr, nullptr, nullptr, 1, n_rays, 0.0, 1.0, n_samples, n_importance,
                            n_importance > 0 ? up_sample_steps : 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                            nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &need, offsets4);
}
int hn_render_dual_compact_offsets(const hn_field* hand, const hn_field* obj, int n_rays, int n_samples, int n_importance,
                                   int up_sample_steps, size_t* offsets2) {
    HN_REQUIRE(hand != nullptr && obj != nullptr && offsets2 != nullptr, "null argument");
    size_t need = 0;
    return render_dual_impl(hand, obj, nullptr, nullptr, nullptr, 1, n_rays, 0.0, 1.0, n_samples, n_importance,
                            n_importance > 0 ? up_sample_steps : 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                            nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &need, nullptr, nullptr, 0, offsets2);
}
int hn_render_dual(const hn_field* hand, const hn_field* obj, const float* rays_o, const float* rays_d,
                   const float* t_rand, int n_frames, int rays_per_frame, double near, double far, int n_samples,
                   int n_importance, int up_sample_steps, const float* bt_inv, const float* T_pose, const float* Ro,
                   const float* To, int batch_quirk, float* color, float* weight_sum, float* sdf_hand, float* sdf_obj,
                   float* grad_hand, float* grad_obj, float* gradient_error, float* z_vals, void* workspace,
                   size_t workspace_bytes, void* tape, size_t tape_bytes, int flags, hn_stream_t stream) {
    HN_REQUIRE(hand != nullptr && obj != nullptr, "null field");
    return render_dual_impl(hand, obj, rays_o, rays_d, t_rand, n_frames, rays_per_frame, near, far, n_samples,
                            n_importance, up_sample_steps, bt_inv, T_pose, Ro, To, batch_quirk, color, weight_sum,
                            sdf_hand, sdf_obj, grad_hand, grad_obj, gradient_error, z_vals, workspace, workspace_bytes,
                            (hipStream_t)stream, nullptr, nullptr, tape, tape_bytes, nullptr, flags);
}
size_t hn_render_dual_tape_bytes(const hn_field* hand, const hn_field* obj, int n_rays, int samples_per_ray) {
    if (hand == nullptr || obj == nullptr || n_rays <= 0 || samples_per_ray <= 0) return 0;
    const size_t N = (size_t)n_rays * samples_per_ray;
    const size_t t = field_tape(hand, hand_cap(hand, N)) + field_tape(obj, (int)N);
    // + rgb / alpha of both fields (+ the compaction record of the hand's samples, hn_field_set_compaction)
    return t == 0 ? 0 : t + 8 * N * sizeof(float) + (hand->compact_far_field ? CompactRec::bytes(N) : 0);
}
size_t hn_render_dual_tape_aux_offset(const hn_field* hand, const hn_field* obj, int n_rays, int samples_per_ray) {
    if (hand == nullptr || obj == nullptr || n_rays <= 0 || samples_per_ray <= 0) return 0;
    const size_t N = (size_t)n_rays * samples_per_ray;
    return field_tape(hand, hand_cap(hand, N)) + field_tape(obj, (int)N);
}

}  // extern "C"
