// v2 MLP building blocks for gfx950: fp16 hi/lo split operands on v_mfma_f32_32x32x16_f16
// ("f16x3": fp32-equivalent products at 16/3 x the f32-MFMA rate), weights streamed
// global -> LDS by DMA (global_load_lds_dwordx4) and shared by the 4 waves of a workgroup,
// activations resident in registers as MFMA B operands.
//
// Numerics.  x = hi + lo/2048 with hi = fp16(x) and lo = fp16((x - hi) * 2048): 22 mantissa
// bits, no range problem for the small second term.  A product W x is evaluated as
//   C1 += Whi xhi            C2 += Whi xlo + Wlo xhi            W x ~= C1 + C2 / 2048
// in fp32 accumulators (the lo*lo term, 2^-22 relative, is dropped): measured 5.6e-7 of the
// result's max on the reference's layers, the same as an fp32 matmul (5.5e-7).
//
// Layout.  One wave owns 32 samples; a workgroup is 4 waves = 128 samples and one workgroup
// runs per CU (512-register kernel).  Activations are [neurons x samples] accumulator tiles of
// the 32x32 MFMA: lane l holds column (sample) l&31; register i of lane half h = l>>5 holds row
// (i&3) + 8 (i>>2) + 4 h.  Registers 8u..8u+7 of tile t, converted to fp16, ARE the B fragment of
// k-step 2t+u of the next layer (element j <-> neuron 32t + 16u + 8 (j>>2) + 4h + (j&3)); the
// packed weights use the same k order, so nothing moves between lanes or through LDS.
//
// Weight stream.  The packer (hn_pack2.hip) lays every matrix out as CHUNKS in the exact order
// the kernel consumes them: chunk = TILES x KS k-step blocks of 2 KiB ([hi fragment 1 KiB][lo
// fragment 1 KiB], lane-linear, so one ds_read_b128 per fragment is conflict-free) + an optional
// 1 KiB tail of fp32 side data (bias of the tile, last-layer rows).  Two LDS buffers: the DMA of
// chunk i+1 is issued right after the barrier that publishes chunk i.
#pragma once
#include <type_traits>
#include <utility>

#include "hn_common.h"

// Timing / bisecting hooks of the field kernels exist only in -DHN_DEBUG_HOOKS builds (tools/); the product
// library carries none of them: HN_DBG(a) folds to 0 and the branches that test it disappear.
#ifdef HN_DEBUG_HOOKS
#define HN_DBG(a) ((a).dbg)
#else
#define HN_DBG(a) 0
#endif

namespace hn {
namespace v2 {

using h8 = _Float16 __attribute__((ext_vector_type(8)));
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int KS_BYTES = 2048;            // one k-step block: hi fragment + lo fragment
constexpr int TAIL_BYTES = 1024;          // 256 floats of side data
constexpr int CHUNK_MAX = 9 * 4096;   // 36 KiB: a chunk is at most 33 KiB; every wave always issues 9 pieces (WStream::piece)
constexpr float LO_SCALE = 2048.f;
constexpr float LO_INV = 1.f / 2048.f;
// The reverse sweep carries d sdf / d z scaled by 2^8 (exact): its entries are small in ABSOLUTE terms
// (1e-5 .. 1e-2), and whatever falls below the fp16 normal range keeps only the 11 bits of the lo part
// (see hi_part); scaled, that floor drops to 2.4e-7 while |dz| < 255 stays clear of fp16 overflow.
constexpr float BWD_SCALE = 256.f;
constexpr float BWD_INV = 1.f / 256.f;
constexpr int WG_WAVES = 4;
constexpr int WG_SAMPLES = 32 * WG_WAVES;

__host__ __device__ constexpr int chunk_bytes(int tiles, int ks, bool tail) {
    return tiles * ks * KS_BYTES + (tail ? TAIL_BYTES : 0);
}

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// lane * 16, recomputed where it is needed (3 VALU instructions) instead of being kept in a register for the whole
// kernel: as a long-lived, rarely-read value it is the register allocator's first choice for a spill in these
// 512-register kernels, and the reload -- a scratch_load with an s_waitcnt vmcnt(0) behind it in front of every stash
// access -- drains the whole memory pipeline each time (measured: every stash prefetch serialised).  asm volatile:
// the result must not be CSE'd back into one long-lived value.
__device__ __forceinline__ int lane_x16() {
    int v;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(v));
    return v;
}

// ---- MFMA shape ------------------------------------------------------------------------------------------------
// HN_MFMA16 (per translation unit): the matrix products run on v_mfma_f32_16x16x32_f16 instead of 32x32x16.  Same
// MACs per cycle, same A-fragment bytes from LDS, same register counts -- but the chip holds a higher clock on this
// shape and the wave loses fewer cycles around it (tools/cpp/layer_bench.hip, -DHN_EXP_MFMA16: a hidden layer 17 - 22 %
// shorter in wall time, a feature pass 32 %; profiles/r02/README.md).  What changes is which lane holds what:
//   32x32x16 ("old"): lane = (sample c of 32, half hh);  a C tile's register i is row (i&3) + 8 (i>>2) + 4 hh;  a B
//                     fragment (k-step s of 16) holds k = 16 s + 8 hh + j.
//   16x16x32 ("new"): lane = (c16, g) with g = lane >> 4;  a 32-row x 32-sample tile is four 16x16 blocks, register
//                     i = 8 rb + 4 cb + ii is row 16 rb + 4 g + ii of sample 16 cb + c16;  the B fragment [2 s' + cb] of
//                     k-step pair s' holds k = 32 s' + 8 g + j of sample 16 cb + c16.
// The slot numbering kappa = 16 s + 8 h + j = 32 s' + 8 g + j (s = 2 s' + (g >> 1), h = g & 1) is common to both, and
// so is the rule that a finished tile's registers ARE next layer's B fragments (new: fragment cb, element 4 rb + ii).
// Everything inside the MLP stack is layout-agnostic (element-wise epilogues, stash tiles written and read in the
// same layout); the per-sample code around it (encodings, Jacobian contractions, output sums) is written for the old
// lane <-> sample map and crosses over with the two exchanges below: one v_permlane32_swap + one v_permlane16_swap
// per register pair.
#ifndef HN_MFMA16
#define HN_MFMA16 0
#endif
constexpr bool S16 = HN_MFMA16 != 0;
// (x0, x1): the two registers of a pair -- fragment dwords of k-steps (2 s', 2 s' + 1), or tile registers (i, i + 4).
// old -> new: new x0 = the cb 0 member, x1 = the cb 1 member.  Rows = 16 lanes: row q + 2 hh <-> g.
__device__ __forceinline__ void rows_to16(unsigned& x0, unsigned& x1) {
    const auto r = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);      // x0 rows 2,3 <-> x1 rows 0,1
    const auto q = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false);  // x0 rows 1,3 <-> x1 rows 0,2
    x0 = q[0];
    x1 = q[1];
}
__device__ __forceinline__ void rows_to32(unsigned& x0, unsigned& x1) {   // the inverse
    const auto q = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
    const auto r = __builtin_amdgcn_permlane32_swap(q[0], q[1], false, false);
    x0 = r[0];
    x1 = r[1];
}

// ---- weight stream ---------------------------------------------------------------------------
// The stream is cyclic (after the last chunk of a sample tile comes the first again) and is read
// through a buffer descriptor: buffer_load_dwordx4 ... offen lds with a scalar byte offset per 1 KiB
// piece and M0 as the LDS destination -- two SALU instructions and one VMEM instruction per piece, no
// 64-bit vector address arithmetic.  Piece p of a chunk is fetched by wave p % 4.
//
// Protocol per chunk i (all waves run the same sequence):
//   acquire()       s_waitcnt vmcnt(0): this wave's pieces of chunk i have landed; s_barrier: everybody's
//                   pieces landed, everybody finished reading chunk i-1
//   begin(bytes)    chunk i+1 goes to the other buffer ...
//   piece(k)        ... one 1 KiB piece at a time, spread over the MFMA slots of chunk i by mma_tile
#ifdef HN_TS
// timing aid (tools/ts_report.py): workgroup 0 records s_memtime stamps at the chunk barrier and tile ends
static __device__ unsigned long long g_hn_ts[4 * 8192];
#endif
struct WStream {
#ifdef HN_TS
    int ts_n, ts_on;
    __device__ __forceinline__ void stamp(int id) {
        if (ts_on) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (voff == 0 && ts_n < 8192) g_hn_ts[wave * 8192 + ts_n] = (t & 0x0fffffffffffffffull) | ((unsigned long long)id << 60);
            ts_n++;
        }
    }
#else
    __device__ __forceinline__ void stamp(int) {}
#endif
    __amdgpu_buffer_rsrc_t rsrc;
    int total;          // stream bytes
    int goff;           // byte offset of the next chunk to fetch
    char* lds;          // two buffers of CHUNK_MAX bytes
    int phase;          // buffer that receives the next fetch
    int wave, voff;     // wave id (uniform), lane * 16
    int voff_oob;       // lane * 16 + 0x7f000000: beyond num_records for every stream
    // the fetch in progress
    int f_goff, f_pieces;
    char* f_dst;
    int dbg_nofetch;    // timing experiments only

    __device__ __forceinline__ void init(const char* blob, size_t bytes, char* lds_base, int wave_, int lane) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(blob), 0, (int)bytes, 0x00020000);
        total = (int)bytes;
        goff = 0;
        lds = lds_base;
        phase = 0;
        wave = wave_;
        voff = lane * 16;
        voff_oob = lane * 16 + 0x7f000000;
        f_pieces = 0;
        f_goff = 0;
        f_dst = lds_base;
        dbg_nofetch = 0;
#ifdef HN_TS
        ts_n = 0;
        ts_on = blockIdx.x == 0;
#endif
    }
    __device__ __forceinline__ void begin(int bytes) {
        if (goff == total) goff = 0;
        f_goff = goff + wave * 1024;
        f_dst = lds + phase * CHUNK_MAX + wave * 1024;
#ifdef HN_DEBUG_HOOKS
        f_pieces = dbg_nofetch ? 0 : ((bytes >> 10) + WG_WAVES - 1 - wave) / WG_WAVES;
#else
        f_pieces = ((bytes >> 10) + WG_WAVES - 1 - wave) / WG_WAVES;   // pieces of this wave
#endif
        goff += bytes;
        phase ^= 1;
    }
    // the next chunk is not the sequentially next one (half-layer passes, skipped bones)
    __device__ __forceinline__ void begin_at(int off, int bytes) {
        goff = off;
        begin(bytes);
    }
    // k-th piece of this wave (k < 9: a chunk has at most 33 pieces).  Branch-free on purpose: a piece the
    // chunk does not have is issued with an out-of-range VGPR offset (the buffer range check drops it: no
    // memory traffic; its LDS destination lies in the unused tail of the 36 KiB buffer).  A uniform branch
    // here would end the basic block inside the MFMA stream, and the compiler then sinks the epilogue slices
    // of all earlier slots below the last branch (measured: ~100 VALU instructions in one lump per chunk).
    // BRANCHY: the plain conditional form.  Kept for the layers where the lump it causes happens to give the
    // register allocator a shorter live range than the interleaved order (hn_field2_hand.hip, reverse sweep).
    template <bool BRANCHY = false>
    __device__ __forceinline__ void piece(int k) {
        if constexpr (BRANCHY) {
            if (k < f_pieces) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(f_dst + k * 4096), 16, lane_x16(), f_goff + k * 4096, 0, 0);
            return;
        }
        const int l16 = lane_x16();
        const int vo = (k < f_pieces) ? l16 : l16 + 0x7f000000;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(f_dst + k * 4096), 16, vo, f_goff + k * 4096, 0, 0);
    }
    // ---- compile-time-sized fetch ------------------------------------------------------------------------------
    // Where the size of the next chunk is a constant (every layer of the forward kernels), the fetch needs no
    // per-piece presence logic: the chunk's whole 4 KiB multiples ("body") are split into four contiguous quarters,
    // wave w moving quarter w in 1 KiB x4 pieces, and each remaining 1 KiB (tails) is moved by all four waves
    // together, 256 B each, with dword pieces.  The lane offset is the caller's (the MFMA tile's own lane * 16), pieces
    // of one 4 KiB group share their scalar offset and M0 and differ in the instruction's immediate offset: per piece
    // one VMEM instruction and ~0.5 SALU, against 3 + 2 VALU, 2 SALU and the VMEM instruction of the run-time form
    // (one wave per SIMD: every instruction of any kind costs the wave's in-order stream ~4 cycles beside the MFMAs).
    int c_goff, c_rgoff;    // this wave's global offsets: body, remainder
    char *c_dst, *c_rdst;   // and LDS destinations
    template <int BYTES>
    static constexpr int body_pieces() { return BYTES / 4096; }
    template <int BYTES>
    static constexpr int rem_pieces() { return (BYTES % 4096) / 1024; }
    template <int BYTES>
    static constexpr int n_pieces() { return body_pieces<BYTES>() + rem_pieces<BYTES>(); }
    template <int BYTES>
    __device__ __forceinline__ void begin_c() {
        static_assert(BYTES % 1024 == 0 && BYTES <= CHUNK_MAX, "chunk size");
        if (goff == total) goff = 0;
        constexpr int Q = body_pieces<BYTES>() * 1024;   // bytes per wave in the body
        char* const dst = lds + phase * CHUNK_MAX;
        c_goff = goff + wave * Q;
        c_dst = dst + wave * Q;
        c_rgoff = goff + 4 * Q + wave * 256;
        c_rdst = dst + 4 * Q + wave * 256;
        goff += BYTES;
        phase ^= 1;
    }
    template <int BYTES>
    __device__ __forceinline__ void begin_at_c(int off) {
        goff = off;
        begin_c<BYTES>();
    }
    // piece K of n_pieces<BYTES>(); l16 = lane * 16
    template <int BYTES, int K>
    __device__ __forceinline__ void piece_c(int l16) {
        constexpr int NB = body_pieces<BYTES>();
        static_assert(K < n_pieces<BYTES>(), "piece index");
        if constexpr (K < NB) {
            constexpr int g = K / 4, i = K % 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(c_dst + g * 4096), 16, l16, c_goff + g * 4096, i * 1024, 0);
        } else {
            constexpr int j = K - NB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(c_rdst + j * 1024), 4, l16 >> 2, c_rgoff + j * 1024, 0, 0);
        }
    }
    template <int BYTES>
    __device__ __forceinline__ void pieces_all_c() {
        const int l16 = lane_x16();
        static_for<n_pieces<BYTES>()>([&](auto K) { piece_c<BYTES, decltype(K)::value>(l16); });
    }
    template <int BYTES>
    __device__ __forceinline__ void fetch_all_c() {
        begin_c<BYTES>();
        const int l16 = lane_x16();
        static_for<n_pieces<BYTES>()>([&](auto K) { piece_c<BYTES, decltype(K)::value>(l16); });
    }
    __device__ __forceinline__ void pieces_all() {
#pragma unroll
        for (int k = 0; k < 9; ++k) piece(k);
    }
    __device__ __forceinline__ void fetch_all(int bytes) {
        begin(bytes);
        pieces_all();
    }
    // VISIBLE: issue the drain as the s_waitcnt builtin instead of inline asm, so that the compiler's own wait-count
    // bookkeeping sees it.  With the opaque asm it still believes every earlier register load may be pending and
    // (the DMA pieces being conditional) guards the first use with a counted vmcnt that can only be met by draining
    // the loads issued AFTER them.  That matters where register loads are issued one step ahead (the hand field's
    // feature passes: ~2000 cycles per bone); elsewhere the visible form changes the register allocation for the
    // worse (reverse sweep 123 -> 202 us per tile), so it is opt-in.
    // gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14.
    template <int ALLOW, bool VISIBLE = false>
    __device__ __forceinline__ const char* acquire() {
        stamp(1);
        if constexpr (VISIBLE) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt((ALLOW & 15) | (7 << 4) | (15 << 8) | ((ALLOW >> 4) << 14));
        } else {
#ifndef HN_EXP_NOWAIT
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ALLOW) : "memory");
#endif
        }
        stamp(2);
#ifndef HN_EXP_NOBAR   // timing experiments only (tools/cpp/layer_bench.hip)
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        stamp(3);
        return lds + (phase ^ 1) * CHUNK_MAX;
    }
};
constexpr int MAX_PIECES_PER_WAVE = 9;

// ---- fragments -------------------------------------------------------------------------------
// hi part of the split on the DEVICE.  Values below the fp16 normal range (2^-14) get hi = 0 and go
// entirely into the scaled lo part (11 significant bits, absolute error < 1.5e-8): the MFMA reads an fp16
// subnormal as its IEEE value, but the conversions back to fp32 that form the residual (plain, SDWA or
// mix forms, the compiler's choice) do not all agree on subnormals -- measured: a hi part read as x by
// the matrix pipe and as 0 by the residual is an error of up to 6e-5 per element.  (The host-side weight
// packer converts with IEEE semantics and needs no such rule.)
#ifndef HN_F16_FLUSH
#define HN_F16_FLUSH 1
#endif
__host__ __device__ __forceinline__ _Float16 hi_part(float x) {
#if HN_F16_FLUSH && defined(__HIP_DEVICE_COMPILE__)
    return (_Float16)x;   // the kernel runs with fp16 denormals flushed (f16_flush_mode): the conversion itself gives 0
#else
    const float xs = (x < 6.103515625e-5f && x > -6.103515625e-5f) ? 0.f : x;   // select in fp32, then convert
    return (_Float16)xs;
#endif
}
// MODE.FP_DENORM[7:6] (fp16/fp64) = 0: flush input and output denormals; [5:4] (fp32) stays 3 (IEEE)
__device__ __forceinline__ void f16_flush_mode() {
#if HN_F16_FLUSH
    __builtin_amdgcn_s_setreg(((4 - 1) << 11) | (4 << 6) | 1, 3);
#endif
}
// The split of a PAIR of fp32 values in 5 VALU instructions (the plain form takes 10: two conversions each way, two
// subtractions, two multiplications, two packs): hi pair = v_cvt_pk_f16_f32 (RTNE, like the scalar conversion);
// residuals straight from the packed halves with v_fma_mix_f32 (x - float(hi), exact in fp32); scaled lo halves
// written in place by v_fma_mixlo/hi_f16 (r * 2048 rounded once to fp16 -- the same value as the separate multiply
// and conversion, the multiplication by a power of two being exact).  HN_SPLIT_MIX=0 keeps the plain form.
#ifndef HN_SPLIT_MIX
#define HN_SPLIT_MIX 1
#endif
// (one asm statement per group: between separate, dependent asm statements the compiler puts an s_nop -- it cannot see
// what they contain -- and every instruction of any kind is an issue slot of the one wave that also feeds the MFMAs)
__device__ __forceinline__ void split_pair_hi(float a, float b, unsigned& hp, float& r0, float& r1) {
    asm("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
        "v_fma_mix_f32 %1, %0, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hp), "=&v"(r0), "=&v"(r1)
        : "v"(a), "v"(b));
}
__device__ __forceinline__ unsigned split_pair_lo(float r0, float r1, float scale) {
    unsigned lp;
    asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]"
        : "=&v"(lp)
        : "v"(r0), "v"(r1), "v"(scale));
    return lp;
}
using u32x4_ = unsigned __attribute__((ext_vector_type(4)));
template <int J2>   // dword J2 of the fragment (halves 2*J2, 2*J2+1)
__device__ __forceinline__ void set_pair(h8& f, unsigned v) {
    u32x4_ t = __builtin_bit_cast(u32x4_, f);
    t[J2] = v;
    f = __builtin_bit_cast(h8, t);
}
// old <-> new layout of a pair of B fragments (k-steps 2 s', 2 s' + 1  <->  [cb 0], [cb 1]) and of an accumulator tile
__device__ __forceinline__ void frags_to16(h8& f0, h8& f1) {
    u32x4_ a = __builtin_bit_cast(u32x4_, f0), b = __builtin_bit_cast(u32x4_, f1);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned x0 = a[d], x1 = b[d];
        rows_to16(x0, x1);
        a[d] = x0;
        b[d] = x1;
    }
    f0 = __builtin_bit_cast(h8, a);
    f1 = __builtin_bit_cast(h8, b);
}
__device__ __forceinline__ void frags_to32(h8& f0, h8& f1) {
    u32x4_ a = __builtin_bit_cast(u32x4_, f0), b = __builtin_bit_cast(u32x4_, f1);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned x0 = a[d], x1 = b[d];
        rows_to32(x0, x1);
        a[d] = x0;
        b[d] = x1;
    }
    f0 = __builtin_bit_cast(h8, a);
    f1 = __builtin_bit_cast(h8, b);
}
template <bool TO16>
__device__ __forceinline__ void tile_convert(f32x16& t) {
#pragma unroll
    for (int A = 0; A < 2; ++A)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            // (through scalars: __builtin_bit_cast applied to a vector-element expression reads element 0)
            const float f0 = t[8 * A + ii], f1 = t[8 * A + 4 + ii];
            unsigned x0 = __builtin_bit_cast(unsigned, f0), x1 = __builtin_bit_cast(unsigned, f1);
            if constexpr (TO16)
                rows_to16(x0, x1);
            else
                rows_to32(x0, x1);
            t[8 * A + ii] = __builtin_bit_cast(float, x0);
            t[8 * A + 4 + ii] = __builtin_bit_cast(float, x1);
        }
}
// Interface helpers of the kernels: no-ops on the 32x32x16 shape.  in: fragments / tile in the old (per-sample code's)
// layout -> the MFMA layout of this translation unit; out: the reverse.
__device__ __forceinline__ void frags_in(h8& f0, h8& f1) {
    if constexpr (S16) frags_to16(f0, f1);
}
__device__ __forceinline__ void frags_out(h8& f0, h8& f1) {
    if constexpr (S16) frags_to32(f0, f1);
}
__device__ __forceinline__ void tile_in(f32x16& t) {
    if constexpr (S16) tile_convert<true>(t);
}
__device__ __forceinline__ void tile_out(f32x16& t) {
    if constexpr (S16) tile_convert<false>(t);
}
// which fragment (of the tile's two) and which element register i of a finished tile becomes
__host__ __device__ constexpr int frag_u(int i) { return S16 ? ((i >> 2) & 1) : (i >> 3); }
__host__ __device__ constexpr int frag_j(int i) { return S16 ? (4 * (i >> 3) + (i & 3)) : (i & 7); }
// fp16 hi / scaled-lo split of 8 fp32 values (one B fragment)
__device__ __forceinline__ void split8(const float (&x)[8], h8& hi, h8& lo) {
#if HN_SPLIT_MIX && defined(__HIP_DEVICE_COMPILE__)
    u32x4_ H, L;
    float scale = LO_SCALE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r0, r1;
        unsigned hp;
        split_pair_hi(x[2 * j], x[2 * j + 1], hp, r0, r1);
        H[j] = hp;
        L[j] = split_pair_lo(r0, r1, scale);
    }
    hi = __builtin_bit_cast(h8, H);
    lo = __builtin_bit_cast(h8, L);
    return;
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 hj = hi_part(x[j]);
        hi[j] = hj;
        lo[j] = (_Float16)((x[j] - (float)hj) * LO_SCALE);
    }
}
// registers 8u..8u+7 of an accumulator tile -> B fragment of k-step 2t+u
__device__ __forceinline__ void split_tile(const f32x16& a, h8& hi0, h8& lo0, h8& hi1, h8& lo1) {
    // (16x16x32 shape: fragment cb = registers 4 cb + ii and 8 + 4 cb + ii, see frag_u / frag_j)
    const float x0[8] = {a[0], a[1], a[2], a[3], S16 ? a[8] : a[4], S16 ? a[9] : a[5], S16 ? a[10] : a[6], S16 ? a[11] : a[7]};
    const float x1[8] = {S16 ? a[4] : a[8], S16 ? a[5] : a[9], S16 ? a[6] : a[10], S16 ? a[7] : a[11], a[12], a[13], a[14], a[15]};
    split8(x0, hi0, lo0);
    split8(x1, hi1, lo1);
    // keep the conversion where it is written (beside the next tile's MFMAs) instead of letting the
    // compiler sink it to the fragments' first use one layer later
    asm volatile("" : "+v"(hi0), "+v"(lo0), "+v"(hi1), "+v"(lo1));
}
// fp32 value back from a stored fragment element
__device__ __forceinline__ float unsplit(_Float16 hi, _Float16 lo) { return fmaf((float)lo, LO_INV, (float)hi); }

__device__ __forceinline__ f32x16 mfma16(const h8& a, const h8& b, const f32x16& c) {
#ifdef HN_EXP_MFMA16   // timing experiment only (tools/cpp/layer_bench.hip): the same MACs as two 16x16x32 instructions; results meaningless
    using f32x4_ = float __attribute__((ext_vector_type(4)));
    f32x4_ c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    f32x16 r = c;
    r[0] = c0[0]; r[1] = c0[1]; r[2] = c0[2]; r[3] = c0[3];
    r[4] = c1[0]; r[5] = c1[1]; r[6] = c1[2]; r[7] = c1[3];
    return r;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// 16x16x32 shape: row block RB of the tile against both column blocks: c[8 RB + 4 cb ..] += A(16 rows x 32 k) * b[cb]
using f32x4v = float __attribute__((ext_vector_type(4)));
template <int RB>
__device__ __forceinline__ void mfma_blk(const h8& a, const h8& b0, const h8& b1, f32x16& c) {
    f32x4v c0 = {c[8 * RB], c[8 * RB + 1], c[8 * RB + 2], c[8 * RB + 3]};
    f32x4v c1 = {c[8 * RB + 4], c[8 * RB + 5], c[8 * RB + 6], c[8 * RB + 7]};
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, c1, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[8 * RB + i] = c0[i];
        c[8 * RB + 4 + i] = c1[i];
    }
}
// one LDS block (2 KiB: hi + lo fragment) of the tile against the matching B fragments: 3 MFMAs (32x32x16: block =
// k-step s) or 6 (16x16x32: block s = row block s & 1 of k-step pair s >> 1), with a filler slot after each third
// PASSES: 3 = the fp32-equivalent product (hi*hi, hi*lo, lo*hi); 1 = the single-pass throughput mode of HN_PREC_F16 (hi*hi
// only: fp16 operands, fp32 accumulation; the three filler slots are kept so that epilogue slices and DMA pieces are issued
// exactly as in the 3-pass form)
template <int S, int S0, int PASSES = 3, int NX, typename Slot>
__device__ __forceinline__ void mma_block(const h8& ah, const h8& al, const h8 (&xh)[NX], const h8 (&xl)[NX], f32x16& c1, f32x16& c2,
                                          Slot&& slot) {
    static_assert(PASSES == 1 || PASSES == 3, "1 or 3 MFMAs per product");
    if constexpr (PASSES == 1) {
        static_assert(!S16, "the single-pass mode is built on the 32x32x16 shape");
        c1 = mfma16(ah, xh[S0 + S], c1);
        slot(std::integral_constant<int, 3 * S>{});
        slot(std::integral_constant<int, 3 * S + 1>{});
        slot(std::integral_constant<int, 3 * S + 2>{});
    } else if constexpr (S16) {
        static_assert(S0 % 2 == 0, "k-step pairs");
        constexpr int b = S0 + (S & ~1), RB = S & 1;
        mfma_blk<RB>(ah, xh[b], xh[b + 1], c1);
        slot(std::integral_constant<int, 3 * S>{});
        mfma_blk<RB>(ah, xl[b], xl[b + 1], c2);
        slot(std::integral_constant<int, 3 * S + 1>{});
        mfma_blk<RB>(al, xh[b], xh[b + 1], c2);
        slot(std::integral_constant<int, 3 * S + 2>{});
    } else {
        c1 = mfma16(ah, xh[S0 + S], c1);
        slot(std::integral_constant<int, 3 * S>{});
        c2 = mfma16(ah, xl[S0 + S], c2);
        slot(std::integral_constant<int, 3 * S + 1>{});
        c2 = mfma16(al, xh[S0 + S], c2);
        slot(std::integral_constant<int, 3 * S + 2>{});
    }
}

// c1/c2 += W[tile rows, KS k-steps] * x[S0 .. S0+KS)  -- `blk` points at the tile's first k-step block in LDS.
// Hand-scheduled: the two A fragments of k-step s+2 are read from LDS before the MFMAs of k-step s (LDS
// latency ~100+ cycles against 96 MFMA cycles per k-step), and after each MFMA one slice of `epi` -- the
// element-wise epilogue of the PREVIOUS tile -- is issued, so that the VALU work runs in the shadow of the
// matrix pipe.  sched_barrier(0) pins this order (the compiler's own placement serialises the two).
// per-translation-unit choice of the DMA piece form by layer kind (see WStream::piece)
#ifndef HN_BRANCHY_FWD
#define HN_BRANCHY_FWD false
#endif
#ifndef HN_BRANCHY_REV
#define HN_BRANCHY_REV false
#endif
#ifndef HN_BRANCHY_RELU
#define HN_BRANCHY_RELU false
#endif
#ifndef HN_BRANCHY_NOEPI
#define HN_BRANCHY_NOEPI false
#endif
struct NoEpi {
    static constexpr bool branchy = HN_BRANCHY_NOEPI;
    template <int Q, int NQ>
    __device__ __forceinline__ void run() {}
};
// FETCH: the DMA pieces of the next chunk are issued from this tile's MFMA slots (one piece per
// stride of slots, all within the first ~half of the tile so that they land before the next barrier)
// how far apart the DMA pieces of the next chunk are placed in the MFMA slots of this one (3: every third slot, over the
// first ~60 % of a 16-k-step tile; 1: in its first slots, so that they have the whole tile to land)
#ifndef HN_PIECE_STRIDE_MAX
#define HN_PIECE_STRIDE_MAX 3
#endif
// FETCH: 0 none; 1 the run-time form (WStream::begin / piece); >= 1024: the size of the chunk being fetched, a
// constant (WStream::begin_c / piece_c)
template <int KS, int S0, int FETCH, int PASSES = 3, int NX, typename Epi>
__device__ __forceinline__ void mma_tile(WStream& ws, const char* blk, const h8 (&xh)[NX], const h8 (&xl)[NX], f32x16& c1,
                                         f32x16& c2, int lane, Epi& epi) {
    static_assert(S0 + KS <= NX, "k-step range");
    constexpr int NQ = 3 * KS;
    constexpr int NP = FETCH > 1 ? WStream::n_pieces<FETCH>() : MAX_PIECES_PER_WAVE;
    constexpr int STRIDE = HN_PIECE_STRIDE_MAX >= 3 && NQ >= 3 * NP ? 3 : (HN_PIECE_STRIDE_MAX >= 2 && NQ >= 2 * NP ? 2 : 1);
    static_assert(!FETCH || NQ >= NP, "not enough slots for the DMA pieces");
    h8 ah[3], al[3];
    // this lane's LDS read address is formed per tile (see lane_x16): kept across the kernel it gets spilled, and its
    // reload in front of every ds_read carries an s_waitcnt vmcnt(0) that also drains the weight stream's DMA
    (void)lane;
    const int l16 = lane_x16();
    const char* const lblk = blk + l16;
    auto load = [&](auto S) {
        constexpr int s = decltype(S)::value;
        ah[s % 3] = *reinterpret_cast<const h8*>(lblk + s * KS_BYTES);
        if constexpr (PASSES == 3) al[s % 3] = *reinterpret_cast<const h8*>(lblk + s * KS_BYTES + 1024);
    };
    auto slot = [&](auto Q_) {
        constexpr int Q = decltype(Q_)::value;
        if constexpr (FETCH == 1 && Q % STRIDE == STRIDE - 1 && Q / STRIDE < NP) ws.template piece<Epi::branchy>(Q / STRIDE);
        if constexpr (FETCH > 1 && Q % STRIDE == STRIDE - 1 && Q / STRIDE < NP) ws.template piece_c<FETCH, Q / STRIDE>(l16);
        epi.template run<Q, NQ>();
        __builtin_amdgcn_sched_barrier(0);
    };
    load(std::integral_constant<int, 0>{});
    if constexpr (KS > 1) load(std::integral_constant<int, 1>{});
    static_assert(!S16 || KS % 2 == 0, "the 16x16x32 shape consumes k-steps in pairs");
    static_for<KS>([&](auto S) {
        constexpr int s = decltype(S)::value;
        if constexpr (s + 2 < KS) load(std::integral_constant<int, s + 2>{});
        mma_block<s, S0, PASSES>(ah[s % 3], al[s % 3], xh, xl, c1, c2, slot);
    });
    ws.stamp(4);
}
template <int KS, int S0, int FETCH, int PASSES = 3, int NX>
__device__ __forceinline__ void mma_tile(WStream& ws, const char* blk, const h8 (&xh)[NX], const h8 (&xl)[NX], f32x16& c1,
                                         f32x16& c2, int lane) {
    NoEpi e;
    mma_tile<KS, S0, FETCH, PASSES>(ws, blk, xh, xl, c1, c2, lane, e);
}

// A whole chunk of NT tiles x KS k-steps that share the KS fragments xh/xl (the hand field's feature passes):
// c1/c2[t] += W[tile t, KS k-steps] * x.  One continuous software pipeline over the NT*KS blocks of the chunk (they
// are contiguous in LDS): the A-fragment ring keeps running across the tile boundaries, where NT separate mma_tile
// calls would each start with an empty ring and expose the LDS latency again.  The DMA pieces of the next chunk go
// into the first slots.
struct NoMid {
    __device__ __forceinline__ void operator()() const {}
};
// mid(): called once, in the slot behind the last DMA piece -- vector-memory work issued there is YOUNGER than the pieces,
// so the next acquire can let it stay in flight (acquire<ALLOW>)
template <int NT, int KS, int FETCH = 1, int PASSES = 3, int NX, typename Mid = NoMid>
__device__ __forceinline__ void mma_chunk(WStream& ws, const char* buf, const h8 (&xh)[NX], const h8 (&xl)[NX], f32x16* c1,
                                          f32x16* c2, int lane, Mid&& mid = Mid{}) {
    static_assert(KS <= NX, "k-step range");
    static_assert(!S16 || KS % 2 == 0, "the 16x16x32 shape consumes k-steps in pairs");
    constexpr int N = NT * KS;           // blocks in the chunk
    constexpr int NQ = 3 * N;
    constexpr int NP = FETCH > 1 ? WStream::n_pieces<FETCH>() : MAX_PIECES_PER_WAVE;
    constexpr int STRIDE = HN_PIECE_STRIDE_MAX >= 3 && NQ >= 3 * NP ? 3 : (HN_PIECE_STRIDE_MAX >= 2 && NQ >= 2 * NP ? 2 : 1);
    static_assert(NQ >= NP, "not enough slots for the DMA pieces");
    h8 ah[3], al[3];
    (void)lane;
    const int l16 = lane_x16();
    const char* const lbuf = buf + l16;   // see mma_tile
    auto load = [&](auto S) {
        constexpr int s = decltype(S)::value;
        ah[s % 3] = *reinterpret_cast<const h8*>(lbuf + s * KS_BYTES);
        if constexpr (PASSES == 3) al[s % 3] = *reinterpret_cast<const h8*>(lbuf + s * KS_BYTES + 1024);
    };
    auto slot = [&](auto Q_) {
        constexpr int Q = decltype(Q_)::value;
        if constexpr (FETCH == 1 && Q % STRIDE == STRIDE - 1 && Q / STRIDE < NP) ws.template piece<NoEpi::branchy>(Q / STRIDE);
        if constexpr (FETCH > 1 && Q % STRIDE == STRIDE - 1 && Q / STRIDE < NP) ws.template piece_c<FETCH, Q / STRIDE>(l16);
        if constexpr (Q == STRIDE * NP) mid();
        __builtin_amdgcn_sched_barrier(0);
    };
    static_assert(STRIDE * NP < NQ, "no slot left behind the pieces");
    load(std::integral_constant<int, 0>{});
    load(std::integral_constant<int, 1>{});
    static_for<N>([&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr int t = s / KS, k = s % KS;
        if constexpr (s + 2 < N) load(std::integral_constant<int, s + 2>{});
        if constexpr (k == 0) ws.stamp(5);
        // (slot indices continue over the tiles of the chunk: block k of tile t is slot base 3 (t KS + k))
        auto slot_t = [&](auto Q_) { slot(std::integral_constant<int, decltype(Q_)::value + 3 * t * KS>{}); };
        mma_block<k, 0, PASSES>(ah[s % 3], al[s % 3], xh, xl, c1[t], c2[t], slot_t);
    });
    ws.stamp(4);
}

// tail helpers: 32 floats stored [half][16] so that lane half h reads its 16 rows as 4 float4
// (row of register i, half h: (i&3) + 8 (i>>2) + 4 h)
__device__ __forceinline__ f32x16 tail_tile(const char* tail, int slot, int h) {
    if constexpr (S16) {
        // stored [g][8]: the 8 rows 16 rb + 4 g + ii of lane group g; both column blocks get the same row values
        const int g = lane_x16() >> 8;
        const float4* p = reinterpret_cast<const float4*>(tail + slot * 128 + g * 32);
        const float4 w0 = p[0], w1 = p[1];
        f32x16 v;
        v[0] = v[4] = w0.x;
        v[1] = v[5] = w0.y;
        v[2] = v[6] = w0.z;
        v[3] = v[7] = w0.w;
        v[8] = v[12] = w1.x;
        v[9] = v[13] = w1.y;
        v[10] = v[14] = w1.z;
        v[11] = v[15] = w1.w;
        return v;
    }
    f32x16 v;
    const float4* p = reinterpret_cast<const float4*>(tail + slot * 128 + h * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 w = p[q];
        v[4 * q] = w.x;
        v[4 * q + 1] = w.y;
        v[4 * q + 2] = w.z;
        v[4 * q + 3] = w.w;
    }
    return v;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
    return v;
}
__device__ __forceinline__ f32x16 combine(const f32x16& c1, const f32x16& c2) {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = fmaf(c2[i], LO_INV, c1[i]);
    return z;
}

// ---- activations -----------------------------------------------------------------------------
// nn.Softplus(beta=100, threshold=20) (utils/fields.py:125, 310): max(z,0) + log1p(exp(-100|z|))/100.
// Beyond the threshold the log term is below half an ulp of z, i.e. the result IS z, as in torch.
__device__ __forceinline__ float softplus100(float z) {
    // branch-free, 6 VALU ops: exp(-100|z|) -> log2(1 + e) * ln2/100 + max(z, 0).  The rounding of
    // 1 + e costs at most 6e-8 * ln2/100 = 4e-10 absolute, far below the fp32 resolution of the sums
    // this value enters (and 4e-8 on the derivative recovered from it).
    const float e = __builtin_amdgcn_exp2f(-fabsf(z * 144.26950408889634f));
    return fmaf(__builtin_amdgcn_logf(1.f + e), 0.0069314718055994531f, fmaxf(z, 0.f));
}
// sigmoid(100 z) recovered from a = softplus(z): exp(100 a) = 1 + exp(100 z) => s = 1 - exp(-100 a)
__device__ __forceinline__ float dsoftplus_from_act(float a) {
    return 1.f - __builtin_amdgcn_exp2f(a * -144.26950408889634f);
}
__device__ __forceinline__ float sigmoid_fast(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- a layer as a software pipeline over its output tiles ----------------------------------------
constexpr float K100_ = 144.26950408889634f;      // 100 * log2(e)
constexpr float C100_ = 0.0069314718055994531f;   // ln(2) / 100
struct NoData {};
using f32x2 = float __attribute__((ext_vector_type(2)));
#ifndef HN_EPI_PAIRS
#define HN_EPI_PAIRS 1
#endif

// Element-wise epilogue of one tile, cut into 48 phase calls (16 elements x 3 phases) that mma_tile
// spreads over the next tile's MFMA slots.  Phases 0 and 1 are the layer's own (`ph(I, P, st, pd)` turns
// the accumulator pair st.c1/st.c2 into st.v[I]); phase 2 is the fp16 hi/lo split of st.v into the two
// B fragments st.hi/st.lo of the next layer (skipped when FRAGS is false).
struct EpiState {
    f32x16 c1, c2;   // the finished accumulators of the tile
    float z[2], e[2];   // per-element temporaries carried from phase 0 to phase 1 (index I & 1): plain scalars,
                        // a 16-wide vector would pin 16 consecutive registers for two live values
    float v[16];        // results (scalars for the same reason; vec() assembles a tile where one is needed)
    float w[16];        // second result of the adjoint's forward-direction layers (kind 4); unused elsewhere
    __device__ __forceinline__ f32x16 vec() const {
        f32x16 y;
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] = v[i];
        return y;
    }
    __device__ __forceinline__ f32x16 wvec() const {
        f32x16 y;
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] = w[i];
        return y;
    }
    h8 hi[2], lo[2]; // fragments of k-steps 2t, 2t+1
    float r0, r1;    // residuals of the pair being converted
    f32x2 z2, t2, e2;   // pair pipeline (HN_EPI_PAIRS): pre-activation, scaled argument, exponential
    float inv;       // 1/2048, laundered behind the tile's barrier: ties every phase-0 to this side of it
    float scale;     // 2048 in a register (the mix instructions take no literal)
};
template <int I, bool FRAGS>
__device__ __forceinline__ void split_phase(EpiState& st) {
    static_assert(!S16, "the scalar epilogue form is not kept for the 16x16x32 shape");
    if constexpr (FRAGS) {
        constexpr int u = I >> 3, j = I & 7;
        if constexpr ((I & 1) == 1) {
            // hi halves of the pair (I-1, I) and their residuals
            const _Float16 h0 = hi_part(st.v[I - 1]), h1 = hi_part(st.v[I]);
            st.hi[u][j - 1] = h0;
            st.hi[u][j] = h1;
            st.r0 = st.v[I - 1] - (float)h0;
            st.r1 = st.v[I] - (float)h1;
        } else if constexpr (I >= 2) {
            // scaled lo halves of the previous pair (I-2, I-1)
            constexpr int up = (I - 2) >> 3, jp = (I - 2) & 7;
            st.lo[up][jp] = (_Float16)(st.r0 * LO_SCALE);
            st.lo[up][jp + 1] = (_Float16)(st.r1 * LO_SCALE);
        }
    }
}
template <bool FRAGS>
__device__ __forceinline__ void split_finish(EpiState& st) {
    if constexpr (FRAGS && !HN_EPI_PAIRS) {
        st.lo[1][6] = (_Float16)(st.r0 * LO_SCALE);
        st.lo[1][7] = (_Float16)(st.r1 * LO_SCALE);
    }
}
// Pair form of the same epilogue (HN_EPI_PAIRS): the 16 elements are processed as 8 pairs x 6 phases (48 calls, one
// per MFMA slot of a 16-k-step tile) so that the multiply-adds become packed fp32 instructions (v_pk_fma/mul/add_f32:
// two elements per issue slot).  The wave is issue-bound -- every VALU instruction of the epilogue costs ~4 cycles
// of the same in-order stream that has to issue the MFMAs -- so the instruction count, not the ALU rate, is what
// matters.  kind: 0 softplus(beta=100), 1 g * sigma'(z) from the stashed activation, 2 relu, 3 identity;
// adjoint (hn_field2_*_adj): 4 forward-direction sweep  v = z sigma',  w = z (1 - sigma') * (pd.x 100 / 256)  (pd.x = dz:
// w is the second-order source sigma'' u dzb of oracle/field_bwd.py written without a division by sigma');
// 5 second reverse sweep  v = z sigma' + pd.x;  6 relu mask  v = pd.v > 0 ? z : 0.
#ifndef HN_DBG_SOFTPLUS_AS_RELU
#define HN_DBG_SOFTPLUS_AS_RELU 0   // 1 (measurement only, WRONG results): the softplus epilogue issues a ReLU's instructions -- an upper bound on
                                    // what a cheaper activation epilogue could buy the evaluation kernels (round 5, profiles/r05/README.md)
#endif
template <int J, int P, int KIND_, bool FRAGS, typename PD>
__device__ __forceinline__ void pair_phase(EpiState& st, const PD& pd) {
    constexpr int KIND = (HN_DBG_SOFTPLUS_AS_RELU && KIND_ == 0) ? 2 : KIND_;
    constexpr int i0 = 2 * J, i1 = 2 * J + 1;
    if constexpr (P == 0) {
        const f32x2 c1 = {st.c1[i0], st.c1[i1]}, c2 = {st.c2[i0], st.c2[i1]};
        const f32x2 inv = {st.inv, st.inv};
        st.z2 = c2 * inv + c1;
        if constexpr (KIND == 0) st.t2 = st.z2 * f32x2{K100_, K100_};
        if constexpr (KIND == 1 || KIND == 4 || KIND == 5) st.t2 = f32x2{pd.v[i0], pd.v[i1]} * f32x2{-K100_, -K100_};
    } else if constexpr (P == 1) {
        if constexpr (KIND == 0) st.e2 = f32x2{__builtin_amdgcn_exp2f(-fabsf(st.t2[0])), __builtin_amdgcn_exp2f(-fabsf(st.t2[1]))};
        if constexpr (KIND == 1 || KIND == 4 || KIND == 5) st.e2 = f32x2{__builtin_amdgcn_exp2f(st.t2[0]), __builtin_amdgcn_exp2f(st.t2[1])};
    } else if constexpr (P == 2) {
        if constexpr (KIND == 0) {
            const f32x2 u = st.e2 + f32x2{1.f, 1.f};
            st.e2 = f32x2{__builtin_amdgcn_logf(u[0]), __builtin_amdgcn_logf(u[1])};
        }
        if constexpr (KIND == 4 || KIND == 5) st.e2 = st.z2 * st.e2;   // z (1 - sigma')
    } else if constexpr (P == 3) {
        f32x2 v;
        if constexpr (KIND == 0) v = st.e2 * f32x2{C100_, C100_} + f32x2{fmaxf(st.z2[0], 0.f), fmaxf(st.z2[1], 0.f)};
        if constexpr (KIND == 1) v = st.z2 - st.z2 * st.e2;
        if constexpr (KIND == 2) v = f32x2{fmaxf(st.z2[0], 0.f), fmaxf(st.z2[1], 0.f)};
        if constexpr (KIND == 3) v = st.z2;
        if constexpr (KIND == 4) {
            v = st.z2 - st.e2;
            // (pd.x is dz_l as the tape holds it; its factor 100 / 256 is applied HERE, a step after the load: applied where the tile is
            // loaded -- in the layer's `pre` -- it made the wave wait for the tape's round trip in front of the step's MFMAs)
            const f32x2 w = st.e2 * (f32x2{pd.x[i0], pd.x[i1]} * f32x2{100.f * BWD_INV, 100.f * BWD_INV});
            st.w[i0] = w[0];
            st.w[i1] = w[1];
        }
        if constexpr (KIND == 5) v = st.z2 - st.e2 + f32x2{pd.x[i0], pd.x[i1]};
        if constexpr (KIND == 6) v = f32x2{pd.v[i0] > 0.f ? st.z2[0] : 0.f, pd.v[i1] > 0.f ? st.z2[1] : 0.f};
        st.v[i0] = v[0];
        st.v[i1] = v[1];
    } else if constexpr (P == 4) {
        if constexpr (FRAGS) {
            constexpr int u = frag_u(i0), j = frag_j(i0);
#if HN_SPLIT_MIX
            unsigned hp;
            split_pair_hi(st.v[i0], st.v[i1], hp, st.r0, st.r1);
            set_pair<j / 2>(st.hi[u], hp);
#else
            const _Float16 h0 = hi_part(st.v[i0]), h1 = hi_part(st.v[i1]);
            st.hi[u][j] = h0;
            st.hi[u][j + 1] = h1;
            const f32x2 r = f32x2{st.v[i0], st.v[i1]} - f32x2{(float)h0, (float)h1};
            st.r0 = r[0];
            st.r1 = r[1];
#endif
        }
    } else {
        if constexpr (FRAGS) {
            constexpr int u = frag_u(i0), j = frag_j(i0);
#if HN_SPLIT_MIX
            set_pair<j / 2>(st.lo[u], split_pair_lo(st.r0, st.r1, st.scale));
#else
            const f32x2 l = f32x2{st.r0, st.r1} * f32x2{LO_SCALE, LO_SCALE};
            st.lo[u][j] = (_Float16)l[0];
            st.lo[u][j + 1] = (_Float16)l[1];
#endif
        }
    }
}
#ifndef HN_ADJ_EPI_DELAY
#define HN_ADJ_EPI_DELAY 0
#endif
template <typename T, typename = void>
struct ph_delay {
    static constexpr int value = 0;
};
template <typename T>
struct ph_delay<T, std::void_t<decltype(T::delay)>> {
    static constexpr int value = T::delay;
};
template <bool FRAGS, typename Ph, typename PD>
struct Epi {
    static constexpr bool branchy = Ph::branchy;
    EpiState& st;
    Ph& ph;
    const PD& pd;
    template <int C>
    __device__ __forceinline__ void call() {
#if HN_EPI_PAIRS
        pair_phase<C / 6, C % 6, Ph::kind, FRAGS>(st, pd);
#else
        constexpr int I = C / 3, P = C % 3;
        if constexpr (P < 2)
            ph(std::integral_constant<int, I>{}, std::integral_constant<int, P>{}, st, pd);
        else
            split_phase<I, FRAGS>(st);
#endif
    }
    // slots Q of NQ: phase calls [Q*48/NQ, (Q+1)*48/NQ).  A phase with side data from the tape (Ph::delay = D > 0: the adjoint's
    // kinds 4 and 5) leaves the first D of a 48-slot tile's slots empty and packs its 48 calls into the others: the tile's side
    // data was requested one step ago and an HBM round trip under load is longer than a step, so the first call's wait is pushed
    // D MFMAs into the step instead of standing in front of them.  (-DHN_ADJ_EPI_DELAY=D, default 0: measured at 16 and 24 on the
    // fitting step, same box, interleaved runs: 2.53 - 2.59 ms against 2.51 - 2.60 ms -- the stall moves, the step does not.)
    template <int Q, int NQ>
    __device__ __forceinline__ void run() {
        constexpr int D = NQ == 48 ? ph_delay<Ph>::value : 0;
        if constexpr (Q >= D) {
            constexpr int lo = (Q - D) * 48 / (NQ - D), hi = (Q - D + 1) * 48 / (NQ - D);
            static_for<hi - lo>([&](auto K) { call<lo + decltype(K)::value>(); });
        }
    }
    __device__ __forceinline__ void run_all() {
        static_for<48>([&](auto K) { call<decltype(K)::value>(); });
    }
};

// Ties the epilogue of the previous tile to this side of the barrier just passed: every phase 0 multiplies
// by st.inv, which comes out of an opaque asm placed after the barrier.
__device__ __forceinline__ void arm(EpiState& st) {
    float inv = LO_INV;
    asm volatile("" : "+v"(inv));
    st.inv = inv;
    st.scale = LO_SCALE;
}

// OT output tiles of KS k-steps; TPC tiles share one chunk (+ 1 KiB tail if TAIL, whose slot (t % TPC) is
// the tile's bias).  Step t:
//   - publish the chunk when t opens one (s_waitcnt vmcnt(0) + s_barrier);
//   - `store(T-2, held)`: the global stores (stash) of tile t-2, deliberately one step late and in FRONT of
//     this step's DMA pieces, so that the vmcnt(0) of the next barrier finds them long complete (VMEM
//     completion is not ordered between stores and loads, a counted wait cannot skip them);
//   - `pre(T, tail)`: what tile t's epilogue will need (side data from the tail, stashed activations);
//   - the MFMAs of tile t with the DMA of the next chunk and the epilogue of tile t-1 in their shadow;
//   - `held = fin(T-1, st, pd)`: the finished tile's fragments -> registers; returns what `store` needs.
// next_same / next_after: bytes of the chunk that follows a chunk of this layer (another of the same
// layer / the first of the next layer; 0 = none).
// NS / NA >= 0: next_same / next_after as constants (the compile-time-sized fetch of WStream); -1: the run-time arguments.
template <int OT, int KS, int TPC, bool TAIL, bool FRAGS, int NS, int NA, int PASSES, int NX, typename Pre, typename Ph, typename Fin, typename Store>
__device__ __forceinline__ void run_layer_impl(WStream& ws, int next_same, int next_after, const h8 (&xh)[NX], const h8 (&xl)[NX],
                                               int lane, int h, Pre&& pre, Ph&& ph, Fin&& fin, Store&& store) {
    static_assert((NS >= 0) == (NA >= 0), "both sizes constant or both run-time");
    using I0 = std::integral_constant<int, 0>;
    using PD = decltype(pre(I0{}, (const char*)nullptr));
    f32x16 c1[2], c2[2];
    PD pd[2];
    EpiState st;
    using Held = decltype(fin(I0{}, st, pd[0]));
    Held held;
    const char* buf = nullptr;
    static_for<OT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        constexpr bool opens = t % TPC == 0;
        if constexpr (opens) buf = ws.template acquire<0>();
        if constexpr (t >= 2) store(std::integral_constant<int, t - 2>{}, held);
        constexpr int nb = NS < 0 ? -1 : (t + TPC < OT ? NS : NA);   // constant size of the chunk fetched from here
        constexpr int fetch = !opens ? 0 : (nb < 0 ? 1 : nb);
        if constexpr (opens && nb < 0) ws.begin(t + TPC < OT ? next_same : next_after);
        if constexpr (opens && nb > 0) ws.template begin_c<nb>();
        const char* tail = buf + TPC * KS * KS_BYTES;
        arm(st);
        if constexpr (t > 0) {
            st.c1 = c1[(t - 1) & 1];
            st.c2 = c2[(t - 1) & 1];
        }
        c1[t & 1] = TAIL ? tail_tile(tail, t % TPC, h) : zero16();
        c2[t & 1] = zero16();
        pd[t & 1] = pre(T, tail);
        if constexpr (t > 0) {
            Epi<FRAGS, std::remove_reference_t<Ph>, PD> epi{st, ph, pd[(t - 1) & 1]};
            mma_tile<KS, 0, fetch, PASSES>(ws, buf + (t % TPC) * KS * KS_BYTES, xh, xl, c1[t & 1], c2[t & 1], lane, epi);
            split_finish<FRAGS>(st);
            held = fin(std::integral_constant<int, t - 1>{}, st, pd[(t - 1) & 1]);
        } else {
            mma_tile<KS, 0, fetch, PASSES>(ws, buf + (t % TPC) * KS * KS_BYTES, xh, xl, c1[t & 1], c2[t & 1], lane);
        }
    });
    if constexpr (OT >= 2) store(std::integral_constant<int, OT - 2>{}, held);
    arm(st);
    st.c1 = c1[(OT - 1) & 1];
    st.c2 = c2[(OT - 1) & 1];
    Epi<FRAGS, std::remove_reference_t<Ph>, PD> epi{st, ph, pd[(OT - 1) & 1]};
    epi.run_all();
    split_finish<FRAGS>(st);
    held = fin(std::integral_constant<int, OT - 1>{}, st, pd[(OT - 1) & 1]);
    store(std::integral_constant<int, OT - 1>{}, held);
}
template <int OT, int KS, int TPC, bool TAIL, bool FRAGS, int NX, typename Pre, typename Ph, typename Fin, typename Store>
__device__ __forceinline__ void run_layer(WStream& ws, int next_same, int next_after, const h8 (&xh)[NX], const h8 (&xl)[NX],
                                          int lane, int h, Pre&& pre, Ph&& ph, Fin&& fin, Store&& store) {
    run_layer_impl<OT, KS, TPC, TAIL, FRAGS, -1, -1, 3>(ws, next_same, next_after, xh, xl, lane, h, pre, ph, fin, store);
}
template <int OT, int KS, int TPC, bool TAIL, bool FRAGS, int NS, int NA, int PASSES = 3, int NX, typename Pre, typename Ph, typename Fin, typename Store>
__device__ __forceinline__ void run_layer_c(WStream& ws, const h8 (&xh)[NX], const h8 (&xl)[NX], int lane, int h, Pre&& pre,
                                            Ph&& ph, Fin&& fin, Store&& store) {
    run_layer_impl<OT, KS, TPC, TAIL, FRAGS, NS, NA, PASSES>(ws, 0, 0, xh, xl, lane, h, pre, ph, fin, store);
}

// standard phases ---------------------------------------------------------------------------------
constexpr float K100 = 144.26950408889634f;      // 100 * log2(e)
constexpr float C100 = 0.0069314718055994531f;   // ln(2) / 100
// softplus(beta=100): v = max(z,0) + log2(1 + exp2(-|z| K100)) * C100
struct PhSoftplus {
    static constexpr int kind = 0;
    static constexpr bool branchy = HN_BRANCHY_FWD;
    template <typename I_, typename P_, typename PD>
    __device__ __forceinline__ void operator()(I_, P_, EpiState& st, const PD&) const {
        constexpr int I = I_::value, P = P_::value;
        if constexpr (P == 0) {
            st.z[I & 1] = fmaf(st.c2[I], st.inv, st.c1[I]);
            st.e[I & 1] = __builtin_amdgcn_exp2f(-fabsf(st.z[I & 1] * K100));
        } else {
            st.v[I] = fmaf(__builtin_amdgcn_logf(1.f + st.e[I & 1]), C100, fmaxf(st.z[I & 1], 0.f));
        }
    }
};
// reverse sweep: v = g * sigma'(z) with sigma' = 1 - exp(-100 a) from the stashed activation pd.v
struct PhDsig {
    static constexpr int kind = 1;
    static constexpr bool branchy = HN_BRANCHY_REV;
    template <typename I_, typename P_, typename PD>
    __device__ __forceinline__ void operator()(I_, P_, EpiState& st, const PD& pd) const {
        constexpr int I = I_::value, P = P_::value;
        if constexpr (P == 0) {
            st.z[I & 1] = fmaf(st.c2[I], st.inv, st.c1[I]);
            st.e[I & 1] = __builtin_amdgcn_exp2f(pd.v[I] * -K100);
        } else {
            st.v[I] = fmaf(st.z[I & 1], -st.e[I & 1], st.z[I & 1]);
        }
    }
};
// the adjoint's element-wise kinds (pair_phase kinds 4, 5, 6).  PD carries the stashed activation in .v and, for
// 4 and 5, a second tile in .x.
struct PhFwdDir {
    static constexpr int kind = 4;
    static constexpr int delay = HN_ADJ_EPI_DELAY;
    static constexpr bool branchy = HN_BRANCHY_REV;
    template <typename I_, typename P_, typename PD>
    __device__ __forceinline__ void operator()(I_, P_, EpiState& st, const PD& pd) const {
        constexpr int I = I_::value, P = P_::value;
        if constexpr (P == 0) {
            st.z[I & 1] = fmaf(st.c2[I], st.inv, st.c1[I]);
            st.e[I & 1] = __builtin_amdgcn_exp2f(pd.v[I] * -K100) * st.z[I & 1];
        } else {
            st.v[I] = st.z[I & 1] - st.e[I & 1];
            st.w[I] = st.e[I & 1] * (pd.x[I] * (100.f * BWD_INV));
        }
    }
};
struct PhRev2 {
    static constexpr int kind = 5;
    static constexpr int delay = HN_ADJ_EPI_DELAY;
    static constexpr bool branchy = HN_BRANCHY_REV;
    template <typename I_, typename P_, typename PD>
    __device__ __forceinline__ void operator()(I_, P_, EpiState& st, const PD& pd) const {
        constexpr int I = I_::value, P = P_::value;
        if constexpr (P == 0) {
            st.z[I & 1] = fmaf(st.c2[I], st.inv, st.c1[I]);
            st.e[I & 1] = __builtin_amdgcn_exp2f(pd.v[I] * -K100) * st.z[I & 1];
        } else {
            st.v[I] = st.z[I & 1] - st.e[I & 1] + pd.x[I];
        }
    }
};
struct PhMask {
    static constexpr int kind = 6;
    static constexpr bool branchy = HN_BRANCHY_RELU;
    template <typename I_, typename P_, typename PD>
    __device__ __forceinline__ void operator()(I_, P_, EpiState& st, const PD& pd) const {
        constexpr int I = I_::value, P = P_::value;
        if constexpr (P == 0) st.v[I] = pd.v[I] > 0.f ? fmaf(st.c2[I], st.inv, st.c1[I]) : 0.f;
    }
};
struct PhRelu {
    static constexpr int kind = 2;
    static constexpr bool branchy = HN_BRANCHY_RELU;
    template <typename I_, typename P_, typename PD>
    __device__ __forceinline__ void operator()(I_, P_, EpiState& st, const PD&) const {
        constexpr int I = I_::value, P = P_::value;
        if constexpr (P == 0) st.v[I] = fmaxf(fmaf(st.c2[I], st.inv, st.c1[I]), 0.f);
    }
};
struct PhIdentity {
    static constexpr int kind = 3;
    static constexpr bool branchy = HN_BRANCHY_FWD;
    template <typename I_, typename P_, typename PD>
    __device__ __forceinline__ void operator()(I_, P_, EpiState& st, const PD&) const {
        constexpr int I = I_::value, P = P_::value;
        if constexpr (P == 0) st.v[I] = fmaf(st.c2[I], st.inv, st.c1[I]);
    }
};

// ---- per-wave stash in global memory (slots of 32 KiB, every instruction moves 1 KiB) -----------
// Addressed through a buffer descriptor: SGPR base + scalar byte offset + one shared VGPR (lane * 16), so
// the hundreds of distinct stash addresses cost no vector registers (64-bit per-lane addresses were the
// kernel's main source of spills).
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
constexpr size_t SLOT_F4 = 8 * 4 * 64;   // float4 per slot
constexpr int SLOT_BYTES = 32768;
// cache policy of the stash traffic: nt (aux = 2) -- written once, read once or a few times much later, it must
// not evict the weight stream, which every workgroup of the XCD re-reads from L2
#ifndef HN_STASH_AUX
#define HN_STASH_AUX 2
#endif
constexpr int STASH_AUX = HN_STASH_AUX;
// The stash STORES of the per-workgroup stash (evaluation kernels: written and read back within a tile) use the default
// (write-back) policy, the loads stay nt: round-4 A/B on the C2 frame, same box, two runs each -- stores nt 248.6 / 249.8 ms,
// default 242.3 / 244.5, sc0 244.3 / 243.4, sc1 + nt 251.7 / 251.9.  (Both on the default policy was 9 % slower in round 2: it
// is the LOADS' allocation that evicts the weight stream from L2.)  The per-TILE stash of the taped kernels (a fitting step's
// 2.6 GB tape, read back a millisecond later by another launch) stays nt: written back through L2 the fitting step was 8 - 10 %
// slower (2.64 -> 2.91 ms).  ST_AUX: the stores' policy, per kernel.
#ifndef HN_STASH_ST_AUX
#define HN_STASH_ST_AUX 0
#endif
constexpr int STASH_ST_AUX = HN_STASH_ST_AUX;
template <int ST_AUX>
struct StashT {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;   // lane * 16

    __device__ __forceinline__ void init(float4* wave_base, int n_slots, int lane) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(wave_base, 0, n_slots * SLOT_BYTES, 0x00020000);
        voff = lane * 16;
    }
    // 16-byte store.  The byte offset goes into the VGPR operand and soffset stays 0 on purpose: with an
    // SGPR soffset hipcc (ROCm 7.2) places no wait state between a buffer_store_dwordx4 and a following
    // VALU write of its data registers (it assumes the hardware interlocks), and on gfx950 the store then
    // picks up the NEW register contents for its later dwords (observed: dword 1 of a fragment replaced
    // by the residual computed two instructions later).  In this form the compiler's store-data hazard
    // rule applies and the required wait state is inserted.
    // A lane offset that the compiler has to treat as new at every use: the stash addresses are loop-invariant
    // (lane * 16 + constant), and hoisted out of the persistent tile loop -- as LICM does with plain arithmetic --
    // the few hundred of them are all live across the whole kernel, get spilled at its top and are re-read from
    // scratch memory in front of every stash store (measured: 568 scratch loads in the full object kernel).
    __device__ __forceinline__ int fresh_voff() const { return lane_x16(); }
    template <typename T16>
    __device__ __forceinline__ void st16_at(const T16& v, int vo) const {
        static_assert(sizeof(T16) == 16, "16-byte values only");
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, vo, 0, ST_AUX);
    }
    template <typename T16>
    __device__ __forceinline__ void st16(const T16& v, int off) const {
        st16_at(v, fresh_voff() + off);
    }
    __device__ __forceinline__ u32x4 ld16_at(int vo, int off) const { return __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, off, STASH_AUX); }
    __device__ __forceinline__ u32x4 ld16(int off) const { return ld16_at(lane_x16(), off); }
    // fp32 tile t of slot `slot`: [t][q][lane] float4
    __device__ __forceinline__ void tile_store(int slot, int t, const f32x16& y) const {
        using f32x4 = float __attribute__((ext_vector_type(4)));
        const int vo = fresh_voff() + (slot * SLOT_BYTES + t * 4096);   // + 1024 q fits the instruction's offset field
        st16_at((f32x4)__builtin_shufflevector(y, y, 0, 1, 2, 3), vo);
        st16_at((f32x4)__builtin_shufflevector(y, y, 4, 5, 6, 7), vo + 1024);
        st16_at((f32x4)__builtin_shufflevector(y, y, 8, 9, 10, 11), vo + 2048);
        st16_at((f32x4)__builtin_shufflevector(y, y, 12, 13, 14, 15), vo + 3072);
    }
    __device__ __forceinline__ f32x16 tile_load(int slot, int t) const {
        using f32x4 = float __attribute__((ext_vector_type(4)));
        using f32x8 = float __attribute__((ext_vector_type(8)));
        const int off = slot * SLOT_BYTES + t * 4096;
        const int vo = lane_x16();
        const f32x4 a = __builtin_bit_cast(f32x4, ld16_at(vo, off));
        const f32x4 b = __builtin_bit_cast(f32x4, ld16_at(vo, off + 1024));
        const f32x4 c = __builtin_bit_cast(f32x4, ld16_at(vo, off + 2048));
        const f32x4 d = __builtin_bit_cast(f32x4, ld16_at(vo, off + 3072));
        const f32x8 ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
        const f32x8 cd = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_shufflevector(ab, cd, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    }
    // MEASUREMENT BUILD ONLY (-DHN_STASH_HALF=1; not parity-preserving): the same tile as 16 fp16 values, two 16-byte accesses instead
    // of four -- what halving the a1..a7 stash traffic of the evaluation kernel would be worth
    __device__ __forceinline__ void tile_store_half(int slot, int t, const f32x16& y) const {
        const int vo = fresh_voff() + (slot * SLOT_BYTES + t * 4096);
        h8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a[i] = (_Float16)y[i];
            b[i] = (_Float16)y[8 + i];
        }
        st16_at(a, vo);
        st16_at(b, vo + 1024);
    }
    __device__ __forceinline__ f32x16 tile_load_half(int slot, int t) const {
        const int off = slot * SLOT_BYTES + t * 4096;
        const int vo = lane_x16();
        const h8 a = __builtin_bit_cast(h8, ld16_at(vo, off));
        const h8 b = __builtin_bit_cast(h8, ld16_at(vo, off + 1024));
        f32x16 y;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            y[i] = (float)a[i];
            y[8 + i] = (float)b[i];
        }
        return y;
    }
    // fragment block s (byte offset `base` + s * 2 KiB): [hi | lo][lane] 16 B
    __device__ __forceinline__ void frag_store(int base, int s, const h8& hi, const h8& lo) const {
        const int vo = fresh_voff() + (base + s * KS_BYTES);
        st16_at(hi, vo);
        st16_at(lo, vo + 1024);
    }
    __device__ __forceinline__ void frag_load(int base, int s, h8& hi, h8& lo) const {
        const int vo = lane_x16();
        const u32x4 a = ld16_at(vo, base + s * KS_BYTES);
        const u32x4 b = ld16_at(vo, base + s * KS_BYTES + 1024);
        hi = __builtin_bit_cast(h8, a);
        lo = __builtin_bit_cast(h8, b);
    }
    // one float per lane at byte offset off + lane * 4
    __device__ __forceinline__ void f32_store(int off, float v) const {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, lane_x16() >> 2, off, 0);
    }
    __device__ __forceinline__ float f32_load(int off) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_x16() >> 2, off, 0));
    }
};
using Stash = StashT<STASH_AUX>;   // (the taped kernels and every kernel that did not ask for another policy)

// ---- XCD pacing ------------------------------------------------------------------------------------------------
// The weight stream of a tile program (5.4 MB object, 12.4 MB hand) is larger than an XCD's 4 MB of L2: it is served from
// L2 only while the XCD's 32 workgroups walk it within a few MB of one another, and they drift apart tile by tile
// (measured on the hand kernel: 67 - 78 % L2 hits, 0.4 TB of Infinity-Cache traffic per 16.8 M-sample launch, which in
// this power-limited kernel is clock).  So on launches of many tiles per workgroup the workgroups of an XCD meet at every
// tile start: one member count and one arrival counter per XCD (its own L2 keeps them coherent), a bounded spin --
// only in the rounds in which every workgroup still has a tile, and never longer than the timeout, so that an absent
// workgroup costs a delay, not a hang.  Measured (same box, C2 frame): 269.6 -> 253.4 ms, HBM-side traffic 0.94 -> 0.55 TB.
struct XcdPace {
    unsigned* c;   // 16 zeroed counters: [xcd] members, [8 + xcd] arrivals; NULL = off
    int xcd;
    __device__ __forceinline__ void init(unsigned* counters) {
        c = counters;
        xcd = 0;
        if (c != nullptr) {
            xcd = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;   // HW_REG_XCC_ID[3:0]
            if (threadIdx.x == 0) atomicAdd(&c[xcd], 1u);
        }
    }
    // meeting number e (1, 2, ...; every workgroup passes them in the same order): arrive, wait for members * e arrivals.
    // A meeting that times out (a workgroup of the XCD is not running beside the others: a shared or partitioned device)
    // switches the pacing of this workgroup off for the rest of the launch -- the others follow one timeout later.
    __device__ __forceinline__ void meet(int e) {
        __shared__ int timed_out;
        if (threadIdx.x == 0) {
            const unsigned members = __hip_atomic_load(&c[xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&c[8 + xcd], 1u);
            const unsigned target = members * (unsigned)e;
            int spin = 0;
            for (; spin < 3000; ++spin) {
                if (__hip_atomic_load(&c[8 + xcd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
                __builtin_amdgcn_s_sleep(16);
            }
            timed_out = spin == 3000;
        }
        __syncthreads();
        if (timed_out) c = nullptr;
        __syncthreads();   // (the flag is rewritten at the next meeting)
    }
    __device__ __forceinline__ bool on() const { return c != nullptr; }
};
#ifndef HN_XCD_PACING
#define HN_XCD_PACING 1
#endif
constexpr int XCD_PACE_MIN_ROUNDS = 8;   // tiles per workgroup from which a launch is paced
#ifndef HN_XCD_PACE_EVERY
#define HN_XCD_PACE_EVERY 1
#endif
constexpr int XCD_PACE_EVERY = HN_XCD_PACE_EVERY;   // meet at every n-th tile start

// Re-materialises a wave-uniform pointer in SGPRs behind an opaque asm so that the compiler cannot
// hoist the (hundreds of) addresses derived from it out of the persistent tile loop -- hoisted, they
// are all live across the whole loop and spill.
template <typename T>
__device__ __forceinline__ T* launder_uniform(T* p) {
    unsigned lo = (unsigned)(reinterpret_cast<uintptr_t>(p) & 0xffffffffu);
    unsigned hi = (unsigned)(reinterpret_cast<uintptr_t>(p) >> 32);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    asm volatile("" : "+s"(lo), "+s"(hi));
    // rebuilt as a GLOBAL pointer: from a bare integer the compiler would fall back to flat_* accesses
    typedef __attribute__((address_space(1))) T* gptr_t;
    return (T*)(gptr_t)((static_cast<uintptr_t>(hi) << 32) | lo);
}

// sin and cos of arguments up to ~1e4 rad (the encodings reach 2^9 |x|): 3-term Cody-Waite reduction by
// pi/2 with FMAs (exact product k * pi/2 to ~2^-70), then the cephes minimax polynomials on [-pi/4, pi/4].
// ~1 ulp, branch-free, no scratch (ocml's sincosf carries a Payne-Hanek path with a private table).
__device__ __forceinline__ void sincos_cw(float x, float& s, float& c) {
    const float kf = rintf(x * 0.63661977236758134308f);
    float r = fmaf(kf, -1.57079637050628662109375f, x);
    r = fmaf(kf, 4.37113900018624283e-8f, r);
    r = fmaf(kf, 1.71512449965966931e-15f, r);
    const float r2 = r * r;
    const float sp = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float cp = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                          fmaf(r2, -0.5f, 1.f));
    const int k = (int)kf;
    const float ss = (k & 1) ? cp : sp;
    const float cc = (k & 1) ? sp : cp;
    s = (k & 2) ? -ss : ss;
    c = ((k + 1) & 2) ? -cc : cc;
}

// Exchange between the two 32-lane halves with v_permlane32_swap_b32 (gfx950): a VALU instruction, where
// __shfl_xor(v, 32) becomes a ds_bpermute through the LDS crossbar.  The swap of (x, x) leaves {x.lo, x.lo} in
// the first and {x.hi, x.hi} in the second result.
__device__ __forceinline__ float half_sum(float v) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// 16x16x32 shape: a lane's tile registers belong to two samples (cb = (i >> 2) & 1), four lanes (g) share a sample.
// v[cb]: this lane's partial sums over its registers of column block cb -> the sample's total, returned in the lanes
// that own the sample in the old map (sample c16 + 16 q  <->  lanes with (lane >> 4) & 1 == q)
__device__ __forceinline__ float sample_sum(float v0, float v1) {
    if constexpr (!S16) return v0 + v1;   // (callers on the old shape use half_sum)
    unsigned a = __builtin_bit_cast(unsigned, v0), b = __builtin_bit_cast(unsigned, v1);
    auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);   // rows (0,1) and (2,3) paired
    const float s0 = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    const float s1 = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    a = __builtin_bit_cast(unsigned, s0);
    b = __builtin_bit_cast(unsigned, s1);
    r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const float t0 = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    const float t1 = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    const int q = (lane_x16() >> 8) & 1;
    return q ? t1 : t0;
}
// the other half's value of the same column: lane l <- lane l ^ 32  (h = l >> 5)
__device__ __forceinline__ float other_half(float v, int h) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return __builtin_bit_cast(float, (unsigned)(h ? r[0] : r[1]));
}

}  // namespace v2
}  // namespace hn
