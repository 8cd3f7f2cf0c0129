// The swizzled [sample][column] image of k_outer_group_t, written as its staging does and read back as its MFMA fragments are:
// every lane checks that element e of fragment (ks, hh) of column tile `colbase` is sample 16 ks + 8 (lane >> 5) + 4 hh + e, column colbase + (lane & 31).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int ogt_swz(int row) { return (8 * (row & 3)) ^ ((row >> 2) & 7); }
__global__ void k(int* bad) {
    __shared__ __attribute__((aligned(16))) char lds[32 * 256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int u = 0; u < 4; ++u) {   // row-major staging map: row (t >> 5) + 8 u, column quad t & 31
        const int row = (t >> 5) + 8 * u, cq = t & 31;
        short v[4];
        for (int r = 0; r < 4; ++r) v[r] = (short)(row * 128 + 4 * cq + r);
        char* w = lds + row * 256 + ((cq ^ ogt_swz(row)) << 3);
        *reinterpret_cast<uint2*>(w) = make_uint2((unsigned)(unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16), (unsigned)(unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16));
    }
    __syncthreads();
    const int hq = lane >> 5, g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;
    int nb = 0;
    for (int ks = 0; ks < 2; ++ks)
        for (int hh = 0; hh < 2; ++hh)
            for (int x = 0; x < 4; ++x) {
                const int colbase = 32 * x;
                const int row = 16 * ks + 8 * hq + 4 * hh + q4;
                const int cqa = ((colbase + 16 * g16) >> 2) + pp;
                const s16x4 ta = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + row * 256 + ((cqa ^ ogt_swz(row)) << 3)));
                for (int e = 0; e < 4; ++e) {
                    const int want = (16 * ks + 8 * hq + 4 * hh + e) * 128 + colbase + (lane & 31);
                    if (ta[e] != (short)want) ++nb;
                }
            }
    // ... and assembled into the 8-element MFMA operand the way the kernel does it
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    for (int ks = 0; ks < 2; ++ks)
        for (int x = 0; x < 4; ++x) {
            s16x4 th[2];
            for (int hh = 0; hh < 2; ++hh) {
                const int row = 16 * ks + 8 * hq + 4 * hh + q4;
                const int cqa = ((32 * x + 16 * g16) >> 2) + pp;
                th[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + row * 256 + ((cqa ^ ogt_swz(row)) << 3)));
            }
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 A16 = __builtin_shufflevector(th[0], th[1], 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 A = __builtin_bit_cast(bf16x8, A16);   // (what the MFMA takes; checked through the shorts: element access on a
            (void)A;                                            //  __bf16 vector is not what this probe is about)
            for (int e = 0; e < 8; ++e) {
                const int want = (16 * ks + 8 * hq + e) * 128 + 32 * x + (lane & 31);
                if (A16[e] != (short)want) ++nb;
            }
        }
    if (wave == 0) atomicAdd(bad, nb);
}
int main() {
    int* d;
    (void)hipMalloc(&d, sizeof(int));
    (void)hipMemset(d, 0, sizeof(int));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
    int h = -1;
    (void)hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost);
    printf("k_outer_group_t image probe: %d mismatches\n", h);
    return h != 0;
}
