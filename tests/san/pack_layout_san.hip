// Sanitizer harness (CPU, no GPU): the HOST side of the f16x3 weight-stream packer -- the layout planner of
// ho-nerf_amd/csrc/hn_pack2.hip (Builder, build_obj_stream / build_hand_stream: which matrix element goes where, index
// maps, chunk tails) -- compiled host-only with -fsanitize=address,undefined and run over matrices of the two fields'
// real shapes, for every program (mode) of both field kinds.  Checks beside the sanitizers':
// every fragment descriptor lies inside its program, every index-map entry names an element of its matrix (or the pad
// value -1), every program is a whole number of KiB.  TEST INFRASTRUCTURE: tests/test_abi.py builds and runs it
// (tests/san/Makefile); the product library does not contain this file.
#include "../../ho-nerf_amd/csrc/hn_pack2.hip"

namespace hn {
// the pieces of the other translation units the packer's device half refers to: never reached by the planner
void set_error(const char*, ...) {}
hipError_t pool_alloc_on(void**, size_t, hipStream_t) { return hipErrorNotSupported; }
void pool_free(void*) {}
int current_device() { return 0; }
}  // namespace hn

using namespace hn;
using namespace hn::v2;

static std::vector<std::vector<float>> g_store;
static HostMat make(int rows, int cols, int tensor) {
    g_store.emplace_back((size_t)rows * cols + rows);
    std::vector<float>& v = g_store.back();
    for (size_t i = 0; i < v.size(); ++i) v[i] = (float)((tensor * 7919 + (int)i) % 1021) / 1021.f - 0.5f;
    HostMat M;
    M.rows = rows;
    M.cols = cols;
    M.w = v.data();
    M.b = v.data() + (size_t)rows * cols;
    M.dev = nullptr;
    return M;
}

static int check(const Builder& B, const HostMat* S, const HostMat* C, const char* what) {
    int bad = 0;
    if (B.blob.size() % 1024 != 0) {
        fprintf(stderr, "%s: program of %zu bytes is not a whole number of KiB\n", what, B.blob.size());
        ++bad;
    }
    for (const FragBlock& fb : B.blocks) {
        if (fb.dst + KS_BYTES > B.blob.size()) ++bad;
        if (fb.rowmap < 0 || (size_t)fb.rowmap + 32 > B.maps.size() || fb.colslot < 0 || (size_t)fb.colslot + 16 * (fb.s + 1) > B.maps.size()) {
            ++bad;
            continue;
        }
        // the descriptor carries the matrix' column count; find the matrix by its device pointer stand-in (cols) and bound the maps
        int rows = 0;
        for (int l = 0; l < 9; ++l)
            if (S[l].cols == fb.cols && S[l].rows > rows) rows = S[l].rows;
        for (int l = 0; l < 5; ++l)
            if (C[l].cols == fb.cols && C[l].rows > rows) rows = C[l].rows;
        const int n_r = fb.transposed ? fb.cols : rows, n_c = fb.transposed ? rows : fb.cols;
        for (int r = 0; r < 32; ++r) {
            const int v = B.maps[fb.rowmap + r];
            if (v < -1 || v >= n_r) ++bad;
        }
        for (int k = 0; k < 16; ++k) {
            const int v = B.maps[fb.colslot + 16 * fb.s + k];
            if (v < -1 || v >= n_c) ++bad;
        }
    }
    if (bad) fprintf(stderr, "%s: %d descriptor / map entries out of range\n", what, bad);
    printf("%-28s %9zu bytes %6zu fragment blocks %7zu map entries\n", what, B.blob.size(), B.blocks.size(), B.maps.size());
    return bad;
}

int main() {
    int bad = 0;
    // utils/fields.py:252-314 (object nets) and :57-130, 180-240 (hand nets): layer shapes [out, in]
    const int obj_s[9][2] = {{256, 63}, {256, 256}, {256, 256}, {193, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {257, 256}};
    const int obj_c[5][2] = {{256, 373}, {256, 256}, {256, 256}, {256, 256}, {3, 256}};
    const int hand_s[9][2] = {{256, 1386}, {256, 256}, {256, 256}, {256, 256}, {256, 1642}, {256, 256}, {256, 256}, {256, 256}, {257, 256}};
    const int hand_c[5][2] = {{256, 1669}, {256, 256}, {256, 256}, {256, 256}, {3, 256}};
    for (int kind = 0; kind < 2; ++kind) {
        HostMat S[9], C[5];
        for (int l = 0; l < 9; ++l) S[l] = kind == 0 ? make(obj_s[l][0], obj_s[l][1], l) : make(hand_s[l][0], hand_s[l][1], 20 + l);
        for (int l = 0; l < 5; ++l) C[l] = kind == 0 ? make(obj_c[l][0], obj_c[l][1], 10 + l) : make(hand_c[l][0], hand_c[l][1], 30 + l);
        for (int mode = 0; mode < 4; ++mode) {   // the four programs of a field on the product's MFMA shape (32x32x16)
            Builder B;
            if (kind == 0)
                build_obj_stream(B, S, C, mode);
            else
                build_hand_stream(B, S, C, mode);
            char what[64];
            snprintf(what, sizeof(what), "%s mode %d", kind == 0 ? "obj" : "hand", mode);
            bad += check(B, S, C, what);
        }
    }
    printf(bad ? "FAILED\n" : "pack layout: ok\n");
    return bad ? 1 : 0;
}
