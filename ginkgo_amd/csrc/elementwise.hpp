// Element-wise launch framework for HBM-bound vector updates (dense axpy family,
// fused CG / GMRES steps, scalar Jacobi).
//
// Ginkgo's stock GPU launcher (common/cuda_hip/base/kernel_launch.hpp:48-101)
// runs one thread per element with a 64-bit div/mod per element.  Here:
//  * flat path (every operand contiguous, i.e. ld == cols, and the scalars do
//    not vary per column, or cols == 1): grid-stride over 16-byte vectors
//    (double2 / float4), all inputs of an iteration are loaded before the
//    first store, so each lane keeps NIN*16 B in flight;
//  * general path (strided views, per-column scalars with cols > 1): 2-D
//    indexed kernel, threads fastest along the column dimension.
// An OP provides
//    struct scalars;                                       per-column values
//    __device__ scalars load(int64_t col) const;          read device scalars
//    __device__ bool skip(const scalars&) const;          whole column no-op
//    __device__ void apply(const scalars&, const T* in, T* out) const;
// and optionally
//    __device__ void store(int64_t col, const scalars&) const;   write back a
//        per-column scalar computed in load(); called by ONE thread per column
//        that is not skipped.  load() must not depend on what store() writes
//        (or the written value must equal the value read).
//    __device__ void note(int64_t col, const scalars&) const;    like store(), but called for
//        skipped columns as well (a criterion evaluated in load() records its verdict).
// Outputs may alias inputs at the same element index only.
#pragma once
#include "common.hpp"

namespace gkoc {

template <typename T, int NIN, int NOUT>
struct ew_operands {
    const T* in[NIN > 0 ? NIN : 1];
    int64_t ld_in[NIN > 0 ? NIN : 1];
    T* out[NOUT];
    int64_t ld_out[NOUT];
};

#ifdef __HIPCC__

template <typename T>
struct vec16 {
    static constexpr int width = 16 / sizeof(T);
    T v[16 / sizeof(T)];
} __attribute__((aligned(16)));

template <typename OP, typename S>
__device__ __forceinline__ auto ew_store(const OP& op, int64_t col, const S& sc, int)
    -> decltype(op.store(col, sc), void())
{
    op.store(col, sc);
}
template <typename OP, typename S>
__device__ __forceinline__ void ew_store(const OP&, int64_t, const S&, long)
{}

template <typename OP, typename S>
__device__ __forceinline__ auto ew_note(const OP& op, int64_t col, const S& sc, int)
    -> decltype(op.note(col, sc), void())
{
    op.note(col, sc);
}
template <typename OP, typename S>
__device__ __forceinline__ void ew_note(const OP&, int64_t, const S&, long)
{}

template <typename T, typename OP, int NIN, int NOUT>
__global__ __launch_bounds__(256) void ew_flat_vec_kernel(
    int64_t n, ew_operands<T, NIN, NOUT> a, OP op)
{
    using V = vec16<T>;
    constexpr int W = V::width;
    const auto sc = op.load(0);
    if (blockIdx.x == 0 && threadIdx.x == 0) ew_note(op, 0, sc, 0);
    if (op.skip(sc)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) ew_store(op, 0, sc, 0);
    const int64_t n_vec = n / W;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n_vec;
         i += stride) {
        V in[NIN > 0 ? NIN : 1];
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            in[k] = reinterpret_cast<const V*>(a.in[k])[i];
        }
        V out[NOUT];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            T ie[NIN > 0 ? NIN : 1], oe[NOUT];
#pragma unroll
            for (int k = 0; k < NIN; ++k) ie[k] = in[k].v[e];
            op.apply(sc, ie, oe);
#pragma unroll
            for (int k = 0; k < NOUT; ++k) out[k].v[e] = oe[k];
        }
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            reinterpret_cast<V*>(a.out[k])[i] = out[k];
        }
    }
    // scalar tail
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int64_t i = n_vec * W; i < n; ++i) {
            T ie[NIN > 0 ? NIN : 1], oe[NOUT];
#pragma unroll
            for (int k = 0; k < NIN; ++k) ie[k] = a.in[k][i];
            op.apply(sc, ie, oe);
#pragma unroll
            for (int k = 0; k < NOUT; ++k) a.out[k][i] = oe[k];
        }
    }
}

template <typename T, typename OP, int NIN, int NOUT>
__global__ __launch_bounds__(256) void ew_general_kernel(
    int64_t rows, int64_t cols, ew_operands<T, NIN, NOUT> a, OP op)
{
    const int64_t total = rows * cols;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total;
         idx += stride) {
        const int64_t row = idx / cols;
        const int64_t col = idx - row * cols;
        const auto sc = op.load(col);
        if (row == 0) ew_note(op, col, sc, 0);
        if (op.skip(sc)) continue;
        if (row == 0) ew_store(op, col, sc, 0);
        T ie[NIN > 0 ? NIN : 1], oe[NOUT];
#pragma unroll
        for (int k = 0; k < NIN; ++k) ie[k] = a.in[k][row * a.ld_in[k] + col];
        op.apply(sc, ie, oe);
#pragma unroll
        for (int k = 0; k < NOUT; ++k) a.out[k][row * a.ld_out[k] + col] = oe[k];
    }
}

// `uniform_scalars`: the OP's scalars are the same for every column
template <typename T, typename OP, int NIN, int NOUT>
int launch_elementwise(gkoc_stream_t s, int64_t rows, int64_t cols,
                       const ew_operands<T, NIN, NOUT>& a, const OP& op,
                       bool uniform_scalars)
{
    GKOC_REQUIRE(rows >= 0 && cols >= 0, GKOC_E_INVALID, "negative dimension");
    if (rows == 0 || cols == 0) return GKOC_OK;
    bool flat = (cols == 1) || uniform_scalars;
    bool aligned = true;
    for (int k = 0; k < NIN; ++k) {
        GKOC_REQUIRE(a.in[k] && a.ld_in[k] >= cols, GKOC_E_INVALID,
                     "bad input operand");
        flat = flat && a.ld_in[k] == cols;
        aligned = aligned && (reinterpret_cast<uintptr_t>(a.in[k]) % 16 == 0);
    }
    for (int k = 0; k < NOUT; ++k) {
        GKOC_REQUIRE(a.out[k] && a.ld_out[k] >= cols, GKOC_E_INVALID,
                     "bad output operand");
        flat = flat && a.ld_out[k] == cols;
        aligned = aligned && (reinterpret_cast<uintptr_t>(a.out[k]) % 16 == 0);
    }
    const int64_t n = rows * cols;
    if (flat && aligned) {
        const int64_t n_vec = n / vec16<T>::width;
        int64_t blocks = ceildiv(n_vec > 0 ? n_vec : 1, 256);
        if (blocks > max_stream_blocks) blocks = max_stream_blocks;
        ew_flat_vec_kernel<T, OP, NIN, NOUT>
            <<<dim3(unsigned(blocks)), dim3(256), 0, as_stream(s)>>>(n, a, op);
    } else {
        int64_t blocks = ceildiv(n, 256);
        if (blocks > 4 * max_stream_blocks) blocks = 4 * max_stream_blocks;
        ew_general_kernel<T, OP, NIN, NOUT>
            <<<dim3(unsigned(blocks)), dim3(256), 0, as_stream(s)>>>(rows, cols,
                                                                    a, op);
    }
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

#endif  // __HIPCC__

}  // namespace gkoc
