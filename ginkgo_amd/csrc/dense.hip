// Dense BLAS-1 for the Krylov inner loop on gfx950.
//
// Replaces gko::kernels::hip::dense::{fill, copy, scale, inv_scale, add_scaled,
// sub_scaled, compute_dot(_dispatch), compute_conj_dot(_dispatch),
// compute_norm2(_dispatch), compute_squared_norm2, compute_sqrt, row_gather}
// (decl core/matrix/dense_kernels.hpp:34-131; semantics
// reference/matrix/dense_kernels.cpp:96-352; stock GPU versions
// common/unified/matrix/dense_kernels.template.cpp:29-267,449-473 and the
// hipBLAS dispatch common/cuda_hip/matrix/dense_kernels.cpp:667-735).
//
// Element-wise ops are bit-identical to the reference (one multiply + one add
// per element, no FMA contraction).  Reductions use a fixed two-level tree:
//   stage 1: <= 1024 workgroups, each lane accumulates a strided subsequence
//            with 16-byte loads, wave shuffle tree, LDS across the 4 waves;
//   stage 2: one workgroup folds the <= 1024 partials in a fixed order.
// => deterministic for a given (n, nrhs), no atomics, no hipBLAS handle.
// Algorithmic HBM bytes: dot 2*n*sizeof(T), norm2 n*sizeof(T), axpy 3*n*sizeof(T).
#include <cmath>

#include "common.hpp"
#include "elementwise.hpp"

namespace gkoc {
namespace {

// ---------------------------------------------------------------- ew ops
template <typename T>
struct op_fill {
    T value;
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T*, T* out) const
    {
        out[0] = value;
    }
};

template <typename T>
struct op_copy {
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
        out[0] = in[0];
    }
};

// dense::inplace_absolute_dense / outplace_absolute_dense (real types),
// reference/matrix/dense_kernels.cpp:1178-1202
template <typename T>
struct op_abs {
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
        out[0] = in[0] < T(0) ? -in[0] : in[0];
    }
};

// reference/matrix/dense_kernels.cpp:126-148: single alpha == 0 => x = 0
template <typename T>
struct op_scale {
    const T* alpha;
    int64_t alpha_cols;
    struct scalars {
        T a;
        bool zero_fill;
    };
    __device__ scalars load(int64_t col) const
    {
        const T a = alpha[alpha_cols == 1 ? 0 : col];
        return {a, alpha_cols == 1 && a == T(0)};
    }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = s.zero_fill ? T(0) : in[0] * s.a;
    }
};

template <typename T>
struct op_inv_scale {
    const T* alpha;
    int64_t alpha_cols;
    struct scalars {
        T a;
    };
    __device__ scalars load(int64_t col) const
    {
        return {alpha[alpha_cols == 1 ? 0 : col]};
    }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = in[0] / s.a;
    }
};

// y (+/-)= alpha * x ; in[0] = x, in[1] = y
// reference/matrix/dense_kernels.cpp:178-222: single alpha == 0 => no-op
template <typename T, bool SUB>
struct op_axpy {
    const T* alpha;
    int64_t alpha_cols;
    struct scalars {
        T a;
        bool noop;
    };
    __device__ scalars load(int64_t col) const
    {
        const T a = alpha[alpha_cols == 1 ? 0 : col];
        return {a, alpha_cols == 1 && a == T(0)};
    }
    __device__ bool skip(const scalars& s) const { return s.noop; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        const T t = s.a * in[0];
        out[0] = SUB ? in[1] - t : in[1] + t;
    }
};

// ------------------------------------------------------------ reductions
constexpr int red_block = 256;
constexpr int max_partials = 1024;

template <typename T, int SQUARE>
__device__ __forceinline__ T red_term(T x, T y)
{
    // SQUARE: 0 x y (dot), 1 x x (squared norm), 2 |x| (1-norm)
    return SQUARE == 2 ? (x < T(0) ? -x : x) : SQUARE ? x * x : x * y;
}

// flat (ld == 1, one column) stage 1
// FOLD != 0: the block that finishes LAST also does stage 2 (the same 256 threads, the same strided
// sums and the same tree as reduce_stage2: the same bits), FOLD == 2 with the square root.  `ticket`
// is a device word that is 0 between launches (one per stream: launches of a stream do not overlap);
// the partial sums reach memory before the ticket is taken (agent-scope fences on both sides: the
// blocks run on eight XCDs with an L2 each).
template <typename T, int SQUARE, int FOLD = 0>
__global__ __launch_bounds__(red_block) void reduce_flat_stage1(
    int64_t n, const T* __restrict__ x, const T* __restrict__ y,
    T* __restrict__ partial, bool vec_ok, unsigned* __restrict__ ticket = nullptr,
    T* __restrict__ result = nullptr)
{
    __shared__ T lds[red_block / 64];
    using V = vec16<T>;
    constexpr int W = V::width;
    T acc = T(0);
    const int64_t tid = int64_t(blockIdx.x) * red_block + threadIdx.x;
    const int64_t nthreads = int64_t(gridDim.x) * red_block;
    if (vec_ok) {
        const int64_t n_vec = n / W;
        // two independent accumulators / iterations in flight per lane
        T acc2 = T(0);
        int64_t i = tid;
        for (; i + nthreads < n_vec; i += 2 * nthreads) {
            const V xa = reinterpret_cast<const V*>(x)[i];
            const V xb = reinterpret_cast<const V*>(x)[i + nthreads];
            V ya, yb;
            if (!SQUARE) {
                ya = reinterpret_cast<const V*>(y)[i];
                yb = reinterpret_cast<const V*>(y)[i + nthreads];
            }
#pragma unroll
            for (int e = 0; e < W; ++e) {
                acc += red_term<T, SQUARE>(xa.v[e], SQUARE ? xa.v[e] : ya.v[e]);
                acc2 += red_term<T, SQUARE>(xb.v[e], SQUARE ? xb.v[e] : yb.v[e]);
            }
        }
        for (; i < n_vec; i += nthreads) {
            const V xa = reinterpret_cast<const V*>(x)[i];
            V ya;
            if (!SQUARE) ya = reinterpret_cast<const V*>(y)[i];
#pragma unroll
            for (int e = 0; e < W; ++e) {
                acc += red_term<T, SQUARE>(xa.v[e], SQUARE ? xa.v[e] : ya.v[e]);
            }
        }
        acc += acc2;
        if (tid == 0) {
            for (int64_t k = n_vec * W; k < n; ++k) {
                acc += red_term<T, SQUARE>(x[k], SQUARE ? x[k] : y[k]);
            }
        }
    } else {
        for (int64_t i = tid; i < n; i += nthreads) {
            acc += red_term<T, SQUARE>(x[i], SQUARE ? x[i] : y[i]);
        }
    }
    const T r = block_sum<red_block>(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
    if constexpr (FOLD != 0) {
        __shared__ unsigned last;
        if (threadIdx.x == 0) {
            __threadfence();                                  // release: the partial sum first
            last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (last == 0u) return;
        __threadfence();                                      // acquire: every thread that reads the sums
        const T* all = partial;
        T acc2 = T(0);
        for (int i = threadIdx.x; i < int(gridDim.x); i += red_block) {
            acc2 += __builtin_nontemporal_load(all + i);
        }
        __syncthreads();                                      // (lds is reused)
        const T total = block_sum<red_block>(acc2, lds);
        if (threadIdx.x == 0) {
            result[0] = FOLD == 2 ? sqrt(total) : total;
            *ticket = 0u;                                     // ready for the stream's next launch
        }
    }
}

// general (strided, multi-column) stage 1: grid = (blocks_x, cols)
template <typename T, int SQUARE>
__global__ __launch_bounds__(red_block) void reduce_cols_stage1(
    int64_t rows, const T* __restrict__ x, int64_t ldx,
    const T* __restrict__ y, int64_t ldy, T* __restrict__ partial)
{
    __shared__ T lds[red_block / 64];
    const int64_t col = blockIdx.y;
    T acc = T(0);
    const int64_t nthreads = int64_t(gridDim.x) * red_block;
    for (int64_t i = int64_t(blockIdx.x) * red_block + threadIdx.x; i < rows;
         i += nthreads) {
        const T xv = x[i * ldx + col];
        acc += red_term<T, SQUARE>(xv, SQUARE ? xv : y[i * ldy + col]);
    }
    const T r = block_sum<red_block>(acc, lds);
    if (threadIdx.x == 0) partial[col * gridDim.x + blockIdx.x] = r;
}

// stage 2: grid = cols; folds n_partials values per column, optional sqrt
template <typename T, bool SQRT>
__global__ __launch_bounds__(red_block) void reduce_stage2(
    int n_partials, const T* __restrict__ partial, T* __restrict__ result)
{
    __shared__ T lds[red_block / 64];
    const int64_t col = blockIdx.x;
    T acc = T(0);
    for (int i = threadIdx.x; i < n_partials; i += red_block) {
        acc += partial[col * n_partials + i];
    }
    const T r = block_sum<red_block>(acc, lds);
    if (threadIdx.x == 0) result[col] = SQRT ? sqrt(r) : r;
}

template <typename T, int SQUARE, bool SQRT>
int launch_reduce(gkoc_stream_t s, int64_t rows, int64_t cols, const T* x,
                  int64_t ldx, const T* y, int64_t ldy, T* result, void* work,
                  size_t work_bytes)
{
    GKOC_REQUIRE(rows >= 0 && cols >= 0, GKOC_E_INVALID, "negative dimension");
    if (cols == 0) return GKOC_OK;
    GKOC_REQUIRE(result, GKOC_E_INVALID, "null result");
    if (rows == 0) {
        GKOC_HIP(hipMemsetAsync(result, 0, sizeof(T) * cols, as_stream(s)));
        return GKOC_OK;
    }
    GKOC_REQUIRE(x && (SQUARE || y), GKOC_E_INVALID, "null operand");
    GKOC_REQUIRE(work_bytes >= gkoc_reduction_workspace_bytes(rows, cols, sizeof(T)),
                 GKOC_E_WORKSPACE, "reduction workspace too small");
    GKOC_REQUIRE(reinterpret_cast<uintptr_t>(work) % sizeof(T) == 0,
                 GKOC_E_INVALID, "misaligned workspace");
    T* partial = static_cast<T*>(work);
    int n_partials;
    if (cols == 1 && ldx == 1 && (SQUARE || ldy == 1)) {
        const bool vec_ok =
            reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
            (SQUARE || reinterpret_cast<uintptr_t>(y) % 16 == 0);
        const int64_t per_block = int64_t(red_block) * vec16<T>::width * 4;
        int64_t nb = ceildiv(rows, per_block);
        if (nb > max_partials) nb = max_partials;
        n_partials = static_cast<int>(nb);
        if (nb > 1 && tune_value(GKOC_TUNE_REDUCE_ONE_KERNEL) != 0) {
            // one launch: the last block folds (5 us of launch + tiny kernel per reduction otherwise)
            unsigned* ticket = nullptr;
            GKOC_TRY(stream_ticket(as_stream(s), &ticket));
            reduce_flat_stage1<T, SQUARE, SQRT ? 2 : 1>
                <<<dim3(unsigned(nb)), dim3(red_block), 0, as_stream(s)>>>(
                    rows, x, y, partial, vec_ok, ticket, result);
            GKOC_LAUNCH_OK();
            return GKOC_OK;
        }
        reduce_flat_stage1<T, SQUARE>
            <<<dim3(unsigned(nb)), dim3(red_block), 0, as_stream(s)>>>(
                rows, x, y, partial, vec_ok);
    } else {
        int64_t nb = ceildiv(rows, int64_t(red_block) * 4);
        const int64_t cap = max_partials / (cols < max_partials ? cols : max_partials);
        if (nb > cap) nb = cap;
        if (nb < 1) nb = 1;
        n_partials = static_cast<int>(nb);
        reduce_cols_stage1<T, SQUARE>
            <<<dim3(unsigned(nb), unsigned(cols)), dim3(red_block), 0,
               as_stream(s)>>>(rows, x, ldx, y, ldy, partial);
    }
    GKOC_LAUNCH_OK();
    reduce_stage2<T, SQRT>
        <<<dim3(unsigned(cols)), dim3(red_block), 0, as_stream(s)>>>(
            n_partials, partial, result);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename T>
__global__ void sqrt_kernel(int64_t n, T* x)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) x[i] = sqrt(x[i]);
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void row_gather_kernel(
    int64_t n_gather, int64_t cols, const I* __restrict__ rows,
    const T* __restrict__ orig, int64_t ld_orig, T* __restrict__ gathered,
    int64_t ld_gathered)
{
    const int64_t total = n_gather * cols;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total;
         idx += stride) {
        const int64_t r = idx / cols;
        const int64_t c = idx - r * cols;
        gathered[r * ld_gathered + c] = orig[int64_t(rows[r]) * ld_orig + c];
    }
}

template <typename T, typename I>
int launch_row_gather(gkoc_stream_t s, int64_t n_gather, int64_t cols,
                      const I* rows, const T* orig, int64_t ld_orig,
                      T* gathered, int64_t ld_gathered)
{
    if (n_gather <= 0 || cols <= 0) return GKOC_OK;
    int64_t blocks = ceildiv(n_gather * cols, 256);
    if (blocks > 4 * max_stream_blocks) blocks = 4 * max_stream_blocks;
    row_gather_kernel<T, I><<<dim3(unsigned(blocks)), dim3(256), 0, as_stream(s)>>>(
        n_gather, cols, rows, orig, ld_orig, gathered, ld_gathered);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

extern "C" size_t gkoc_reduction_workspace_bytes(int64_t, int64_t nrhs,
                                                 size_t value_size)
{
    const int64_t c = nrhs < 1 ? 1 : nrhs;
    // up to max_partials partials in total, but never fewer than one per column
    const int64_t n = c > max_partials ? c : max_partials;
    return static_cast<size_t>(n) * value_size;
}

#define GKOC_DEF_DENSE(T, TN)                                                  \
    extern "C" int gkoc_dense_fill_##TN(gkoc_stream_t s, int64_t rows,         \
                                        int64_t cols, T* x, int64_t ldx,       \
                                        T value)                               \
    {                                                                          \
        ew_operands<T, 0, 1> a{};                                              \
        a.out[0] = x;                                                          \
        a.ld_out[0] = ldx;                                                     \
        return launch_elementwise<T, op_fill<T>, 0, 1>(s, rows, cols, a,       \
                                                       op_fill<T>{value},      \
                                                       true);                  \
    }                                                                          \
    extern "C" int gkoc_dense_copy_##TN(gkoc_stream_t s, int64_t rows,         \
                                        int64_t cols, const T* x,              \
                                        int64_t ldx, T* y, int64_t ldy)        \
    {                                                                          \
        ew_operands<T, 1, 1> a{};                                              \
        a.in[0] = x;                                                           \
        a.ld_in[0] = ldx;                                                      \
        a.out[0] = y;                                                          \
        a.ld_out[0] = ldy;                                                     \
        return launch_elementwise<T, op_copy<T>, 1, 1>(s, rows, cols, a,       \
                                                       op_copy<T>{}, true);    \
    }                                                                          \
    extern "C" int gkoc_dense_absolute_##TN(gkoc_stream_t s, int64_t rows,      \
                                            int64_t cols, const T* x,          \
                                            int64_t ldx, T* y, int64_t ldy)    \
    {                                                                          \
        ew_operands<T, 1, 1> a{};                                              \
        a.in[0] = x;                                                           \
        a.ld_in[0] = ldx;                                                      \
        a.out[0] = y;                                                          \
        a.ld_out[0] = ldy;                                                     \
        return launch_elementwise<T, op_abs<T>, 1, 1>(s, rows, cols, a,        \
                                                      op_abs<T>{}, true);      \
    }                                                                          \
    extern "C" int gkoc_dense_scale_##TN(gkoc_stream_t s, int64_t rows,        \
                                         int64_t cols, const T* alpha,         \
                                         int64_t alpha_cols, T* x,             \
                                         int64_t ldx)                          \
    {                                                                          \
        if (rows <= 0 || cols <= 0) return GKOC_OK; /* nothing to do: alpha may be 1 x 0 / NULL */ \
        GKOC_REQUIRE(alpha && (alpha_cols == 1 || alpha_cols == cols),         \
                     GKOC_E_INVALID, "bad alpha");                             \
        ew_operands<T, 1, 1> a{};                                              \
        a.in[0] = x;                                                           \
        a.ld_in[0] = ldx;                                                      \
        a.out[0] = x;                                                          \
        a.ld_out[0] = ldx;                                                     \
        return launch_elementwise<T, op_scale<T>, 1, 1>(                       \
            s, rows, cols, a, op_scale<T>{alpha, alpha_cols},                  \
            alpha_cols == 1);                                                  \
    }                                                                          \
    extern "C" int gkoc_dense_inv_scale_##TN(gkoc_stream_t s, int64_t rows,    \
                                             int64_t cols, const T* alpha,     \
                                             int64_t alpha_cols, T* x,         \
                                             int64_t ldx)                      \
    {                                                                          \
        if (rows <= 0 || cols <= 0) return GKOC_OK; /* nothing to do: alpha may be 1 x 0 / NULL */ \
        GKOC_REQUIRE(alpha && (alpha_cols == 1 || alpha_cols == cols),         \
                     GKOC_E_INVALID, "bad alpha");                             \
        ew_operands<T, 1, 1> a{};                                              \
        a.in[0] = x;                                                           \
        a.ld_in[0] = ldx;                                                      \
        a.out[0] = x;                                                          \
        a.ld_out[0] = ldx;                                                     \
        return launch_elementwise<T, op_inv_scale<T>, 1, 1>(                   \
            s, rows, cols, a, op_inv_scale<T>{alpha, alpha_cols},              \
            alpha_cols == 1);                                                  \
    }                                                                          \
    extern "C" int gkoc_dense_add_scaled_##TN(                                 \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* alpha,           \
        int64_t alpha_cols, const T* x, int64_t ldx, T* y, int64_t ldy)        \
    {                                                                          \
        if (rows <= 0 || cols <= 0) return GKOC_OK; /* nothing to do: alpha may be 1 x 0 / NULL */ \
        GKOC_REQUIRE(alpha && (alpha_cols == 1 || alpha_cols == cols),         \
                     GKOC_E_INVALID, "bad alpha");                             \
        ew_operands<T, 2, 1> a{};                                              \
        a.in[0] = x;                                                           \
        a.ld_in[0] = ldx;                                                      \
        a.in[1] = y;                                                           \
        a.ld_in[1] = ldy;                                                      \
        a.out[0] = y;                                                          \
        a.ld_out[0] = ldy;                                                     \
        return launch_elementwise<T, op_axpy<T, false>, 2, 1>(                 \
            s, rows, cols, a, op_axpy<T, false>{alpha, alpha_cols},            \
            alpha_cols == 1);                                                  \
    }                                                                          \
    extern "C" int gkoc_dense_sub_scaled_##TN(                                 \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* alpha,           \
        int64_t alpha_cols, const T* x, int64_t ldx, T* y, int64_t ldy)        \
    {                                                                          \
        if (rows <= 0 || cols <= 0) return GKOC_OK; /* nothing to do: alpha may be 1 x 0 / NULL */ \
        GKOC_REQUIRE(alpha && (alpha_cols == 1 || alpha_cols == cols),         \
                     GKOC_E_INVALID, "bad alpha");                             \
        ew_operands<T, 2, 1> a{};                                              \
        a.in[0] = x;                                                           \
        a.ld_in[0] = ldx;                                                      \
        a.in[1] = y;                                                           \
        a.ld_in[1] = ldy;                                                      \
        a.out[0] = y;                                                          \
        a.ld_out[0] = ldy;                                                     \
        return launch_elementwise<T, op_axpy<T, true>, 2, 1>(                  \
            s, rows, cols, a, op_axpy<T, true>{alpha, alpha_cols},             \
            alpha_cols == 1);                                                  \
    }                                                                          \
    extern "C" int gkoc_dense_compute_dot_##TN(                                \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* x, int64_t ldx,  \
        const T* y, int64_t ldy, T* result, void* work, size_t work_bytes)     \
    {                                                                          \
        return launch_reduce<T, false, false>(s, rows, cols, x, ldx, y, ldy,   \
                                              result, work, work_bytes);       \
    }                                                                          \
    extern "C" int gkoc_dense_compute_norm2_##TN(                              \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* x, int64_t ldx,  \
        T* result, void* work, size_t work_bytes)                              \
    {                                                                          \
        return launch_reduce<T, true, true>(s, rows, cols, x, ldx, nullptr, 0, \
                                            result, work, work_bytes);         \
    }                                                                          \
    extern "C" int gkoc_dense_compute_norm1_##TN(                              \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* x, int64_t ldx,  \
        T* result, void* work, size_t work_bytes)                              \
    {                                                                          \
        return launch_reduce<T, 2, false>(s, rows, cols, x, ldx, nullptr, 0,   \
                                          result, work, work_bytes);           \
    }                                                                          \
    extern "C" int gkoc_dense_compute_squared_norm2_##TN(                      \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* x, int64_t ldx,  \
        T* result, void* work, size_t work_bytes)                              \
    {                                                                          \
        return launch_reduce<T, true, false>(s, rows, cols, x, ldx, nullptr,   \
                                             0, result, work, work_bytes);     \
    }                                                                          \
    extern "C" int gkoc_dense_compute_sqrt_##TN(gkoc_stream_t s,               \
                                                int64_t cols, T* x)            \
    {                                                                          \
        if (cols <= 0) return GKOC_OK;                                         \
        sqrt_kernel<T><<<dim3(unsigned(ceildiv(cols, 256))), dim3(256), 0,     \
                         as_stream(s)>>>(cols, x);                             \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_dense_row_gather_##TN##_i32(                           \
        gkoc_stream_t s, int64_t n_gather, int64_t cols, const int32_t* rows,  \
        const T* orig, int64_t ld_orig, T* gathered, int64_t ld_gathered)      \
    {                                                                          \
        return launch_row_gather<T, int32_t>(s, n_gather, cols, rows, orig,    \
                                             ld_orig, gathered, ld_gathered);  \
    }                                                                          \
    extern "C" int gkoc_dense_row_gather_##TN##_i64(                           \
        gkoc_stream_t s, int64_t n_gather, int64_t cols, const int64_t* rows,  \
        const T* orig, int64_t ld_orig, T* gathered, int64_t ld_gathered)      \
    {                                                                          \
        return launch_row_gather<T, int64_t>(s, n_gather, cols, rows, orig,    \
                                             ld_orig, gathered, ld_gathered);  \
    }

GKOC_DEF_DENSE(double, f64)
GKOC_DEF_DENSE(float, f32)

// ---- complex pairs: fill, fill_seq, column 2-norms ------------------------------------------
namespace gkoc {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void fill_pair_kernel(int64_t n, T* __restrict__ data, T value, int seq)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
        T v = value;
        if (seq) {
            v.re = decltype(v.re)(i);
            v.im = 0;
        }
        data[i] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fill_pair_2d_kernel(int64_t rows, int64_t cols, T* __restrict__ x,
                                                          int64_t ld, T value)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < rows * cols; i += stride) {
        x[(i / cols) * ld + i % cols] = value;
    }
}

// stage 1 of the column norms of a complex matrix: x as real numbers, ld in real elements
template <typename R>
__global__ __launch_bounds__(red_block) void cnorm2_stage1(int64_t rows, const R* __restrict__ x,
                                                           int64_t ld, R* __restrict__ partial)
{
    __shared__ R lds[red_block / 64];
    const int64_t col = blockIdx.y;
    R acc = R(0);
    const int64_t nthreads = int64_t(gridDim.x) * red_block;
    for (int64_t i = int64_t(blockIdx.x) * red_block + threadIdx.x; i < rows; i += nthreads) {
        const R re = x[i * ld + 2 * col], im = x[i * ld + 2 * col + 1];
        acc += re * re + im * im;   // squared_norm(z) = real(conj(z) z)
    }
    const R r = block_sum<red_block>(acc, lds);
    if (threadIdx.x == 0) partial[col * gridDim.x + blockIdx.x] = r;
}

template <typename R>
int launch_cnorm2(gkoc_stream_t s, int64_t rows, int64_t cols, const R* x, int64_t ld_complex, R* result,
                  void* work, size_t work_bytes)
{
    GKOC_REQUIRE(rows >= 0 && cols >= 0, GKOC_E_INVALID, "negative dimension");
    if (cols == 0) return GKOC_OK;
    GKOC_REQUIRE(result, GKOC_E_INVALID, "null result");
    if (rows == 0) {
        GKOC_HIP(hipMemsetAsync(result, 0, sizeof(R) * cols, as_stream(s)));
        return GKOC_OK;
    }
    GKOC_REQUIRE(x && work_bytes >= gkoc_reduction_workspace_bytes(rows, cols, sizeof(R)), GKOC_E_WORKSPACE,
                 "reduction workspace too small");
    R* partial = static_cast<R*>(work);
    int64_t nb = ceildiv(rows, int64_t(red_block) * 4);
    const int64_t cap = max_partials / (cols < max_partials ? cols : max_partials);
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    cnorm2_stage1<R><<<dim3(unsigned(nb), unsigned(cols)), dim3(red_block), 0, as_stream(s)>>>(
        rows, x, 2 * ld_complex, partial);
    GKOC_LAUNCH_OK();
    reduce_stage2<R, true><<<dim3(unsigned(cols)), dim3(red_block), 0, as_stream(s)>>>(int(nb), partial,
                                                                                       result);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

#define GKOC_DEF_COMPLEX(T, TN, R)                                                                   \
    extern "C" int gkoc_fill_array_##TN(gkoc_stream_t s, T* data, int64_t n, T value)                \
    {                                                                                                \
        if (n <= 0) return GKOC_OK;                                                                  \
        int64_t nb = gkoc::ceildiv(n, 256);                                                          \
        if (nb > gkoc::max_stream_blocks) nb = gkoc::max_stream_blocks;                              \
        gkoc::fill_pair_kernel<T><<<dim3(unsigned(nb)), dim3(256), 0, gkoc::as_stream(s)>>>(n, data, \
                                                                                            value, 0); \
        GKOC_LAUNCH_OK();                                                                            \
        return GKOC_OK;                                                                              \
    }                                                                                                \
    extern "C" int gkoc_dense_fill_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, T* x,           \
                                        int64_t ldx, T value)                                        \
    {                                                                                                \
        if (rows <= 0 || cols <= 0) return GKOC_OK;                                                  \
        int64_t nb = gkoc::ceildiv(rows * cols, 256);                                                \
        if (nb > gkoc::max_stream_blocks) nb = gkoc::max_stream_blocks;                              \
        gkoc::fill_pair_2d_kernel<T><<<dim3(unsigned(nb)), dim3(256), 0, gkoc::as_stream(s)>>>(      \
            rows, cols, x, ldx, value);                                                              \
        GKOC_LAUNCH_OK();                                                                            \
        return GKOC_OK;                                                                              \
    }                                                                                                \
    extern "C" int gkoc_fill_seq_array_##TN(gkoc_stream_t s, T* data, int64_t n)                     \
    {                                                                                                \
        if (n <= 0) return GKOC_OK;                                                                  \
        int64_t nb = gkoc::ceildiv(n, 256);                                                          \
        if (nb > gkoc::max_stream_blocks) nb = gkoc::max_stream_blocks;                              \
        gkoc::fill_pair_kernel<T><<<dim3(unsigned(nb)), dim3(256), 0, gkoc::as_stream(s)>>>(         \
            n, data, T{0, 0}, 1);                                                                    \
        GKOC_LAUNCH_OK();                                                                            \
        return GKOC_OK;                                                                              \
    }                                                                                                \
    extern "C" int gkoc_dense_compute_norm2_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,        \
                                                 const T* x, int64_t ldx, R* result, void* work,     \
                                                 size_t work_bytes)                                  \
    {                                                                                                \
        return gkoc::launch_cnorm2<R>(s, rows, cols, reinterpret_cast<const R*>(x), ldx, result,     \
                                      work, work_bytes);                                             \
    }
GKOC_DEF_COMPLEX(gkoc_c128, c128, double)
GKOC_DEF_COMPLEX(gkoc_c64, c64, float)

extern "C" int gkoc_fill_array_f64(gkoc_stream_t s, double* data, int64_t n,
                                   double value)
{
    return gkoc_dense_fill_f64(s, n, 1, data, 1, value);
}

extern "C" int gkoc_fill_array_f32(gkoc_stream_t s, float* data, int64_t n,
                                   float value)
{
    return gkoc_dense_fill_f32(s, n, 1, data, 1, value);
}

// dense::compute_sqrt for complex values (principal root, complex_type.hpp)
#define GKOC_DEF_CSQRT(T, TN)                                                  \
    extern "C" int gkoc_dense_compute_sqrt_##TN(gkoc_stream_t s, int64_t cols, T* x) \
    {                                                                          \
        if (cols <= 0) return GKOC_OK;                                         \
        gkoc::sqrt_kernel<T><<<dim3(unsigned(gkoc::ceildiv(cols, 256))), dim3(256), 0, gkoc::as_stream(s)>>>(cols, x); \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }
GKOC_DEF_CSQRT(gkoc_c128, c128)
GKOC_DEF_CSQRT(gkoc_c64, c64)
