// Dense BLAS-1 on complex value types (std::complex<double / float> as pairs gkoc_c128 / gkoc_c64):
// the complex instantiations of dense::{scale, inv_scale, add_scaled, sub_scaled} (scalar either
// complex or real: ValueType / ScalarType of core/matrix/dense_kernels.hpp:34-61), compute_dot,
// compute_conj_dot, compute_squared_norm2, compute_mean, make_complex, get_real, get_imag,
// conj_transpose, row_gather, fill_in_matrix_data.  Semantics: reference/matrix/dense_kernels.cpp
// (:184-300 element-wise, :300-440 reductions, :832-841, :916-925, :1208-1250).
// What Ginkgo's distributed solvers and their tests need around REAL systems: a real solver applied
// to complex vectors works on their real views, but residual checks, updates and reductions of the
// complex vectors themselves go through these kernels.  Plain 2-D indexed kernels (strided views,
// per-column scalars), reductions as a fixed two-level tree per column; complex products as
// (ac - bd, ad + bc) with separate multiplies and adds, quotients through the conjugate.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace gkoc {
namespace {

template <typename R>
struct cx {
    R re, im;
};

template <typename R>
__device__ __forceinline__ cx<R> operator+(cx<R> a, cx<R> b) { return {a.re + b.re, a.im + b.im}; }
template <typename R>
__device__ __forceinline__ cx<R> operator-(cx<R> a, cx<R> b) { return {a.re - b.re, a.im - b.im}; }
template <typename R>
__device__ __forceinline__ cx<R> operator*(cx<R> a, cx<R> b)
{
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename R>
__device__ __forceinline__ cx<R> operator*(cx<R> a, R b) { return {a.re * b, a.im * b}; }
template <typename R>
__device__ __forceinline__ cx<R> operator/(cx<R> a, R b) { return {a.re / b, a.im / b}; }
template <typename R>
__device__ __forceinline__ cx<R> operator/(cx<R> a, cx<R> b)
{
    const R d = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
template <typename R>
__device__ __forceinline__ cx<R> conj_of(cx<R> a) { return {a.re, -a.im}; }
template <typename R>
__device__ __forceinline__ bool is_zero(cx<R> a) { return a.re == R(0) && a.im == R(0); }
template <typename R>
__device__ __forceinline__ bool is_zero(R a) { return a == R(0); }

inline unsigned grid_of(int64_t n)
{
    int64_t b = ceildiv(n > 0 ? n : 1, 256);
    if (b > 4 * max_stream_blocks) b = 4 * max_stream_blocks;
    return unsigned(b);
}

#define GKOC_FOR2(idx, total)                                                                          \
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x, idx##_s = int64_t(gridDim.x) * 256;     \
         idx < (total); idx += idx##_s)

// op: 0 scale, 1 inv_scale, 2 add_scaled (y += a x), 3 sub_scaled (y -= a x); S = scalar type
template <typename R, typename S, int OP>
__global__ __launch_bounds__(256) void cx_axpy_kernel(int64_t rows, int64_t cols, const S* __restrict__ alpha,
                                                     int64_t alpha_cols, const cx<R>* __restrict__ x,
                                                     int64_t ldx, cx<R>* __restrict__ y, int64_t ldy)
{
    GKOC_FOR2(i, rows * cols)
    {
        const int64_t r = i / cols, c = i - r * cols;
        const S a = alpha[alpha_cols == 1 ? 0 : c];
        cx<R>& yy = y[r * ldy + c];
        if (OP == 0) {
            // (the reference scales by an exact zero like by any value; NaN * 0 stays NaN)
            yy = yy * a;
        } else if (OP == 1) {
            yy = yy / a;
        } else if (OP == 2) {
            yy = yy + x[r * ldx + c] * a;
        } else {
            yy = yy - x[r * ldx + c] * a;
        }
    }
}

// reductions over the rows of one column: kind 0 x*y, 1 conj(x)*y, 2 |x|^2 (real), 3 x (mean)
template <typename R, int KIND>
__global__ __launch_bounds__(256) void cx_reduce_stage1(int64_t rows, int64_t cols, const cx<R>* __restrict__ x,
                                                       int64_t ldx, const cx<R>* __restrict__ y, int64_t ldy,
                                                       cx<R>* __restrict__ partial)
{
    __shared__ R lds[4];
    const int64_t col = blockIdx.y;
    R are = R(0), aim = R(0);
    for (int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x; r < rows; r += int64_t(gridDim.x) * 256) {
        const cx<R> a = x[r * ldx + col];
        cx<R> t;
        if (KIND == 0) {
            t = a * y[r * ldy + col];
        } else if (KIND == 1) {
            t = conj_of(a) * y[r * ldy + col];
        } else if (KIND == 2) {
            t = {a.re * a.re + a.im * a.im, R(0)};
        } else {
            t = a;
        }
        are += t.re;
        aim += t.im;
    }
    const R sre = block_sum<256>(are, lds);
    __syncthreads();
    const R sim = block_sum<256>(aim, lds);
    if (threadIdx.x == 0) partial[col * gridDim.x + blockIdx.x] = {sre, sim};
}

// mode 0: complex result; 1: real part only (squared norm); 2: complex result / rows (mean)
template <typename R>
__global__ __launch_bounds__(256) void cx_reduce_stage2(int n_partials, const cx<R>* __restrict__ partial,
                                                       void* __restrict__ result, int mode, int64_t rows)
{
    __shared__ R lds[4];
    const int64_t col = blockIdx.x;
    R are = R(0), aim = R(0);
    for (int i = threadIdx.x; i < n_partials; i += 256) {
        are += partial[col * n_partials + i].re;
        aim += partial[col * n_partials + i].im;
    }
    const R sre = block_sum<256>(are, lds);
    __syncthreads();
    const R sim = block_sum<256>(aim, lds);
    if (threadIdx.x == 0) {
        if (mode == 1) {
            static_cast<R*>(result)[col] = sre;
        } else if (mode == 2) {
            static_cast<cx<R>*>(result)[col] = {sre / R(rows), sim / R(rows)};
        } else {
            static_cast<cx<R>*>(result)[col] = {sre, sim};
        }
    }
}

template <typename R, int KIND>
int cx_reduce(gkoc_stream_t s, int64_t rows, int64_t cols, const cx<R>* x, int64_t ldx, const cx<R>* y,
              int64_t ldy, void* result, int mode)
{
    GKOC_REQUIRE(rows >= 0 && cols >= 0, GKOC_E_INVALID, "negative dimension");
    if (cols == 0) return GKOC_OK;
    hipStream_t st = as_stream(s);
    if (rows == 0) {
        GKOC_HIP(hipMemsetAsync(result, 0, size_t(cols) * (mode == 1 ? sizeof(R) : sizeof(cx<R>)), st));
        return GKOC_OK;
    }
    int64_t nb = ceildiv(rows, 2048);
    if (nb > 256) nb = 256;
    cx<R>* partial = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&partial), size_t(nb * cols) * sizeof(cx<R>)));
    cx_reduce_stage1<R, KIND><<<dim3(unsigned(nb), unsigned(cols)), dim3(256), 0, st>>>(rows, cols, x, ldx, y, ldy,
                                                                                        partial);
    cx_reduce_stage2<R><<<dim3(unsigned(cols)), dim3(256), 0, st>>>(int(nb), partial, result, mode, rows);
    hipError_t e = hipGetLastError();
    (void)scratch_free(st, partial);
    GKOC_HIP(e);
    return GKOC_OK;
}

// real columns: mean (the only real reduction that was missing)
template <typename R>
__global__ __launch_bounds__(256) void real_mean_stage1(int64_t rows, const R* __restrict__ x, int64_t ldx,
                                                       R* __restrict__ partial)
{
    __shared__ R lds[4];
    const int64_t col = blockIdx.y;
    R acc = R(0);
    for (int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x; r < rows; r += int64_t(gridDim.x) * 256) {
        acc += x[r * ldx + col];
    }
    const R sum = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) partial[col * gridDim.x + blockIdx.x] = sum;
}

template <typename R>
__global__ __launch_bounds__(256) void real_mean_stage2(int n_partials, const R* __restrict__ partial,
                                                       R* __restrict__ result, int64_t rows)
{
    __shared__ R lds[4];
    R acc = R(0);
    for (int i = threadIdx.x; i < n_partials; i += 256) acc += partial[int64_t(blockIdx.x) * n_partials + i];
    const R sum = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) result[blockIdx.x] = sum / R(rows);
}

// mode 0: make_complex (real in), 1: get_real, 2: get_imag, 3: conj transpose, 4: row gather
template <typename R>
__global__ __launch_bounds__(256) void cx_convert_kernel(int64_t rows, int64_t cols, const void* __restrict__ in,
                                                        int64_t ld_in, void* __restrict__ out, int64_t ld_out,
                                                        int mode)
{
    GKOC_FOR2(i, rows * cols)
    {
        const int64_t r = i / cols, c = i - r * cols;
        if (mode == 0) {
            static_cast<cx<R>*>(out)[r * ld_out + c] = {static_cast<const R*>(in)[r * ld_in + c], R(0)};
        } else if (mode == 1) {
            static_cast<R*>(out)[r * ld_out + c] = static_cast<const cx<R>*>(in)[r * ld_in + c].re;
        } else if (mode == 2) {
            static_cast<R*>(out)[r * ld_out + c] = static_cast<const cx<R>*>(in)[r * ld_in + c].im;
        } else {
            // out is cols x rows
            static_cast<cx<R>*>(out)[c * ld_out + r] = conj_of(static_cast<const cx<R>*>(in)[r * ld_in + c]);
        }
    }
}

template <typename R, typename I>
__global__ __launch_bounds__(256) void cx_row_gather_kernel(int64_t n_gather, int64_t cols, const I* __restrict__ rows,
                                                           const cx<R>* __restrict__ orig, int64_t ld_orig,
                                                           cx<R>* __restrict__ out, int64_t ld_out)
{
    GKOC_FOR2(i, n_gather * cols)
    {
        const int64_t r = i / cols, c = i - r * cols;
        out[r * ld_out + c] = orig[int64_t(rows[r]) * ld_orig + c];
    }
}

template <typename R, typename I>
__global__ __launch_bounds__(256) void cx_fill_in_kernel(int64_t nnz, const I* __restrict__ rows,
                                                        const I* __restrict__ cols, const cx<R>* __restrict__ vals,
                                                        cx<R>* __restrict__ out, int64_t ld)
{
    GKOC_FOR2(i, nnz) out[int64_t(rows[i]) * ld + int64_t(cols[i])] = vals[i];
}

// |x| as hypot (what std::abs of a complex value computes)
template <typename R>
__device__ __forceinline__ R abs_of(cx<R> a) { return hypot(a.re, a.im); }
template <typename R>
__device__ __forceinline__ R abs_of(R a) { return a < R(0) ? -a : a; }

// mode 0: x = |x| + 0i in place; 1: out (reals) = |x|
template <typename R>
__global__ __launch_bounds__(256) void cx_abs_kernel(int64_t rows, int64_t cols, cx<R>* __restrict__ x, int64_t ldx,
                                                    R* __restrict__ out, int64_t ld_out, int mode)
{
    GKOC_FOR2(i, rows * cols)
    {
        const int64_t r = i / cols, c = i - r * cols;
        const R a = abs_of(x[r * ldx + c]);
        if (mode == 0) {
            x[r * ldx + c] = {a, R(0)};
        } else {
            out[r * ld_out + c] = a;
        }
    }
}

template <typename R>
__global__ __launch_bounds__(256) void cx_norm1_stage1(int64_t rows, const cx<R>* __restrict__ x, int64_t ldx,
                                                      R* __restrict__ partial)
{
    __shared__ R lds[4];
    const int64_t col = blockIdx.y;
    R acc = R(0);
    for (int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x; r < rows; r += int64_t(gridDim.x) * 256) {
        acc += abs_of(x[r * ldx + col]);
    }
    const R sum = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) partial[col * gridDim.x + blockIdx.x] = sum;
}

// CSR with complex values, one thread per (row, right-hand side): y = A x, or y = alpha A x + beta y
template <typename R, typename I, bool ADVANCED>
__global__ __launch_bounds__(256) void cx_csr_spmv_kernel(int64_t rows, int64_t nrhs, const I* __restrict__ row_ptrs,
                                                         const I* __restrict__ col_idxs,
                                                         const cx<R>* __restrict__ vals, const cx<R>* __restrict__ alpha,
                                                         const cx<R>* __restrict__ x, int64_t ldx,
                                                         const cx<R>* __restrict__ beta, cx<R>* __restrict__ y,
                                                         int64_t ldy)
{
    GKOC_FOR2(i, rows * nrhs)
    {
        const int64_t r = i / nrhs, c = i - r * nrhs;
        cx<R> acc{R(0), R(0)};
        for (I k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) acc = acc + vals[k] * x[int64_t(col_idxs[k]) * ldx + c];
        if (ADVANCED) {
            const cx<R> b = beta[0];
            const cx<R> ax = alpha[0] * acc;
            y[r * ldy + c] = is_zero(b) ? ax : ax + b * y[r * ldy + c];
        } else {
            y[r * ldy + c] = acc;
        }
    }
}

template <typename V>
struct real_of {
    using type = V;
};
template <typename R>
struct real_of<cx<R>> {
    using type = R;
};

// mode 0: diag[row] = a(row,row) where stored; 1: sum[row] = sum_k |a_k| (+ 0i)
template <typename V, typename I>
__global__ __launch_bounds__(256) void csr_row_scan_kernel(int64_t rows, const I* __restrict__ row_ptrs,
                                                          const I* __restrict__ col_idxs,
                                                          const V* __restrict__ vals, V* __restrict__ out, int mode)
{
    GKOC_FOR2(r, rows)
    {
        if (mode == 0) {
            for (I k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
                if (int64_t(col_idxs[k]) == r) {
                    out[r] = vals[k];
                    break;
                }
            }
        } else {
            typename real_of<V>::type acc = 0;
            for (I k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) acc += abs_of(vals[k]);
            out[r] = V{acc};
        }
    }
}

// scalar Jacobi and Diagonal products on complex values (jacobi::{invert_diagonal,
// simple_scalar_apply, scalar_apply}, diagonal::{apply_to_csr, right_apply_to_csr};
// reference/preconditioner/jacobi_kernels.cpp:535-592, reference/matrix/diagonal_kernels.cpp:60-102)
template <typename R>
__global__ __launch_bounds__(256) void cx_invert_kernel(int64_t n, const cx<R>* __restrict__ d, cx<R>* __restrict__ inv)
{
    GKOC_FOR2(i, n)
    {
        const cx<R> v = is_zero(d[i]) ? cx<R>{R(1), R(0)} : d[i];
        inv[i] = cx<R>{R(1), R(0)} / v;
    }
}

// x(i,j) = b(i,j) d[i], or beta x(i,j) + (alpha b(i,j)) d[i]
template <typename R, bool ADVANCED>
__global__ __launch_bounds__(256) void cx_row_scale_kernel(int64_t rows, int64_t cols, const cx<R>* __restrict__ d,
                                                          const cx<R>* __restrict__ alpha,
                                                          const cx<R>* __restrict__ b, int64_t ldb,
                                                          const cx<R>* __restrict__ beta, cx<R>* __restrict__ x,
                                                          int64_t ldx)
{
    GKOC_FOR2(i, rows * cols)
    {
        const int64_t r = i / cols, c = i - r * cols;
        if (ADVANCED) {
            x[r * ldx + c] = beta[0] * x[r * ldx + c] + (alpha[0] * b[r * ldb + c]) * d[r];
        } else {
            x[r * ldx + c] = b[r * ldb + c] * d[r];
        }
    }
}

// mode 0: vals[k] *= d[row]; 1: vals[k] *= 1 / d[row]; 2: vals[k] *= d[col[k]]
template <typename R, typename I>
__global__ __launch_bounds__(256) void cx_csr_scale_kernel(int64_t n_rows, const I* __restrict__ row_ptrs,
                                                          const I* __restrict__ cols, const cx<R>* __restrict__ d,
                                                          int mode, cx<R>* __restrict__ vals)
{
    GKOC_FOR2(r, n_rows)
    {
        const cx<R> s = mode == 1 ? cx<R>{R(1), R(0)} / d[r] : d[r];
        for (I k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
            vals[k] = vals[k] * (mode == 2 ? d[cols[k]] : s);
        }
    }
}

// Dense -> Csr with complex values (dense::count_nonzeros_per_row / convert_to_csr): one lane per
// row; the entries of a row keep their column order
template <typename R>
__global__ __launch_bounds__(256) void cx_dense_count_kernel(int64_t rows, int64_t cols, const cx<R>* __restrict__ in,
                                                            int64_t ld, void* __restrict__ out, int out_bytes)
{
    GKOC_FOR2(r, rows)
    {
        int64_t n = 0;
        for (int64_t c = 0; c < cols; ++c) n += !is_zero(in[r * ld + c]);
        if (out_bytes == 4) {
            static_cast<int32_t*>(out)[r] = int32_t(n);
        } else {
            static_cast<int64_t*>(out)[r] = n;
        }
    }
}

template <typename R, typename I>
__global__ __launch_bounds__(256) void cx_dense_to_csr_kernel(int64_t rows, int64_t cols, const cx<R>* __restrict__ in,
                                                             int64_t ld, const I* __restrict__ row_ptrs,
                                                             I* __restrict__ out_cols, cx<R>* __restrict__ out_vals)
{
    GKOC_FOR2(r, rows)
    {
        int64_t k = row_ptrs[r];
        for (int64_t c = 0; c < cols; ++c) {
            const cx<R> v = in[r * ld + c];
            if (!is_zero(v)) {
                out_cols[k] = I(c);
                out_vals[k] = v;
                ++k;
            }
        }
    }
}

// Coo with complex values: c(row, j) += [alpha] val b(col, j), entry by entry with atomic adds of the
// real and imaginary parts (the only kernels of this library whose summation order is not fixed:
// the complex instantiations exist so that Ginkgo's distributed Matrix runs with a Coo non-local
// part on complex data, they are not a tuned path)
template <typename R, typename I>
__global__ __launch_bounds__(256) void cx_coo_spmv2_kernel(int64_t nnz, int64_t nrhs, const I* __restrict__ rows,
                                                          const I* __restrict__ cols,
                                                          const cx<R>* __restrict__ vals,
                                                          const cx<R>* __restrict__ alpha,
                                                          const cx<R>* __restrict__ b, int64_t ldb,
                                                          cx<R>* __restrict__ c, int64_t ldc)
{
    GKOC_FOR2(i, nnz * nrhs)
    {
        const int64_t k = i / nrhs, j = i - k * nrhs;
        cx<R> t = vals[k] * b[int64_t(cols[k]) * ldb + j];
        if (alpha) t = alpha[0] * t;
        cx<R>* dst = c + int64_t(rows[k]) * ldc + j;
        atomicAdd(&dst->re, t.re);
        atomicAdd(&dst->im, t.im);
    }
}

template <typename R, typename S, int OP>
int cx_axpy(gkoc_stream_t s, int64_t rows, int64_t cols, const void* alpha, int64_t alpha_cols, const void* x,
            int64_t ldx, void* y, int64_t ldy)
{
    if (rows <= 0 || cols <= 0) return GKOC_OK;
    GKOC_REQUIRE(alpha && (alpha_cols == 1 || alpha_cols == cols), GKOC_E_INVALID, "bad alpha");
    cx_axpy_kernel<R, S, OP><<<dim3(grid_of(rows * cols)), dim3(256), 0, as_stream(s)>>>(
        rows, cols, static_cast<const S*>(alpha), alpha_cols, static_cast<const cx<R>*>(x), ldx,
        static_cast<cx<R>*>(y), ldy);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

// scalar_is_real: alpha is a matrix of R (remove_complex<ValueType>) instead of complex values
#define GKOC_DEF_CBLAS(P, TN, R)                                                                            \
    extern "C" int gkoc_cdense_scale_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const void* alpha,   \
                                          int64_t alpha_cols, int scalar_is_real, P* x, int64_t ldx)        \
    {                                                                                                       \
        return scalar_is_real ? cx_axpy<R, R, 0>(s, rows, cols, alpha, alpha_cols, nullptr, 0, x, ldx)      \
                              : cx_axpy<R, cx<R>, 0>(s, rows, cols, alpha, alpha_cols, nullptr, 0, x, ldx); \
    }                                                                                                       \
    extern "C" int gkoc_cdense_inv_scale_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,                  \
                                              const void* alpha, int64_t alpha_cols, int scalar_is_real,    \
                                              P* x, int64_t ldx)                                            \
    {                                                                                                       \
        return scalar_is_real ? cx_axpy<R, R, 1>(s, rows, cols, alpha, alpha_cols, nullptr, 0, x, ldx)      \
                              : cx_axpy<R, cx<R>, 1>(s, rows, cols, alpha, alpha_cols, nullptr, 0, x, ldx); \
    }                                                                                                       \
    extern "C" int gkoc_cdense_add_scaled_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,                 \
                                               const void* alpha, int64_t alpha_cols, int scalar_is_real,   \
                                               const P* x, int64_t ldx, P* y, int64_t ldy)                  \
    {                                                                                                       \
        return scalar_is_real ? cx_axpy<R, R, 2>(s, rows, cols, alpha, alpha_cols, x, ldx, y, ldy)          \
                              : cx_axpy<R, cx<R>, 2>(s, rows, cols, alpha, alpha_cols, x, ldx, y, ldy);     \
    }                                                                                                       \
    extern "C" int gkoc_cdense_sub_scaled_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,                 \
                                               const void* alpha, int64_t alpha_cols, int scalar_is_real,   \
                                               const P* x, int64_t ldx, P* y, int64_t ldy)                  \
    {                                                                                                       \
        return scalar_is_real ? cx_axpy<R, R, 3>(s, rows, cols, alpha, alpha_cols, x, ldx, y, ldy)          \
                              : cx_axpy<R, cx<R>, 3>(s, rows, cols, alpha, alpha_cols, x, ldx, y, ldy);     \
    }                                                                                                       \
    extern "C" int gkoc_cdense_compute_dot_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* x,    \
                                                int64_t ldx, const P* y, int64_t ldy, P* result,            \
                                                int conjugate_x)                                            \
    {                                                                                                       \
        return conjugate_x                                                                                  \
                   ? cx_reduce<R, 1>(s, rows, cols, reinterpret_cast<const cx<R>*>(x), ldx,                 \
                                     reinterpret_cast<const cx<R>*>(y), ldy, result, 0)                     \
                   : cx_reduce<R, 0>(s, rows, cols, reinterpret_cast<const cx<R>*>(x), ldx,                 \
                                     reinterpret_cast<const cx<R>*>(y), ldy, result, 0);                    \
    }                                                                                                       \
    extern "C" int gkoc_cdense_compute_squared_norm2_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,      \
                                                          const P* x, int64_t ldx, R* result)               \
    {                                                                                                       \
        return cx_reduce<R, 2>(s, rows, cols, reinterpret_cast<const cx<R>*>(x), ldx, nullptr, 0, result,   \
                               1);                                                                          \
    }                                                                                                       \
    extern "C" int gkoc_cdense_compute_sum_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* x,    \
                                                int64_t ldx, P* result)                                     \
    {                                                                                                       \
        return cx_reduce<R, 3>(s, rows, cols, reinterpret_cast<const cx<R>*>(x), ldx, nullptr, 0, result,   \
                               0);                                                                          \
    }                                                                                                       \
    extern "C" int gkoc_cdense_compute_mean_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* x,   \
                                                 int64_t ldx, P* result)                                    \
    {                                                                                                       \
        return cx_reduce<R, 3>(s, rows, cols, reinterpret_cast<const cx<R>*>(x), ldx, nullptr, 0, result,   \
                               2);                                                                          \
    }                                                                                                       \
    /* mode 0 make_complex (in: R), 1 get_real, 2 get_imag (out: R), 3 conj_transpose (out cols x rows) */ \
    extern "C" int gkoc_cdense_convert_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const void* in,    \
                                            int64_t ld_in, void* out, int64_t ld_out, int mode)             \
    {                                                                                                       \
        GKOC_REQUIRE(mode >= 0 && mode <= 3, GKOC_E_INVALID, "mode");                                       \
        if (rows <= 0 || cols <= 0) return GKOC_OK;                                                         \
        cx_convert_kernel<R><<<dim3(grid_of(rows * cols)), dim3(256), 0, as_stream(s)>>>(rows, cols, in,    \
                                                                                        ld_in, out, ld_out, \
                                                                                        mode);              \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_CBLAS(gkoc_c128, c128, double)
GKOC_DEF_CBLAS(gkoc_c64, c64, float)

#define GKOC_DEF_CBLAS_I(P, TN, R, I, IN)                                                                   \
    extern "C" int gkoc_cdense_row_gather_##TN##_##IN(gkoc_stream_t s, int64_t n_gather, int64_t cols,      \
                                                      const I* rows, const P* orig, int64_t ld_orig,        \
                                                      P* gathered, int64_t ld_gathered)                     \
    {                                                                                                       \
        if (n_gather <= 0 || cols <= 0) return GKOC_OK;                                                     \
        cx_row_gather_kernel<R, I><<<dim3(grid_of(n_gather * cols)), dim3(256), 0, as_stream(s)>>>(         \
            n_gather, cols, rows, reinterpret_cast<const cx<R>*>(orig), ld_orig,                            \
            reinterpret_cast<cx<R>*>(gathered), ld_gathered);                                               \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    extern "C" int gkoc_cdense_fill_in_matrix_data_##TN##_##IN(gkoc_stream_t s, int64_t nnz, const I* rows, \
                                                               const I* cols, const P* vals, P* out,        \
                                                               int64_t ld)                                  \
    {                                                                                                       \
        if (nnz <= 0) return GKOC_OK;                                                                       \
        cx_fill_in_kernel<R, I><<<dim3(grid_of(nnz)), dim3(256), 0, as_stream(s)>>>(                        \
            nnz, rows, cols, reinterpret_cast<const cx<R>*>(vals), reinterpret_cast<cx<R>*>(out), ld);      \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_CBLAS_I(gkoc_c128, c128, double, int32_t, i32)
GKOC_DEF_CBLAS_I(gkoc_c128, c128, double, int64_t, i64)
GKOC_DEF_CBLAS_I(gkoc_c64, c64, float, int32_t, i32)
GKOC_DEF_CBLAS_I(gkoc_c64, c64, float, int64_t, i64)

#define GKOC_DEF_MEAN(R, RN)                                                                                \
    extern "C" int gkoc_dense_compute_mean_##RN(gkoc_stream_t s, int64_t rows, int64_t cols, const R* x,    \
                                                int64_t ldx, R* result)                                     \
    {                                                                                                       \
        GKOC_REQUIRE(rows >= 0 && cols >= 0, GKOC_E_INVALID, "negative dimension");                         \
        if (cols == 0) return GKOC_OK;                                                                      \
        hipStream_t st = as_stream(s);                                                                      \
        if (rows == 0) {                                                                                    \
            GKOC_HIP(hipMemsetAsync(result, 0, size_t(cols) * sizeof(R), st));                              \
            return GKOC_OK;                                                                                 \
        }                                                                                                   \
        int64_t nb = ceildiv(rows, 2048);                                                                   \
        if (nb > 256) nb = 256;                                                                             \
        R* partial = nullptr;                                                                               \
        GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&partial), size_t(nb * cols) * sizeof(R)));    \
        real_mean_stage1<R><<<dim3(unsigned(nb), unsigned(cols)), dim3(256), 0, st>>>(rows, x, ldx,         \
                                                                                      partial);            \
        real_mean_stage2<R><<<dim3(unsigned(cols)), dim3(256), 0, st>>>(int(nb), partial, result, rows);    \
        hipError_t e = hipGetLastError();                                                                   \
        (void)scratch_free(st, partial);                                                                    \
        GKOC_HIP(e);                                                                                        \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_MEAN(double, f64)
GKOC_DEF_MEAN(float, f32)

#define GKOC_DEF_CABS(P, TN, R)                                                                             \
    /* mode 0: x = |x| in place (imaginary parts 0); 1: out (reals, ld_out) = |x| */                        \
    extern "C" int gkoc_cdense_absolute_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, P* x,             \
                                             int64_t ldx, R* out, int64_t ld_out, int mode)                 \
    {                                                                                                       \
        GKOC_REQUIRE(mode == 0 || mode == 1, GKOC_E_INVALID, "mode");                                       \
        if (rows <= 0 || cols <= 0) return GKOC_OK;                                                         \
        cx_abs_kernel<R><<<dim3(grid_of(rows * cols)), dim3(256), 0, as_stream(s)>>>(                       \
            rows, cols, reinterpret_cast<cx<R>*>(x), ldx, out, ld_out, mode);                               \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    extern "C" int gkoc_cdense_compute_norm1_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* x,  \
                                                  int64_t ldx, R* result)                                   \
    {                                                                                                       \
        GKOC_REQUIRE(rows >= 0 && cols >= 0, GKOC_E_INVALID, "negative dimension");                         \
        if (cols == 0) return GKOC_OK;                                                                      \
        hipStream_t st = as_stream(s);                                                                      \
        if (rows == 0) {                                                                                    \
            GKOC_HIP(hipMemsetAsync(result, 0, size_t(cols) * sizeof(R), st));                              \
            return GKOC_OK;                                                                                 \
        }                                                                                                   \
        int64_t nb = ceildiv(rows, 2048);                                                                   \
        if (nb > 256) nb = 256;                                                                             \
        R* partial = nullptr;                                                                               \
        GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&partial), size_t(nb * cols) * sizeof(R)));    \
        cx_norm1_stage1<R><<<dim3(unsigned(nb), unsigned(cols)), dim3(256), 0, st>>>(                       \
            rows, reinterpret_cast<const cx<R>*>(x), ldx, partial);                                         \
        /* the second stage of the mean with a divisor of 1 */                                              \
        real_mean_stage2<R><<<dim3(unsigned(cols)), dim3(256), 0, st>>>(int(nb), partial, result, 1);       \
        hipError_t e = hipGetLastError();                                                                   \
        (void)scratch_free(st, partial);                                                                    \
        GKOC_HIP(e);                                                                                        \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_CABS(gkoc_c128, c128, double)
GKOC_DEF_CABS(gkoc_c64, c64, float)

#define GKOC_DEF_CCSR(P, TN, R, I, IN)                                                                      \
    /* alpha == NULL: y = A x; else y = alpha[0] A x + beta[0] y (scalars on the device) */                 \
    extern "C" int gkoc_ccsr_spmv_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t nrhs,                  \
                                              const I* row_ptrs, const I* col_idxs, const P* vals,          \
                                              const P* alpha, const P* x, int64_t ldx, const P* beta, P* y, \
                                              int64_t ldy)                                                  \
    {                                                                                                       \
        if (rows <= 0 || nrhs <= 0) return GKOC_OK;                                                         \
        GKOC_REQUIRE((alpha == nullptr) == (beta == nullptr), GKOC_E_INVALID, "alpha and beta go together");\
        /* the row-segment kernel of the real types (csr_spmv.hip) unless the arrays are not aligned for   \
           it or GKOC_TUNE_CCSR_THREAD_PER_ROW asks for round 5's kernel (A/B runs) */                      \
        if (tune_value(GKOC_TUNE_CCSR_THREAD_PER_ROW) == 0) {                                               \
            const int rc_ = csr_spmv_complex<P, I>(s, rows, nrhs, row_ptrs, col_idxs, vals, alpha, x, ldx,  \
                                                   beta, y, ldy);                                           \
            if (rc_ != GKOC_E_NOT_SUPPORTED) return rc_;                                                    \
        }                                                                                                   \
        const dim3 g(grid_of(rows * nrhs));                                                                 \
        if (alpha) {                                                                                        \
            cx_csr_spmv_kernel<R, I, true><<<g, dim3(256), 0, as_stream(s)>>>(                              \
                rows, nrhs, row_ptrs, col_idxs, reinterpret_cast<const cx<R>*>(vals),                       \
                reinterpret_cast<const cx<R>*>(alpha), reinterpret_cast<const cx<R>*>(x), ldx,              \
                reinterpret_cast<const cx<R>*>(beta), reinterpret_cast<cx<R>*>(y), ldy);                    \
        } else {                                                                                            \
            cx_csr_spmv_kernel<R, I, false><<<g, dim3(256), 0, as_stream(s)>>>(                             \
                rows, nrhs, row_ptrs, col_idxs, reinterpret_cast<const cx<R>*>(vals), nullptr,              \
                reinterpret_cast<const cx<R>*>(x), ldx, nullptr, reinterpret_cast<cx<R>*>(y), ldy);         \
        }                                                                                                   \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    /* mode 0: diag[row] = a(row, row) where it is stored (rows of the diagonal); 1: sums[row] = sum |a| */ \
    extern "C" int gkoc_ccsr_row_scan_##TN##_##IN(gkoc_stream_t s, int64_t rows, const I* row_ptrs,         \
                                                  const I* col_idxs, const P* vals, P* out, int mode)       \
    {                                                                                                       \
        GKOC_REQUIRE(mode == 0 || mode == 1, GKOC_E_INVALID, "mode");                                       \
        if (rows <= 0) return GKOC_OK;                                                                      \
        csr_row_scan_kernel<cx<R>, I><<<dim3(grid_of(rows)), dim3(256), 0, as_stream(s)>>>(                 \
            rows, row_ptrs, col_idxs, reinterpret_cast<const cx<R>*>(vals), reinterpret_cast<cx<R>*>(out),  \
            mode);                                                                                          \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_CCSR(gkoc_c128, c128, double, int32_t, i32)
GKOC_DEF_CCSR(gkoc_c128, c128, double, int64_t, i64)
GKOC_DEF_CCSR(gkoc_c64, c64, float, int32_t, i32)
GKOC_DEF_CCSR(gkoc_c64, c64, float, int64_t, i64)

// csr::row_wise_absolute_sum for real values (core/matrix/csr_kernels.hpp:287-290): the L1 smoother
// of the Schwarz preconditioner
#define GKOC_DEF_RWAS(T, TN, I, IN)                                                                         \
    extern "C" int gkoc_csr_row_wise_absolute_sum_##TN##_##IN(gkoc_stream_t s, int64_t rows,                \
                                                              const I* row_ptrs, const T* vals, T* sums)    \
    {                                                                                                       \
        if (rows <= 0) return GKOC_OK;                                                                      \
        csr_row_scan_kernel<T, I><<<dim3(grid_of(rows)), dim3(256), 0, as_stream(s)>>>(rows, row_ptrs,      \
                                                                                      nullptr, vals, sums,  \
                                                                                      1);                   \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_RWAS(double, f64, int32_t, i32)
GKOC_DEF_RWAS(double, f64, int64_t, i64)
GKOC_DEF_RWAS(float, f32, int32_t, i32)
GKOC_DEF_RWAS(float, f32, int64_t, i64)

#define GKOC_DEF_CJAC(P, TN, R)                                                                             \
    extern "C" int gkoc_cjacobi_invert_diagonal_##TN(gkoc_stream_t s, int64_t n, const P* diag, P* inv)     \
    {                                                                                                       \
        if (n <= 0) return GKOC_OK;                                                                         \
        cx_invert_kernel<R><<<dim3(grid_of(n)), dim3(256), 0, as_stream(s)>>>(                              \
            n, reinterpret_cast<const cx<R>*>(diag), reinterpret_cast<cx<R>*>(inv));                        \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    /* alpha == NULL: x = diag b (row-wise); else x = beta x + alpha b diag */                              \
    extern "C" int gkoc_cjacobi_scalar_apply_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,              \
                                                  const P* diag, const P* alpha, const P* b, int64_t ldb,   \
                                                  const P* beta, P* x, int64_t ldx)                         \
    {                                                                                                       \
        if (rows <= 0 || cols <= 0) return GKOC_OK;                                                         \
        GKOC_REQUIRE((alpha == nullptr) == (beta == nullptr), GKOC_E_INVALID, "alpha and beta go together");\
        const dim3 g(grid_of(rows * cols));                                                                 \
        if (alpha) {                                                                                        \
            cx_row_scale_kernel<R, true><<<g, dim3(256), 0, as_stream(s)>>>(                                \
                rows, cols, reinterpret_cast<const cx<R>*>(diag), reinterpret_cast<const cx<R>*>(alpha),    \
                reinterpret_cast<const cx<R>*>(b), ldb, reinterpret_cast<const cx<R>*>(beta),               \
                reinterpret_cast<cx<R>*>(x), ldx);                                                          \
        } else {                                                                                            \
            cx_row_scale_kernel<R, false><<<g, dim3(256), 0, as_stream(s)>>>(                               \
                rows, cols, reinterpret_cast<const cx<R>*>(diag), nullptr,                                  \
                reinterpret_cast<const cx<R>*>(b), ldb, nullptr, reinterpret_cast<cx<R>*>(x), ldx);         \
        }                                                                                                   \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_CJAC(gkoc_c128, c128, double)
GKOC_DEF_CJAC(gkoc_c64, c64, float)

#define GKOC_DEF_CCSR_SCALE(P, TN, R, I, IN)                                                                \
    /* mode 0: vals *= diag[row]; 1: vals *= 1 / diag[row]; 2: vals *= diag[col] */                         \
    extern "C" int gkoc_ccsr_scale_by_diagonal_##TN##_##IN(gkoc_stream_t s, int64_t n_rows,                 \
                                                           const I* row_ptrs, const I* col_idxs,            \
                                                           const P* diag, int mode, P* vals)                \
    {                                                                                                       \
        GKOC_REQUIRE(mode >= 0 && mode <= 2, GKOC_E_INVALID, "mode");                                       \
        if (n_rows <= 0) return GKOC_OK;                                                                    \
        cx_csr_scale_kernel<R, I><<<dim3(grid_of(n_rows)), dim3(256), 0, as_stream(s)>>>(                   \
            n_rows, row_ptrs, col_idxs, reinterpret_cast<const cx<R>*>(diag), mode,                         \
            reinterpret_cast<cx<R>*>(vals));                                                                \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_CCSR_SCALE(gkoc_c128, c128, double, int32_t, i32)
GKOC_DEF_CCSR_SCALE(gkoc_c128, c128, double, int64_t, i64)
GKOC_DEF_CCSR_SCALE(gkoc_c64, c64, float, int32_t, i32)
GKOC_DEF_CCSR_SCALE(gkoc_c64, c64, float, int64_t, i64)

#define GKOC_DEF_CDENSE_CSR(P, TN, R)                                                                       \
    /* out: int32 / int64 / uint64 counts (out_bytes 4 or 8) */                                             \
    extern "C" int gkoc_cdense_count_nonzeros_per_row_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,     \
                                                           const P* in, int64_t ld, void* out,              \
                                                           int out_bytes)                                   \
    {                                                                                                       \
        GKOC_REQUIRE(out_bytes == 4 || out_bytes == 8, GKOC_E_INVALID, "out_bytes");                        \
        if (rows <= 0) return GKOC_OK;                                                                      \
        cx_dense_count_kernel<R><<<dim3(grid_of(rows)), dim3(256), 0, as_stream(s)>>>(                      \
            rows, cols, reinterpret_cast<const cx<R>*>(in), ld, out, out_bytes);                            \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_CDENSE_CSR(gkoc_c128, c128, double)
GKOC_DEF_CDENSE_CSR(gkoc_c64, c64, float)
#define GKOC_DEF_CDENSE_CSR_I(P, TN, R, I, IN)                                                              \
    extern "C" int gkoc_cdense_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols, const P* in, \
                                                  int64_t ld, const I* row_ptrs, I* out_cols, P* out_vals)  \
    {                                                                                                       \
        if (rows <= 0) return GKOC_OK;                                                                      \
        cx_dense_to_csr_kernel<R, I><<<dim3(grid_of(rows)), dim3(256), 0, as_stream(s)>>>(                  \
            rows, cols, reinterpret_cast<const cx<R>*>(in), ld, row_ptrs, out_cols,                         \
            reinterpret_cast<cx<R>*>(out_vals));                                                            \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_CDENSE_CSR_I(gkoc_c128, c128, double, int32_t, i32)
GKOC_DEF_CDENSE_CSR_I(gkoc_c128, c128, double, int64_t, i64)
GKOC_DEF_CDENSE_CSR_I(gkoc_c64, c64, float, int32_t, i32)
GKOC_DEF_CDENSE_CSR_I(gkoc_c64, c64, float, int64_t, i64)

#define GKOC_DEF_CCOO(P, TN, R, I, IN)                                                                      \
    /* c += [alpha] A b (alpha == NULL: 1); for c = A b clear c first, for c = alpha A b + beta c scale it */ \
    extern "C" int gkoc_ccoo_spmv2_##TN##_##IN(gkoc_stream_t s, int64_t nnz, int64_t nrhs, const I* rows,   \
                                               const I* cols, const P* vals, const P* alpha, const P* b,    \
                                               int64_t ldb, P* c, int64_t ldc)                              \
    {                                                                                                       \
        if (nnz <= 0 || nrhs <= 0) return GKOC_OK;                                                          \
        cx_coo_spmv2_kernel<R, I><<<dim3(grid_of(nnz * nrhs)), dim3(256), 0, as_stream(s)>>>(               \
            nnz, nrhs, rows, cols, reinterpret_cast<const cx<R>*>(vals),                                    \
            reinterpret_cast<const cx<R>*>(alpha), reinterpret_cast<const cx<R>*>(b), ldb,                  \
            reinterpret_cast<cx<R>*>(c), ldc);                                                              \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_CCOO(gkoc_c128, c128, double, int32_t, i32)
GKOC_DEF_CCOO(gkoc_c128, c128, double, int64_t, i64)
GKOC_DEF_CCOO(gkoc_c64, c64, float, int32_t, i32)
GKOC_DEF_CCOO(gkoc_c64, c64, float, int64_t, i64)
