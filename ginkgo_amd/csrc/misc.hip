// Small kernels around the hot path that Ginkgo's generic LinOp / solver test-suites reach
// (test/matrix/{diagonal,sparsity_csr}_kernels.cpp, test/components/{reduce_array,
// precision_conversion}_kernels.cpp, test/matrix/matrix.cpp, test/solver/solver.cpp):
//   matrix::Diagonal   reference/matrix/diagonal_kernels.cpp:20-170
//   SparsityCsr        reference/matrix/sparsity_csr_kernels.cpp (diagonal_element_prefix_sum,
//                      remove_diagonal_elements)
//   components         reference/components/{reduce_array,precision_conversion}_kernels.cpp
// Element-wise results carry the reference's expression (one rounding per operation,
// -ffp-contract=off); reduce_add_array is a fixed two-level tree (exact for the integer types).
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace gkoc {
namespace {

inline unsigned grid_of(int64_t n)
{
    int64_t b = ceildiv(n > 0 ? n : 1, 256);
    if (b > 4 * max_stream_blocks) b = 4 * max_stream_blocks;
    return unsigned(b);
}

#define GKOC_FOR(i, n)                                                                               \
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x, i##_stride = int64_t(gridDim.x) * 256; \
         i < (n); i += i##_stride)

// ---- Diagonal
template <typename T>
__global__ __launch_bounds__(256) void diag_dense_kernel(int64_t rows, int64_t cols, const T* __restrict__ diag,
                                                        const T* __restrict__ b, int64_t ldb, T* __restrict__ c,
                                                        int64_t ldc, int mode)
{
    // mode 0: c = b * diag[row]; 1: c = b * (1 / diag[row]); 2: c = b * diag[col]
    GKOC_FOR(i, rows * cols)
    {
        const int64_t r = i / cols, j = i % cols;
        const T scal = mode == 2 ? diag[j] : (mode == 1 ? T(1) / diag[r] : diag[r]);
        c[r * ldc + j] = b[r * ldb + j] * scal;
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void diag_csr_rows_kernel(int64_t n_rows, const T* __restrict__ diag,
                                                           const I* __restrict__ row_ptrs, T* __restrict__ vals,
                                                           int inverse)
{
    // one wave per row
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t(blockIdx.x) * 256 + threadIdx.x) >> 6;
    const int64_t n_waves = (int64_t(gridDim.x) * 256) >> 6;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const T scal = inverse ? T(1) / diag[r] : diag[r];
        for (int64_t k = int64_t(row_ptrs[r]) + lane; k < int64_t(row_ptrs[r + 1]); k += 64) vals[k] *= scal;
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void diag_csr_cols_kernel(int64_t nnz, const T* __restrict__ diag,
                                                           const I* __restrict__ cols, T* __restrict__ vals)
{
    GKOC_FOR(k, nnz) vals[k] *= diag[cols[k]];
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void diag_to_csr_kernel(int64_t n, const T* __restrict__ diag,
                                                         I* __restrict__ row_ptrs, I* __restrict__ cols,
                                                         T* __restrict__ vals)
{
    GKOC_FOR(i, n + 1)
    {
        row_ptrs[i] = I(i);
        if (i < n) {
            cols[i] = I(i);
            vals[i] = diag[i];
        }
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void diag_fill_kernel(int64_t nnz, const I* __restrict__ rows,
                                                       const I* __restrict__ cols, const T* __restrict__ vals,
                                                       T* __restrict__ diag)
{
    GKOC_FOR(i, nnz)
    if (rows[i] == cols[i]) diag[rows[i]] = vals[i];
}

// ---- SparsityCsr
template <typename I>
__global__ __launch_bounds__(256) void count_diag_kernel(int64_t n_rows, const I* __restrict__ row_ptrs,
                                                        const I* __restrict__ cols, I* __restrict__ out)
{
    GKOC_FOR(r, n_rows + 1)
    {
        I c = 0;
        if (r < n_rows) {
            for (int64_t k = row_ptrs[r]; k < int64_t(row_ptrs[r + 1]); ++k) c += cols[k] == I(r) ? 1 : 0;
        }
        out[r] = c;
    }
}

template <typename I>
__global__ __launch_bounds__(256) void remove_diag_kernel(int64_t n_rows, const I* __restrict__ row_ptrs,
                                                         const I* __restrict__ cols, const I* __restrict__ prefix,
                                                         I* __restrict__ adj_ptrs, I* __restrict__ adj_idxs)
{
    GKOC_FOR(r, n_rows + 1)
    {
        adj_ptrs[r] = row_ptrs[r] - prefix[r];
        if (r < n_rows) {
            int64_t o = int64_t(row_ptrs[r]) - int64_t(prefix[r]);
            for (int64_t k = row_ptrs[r]; k < int64_t(row_ptrs[r + 1]); ++k) {
                if (cols[k] != I(r)) adj_idxs[o++] = cols[k];
            }
        }
    }
}

// ---- components
template <typename S, typename T>
__global__ __launch_bounds__(256) void convert_kernel(int64_t n, const S* __restrict__ in, T* __restrict__ out)
{
    GKOC_FOR(i, n) out[i] = T(in[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void reduce_add_stage1(int64_t n, const T* __restrict__ arr, T* __restrict__ partial)
{
    __shared__ T lds[4];
    T acc = T(0);
    GKOC_FOR(i, n) acc += arr[i];
    const T r = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

template <typename T>
__global__ __launch_bounds__(256) void reduce_add_stage2(int n_partials, const T* __restrict__ partial,
                                                        T* __restrict__ val)
{
    __shared__ T lds[4];
    T acc = T(0);
    for (int i = threadIdx.x; i < n_partials; i += 256) acc += partial[i];
    const T r = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) val[0] += r;
}

template <typename T>
int reduce_add(gkoc_stream_t s, int64_t n, const T* arr, T* val)
{
    GKOC_REQUIRE(val && n >= 0, GKOC_E_INVALID, "bad argument");
    if (n == 0) return GKOC_OK;
    hipStream_t st = as_stream(s);
    int64_t nb = ceildiv(n, 2048);
    if (nb > 1024) nb = 1024;
    T* partial = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&partial), size_t(nb) * sizeof(T)));
    reduce_add_stage1<T><<<dim3(unsigned(nb)), dim3(256), 0, st>>>(n, arr, partial);
    reduce_add_stage2<T><<<dim3(1), dim3(256), 0, st>>>(int(nb), partial, val);
    hipError_t e = hipGetLastError();
    (void)scratch_free(st, partial);
    GKOC_HIP(e);
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_MISC_T(T, TN)                                                                              \
    extern "C" int gkoc_diagonal_apply_to_dense_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,           \
                                                     const T* diag, const T* b, int64_t ldb, T* c,          \
                                                     int64_t ldc, int inverse)                              \
    {                                                                                                       \
        if (rows <= 0 || cols <= 0) return GKOC_OK;                                                         \
        diag_dense_kernel<T><<<dim3(grid_of(rows * cols)), dim3(256), 0, as_stream(s)>>>(                   \
            rows, cols, diag, b, ldb, c, ldc, inverse ? 1 : 0);                                             \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    extern "C" int gkoc_diagonal_right_apply_to_dense_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,     \
                                                           const T* diag, const T* b, int64_t ldb, T* c,    \
                                                           int64_t ldc)                                     \
    {                                                                                                       \
        if (rows <= 0 || cols <= 0) return GKOC_OK;                                                         \
        diag_dense_kernel<T><<<dim3(grid_of(rows * cols)), dim3(256), 0, as_stream(s)>>>(                   \
            rows, cols, diag, b, ldb, c, ldc, 2);                                                           \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    extern "C" int gkoc_reduce_add_array_##TN(gkoc_stream_t s, int64_t n, const T* arr, T* val)             \
    {                                                                                                       \
        return reduce_add<T>(s, n, arr, val);                                                               \
    }
GKOC_DEF_MISC_T(double, f64)
GKOC_DEF_MISC_T(float, f32)
GKOC_DEF_MISC_T(gkoc_c128, c128)
GKOC_DEF_MISC_T(gkoc_c64, c64)
extern "C" int gkoc_reduce_add_array_i32(gkoc_stream_t s, int64_t n, const int32_t* arr, int32_t* val)
{
    return reduce_add<int32_t>(s, n, arr, val);
}
extern "C" int gkoc_reduce_add_array_i64(gkoc_stream_t s, int64_t n, const int64_t* arr, int64_t* val)
{
    return reduce_add<int64_t>(s, n, arr, val);
}
extern "C" int gkoc_reduce_add_array_u64(gkoc_stream_t s, int64_t n, const uint64_t* arr, uint64_t* val)
{
    return reduce_add<uint64_t>(s, n, arr, val);
}
extern "C" int gkoc_convert_precision_f32_f64(gkoc_stream_t s, int64_t n, const float* in, double* out)
{
    if (n <= 0) return GKOC_OK;
    convert_kernel<float, double><<<dim3(grid_of(n)), dim3(256), 0, as_stream(s)>>>(n, in, out);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}
extern "C" int gkoc_convert_precision_f64_f32(gkoc_stream_t s, int64_t n, const double* in, float* out)
{
    if (n <= 0) return GKOC_OK;
    convert_kernel<double, float><<<dim3(grid_of(n)), dim3(256), 0, as_stream(s)>>>(n, in, out);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

#define GKOC_DEF_MISC_TI(T, TN, I, IN)                                                                      \
    extern "C" int gkoc_diagonal_apply_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const T* diag,   \
                                                          const I* row_ptrs, T* vals, int inverse)          \
    {                                                                                                       \
        if (n_rows <= 0) return GKOC_OK;                                                                    \
        diag_csr_rows_kernel<T, I><<<dim3(grid_of(n_rows * 64)), dim3(256), 0, as_stream(s)>>>(             \
            n_rows, diag, row_ptrs, vals, inverse);                                                         \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    extern "C" int gkoc_diagonal_right_apply_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t nnz,               \
                                                                const T* diag, const I* cols, T* vals)      \
    {                                                                                                       \
        if (nnz <= 0) return GKOC_OK;                                                                       \
        diag_csr_cols_kernel<T, I><<<dim3(grid_of(nnz)), dim3(256), 0, as_stream(s)>>>(nnz, diag, cols,     \
                                                                                      vals);                \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    extern "C" int gkoc_diagonal_convert_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t n, const T* diag,      \
                                                            I* row_ptrs, I* cols, T* vals)                  \
    {                                                                                                       \
        diag_to_csr_kernel<T, I><<<dim3(grid_of(n + 1)), dim3(256), 0, as_stream(s)>>>(n, diag, row_ptrs,   \
                                                                                      cols, vals);          \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    extern "C" int gkoc_diagonal_fill_in_matrix_data_##TN##_##IN(gkoc_stream_t s, int64_t nnz,              \
                                                                 const I* rows, const I* cols,              \
                                                                 const T* vals, T* diag)                    \
    {                                                                                                       \
        if (nnz <= 0) return GKOC_OK;                                                                       \
        diag_fill_kernel<T, I><<<dim3(grid_of(nnz)), dim3(256), 0, as_stream(s)>>>(nnz, rows, cols, vals,   \
                                                                                  diag);                    \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_MISC_TI(double, f64, int32_t, i32)
GKOC_DEF_MISC_TI(double, f64, int64_t, i64)
GKOC_DEF_MISC_TI(float, f32, int32_t, i32)
GKOC_DEF_MISC_TI(float, f32, int64_t, i64)
GKOC_DEF_MISC_TI(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_MISC_TI(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_MISC_TI(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_MISC_TI(gkoc_c64, c64, int64_t, i64)

#define GKOC_DEF_MISC_I(I, IN)                                                                              \
    extern "C" int gkoc_sparsity_csr_count_diagonal_##IN(gkoc_stream_t s, int64_t n_rows,                   \
                                                         const I* row_ptrs, const I* cols, I* counts)       \
    {                                                                                                       \
        count_diag_kernel<I><<<dim3(grid_of(n_rows + 1)), dim3(256), 0, as_stream(s)>>>(n_rows, row_ptrs,   \
                                                                                       cols, counts);       \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    extern "C" int gkoc_sparsity_csr_remove_diagonal_##IN(gkoc_stream_t s, int64_t n_rows,                  \
                                                          const I* row_ptrs, const I* cols,                 \
                                                          const I* diag_prefix_sum, I* adj_ptrs,            \
                                                          I* adj_idxs)                                      \
    {                                                                                                       \
        remove_diag_kernel<I><<<dim3(grid_of(n_rows + 1)), dim3(256), 0, as_stream(s)>>>(                   \
            n_rows, row_ptrs, cols, diag_prefix_sum, adj_ptrs, adj_idxs);                                   \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_MISC_I(int32_t, i32)
GKOC_DEF_MISC_I(int64_t, i64)

// components::fill_array for the small integer types Ginkgo's arrays are instantiated with (bool,
// char, uint16, uint32: core/components/fill_array_kernels.hpp:18-21 over
// GKO_INSTANTIATE_FOR_EACH_TEMPLATE_TYPE); elem_bytes 1, 2 or 4, the low bytes of `pattern`
namespace gkoc {
namespace {
template <typename U>
__global__ __launch_bounds__(256) void fill_small_kernel(int64_t n, U* __restrict__ data, U value)
{
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) data[i] = value;
}
}  // namespace
}  // namespace gkoc

extern "C" int gkoc_fill_array_small(gkoc_stream_t s, void* data, int64_t n, int elem_bytes, uint32_t pattern)
{
    GKOC_REQUIRE(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4, GKOC_E_INVALID, "elem_bytes");
    if (n <= 0) return GKOC_OK;
    const dim3 g(grid_of(n));
    if (elem_bytes == 1) {
        fill_small_kernel<uint8_t><<<g, dim3(256), 0, as_stream(s)>>>(n, static_cast<uint8_t*>(data), uint8_t(pattern));
    } else if (elem_bytes == 2) {
        fill_small_kernel<uint16_t><<<g, dim3(256), 0, as_stream(s)>>>(n, static_cast<uint16_t*>(data),
                                                                       uint16_t(pattern));
    } else {
        fill_small_kernel<uint32_t><<<g, dim3(256), 0, as_stream(s)>>>(n, static_cast<uint32_t*>(data), pattern);
    }
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// csr::spgemm_reuse / advanced_spgemm_reuse / spgeam_numeric (core/matrix/csr_kernels.hpp:60-92;
// reference/matrix/csr_kernels.cpp:304-436, :474-499): the values of a product or sum whose
// sparsity pattern exists already.  One lane per row of C, entries added in the reference's
// order.  The position of a column in C's row is found by bisection of its (sorted) column
// indices - Ginkgo's per-row lookup structures (bitmaps / hash tables, csr_lookup.hpp) are
// consumed by device kernels only, so this backend keeps none (build_lookup_offsets /
// build_lookup of the binding write empty ones).
namespace gkoc {
namespace {

template <typename I>
__device__ __forceinline__ int64_t find_col(const I* __restrict__ cols, int64_t b, int64_t e, I col)
{
    int64_t lo = b, hi = e;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cols[mid] < col) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    if (lo < e && cols[lo] == col) return lo;
    for (int64_t k = b; k < e; ++k) {   // an unsorted row
        if (cols[k] == col) return k;
    }
    return -1;
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void spgemm_reuse_kernel(
    int64_t n_rows, const I* __restrict__ a_ptrs, const I* __restrict__ a_cols, const T* __restrict__ a_vals,
    const I* __restrict__ b_ptrs, const I* __restrict__ b_cols, const T* __restrict__ b_vals,
    const T* __restrict__ alpha, const T* __restrict__ beta, const I* __restrict__ d_ptrs,
    const I* __restrict__ d_cols, const T* __restrict__ d_vals, const I* __restrict__ c_ptrs,
    const I* __restrict__ c_cols, T* __restrict__ c_vals)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n_rows) return;
    const int64_t cb = c_ptrs[row], ce = c_ptrs[row + 1];
    for (int64_t k = cb; k < ce; ++k) c_vals[k] = T(0);
    const bool adv = alpha != nullptr;
    const T va = adv ? alpha[0] : T(1);
    for (int64_t an = a_ptrs[row]; an < a_ptrs[row + 1]; ++an) {
        const int64_t ac = a_cols[an];
        const T av = a_vals[an];
        for (int64_t bn = b_ptrs[ac]; bn < b_ptrs[ac + 1]; ++bn) {
            const int64_t pos = find_col<I>(c_cols, cb, ce, b_cols[bn]);
            if (pos >= 0) c_vals[pos] += adv ? va * av * b_vals[bn] : av * b_vals[bn];
        }
    }
    if (adv) {
        const T vb = beta[0];
        for (int64_t dn = d_ptrs[row]; dn < d_ptrs[row + 1]; ++dn) {
            const int64_t pos = find_col<I>(c_cols, cb, ce, d_cols[dn]);
            if (pos >= 0) c_vals[pos] += vb * d_vals[dn];
        }
    }
}

// merge of the sorted rows of a and b; C's row holds the union of their columns in order
template <typename T, typename I>
__global__ __launch_bounds__(256) void spgeam_numeric_kernel(
    int64_t n_rows, const T* __restrict__ alpha, const I* __restrict__ a_ptrs, const I* __restrict__ a_cols,
    const T* __restrict__ a_vals, const T* __restrict__ beta, const I* __restrict__ b_ptrs,
    const I* __restrict__ b_cols, const T* __restrict__ b_vals, const I* __restrict__ c_ptrs,
    T* __restrict__ c_vals)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n_rows) return;
    const T va = alpha[0], vb = beta[0];
    int64_t ia = a_ptrs[row], ib = b_ptrs[row], out = c_ptrs[row];
    const int64_t ea = a_ptrs[row + 1], eb = b_ptrs[row + 1], ec = c_ptrs[row + 1];
    while ((ia < ea || ib < eb) && out < ec) {
        const int64_t ca = ia < ea ? int64_t(a_cols[ia]) : INT64_MAX;
        const int64_t cbv = ib < eb ? int64_t(b_cols[ib]) : INT64_MAX;
        const int64_t col = ca < cbv ? ca : cbv;
        const T av = ca == col ? a_vals[ia] : T(0);
        const T bv = cbv == col ? b_vals[ib] : T(0);
        c_vals[out++] = va * av + vb * bv;
        ia += ca == col;
        ib += cbv == col;
    }
}

}  // namespace
}  // namespace gkoc

#define GKOC_DEF_REUSE(T, TN, I, IN)                                                                        \
    extern "C" int gkoc_csr_spgemm_reuse_##TN##_##IN(                                                       \
        gkoc_stream_t s, int64_t n_rows, const I* a_ptrs, const I* a_cols, const T* a_vals,                 \
        const I* b_ptrs, const I* b_cols, const T* b_vals, const T* alpha, const T* beta,                   \
        const I* d_ptrs, const I* d_cols, const T* d_vals, const I* c_ptrs, const I* c_cols, T* c_vals)     \
    {                                                                                                       \
        if (n_rows <= 0) return GKOC_OK;                                                                    \
        GKOC_REQUIRE((alpha == nullptr) == (beta == nullptr) && (alpha == nullptr || d_ptrs),               \
                     GKOC_E_INVALID, "alpha, beta and d go together");                                      \
        spgemm_reuse_kernel<T, I><<<dim3(grid_of(n_rows)), dim3(256), 0, as_stream(s)>>>(                   \
            n_rows, a_ptrs, a_cols, a_vals, b_ptrs, b_cols, b_vals, alpha, beta, d_ptrs, d_cols, d_vals,    \
            c_ptrs, c_cols, c_vals);                                                                        \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }                                                                                                       \
    extern "C" int gkoc_csr_spgeam_numeric_##TN##_##IN(                                                     \
        gkoc_stream_t s, int64_t n_rows, const T* alpha, const I* a_ptrs, const I* a_cols,                  \
        const T* a_vals, const T* beta, const I* b_ptrs, const I* b_cols, const T* b_vals,                  \
        const I* c_ptrs, T* c_vals)                                                                         \
    {                                                                                                       \
        if (n_rows <= 0) return GKOC_OK;                                                                    \
        spgeam_numeric_kernel<T, I><<<dim3(grid_of(n_rows)), dim3(256), 0, as_stream(s)>>>(                 \
            n_rows, alpha, a_ptrs, a_cols, a_vals, beta, b_ptrs, b_cols, b_vals, c_ptrs, c_vals);           \
        GKOC_LAUNCH_OK();                                                                                   \
        return GKOC_OK;                                                                                     \
    }
GKOC_DEF_REUSE(double, f64, int32_t, i32)
GKOC_DEF_REUSE(double, f64, int64_t, i64)
GKOC_DEF_REUSE(float, f32, int32_t, i32)
GKOC_DEF_REUSE(float, f32, int64_t, i64)
GKOC_DEF_REUSE(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_REUSE(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_REUSE(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_REUSE(gkoc_c64, c64, int64_t, i64)
