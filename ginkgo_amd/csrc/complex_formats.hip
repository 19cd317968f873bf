// ELL and SELL-P products on complex values (ell::{spmv, advanced_spmv}, sellp::{spmv, advanced_spmv}
// for complex<float> / complex<double>; core/matrix/ell_kernels.hpp:20-34, sellp_kernels.hpp:20-31;
// semantics reference/matrix/ell_kernels.cpp:29-120, sellp_kernels.cpp:27-100).  Same argument lists
// as the real entries of formats.hip.  One lane per (row, right-hand side); consecutive lanes read
// consecutive rows of the column-major value / index arrays, i.e. coalesced - a complex product moves
// 16 + 4 bytes per stored entry and is bound by HBM with this layout as it stands (no LDS staging, no
// fragments: the tuned kernels of formats.hip exist for the real types the configurations use).
// A row's entries are added in storage order with the textbook complex product: the reference's
// result to rounding.  Padding (column -1) is skipped.  beta == 0 does not read c.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace gkoc {
namespace {

template <typename T>
__device__ __forceinline__ void cx_store(T* __restrict__ c, bool adv, T alpha, T beta, T sum)
{
    if (!adv) {
        *c = sum;
    } else {
        const T ax = alpha * sum;
        *c = beta == T(0) ? ax : ax + beta * *c;
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void cx_ell_spmv_kernel(int64_t rows, int64_t nrhs, int64_t per_row,
                                                         int64_t stride, const I* __restrict__ cols,
                                                         const T* __restrict__ vals,
                                                         const T* __restrict__ alpha_p,
                                                         const T* __restrict__ b, int64_t ldb,
                                                         const T* __restrict__ beta_p, T* __restrict__ c,
                                                         int64_t ldc)
{
    const bool adv = alpha_p != nullptr;
    const T alpha = adv ? alpha_p[0] : T(1), beta = adv ? beta_p[0] : T(0);
    const int64_t total = rows * nrhs, step = int64_t(gridDim.x) * 256;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += step) {
        // rows fastest: the lanes of a wave read one contiguous run of every stored column
        const int64_t j = idx / rows, row = idx - j * rows;
        T sum = T(0);
        for (int64_t k = 0; k < per_row; ++k) {
            const I col = cols[row + k * stride];
            if (col != I(-1)) sum += vals[row + k * stride] * b[int64_t(col) * ldb + j];
        }
        cx_store(c + row * ldc + j, adv, alpha, beta, sum);
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void cx_sellp_spmv_kernel(int64_t rows, int64_t nrhs, int64_t slice_size,
                                                           const uint64_t* __restrict__ slice_sets,
                                                           const uint64_t* __restrict__ slice_lengths,
                                                           const I* __restrict__ cols,
                                                           const T* __restrict__ vals,
                                                           const T* __restrict__ alpha_p,
                                                           const T* __restrict__ b, int64_t ldb,
                                                           const T* __restrict__ beta_p, T* __restrict__ c,
                                                           int64_t ldc)
{
    const bool adv = alpha_p != nullptr;
    const T alpha = adv ? alpha_p[0] : T(1), beta = adv ? beta_p[0] : T(0);
    const int64_t total = rows * nrhs, step = int64_t(gridDim.x) * 256;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += step) {
        const int64_t j = idx / rows, row = idx - j * rows;
        const int64_t slice = row / slice_size, in_slice = row - slice * slice_size;
        const int64_t base = int64_t(slice_sets[slice]) * slice_size + in_slice;
        const int64_t len = int64_t(slice_lengths[slice]);
        T sum = T(0);
        for (int64_t k = 0; k < len; ++k) {
            const I col = cols[base + k * slice_size];
            if (col != I(-1)) sum += vals[base + k * slice_size] * b[int64_t(col) * ldb + j];
        }
        cx_store(c + row * ldc + j, adv, alpha, beta, sum);
    }
}

inline unsigned blocks_for(int64_t n)
{
    int64_t b = ceildiv(n > 0 ? n : 1, 256);
    if (b > 8 * max_stream_blocks) b = 8 * max_stream_blocks;
    return unsigned(b);
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_CFMT(T, TN, I, IN)                                                                       \
    extern "C" int gkoc_ell_spmv_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t n_cols,             \
                                             int64_t per_row, int64_t stride, const I* col_idxs,          \
                                             const T* vals, const T* b, int64_t ldb, T* c, int64_t ldc,   \
                                             int64_t nrhs)                                                \
    {                                                                                                     \
        if (n_rows <= 0 || nrhs <= 0) return GKOC_OK;                                                     \
        GKOC_REQUIRE(per_row >= 0 && stride >= n_rows && c && (per_row == 0 || (col_idxs && vals && b)),  \
                     GKOC_E_INVALID, "bad argument");                                                     \
        cx_ell_spmv_kernel<T, I><<<dim3(blocks_for(n_rows * nrhs)), dim3(256), 0, as_stream(s)>>>(        \
            n_rows, nrhs, per_row, stride, col_idxs, vals, nullptr, b, ldb, nullptr, c, ldc);             \
        GKOC_LAUNCH_OK();                                                                                 \
        return GKOC_OK;                                                                                   \
    }                                                                                                     \
    extern "C" int gkoc_ell_advanced_spmv_##TN##_##IN(                                                    \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t per_row, int64_t stride, const T* alpha, \
        const I* col_idxs, const T* vals, const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc,      \
        int64_t nrhs)                                                                                     \
    {                                                                                                     \
        if (n_rows <= 0 || nrhs <= 0) return GKOC_OK;                                                     \
        GKOC_REQUIRE(per_row >= 0 && stride >= n_rows && c && alpha && beta &&                            \
                         (per_row == 0 || (col_idxs && vals && b)),                                       \
                     GKOC_E_INVALID, "bad argument");                                                     \
        cx_ell_spmv_kernel<T, I><<<dim3(blocks_for(n_rows * nrhs)), dim3(256), 0, as_stream(s)>>>(        \
            n_rows, nrhs, per_row, stride, col_idxs, vals, alpha, b, ldb, beta, c, ldc);                  \
        GKOC_LAUNCH_OK();                                                                                 \
        return GKOC_OK;                                                                                   \
    }                                                                                                     \
    extern "C" int gkoc_sellp_spmv_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t n_cols,           \
                                               int64_t slice_size, const uint64_t* slice_sets,            \
                                               const uint64_t* slice_lengths, const I* col_idxs,          \
                                               const T* vals, const T* b, int64_t ldb, T* c, int64_t ldc, \
                                               int64_t nrhs)                                              \
    {                                                                                                     \
        if (n_rows <= 0 || nrhs <= 0) return GKOC_OK;                                                     \
        GKOC_REQUIRE(slice_size > 0 && slice_sets && slice_lengths && c, GKOC_E_INVALID, "bad argument"); \
        cx_sellp_spmv_kernel<T, I><<<dim3(blocks_for(n_rows * nrhs)), dim3(256), 0, as_stream(s)>>>(      \
            n_rows, nrhs, slice_size, slice_sets, slice_lengths, col_idxs, vals, nullptr, b, ldb,         \
            nullptr, c, ldc);                                                                             \
        GKOC_LAUNCH_OK();                                                                                 \
        return GKOC_OK;                                                                                   \
    }                                                                                                     \
    extern "C" int gkoc_sellp_advanced_spmv_##TN##_##IN(                                                  \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t slice_size, const T* alpha,              \
        const uint64_t* slice_sets, const uint64_t* slice_lengths, const I* col_idxs, const T* vals,      \
        const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc, int64_t nrhs)                          \
    {                                                                                                     \
        if (n_rows <= 0 || nrhs <= 0) return GKOC_OK;                                                     \
        GKOC_REQUIRE(slice_size > 0 && slice_sets && slice_lengths && c && alpha && beta,                 \
                     GKOC_E_INVALID, "bad argument");                                                     \
        cx_sellp_spmv_kernel<T, I><<<dim3(blocks_for(n_rows * nrhs)), dim3(256), 0, as_stream(s)>>>(      \
            n_rows, nrhs, slice_size, slice_sets, slice_lengths, col_idxs, vals, alpha, b, ldb, beta, c,  \
            ldc);                                                                                         \
        GKOC_LAUNCH_OK();                                                                                 \
        return GKOC_OK;                                                                                   \
    }
GKOC_DEF_CFMT(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_CFMT(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_CFMT(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_CFMT(gkoc_c64, c64, int64_t, i64)
