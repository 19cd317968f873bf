// CSR SpMV for gfx950: row-segment-per-wavefront streaming kernel.
//
// Replaces gko::kernels::hip::csr::{spmv, advanced_spmv}
// (decl core/matrix/csr_kernels.hpp:29-43; semantics
// reference/matrix/csr_kernels.cpp:49-118; stock GPU version
// common/cuda_hip/matrix/csr_kernels.template.cpp:206-586,2351-2468).
//
// Design (not a translation of the stock classical/load-balance kernels):
//  * one 64-lane wavefront owns a segment of 64 consecutive rows; its nnz
//    range [row_ptrs[r0], row_ptrs[r0+64]) is contiguous in val / col_idx, so
//    the wave streams it with perfectly coalesced 512 B (val) / 256 B (col)
//    loads, lane = nnz index (no per-row alignment loss, no idle lanes on
//    27-nnz rows);
//  * the products val[k]*b[col[k]] are staged in LDS (14 KB per wave);
//  * then lane = row: each lane adds its row's products from LDS in k order.
//    => same summation order and same roundings as the sequential reference
//    (separate multiply and add; this file is compiled with
//    -ffp-contract=off), i.e. BIT-IDENTICAL results, no atomics, and one
//    coalesced 512 B store of y per wave;
//  * segments whose nnz exceed the LDS tile are processed tile by tile with the
//    per-row partial sum carried in a register (order still sequential);
//  * rows longer than GKOC_CSR_LONG_ROW are summed cooperatively by the whole
//    wave (tolerance instead of bit-exactness for those rows only).
//
// Algorithmic HBM bytes: nnz*(sizeof(T)+sizeof(I)) + (n+1)*sizeof(I)
//                        + n_cols*sizeof(T) (b once) + n*sizeof(T) (c).
#include "common.hpp"

namespace gkoc {
namespace {

template <typename T>
struct tile_cap {
    // products per wave: 14 KB of LDS => 11 single-wave workgroups per CU
    static constexpr int value = 14336 / sizeof(T);
};

// bijective XCD-aware remap of the workgroup id: dispatcher places block b on
// XCD b % 8 (MI355X_MICROARCH.md); give every XCD a contiguous band of row
// segments so that the b-vector lines shared by neighbouring segments stay in
// one XCD's L2.
__device__ __forceinline__ int64_t xcd_band_remap(int64_t bid, int64_t n)
{
    constexpr int64_t nx = 8;
    const int64_t q = n / nx, r = n % nx;
    const int64_t xcd = bid % nx, idx = bid / nx;
    const int64_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <typename T, typename I, bool ADV, bool REMAP, int UNROLL>
__global__ __launch_bounds__(64) void csr_spmv_wave_kernel(
    int64_t n_rows, int64_t n_segments, const I* __restrict__ row_ptrs,
    const I* __restrict__ cols, const T* __restrict__ vals,
    const T* __restrict__ b, int64_t ldb, T* __restrict__ c, int64_t ldc,
    int nrhs, const T* __restrict__ alpha_p, const T* __restrict__ beta_p)
{
    constexpr int CAP = tile_cap<T>::value;
    __shared__ T prod[CAP];
    const int lane = threadIdx.x;
    const int64_t seg =
        REMAP ? xcd_band_remap(blockIdx.x, n_segments) : int64_t(blockIdx.x);
    const int64_t r0 = seg * 64;
    const int64_t row = r0 + lane;
    const bool valid = row < n_rows;
    const int64_t r_last = (r0 + 64 < n_rows) ? r0 + 64 : n_rows;
    const int64_t rs = row_ptrs[valid ? row : r_last];
    const int64_t re = row_ptrs[valid ? row + 1 : r_last];
    const int64_t k0 = __shfl(rs, 0, 64);
    const int64_t k1 = __shfl(re, int(r_last - r0 - 1), 64);
    const bool is_long = (re - rs) > GKOC_CSR_LONG_ROW;
    const unsigned long long long_mask = __ballot(is_long);

    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }

    for (int j = 0; j < nrhs; ++j) {
        T sum = T(0);
        if (ADV && valid && beta != T(0)) {
            sum = c[row * ldc + j] * beta;
        }
        for (int64_t t0 = k0; t0 < k1; t0 += CAP) {
            const int64_t t1 = (t0 + CAP < k1) ? t0 + CAP : k1;
            // phase 1: lane = nnz; coalesced stream of val/col, gather b
            for (int64_t base = t0; base < t1; base += 64 * UNROLL) {
                T v[UNROLL];
                I cc[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    int64_t k = base + u * 64 + lane;
                    k = k < t1 ? k : t1 - 1;
                    v[u] = vals[k];
                    cc[u] = cols[k];
                }
                T xv[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    xv[u] = b[int64_t(cc[u]) * ldb + j];
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int64_t k = base + u * 64 + lane;
                    if (k < t1) {
                        prod[k - t0] = ADV ? (alpha * v[u]) * xv[u]
                                           : v[u] * xv[u];
                    }
                }
            }
            wave_lds_sync();
            // phase 2: lane = row; sequential (reference-order) row sums
            if (!is_long) {
                const int64_t a = rs > t0 ? rs : t0;
                const int64_t e = re < t1 ? re : t1;
                for (int64_t k = a; k < e; ++k) {
                    sum += prod[k - t0];
                }
            }
            wave_lds_sync();
        }
        // long rows: whole-wave cooperative dot straight from global memory
        unsigned long long m = long_mask;
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const int64_t lrs = __shfl(rs, src, 64);
            const int64_t lre = __shfl(re, src, 64);
            T part = T(0);
            for (int64_t k = lrs + lane; k < lre; k += 64) {
                const T p = ADV ? (alpha * vals[k]) * b[int64_t(cols[k]) * ldb + j]
                                : vals[k] * b[int64_t(cols[k]) * ldb + j];
                part += p;
            }
            part = wave_sum(part);
            if (lane == src) sum += part;
        }
        if (valid) {
            c[row * ldc + j] = sum;
        }
    }
}

template <typename T, typename I, bool ADV>
int launch_csr(gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const T* alpha,
               const I* row_ptrs, const I* col_idxs, const T* vals, const T* b,
               int64_t ldb, const T* beta, T* c, int64_t ldc, int64_t nrhs)
{
    GKOC_REQUIRE(n_rows >= 0 && n_cols >= 0 && nrhs >= 0, GKOC_E_INVALID,
                 "negative dimension");
    if (n_rows == 0 || nrhs == 0) return GKOC_OK;
    GKOC_REQUIRE(row_ptrs && c, GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(ldc >= nrhs && (n_cols == 0 || ldb >= nrhs), GKOC_E_INVALID,
                 "stride smaller than nrhs");
    if (ADV) GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha/beta");
    const int64_t n_seg = ceildiv(n_rows, 64);
    GKOC_REQUIRE(n_seg < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED,
                 "more than 2^31 row segments");
    // XCD band remap pays once every XCD gets a long band; tiny matrices keep
    // the natural order
    const bool remap = n_seg >= 4096;
    dim3 grid(static_cast<unsigned>(n_seg)), block(64);
    if (remap) {
        csr_spmv_wave_kernel<T, I, ADV, true, 9><<<grid, block, 0, as_stream(s)>>>(
            n_rows, n_seg, row_ptrs, col_idxs, vals, b, ldb, c, ldc,
            static_cast<int>(nrhs), alpha, beta);
    } else {
        csr_spmv_wave_kernel<T, I, ADV, false, 9><<<grid, block, 0, as_stream(s)>>>(
            n_rows, n_seg, row_ptrs, col_idxs, vals, b, ldb, c, ldc,
            static_cast<int>(nrhs), alpha, beta);
    }
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// ---- diagonal extraction / sortedness / per-row sort ---------------------

template <typename T, typename I>
__global__ __launch_bounds__(256) void extract_diag_kernel(
    int64_t n_diag, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, T* __restrict__ diag)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n_diag) return;
    T d = T(0);
    for (int64_t k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
        if (int64_t(cols[k]) == row) {
            d = vals[k];
            break;
        }
    }
    diag[row] = d;
}

template <typename I>
__global__ __launch_bounds__(256) void is_sorted_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    int* __restrict__ flag)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n_rows) return;
    bool ok = true;
    for (int64_t k = row_ptrs[row] + 1; k < row_ptrs[row + 1]; ++k) {
        if (cols[k - 1] > cols[k]) {
            ok = false;
            break;
        }
    }
    if (!ok) atomicAnd(flag, 0);
}

// stable insertion sort per row (rows are short on this path; Ginkgo only
// calls it when is_sorted_by_column_index returned false)
template <typename T, typename I>
__global__ __launch_bounds__(256) void sort_rows_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, I* __restrict__ cols,
    T* __restrict__ vals)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n_rows) return;
    const int64_t a = row_ptrs[row], e = row_ptrs[row + 1];
    for (int64_t i = a + 1; i < e; ++i) {
        const I ci = cols[i];
        const T vi = vals[i];
        int64_t k = i - 1;
        while (k >= a && cols[k] > ci) {
            cols[k + 1] = cols[k];
            vals[k + 1] = vals[k];
            --k;
        }
        cols[k + 1] = ci;
        vals[k + 1] = vi;
    }
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_CSR(T, TN, I, IN)                                             \
    extern "C" int gkoc_csr_spmv_##TN##_##IN(                                  \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const I* row_ptrs,    \
        const I* col_idxs, const T* vals, const T* b, int64_t ldb, T* c,       \
        int64_t ldc, int64_t nrhs)                                             \
    {                                                                          \
        return launch_csr<T, I, false>(s, n_rows, n_cols, nullptr, row_ptrs,   \
                                       col_idxs, vals, b, ldb, nullptr, c,     \
                                       ldc, nrhs);                             \
    }                                                                          \
    extern "C" int gkoc_csr_advanced_spmv_##TN##_##IN(                         \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const T* alpha,       \
        const I* row_ptrs, const I* col_idxs, const T* vals, const T* b,       \
        int64_t ldb, const T* beta, T* c, int64_t ldc, int64_t nrhs)           \
    {                                                                          \
        return launch_csr<T, I, true>(s, n_rows, n_cols, alpha, row_ptrs,      \
                                      col_idxs, vals, b, ldb, beta, c, ldc,    \
                                      nrhs);                                   \
    }                                                                          \
    extern "C" int gkoc_csr_extract_diagonal_##TN##_##IN(                      \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const I* row_ptrs,    \
        const I* col_idxs, const T* vals, T* diag)                             \
    {                                                                          \
        const int64_t nd = n_rows < n_cols ? n_rows : n_cols;                  \
        if (nd <= 0) return GKOC_OK;                                           \
        extract_diag_kernel<T, I>                                              \
            <<<dim3(unsigned(ceildiv(nd, 256))), dim3(256), 0, as_stream(s)>>>( \
                nd, row_ptrs, col_idxs, vals, diag);                           \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_csr_is_sorted_by_column_index_##TN##_##IN(             \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                    \
        const I* col_idxs, int* is_sorted_host)                                \
    {                                                                          \
        GKOC_REQUIRE(is_sorted_host, GKOC_E_INVALID, "null result");           \
        *is_sorted_host = 1;                                                   \
        if (n_rows <= 0) return GKOC_OK;                                       \
        int* flag = nullptr;                                                   \
        GKOC_HIP(hipMallocAsync(reinterpret_cast<void**>(&flag), sizeof(int),  \
                                as_stream(s)));                                \
        int one = 1;                                                           \
        GKOC_HIP(hipMemcpyAsync(flag, &one, sizeof(int),                       \
                                hipMemcpyHostToDevice, as_stream(s)));         \
        is_sorted_kernel<I>                                                    \
            <<<dim3(unsigned(ceildiv(n_rows, 256))), dim3(256), 0,             \
               as_stream(s)>>>(n_rows, row_ptrs, col_idxs, flag);              \
        GKOC_LAUNCH_OK();                                                      \
        GKOC_HIP(hipMemcpyAsync(is_sorted_host, flag, sizeof(int),             \
                                hipMemcpyDeviceToHost, as_stream(s)));         \
        GKOC_HIP(hipStreamSynchronize(as_stream(s)));                          \
        GKOC_HIP(hipFreeAsync(flag, as_stream(s)));                            \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_csr_sort_by_column_index_##TN##_##IN(                  \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, I* col_idxs,       \
        T* vals)                                                               \
    {                                                                          \
        if (n_rows <= 0) return GKOC_OK;                                       \
        sort_rows_kernel<T, I>                                                 \
            <<<dim3(unsigned(ceildiv(n_rows, 256))), dim3(256), 0,             \
               as_stream(s)>>>(n_rows, row_ptrs, col_idxs, vals);              \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }

GKOC_DEF_CSR(double, f64, int32_t, i32)
GKOC_DEF_CSR(double, f64, int64_t, i64)
GKOC_DEF_CSR(float, f32, int32_t, i32)
GKOC_DEF_CSR(float, f32, int64_t, i64)
