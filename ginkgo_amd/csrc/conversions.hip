// Format conversions and small matrix utilities on either side of the SpMV hot path
// ("callers and data formats"): everything Ginkgo's Dense / Csr / Coo / Ell /
// Sellp / Hybrid classes ask the device for when a matrix moves between formats, plus the
// diagonal / absolute / transpose helpers.  Restated from the reference kernels cited per function
// (reference/matrix/{dense,csr,coo,ell,sellp,hybrid}_kernels.cpp); all of it is copy / index work,
// reproduced entry for entry (output order = the reference's row-by-row walk).  None of this is on
// the iteration path: kernels are one thread per row (or per entry), grid-stride.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "scan.hpp"

namespace gkoc {
namespace {

inline unsigned cv_grid(int64_t n)
{
    int64_t b = ceildiv(n, 256);
    if (b > 4 * max_stream_blocks) b = 4 * max_stream_blocks;
    return unsigned(b < 1 ? 1 : b);
}

#define GKOC_FOR_EACH(i, n) \
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < (n); i += int64_t(gridDim.x) * 256)

template <typename T>
__device__ __forceinline__ bool nonzero(T v)
{
    return v != T(0);   // is_nonzero (include/ginkgo/core/base/math.hpp): NaN counts, -0 does not
}

// ------------------------------------------------------------------ arrays
template <typename T>
__global__ __launch_bounds__(256) void fill_seq_kernel(int64_t n, T* data)
{
    GKOC_FOR_EACH(i, n) data[i] = T(i);
}

// ------------------------------------------------------------------- dense
template <typename T>
__global__ __launch_bounds__(256) void dense_transpose_kernel(int64_t rows, int64_t cols,
                                                              const T* __restrict__ in, int64_t ldi,
                                                              T* __restrict__ out, int64_t ldo)
{
    // one 16 x 16 tile per 256 threads through LDS would coalesce both sides; the matrices that
    // come here are right-hand-side blocks and test fixtures
    GKOC_FOR_EACH(t, rows * cols)
    {
        const int64_t i = t / cols, j = t - i * cols;
        out[j * ldo + i] = in[i * ldi + j];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dense_extract_diagonal_kernel(int64_t n, const T* __restrict__ in,
                                                                     int64_t ld, T* __restrict__ diag)
{
    GKOC_FOR_EACH(i, n) diag[i] = in[i * ld + i];
}

// reference/matrix/dense_kernels.cpp add_scaled_identity: m *= beta, diagonal += alpha
template <typename T>
__global__ __launch_bounds__(256) void dense_add_scaled_identity_kernel(int64_t rows, int64_t cols,
                                                                        const T* __restrict__ alpha,
                                                                        const T* __restrict__ beta,
                                                                        T* __restrict__ m, int64_t ld)
{
    const T a = alpha[0], b = beta[0];
    GKOC_FOR_EACH(t, rows * cols)
    {
        const int64_t i = t / cols, j = t - i * cols;
        T v = m[i * ld + j] * b;
        if (i == j) v += a;
        m[i * ld + j] = v;
    }
}

// add_scaled_diag / sub_scaled_diag (:229-257): nothing happens for alpha == 0
template <typename T, bool SUB>
__global__ __launch_bounds__(256) void dense_scaled_diag_kernel(int64_t n, const T* __restrict__ alpha,
                                                                const T* __restrict__ diag,
                                                                T* __restrict__ y, int64_t ld)
{
    const T a = alpha[0];
    if (!nonzero(a)) return;
    GKOC_FOR_EACH(i, n)
    {
        const T p = a * diag[i];
        y[i * ld + i] = SUB ? y[i * ld + i] - p : y[i * ld + i] + p;
    }
}

template <typename T, typename O>
__global__ __launch_bounds__(256) void dense_count_nnz_kernel(int64_t rows, int64_t cols,
                                                              const T* __restrict__ in, int64_t ld,
                                                              O* __restrict__ out)
{
    GKOC_FOR_EACH(r, rows)
    {
        O c = 0;
        for (int64_t j = 0; j < cols; ++j) c += nonzero(in[r * ld + j]) ? O(1) : O(0);
        out[r] = c;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dense_max_nnz_kernel(int64_t rows, int64_t cols,
                                                            const T* __restrict__ in, int64_t ld,
                                                            unsigned long long* __restrict__ out)
{
    GKOC_FOR_EACH(r, rows)
    {
        unsigned long long c = 0;
        for (int64_t j = 0; j < cols; ++j) c += nonzero(in[r * ld + j]) ? 1 : 0;
        atomicMax(out, c);
    }
}

// compute_slice_sets (:733-758): slice length = max over the slice's rows of the row count
// rounded up to stride_factor
template <typename T>
__global__ __launch_bounds__(256) void dense_slice_lengths_kernel(int64_t rows, int64_t cols,
                                                                  const T* __restrict__ in, int64_t ld,
                                                                  int64_t slice_size, int64_t stride_factor,
                                                                  int64_t num_slices,
                                                                  uint64_t* __restrict__ lengths,
                                                                  uint64_t* __restrict__ sets)
{
    GKOC_FOR_EACH(sl, num_slices + 1)
    {
        uint64_t len = 0;
        if (sl < num_slices) {
            for (int64_t lr = 0; lr < slice_size; ++lr) {
                const int64_t r = sl * slice_size + lr;
                if (r >= rows) break;
                uint64_t c = 0;
                for (int64_t j = 0; j < cols; ++j) c += nonzero(in[r * ld + j]) ? 1 : 0;
                const uint64_t up = (c + stride_factor - 1) / stride_factor * stride_factor;
                len = up > len ? up : len;
            }
            lengths[sl] = len;
        }
        sets[sl] = len;   // scanned in place afterwards; the entry past the end starts at 0
    }
}

// Dense -> Csr / SparsityCsr (vals == NULL) / Coo (rows_out != NULL): row pointers come from the
// caller (count + prefix sum, as core/matrix/dense.cpp does), entries in column order
template <typename T, typename I, typename P>
__global__ __launch_bounds__(256) void dense_to_rows_kernel(int64_t rows, int64_t cols,
                                                            const T* __restrict__ in, int64_t ld,
                                                            const P* __restrict__ row_ptrs,
                                                            I* __restrict__ rows_out,
                                                            I* __restrict__ cols_out,
                                                            T* __restrict__ vals_out)
{
    GKOC_FOR_EACH(r, rows)
    {
        int64_t k = int64_t(row_ptrs[r]);
        for (int64_t j = 0; j < cols; ++j) {
            const T v = in[r * ld + j];
            if (nonzero(v)) {
                if (rows_out) rows_out[k] = I(r);
                cols_out[k] = I(j);
                if (vals_out) vals_out[k] = v;
                ++k;
            }
        }
    }
}

// Dense -> Ell (:518-543) and the Ell part of Dense -> Hybrid (:597-639, coo_* != NULL): the first
// ell_lim non-zeros of a row, the padding (0, invalid_index = -1) over the whole stride; the rest
// goes to Coo at coo_row_ptrs[row]
template <typename T, typename I>
__global__ __launch_bounds__(256) void dense_to_ell_kernel(int64_t rows, int64_t cols,
                                                           const T* __restrict__ in, int64_t ld,
                                                           int64_t ell_k, int64_t ell_lim, int64_t stride,
                                                           I* __restrict__ ell_cols,
                                                           T* __restrict__ ell_vals,
                                                           const int64_t* __restrict__ coo_row_ptrs,
                                                           I* __restrict__ coo_rows,
                                                           I* __restrict__ coo_cols,
                                                           T* __restrict__ coo_vals)
{
    GKOC_FOR_EACH(r, stride)
    {
        int64_t k = 0;
        int64_t c = coo_row_ptrs ? coo_row_ptrs[r < rows ? r : rows] : 0;
        if (r < rows) {
            for (int64_t j = 0; j < cols; ++j) {
                const T v = in[r * ld + j];
                if (!nonzero(v)) continue;
                if (k < ell_lim) {
                    ell_cols[r + k * stride] = I(j);
                    ell_vals[r + k * stride] = v;
                    ++k;
                } else if (coo_row_ptrs) {
                    coo_rows[c] = I(r);
                    coo_cols[c] = I(j);
                    coo_vals[c] = v;
                    ++c;
                }
            }
        }
        for (; k < ell_k; ++k) {
            ell_cols[r + k * stride] = I(-1);
            ell_vals[r + k * stride] = T(0);
        }
    }
}

// Dense -> Sellp (:646-675)
template <typename T, typename I>
__global__ __launch_bounds__(256) void dense_to_sellp_kernel(int64_t rows, int64_t cols,
                                                             const T* __restrict__ in, int64_t ld,
                                                             int64_t slice_size,
                                                             const uint64_t* __restrict__ slice_sets,
                                                             I* __restrict__ cols_out,
                                                             T* __restrict__ vals_out)
{
    GKOC_FOR_EACH(r, rows)
    {
        const int64_t sl = r / slice_size, lr = r - sl * slice_size;
        int64_t at = int64_t(slice_sets[sl]) * slice_size + lr;
        const int64_t end = int64_t(slice_sets[sl + 1]) * slice_size + lr;
        for (int64_t j = 0; j < cols; ++j) {
            const T v = in[r * ld + j];
            if (nonzero(v)) {
                cols_out[at] = I(j);
                vals_out[at] = v;
                at += slice_size;
            }
        }
        for (; at < end; at += slice_size) {
            cols_out[at] = I(-1);
            vals_out[at] = T(0);
        }
    }
}

// --------------------------------------------------------- sparse -> dense
// fill_in_dense: the caller has zeroed the result (core/matrix/*.cpp convert_to(Dense))
template <typename T, typename I>
__global__ __launch_bounds__(256) void csr_fill_in_dense_kernel(int64_t n_rows,
                                                                const I* __restrict__ row_ptrs,
                                                                const I* __restrict__ cols,
                                                                const T* __restrict__ vals,
                                                                T* __restrict__ out, int64_t ld)
{
    GKOC_FOR_EACH(r, n_rows)
    {
        for (int64_t k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) out[r * ld + cols[k]] = vals[k];
    }
}

template <typename T>
__device__ __forceinline__ void atomic_add_value(T* p, T v)
{
    atomicAdd(p, v);
}
template <typename R>
__device__ __forceinline__ void atomic_add_value(gkoc_cplx<R>* p, gkoc_cplx<R> v)
{
    atomicAdd(&p->re, v.re);
    atomicAdd(&p->im, v.im);
}

// coo (reference/matrix/coo_kernels.cpp:104-114): +=, so duplicates add up
template <typename T, typename I>
__global__ __launch_bounds__(256) void coo_fill_in_dense_kernel(int64_t nnz, const I* __restrict__ rows,
                                                                const I* __restrict__ cols,
                                                                const T* __restrict__ vals,
                                                                T* __restrict__ out, int64_t ld)
{
    GKOC_FOR_EACH(k, nnz) atomic_add_value(&out[int64_t(rows[k]) * ld + cols[k]], vals[k]);
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void ell_fill_in_dense_kernel(int64_t n_rows, int64_t ell_k,
                                                                int64_t stride,
                                                                const I* __restrict__ cols,
                                                                const T* __restrict__ vals,
                                                                T* __restrict__ out, int64_t ld)
{
    GKOC_FOR_EACH(r, n_rows)
    {
        for (int64_t k = 0; k < ell_k; ++k) {
            const I c = cols[r + k * stride];
            if (c != I(-1)) out[r * ld + c] = vals[r + k * stride];
        }
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void sellp_fill_in_dense_kernel(int64_t n_rows, int64_t slice_size,
                                                                  const uint64_t* __restrict__ slice_sets,
                                                                  const I* __restrict__ cols,
                                                                  const T* __restrict__ vals,
                                                                  T* __restrict__ out, int64_t ld)
{
    GKOC_FOR_EACH(r, n_rows)
    {
        const int64_t sl = r / slice_size, lr = r - sl * slice_size;
        for (int64_t i = int64_t(slice_sets[sl]); i < int64_t(slice_sets[sl + 1]); ++i) {
            const I c = cols[lr + i * slice_size];
            if (c != I(-1)) out[r * ld + c] = vals[lr + i * slice_size];
        }
    }
}

// ------------------------------------------------------------ diagonals
// extract_diagonal: the first entry of row r with column r (ell / sellp: the reference breaks at
// the first match; csr: csr_spmv.hip); coo: every entry with row == column (one per row in a valid
// matrix)
template <typename T, typename I>
__global__ __launch_bounds__(256) void ell_extract_diagonal_kernel(int64_t n, int64_t ell_k, int64_t stride,
                                                                   const I* __restrict__ cols,
                                                                   const T* __restrict__ vals,
                                                                   T* __restrict__ diag)
{
    GKOC_FOR_EACH(r, n)
    {
        for (int64_t k = 0; k < ell_k; ++k) {
            if (int64_t(cols[r + k * stride]) == r) {
                diag[r] = vals[r + k * stride];
                break;
            }
        }
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void sellp_extract_diagonal_kernel(
    int64_t n, int64_t slice_size, const uint64_t* __restrict__ slice_sets, const I* __restrict__ cols,
    const T* __restrict__ vals, T* __restrict__ diag)
{
    GKOC_FOR_EACH(r, n)
    {
        const int64_t sl = r / slice_size, lr = r - sl * slice_size;
        for (int64_t i = int64_t(slice_sets[sl]); i < int64_t(slice_sets[sl + 1]); ++i) {
            if (int64_t(cols[lr + i * slice_size]) == r) {
                diag[r] = vals[lr + i * slice_size];
                break;
            }
        }
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void coo_extract_diagonal_kernel(int64_t nnz, const I* __restrict__ rows,
                                                                   const I* __restrict__ cols,
                                                                   const T* __restrict__ vals,
                                                                   T* __restrict__ diag)
{
    GKOC_FOR_EACH(k, nnz)
    {
        if (rows[k] == cols[k]) diag[rows[k]] = vals[k];
    }
}

// check_diagonal_entries_exist (reference/matrix/csr_kernels.cpp:1375-1395): *missing != 0 if a row
// below min(rows, cols) has no entry on the diagonal
template <typename I>
__global__ __launch_bounds__(256) void csr_missing_diagonal_kernel(int64_t n, const I* __restrict__ row_ptrs,
                                                                   const I* __restrict__ cols,
                                                                   int* __restrict__ missing)
{
    GKOC_FOR_EACH(r, n)
    {
        bool found = false;
        for (int64_t k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) found |= int64_t(cols[k]) == r;
        if (!found) *missing = 1;
    }
}

// csr::add_scaled_identity (:1402-1418): every stored value *= beta, the diagonal ones += alpha
template <typename T, typename I>
__global__ __launch_bounds__(256) void csr_add_scaled_identity_kernel(int64_t n_rows,
                                                                      const I* __restrict__ row_ptrs,
                                                                      const I* __restrict__ cols,
                                                                      T* __restrict__ vals,
                                                                      const T* __restrict__ alpha,
                                                                      const T* __restrict__ beta)
{
    const T a = alpha[0], b = beta[0];
    GKOC_FOR_EACH(r, n_rows)
    {
        for (int64_t k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
            T v = vals[k] * b;
            if (int64_t(cols[k]) == r) v += a;
            vals[k] = v;
        }
    }
}

// ------------------------------------------------------- Ell / Sellp / Hybrid -> Csr
template <typename I>
__global__ __launch_bounds__(256) void ell_count_kernel(int64_t n_rows, int64_t ell_k, int64_t stride,
                                                        const I* __restrict__ cols, I* __restrict__ out)
{
    GKOC_FOR_EACH(r, n_rows)
    {
        I c = 0;
        for (int64_t k = 0; k < ell_k; ++k) c += cols[r + k * stride] != I(-1) ? I(1) : I(0);
        out[r] = c;
    }
}

template <typename I>
__global__ __launch_bounds__(256) void sellp_count_kernel(int64_t n_rows, int64_t slice_size,
                                                          const uint64_t* __restrict__ slice_sets,
                                                          const I* __restrict__ cols, I* __restrict__ out)
{
    GKOC_FOR_EACH(r, n_rows)
    {
        const int64_t sl = r / slice_size, lr = r - sl * slice_size;
        I c = 0;
        for (int64_t i = int64_t(slice_sets[sl]); i < int64_t(slice_sets[sl + 1]); ++i) {
            c += cols[lr + i * slice_size] != I(-1) ? I(1) : I(0);
        }
        out[r] = c;
    }
}

// the stored entries of a row in storage order, padding skipped wherever it sits
// (reference/matrix/ell_kernels.cpp:212-237, sellp_kernels.cpp:246-285)
template <typename T, typename I>
__global__ __launch_bounds__(256) void ell_to_csr_kernel(int64_t n_rows, int64_t ell_k, int64_t stride,
                                                         const I* __restrict__ cols,
                                                         const T* __restrict__ vals,
                                                         const I* __restrict__ row_ptrs,
                                                         I* __restrict__ out_cols,
                                                         T* __restrict__ out_vals)
{
    GKOC_FOR_EACH(r, n_rows)
    {
        int64_t at = row_ptrs[r];
        for (int64_t k = 0; k < ell_k; ++k) {
            const I c = cols[r + k * stride];
            if (c != I(-1)) {
                out_cols[at] = c;
                out_vals[at] = vals[r + k * stride];
                ++at;
            }
        }
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void sellp_to_csr_kernel(int64_t n_rows, int64_t slice_size,
                                                           const uint64_t* __restrict__ slice_sets,
                                                           const I* __restrict__ cols,
                                                           const T* __restrict__ vals,
                                                           const I* __restrict__ row_ptrs,
                                                           I* __restrict__ out_cols,
                                                           T* __restrict__ out_vals)
{
    GKOC_FOR_EACH(r, n_rows)
    {
        const int64_t sl = r / slice_size, lr = r - sl * slice_size;
        int64_t at = row_ptrs[r];
        for (int64_t i = int64_t(slice_sets[sl]); i < int64_t(slice_sets[sl + 1]); ++i) {
            const I c = cols[lr + i * slice_size];
            if (c != I(-1)) {
                out_cols[at] = c;
                out_vals[at] = vals[lr + i * slice_size];
                ++at;
            }
        }
    }
}

// hybrid::convert_to_csr (reference/matrix/hybrid_kernels.cpp:94-131): a row's Ell entries, then
// its Coo entries; ell_row_ptrs / coo_row_ptrs are the exclusive sums core/matrix/hybrid.cpp:300-308
// computes, out_row_ptrs = their sum
template <typename T, typename I>
__global__ __launch_bounds__(256) void hybrid_to_csr_kernel(
    int64_t n_rows, int64_t ell_k, int64_t stride, const I* __restrict__ ell_cols,
    const T* __restrict__ ell_vals, const I* __restrict__ coo_cols, const T* __restrict__ coo_vals,
    const I* __restrict__ ell_row_ptrs, const I* __restrict__ coo_row_ptrs, I* __restrict__ out_row_ptrs,
    I* __restrict__ out_cols, T* __restrict__ out_vals)
{
    GKOC_FOR_EACH(r, n_rows + 1)
    {
        int64_t at = int64_t(ell_row_ptrs[r]) + int64_t(coo_row_ptrs[r]);
        out_row_ptrs[r] = I(at);
        if (r == n_rows) continue;
        for (int64_t k = 0; k < ell_k; ++k) {
            const I c = ell_cols[r + k * stride];
            if (c != I(-1)) {
                out_cols[at] = c;
                out_vals[at] = ell_vals[r + k * stride];
                ++at;
            }
        }
        for (int64_t k = coo_row_ptrs[r]; k < coo_row_ptrs[r + 1]; ++k) {
            out_cols[at] = coo_cols[k];
            out_vals[at] = coo_vals[k];
            ++at;
        }
    }
}

// ------------------------------------------------------------ permutations
// Dense (reference/matrix/dense_kernels.cpp:838-1160): with si = row_perm[i] (or i), sj = col_perm[j]
// (or j): forward out(i, j) = scale * in(si, sj), inverse out(si, sj) = in(i, j) / scale, where scale
// = row_scale[si] * col_scale[sj], or the one of the two that is given, or nothing.
template <typename T, typename I, bool INVERSE>
__global__ __launch_bounds__(256) void dense_permute_kernel(int64_t rows, int64_t cols,
                                                            const T* __restrict__ in, int64_t ldi,
                                                            T* __restrict__ out, int64_t ldo,
                                                            const I* __restrict__ row_perm,
                                                            const I* __restrict__ col_perm,
                                                            const T* __restrict__ row_scale,
                                                            const T* __restrict__ col_scale)
{
    GKOC_FOR_EACH(t, rows * cols)
    {
        const int64_t i = t / cols, j = t - i * cols;
        const int64_t si = row_perm ? int64_t(row_perm[i]) : i;
        const int64_t sj = col_perm ? int64_t(col_perm[j]) : j;
        T v = INVERSE ? in[i * ldi + j] : in[si * ldi + sj];
        if (row_scale && col_scale) {
            const T sc = row_scale[si] * col_scale[sj];
            v = INVERSE ? v / sc : sc * v;
        } else if (row_scale || col_scale) {
            const T sc = row_scale ? row_scale[si] : col_scale[sj];
            v = INVERSE ? v / sc : sc * v;
        }
        if (INVERSE) {
            out[si * ldo + sj] = v;
        } else {
            out[i * ldo + j] = v;
        }
    }
}

// advanced_row_gather (:932-950): out(i, :) = alpha in(rows[i], :) + beta out(i, :)
template <typename T, typename I>
__global__ __launch_bounds__(256) void dense_advanced_row_gather_kernel(
    int64_t n_gather, int64_t cols, const T* __restrict__ alpha, const I* __restrict__ rows_idx,
    const T* __restrict__ in, int64_t ldi, const T* __restrict__ beta, T* __restrict__ out, int64_t ldo)
{
    const T a = alpha[0], b = beta[0];
    GKOC_FOR_EACH(t, n_gather * cols)
    {
        const int64_t i = t / cols, j = t - i * cols;
        out[i * ldo + j] = a * in[int64_t(rows_idx[i]) * ldi + j] + b * out[i * ldo + j];
    }
}

// Csr (reference/matrix/csr_kernels.cpp:962-1262).  Rows: forward (row_permute) out row i = in row
// perm[i]; inverse out row perm[i] = in row i.  Columns only ever inverse: new column =
// col_perm[old column] (entries keep their place; the caller sorts afterwards).
template <typename I>
__global__ __launch_bounds__(256) void csr_permute_lengths_kernel(int64_t n_rows,
                                                                  const I* __restrict__ in_rp,
                                                                  const I* __restrict__ row_perm,
                                                                  bool row_inverse, I* __restrict__ out_rp)
{
    GKOC_FOR_EACH(r, n_rows + 1)
    {
        if (r == n_rows) {
            out_rp[n_rows] = 0;
            continue;
        }
        const int64_t src = (row_perm && !row_inverse) ? int64_t(row_perm[r]) : r;
        const int64_t dst = (row_perm && row_inverse) ? int64_t(row_perm[r]) : r;
        out_rp[dst] = in_rp[src + 1] - in_rp[src];
    }
}

// scale_mode 1: value * row_scale[source row] (row_scale_permute); 2: value / (row_scale[new row] *
// col_scale[new column]), absent factors left out (the inv_*_scale_permute family)
template <typename T, typename I>
__global__ __launch_bounds__(256) void csr_permute_fill_kernel(
    int64_t n_rows, const I* __restrict__ in_rp, const I* __restrict__ in_ci, const T* __restrict__ in_v,
    const I* __restrict__ row_perm, bool row_inverse, const I* __restrict__ col_perm,
    const T* __restrict__ row_scale, const T* __restrict__ col_scale, int scale_mode,
    const I* __restrict__ out_rp, I* __restrict__ out_ci, T* __restrict__ out_v)
{
    GKOC_FOR_EACH(r, n_rows)
    {
        const int64_t src = (row_perm && !row_inverse) ? int64_t(row_perm[r]) : r;
        const int64_t dst = (row_perm && row_inverse) ? int64_t(row_perm[r]) : r;
        const int64_t sb = in_rp[src], len = in_rp[src + 1] - sb, db = out_rp[dst];
        for (int64_t k = 0; k < len; ++k) {
            const I c = col_perm ? col_perm[in_ci[sb + k]] : in_ci[sb + k];
            T v = in_v[sb + k];
            if (scale_mode == 1) {
                v = v * row_scale[src];
            } else if (scale_mode == 2) {
                if (row_scale && col_scale) {
                    v = v / (row_scale[dst] * col_scale[c]);
                } else {
                    v = v / (row_scale ? row_scale[dst] : col_scale[c]);
                }
            }
            out_ci[db + k] = c;
            out_v[db + k] = v;
        }
    }
}

// permutation::invert / compose, scaled_permutation::invert / compose
// (reference/matrix/permutation_kernels.cpp, scaled_permutation_kernels.cpp)
template <typename T, typename I>
__global__ __launch_bounds__(256) void permutation_invert_kernel(int64_t n, const T* __restrict__ in_scale,
                                                                 const I* __restrict__ perm,
                                                                 T* __restrict__ out_scale,
                                                                 I* __restrict__ out_perm)
{
    GKOC_FOR_EACH(i, n)
    {
        const I ip = perm[i];
        out_perm[ip] = I(i);
        if (in_scale) out_scale[i] = T(1) / in_scale[ip];
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void permutation_compose_kernel(
    int64_t n, const T* __restrict__ first_scale, const I* __restrict__ first,
    const T* __restrict__ second_scale, const I* __restrict__ second, T* __restrict__ out_scale,
    I* __restrict__ out_perm)
{
    GKOC_FOR_EACH(i, n)
    {
        const I sp = second[i];
        const I cp = first[sp];
        out_perm[i] = cp;
        if (first_scale) out_scale[cp] = first_scale[cp] * second_scale[sp];
    }
}

// csr::calculate_nonzeros_per_row_in_span / compute_submatrix (:1468-1530): the entries of rows
// [row0, row0 + n) with columns in [col0, col1), in storage order, columns shifted by col0
template <typename T, typename I, bool FILL>
__global__ __launch_bounds__(256) void csr_span_kernel(int64_t n, int64_t row0, int64_t col0, int64_t col1,
                                                       const I* __restrict__ in_rp,
                                                       const I* __restrict__ in_ci,
                                                       const T* __restrict__ in_v, I* __restrict__ counts,
                                                       const I* __restrict__ out_rp,
                                                       I* __restrict__ out_ci, T* __restrict__ out_v)
{
    GKOC_FOR_EACH(r, n)
    {
        int64_t at = FILL ? int64_t(out_rp[r]) : 0;
        for (int64_t k = in_rp[row0 + r]; k < in_rp[row0 + r + 1]; ++k) {
            const int64_t c = in_ci[k];
            if (c >= col0 && c < col1) {
                if (FILL) {
                    out_ci[at] = I(c - col0);
                    out_v[at] = in_v[k];
                }
                ++at;
            }
        }
        if (!FILL) counts[r] = I(at);
    }
}


// csr::calculate_nonzeros_per_row_in_index_set / compute_submatrix_from_index_set
// (reference/matrix/csr_kernels.cpp:772-812, 853-904): result row t is the t-th row of the row index
// set (subset j holds the rows [row_begin[j], row_end[j]), its first result row is row_superset[j]); an
// entry is kept when its column lies in a subset of the column index set and gets the column
// col_superset[subset] + (column - col_begin[subset]).  One lane per result row, two binary searches.
template <typename I>
__device__ __forceinline__ int64_t last_not_above(const I* __restrict__ a, int64_t n, int64_t v)
{
    // std::upper_bound(a, a + n, v) - 1, clamped at 0 (the reference's shifted_bucket)
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (int64_t(a[mid]) <= v) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    return lo == 0 ? 0 : lo - 1;
}

template <typename T, typename I, bool FILL>
__global__ __launch_bounds__(256) void csr_index_set_kernel(
    int64_t n_result_rows, int64_t n_row_subsets, const I* __restrict__ row_begin,
    const I* __restrict__ row_superset, int64_t n_col_subsets, const I* __restrict__ col_begin,
    const I* __restrict__ col_end, const I* __restrict__ col_superset, int64_t col_set_size,
    const I* __restrict__ in_rp, const I* __restrict__ in_ci, const T* __restrict__ in_v, I* __restrict__ counts,
    const I* __restrict__ out_rp, I* __restrict__ out_ci, T* __restrict__ out_v)
{
    GKOC_FOR_EACH(t, n_result_rows)
    {
        const int64_t set = last_not_above(row_superset, n_row_subsets, t);
        const int64_t row = int64_t(row_begin[set]) + (t - int64_t(row_superset[set]));
        int64_t at = FILL ? int64_t(out_rp[t]) : 0;
        for (int64_t k = in_rp[row]; k < in_rp[row + 1]; ++k) {
            const int64_t c = in_ci[k];
            if (c >= col_set_size) continue;
            const int64_t b = last_not_above(col_begin, n_col_subsets, c);
            if (int64_t(col_end[b]) <= c || c < int64_t(col_begin[b])) continue;
            if (FILL) {
                out_ci[at] = I(c - int64_t(col_begin[b]) + int64_t(col_superset[b]));
                out_v[at] = in_v[k];
            }
            ++at;
        }
        if (!FILL) counts[t] = I(at);
    }
}


// ------------------------------------------------------- SpGEMM / SpGEAM as triplets
// C = alpha A B + beta D (csr::spgemm / advanced_spgemm, reference/matrix/csr_kernels.cpp:156-300)
// and C = alpha A + beta B (csr::spgeam, :425-468) accumulate a row's contributions in a map by
// column: value = 0 + contributions in the order they are met (D's entries first, then the products
// a_ik b_kj in storage order), pattern = union, columns ascending.  Here: every contribution of row
// i is written as a triplet (i, column, value) at offsets[i] ..., then the library's own
// sort_row_major (stable) and sum_duplicates (assembly.hip: 0 + v0 + v1 + ... in storage order) give
// exactly that.  B == NULL (b_rp): A's entries themselves are the contributions (SpGEAM).
template <typename I>
__global__ __launch_bounds__(256) void spgemm_count_kernel(int64_t n_rows, const I* __restrict__ a_rp,
                                                           const I* __restrict__ a_ci,
                                                           const I* __restrict__ b_rp,
                                                           const I* __restrict__ d_rp,
                                                           int64_t* __restrict__ offsets)
{
    GKOC_FOR_EACH(r, n_rows + 1)
    {
        int64_t c = 0;
        if (r < n_rows) {
            if (d_rp) c += d_rp[r + 1] - d_rp[r];
            if (b_rp) {
                for (int64_t k = a_rp[r]; k < a_rp[r + 1]; ++k) c += b_rp[a_ci[k] + 1] - b_rp[a_ci[k]];
            } else {
                c += a_rp[r + 1] - a_rp[r];
            }
        }
        offsets[r] = c;
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void spgemm_expand_kernel(
    int64_t n_rows, const T* __restrict__ alpha_p, const I* __restrict__ a_rp, const I* __restrict__ a_ci,
    const T* __restrict__ a_v, const I* __restrict__ b_rp, const I* __restrict__ b_ci,
    const T* __restrict__ b_v, const T* __restrict__ beta_p, const I* __restrict__ d_rp,
    const I* __restrict__ d_ci, const T* __restrict__ d_v, const int64_t* __restrict__ offsets,
    I* __restrict__ t_rows, I* __restrict__ t_cols, T* __restrict__ t_vals)
{
    const T alpha = alpha_p ? alpha_p[0] : T(1);
    const T beta = beta_p ? beta_p[0] : T(1);
    GKOC_FOR_EACH(r, n_rows)
    {
        int64_t at = offsets[r];
        if (d_rp) {
            for (int64_t k = d_rp[r]; k < d_rp[r + 1]; ++k, ++at) {
                t_rows[at] = I(r);
                t_cols[at] = d_ci[k];
                t_vals[at] = beta * d_v[k];
            }
        }
        for (int64_t k = a_rp[r]; k < a_rp[r + 1]; ++k) {
            const T sa = alpha * a_v[k];
            if (!b_rp) {
                t_rows[at] = I(r);
                t_cols[at] = a_ci[k];
                t_vals[at] = sa;
                ++at;
                continue;
            }
            const int64_t br = a_ci[k];
            for (int64_t l = b_rp[br]; l < b_rp[br + 1]; ++l, ++at) {
                t_rows[at] = I(r);
                t_cols[at] = b_ci[l];
                t_vals[at] = sa * b_v[l];
            }
        }
    }
}


// ------------------------------------------------------------ L1 block-Jacobi
// jacobi::scalar_l1 / block_l1 (reference/preconditioner/jacobi_kernels.cpp:728-780): the diagonal
// entry of every row grows by the sum of |a_ij| over the entries OUTSIDE its diagonal block (scalar:
// j != i), added in storage order.  factorization::add_diagonal_elements
// (reference/factorization/factorization_kernels.cpp:55-128) makes sure every row has one: a missing
// diagonal is inserted as an explicit zero before the first larger column (at the end of the row if
// there is none).
template <typename T, typename I>
__global__ __launch_bounds__(256) void jacobi_scalar_l1_kernel(int64_t n_rows, const I* __restrict__ rp,
                                                               const I* __restrict__ ci,
                                                               const T* __restrict__ v, T* __restrict__ diag)
{
    GKOC_FOR_EACH(r, n_rows)
    {
        real_t<T> off = 0;
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
            if (int64_t(ci[k]) != r) off += abs_v(v[k]);
        }
        diag[r] += T(off);
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void jacobi_block_l1_kernel(int64_t num_blocks,
                                                              const I* __restrict__ block_ptrs,
                                                              const I* __restrict__ rp,
                                                              const I* __restrict__ ci, T* __restrict__ v)
{
    GKOC_FOR_EACH(b, num_blocks)
    {
        const int64_t start = block_ptrs[b], end = block_ptrs[b + 1];
        for (int64_t r = start; r < end; ++r) {
            real_t<T> off = 0;
            int64_t diag_at = -1;
            for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
                const int64_t c = ci[k];
                if (c >= start && c < end) {
                    if (c == r) diag_at = k;
                    continue;
                }
                off += abs_v(v[k]);
            }
            if (diag_at >= 0) v[diag_at] += T(off);
        }
    }
}

template <typename I>
__global__ __launch_bounds__(256) void missing_diagonal_count_kernel(int64_t n_rows, int64_t n_cols,
                                                                     const I* __restrict__ rp,
                                                                     const I* __restrict__ ci,
                                                                     I* __restrict__ shift)
{
    GKOC_FOR_EACH(r, n_rows + 1)
    {
        I miss = 0;
        if (r < n_rows && r < n_cols) {
            miss = 1;
            for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
                if (int64_t(ci[k]) == r) miss = 0;
            }
        }
        shift[r] = miss;
    }
}

// shift = exclusive sums of the per-row counts: row r moves to old start + shift[r]
template <typename T, typename I>
__global__ __launch_bounds__(256) void add_diagonal_fill_kernel(int64_t n_rows, const I* __restrict__ rp,
                                                                const I* __restrict__ ci,
                                                                const T* __restrict__ v,
                                                                const I* __restrict__ shift,
                                                                I* __restrict__ new_rp,
                                                                I* __restrict__ new_ci,
                                                                T* __restrict__ new_v)
{
    GKOC_FOR_EACH(r, n_rows + 1)
    {
        new_rp[r] = rp[r] + shift[r];
        if (r == n_rows) continue;
        const bool missing = shift[r + 1] != shift[r];
        int64_t at = int64_t(rp[r]) + shift[r];
        bool placed = !missing;
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
            if (!placed && int64_t(ci[k]) > r) {
                new_ci[at] = I(r);
                new_v[at] = T(0);
                ++at;
                placed = true;
            }
            new_ci[at] = ci[k];
            new_v[at] = v[k];
            ++at;
        }
        if (!placed) {
            new_ci[at] = I(r);
            new_v[at] = T(0);
        }
    }
}


#define CV_LAUNCH(kernel, n, ...)                                                     \
    do {                                                                              \
        if ((n) > 0) {                                                                \
            kernel<<<dim3(cv_grid(n)), dim3(256), 0, as_stream(s)>>>(__VA_ARGS__);    \
            GKOC_LAUNCH_OK();                                                         \
        }                                                                             \
    } while (0)

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_FILL_SEQ(T, TN)                                                  \
    extern "C" int gkoc_fill_seq_array_##TN(gkoc_stream_t s, T* data, int64_t n)  \
    {                                                                             \
        GKOC_REQUIRE(n >= 0 && (n == 0 || data), GKOC_E_INVALID, "bad argument"); \
        CV_LAUNCH(fill_seq_kernel<T>, n, n, data);                                \
        return GKOC_OK;                                                           \
    }
GKOC_DEF_FILL_SEQ(double, f64)
GKOC_DEF_FILL_SEQ(float, f32)
GKOC_DEF_FILL_SEQ(uint64_t, u64)

#define GKOC_DEF_CV_DENSE(T, TN)                                                                       \
    extern "C" int gkoc_dense_transpose_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, const T* in, \
                                             int64_t ldi, T* out, int64_t ldo)                         \
    {                                                                                                  \
        GKOC_REQUIRE(rows >= 0 && cols >= 0, GKOC_E_INVALID, "negative dimension");                    \
        GKOC_REQUIRE(rows * cols == 0 || (in && out && in != out), GKOC_E_INVALID, "bad operand");     \
        CV_LAUNCH(dense_transpose_kernel<T>, rows* cols, rows, cols, in, ldi, out, ldo);               \
        return GKOC_OK;                                                                                \
    }                                                                                                  \
    extern "C" int gkoc_dense_extract_diagonal_##TN(gkoc_stream_t s, int64_t n, const T* in,           \
                                                    int64_t ld, T* diag)                               \
    {                                                                                                  \
        GKOC_REQUIRE(n >= 0 && (n == 0 || (in && diag)), GKOC_E_INVALID, "bad argument");              \
        CV_LAUNCH(dense_extract_diagonal_kernel<T>, n, n, in, ld, diag);                               \
        return GKOC_OK;                                                                                \
    }                                                                                                  \
    extern "C" int gkoc_dense_add_scaled_identity_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,    \
                                                       const T* alpha, const T* beta, T* m,            \
                                                       int64_t ld)                                     \
    {                                                                                                  \
        GKOC_REQUIRE(rows >= 0 && cols >= 0 && alpha && beta, GKOC_E_INVALID, "bad argument");         \
        CV_LAUNCH(dense_add_scaled_identity_kernel<T>, rows* cols, rows, cols, alpha, beta, m, ld);    \
        return GKOC_OK;                                                                                \
    }                                                                                                  \
    extern "C" int gkoc_dense_add_scaled_diag_##TN(gkoc_stream_t s, int64_t n, const T* alpha,         \
                                                   const T* diag, T* y, int64_t ldy, int subtract)     \
    {                                                                                                  \
        GKOC_REQUIRE(n >= 0 && alpha && (n == 0 || (diag && y)), GKOC_E_INVALID, "bad argument");      \
        if (subtract) {                                                                                \
            CV_LAUNCH((dense_scaled_diag_kernel<T, true>), n, n, alpha, diag, y, ldy);                 \
        } else {                                                                                       \
            CV_LAUNCH((dense_scaled_diag_kernel<T, false>), n, n, alpha, diag, y, ldy);                \
        }                                                                                              \
        return GKOC_OK;                                                                                \
    }                                                                                                  \
    extern "C" int gkoc_dense_count_nonzeros_per_row_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, \
                                                          const T* in, int64_t ld, void* out,          \
                                                          int out_bytes)                               \
    {                                                                                                  \
        GKOC_REQUIRE(rows >= 0 && cols >= 0 && (out_bytes == 4 || out_bytes == 8), GKOC_E_INVALID,     \
                     "bad argument");                                                                  \
        if (out_bytes == 4) {                                                                          \
            CV_LAUNCH((dense_count_nnz_kernel<T, int32_t>), rows, rows, cols, in, ld,                  \
                      static_cast<int32_t*>(out));                                                     \
        } else {                                                                                       \
            CV_LAUNCH((dense_count_nnz_kernel<T, int64_t>), rows, rows, cols, in, ld,                  \
                      static_cast<int64_t*>(out));                                                     \
        }                                                                                              \
        return GKOC_OK;                                                                                \
    }                                                                                                  \
    extern "C" int gkoc_dense_max_nnz_per_row_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,        \
                                                   const T* in, int64_t ld, uint64_t* result_host)     \
    {                                                                                                  \
        GKOC_REQUIRE(rows >= 0 && cols >= 0 && result_host, GKOC_E_INVALID, "bad argument");           \
        *result_host = 0;                                                                              \
        if (rows == 0 || cols == 0) return GKOC_OK;                                                    \
        void* d = nullptr;                                                                             \
        GKOC_TRY(scratch_malloc(as_stream(s), &d, 8));                                                 \
        GKOC_HIP(hipMemsetAsync(d, 0, 8, as_stream(s)));                                               \
        CV_LAUNCH(dense_max_nnz_kernel<T>, rows, rows, cols, in, ld,                                   \
                  static_cast<unsigned long long*>(d));                                                \
        GKOC_HIP(hipMemcpyAsync(result_host, d, 8, hipMemcpyDeviceToHost, as_stream(s)));              \
        GKOC_HIP(hipStreamSynchronize(as_stream(s)));                                                  \
        return scratch_free(as_stream(s), d);                                                          \
    }                                                                                                  \
    extern "C" int gkoc_dense_compute_slice_sets_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,     \
                                                      const T* in, int64_t ld, int64_t slice_size,     \
                                                      int64_t stride_factor, uint64_t* slice_sets,     \
                                                      uint64_t* slice_lengths)                         \
    {                                                                                                  \
        GKOC_REQUIRE(rows >= 0 && cols >= 0 && slice_size > 0 && stride_factor > 0 && slice_sets,      \
                     GKOC_E_INVALID, "bad argument");                                                  \
        const int64_t ns = ceildiv(rows, slice_size);                                                  \
        GKOC_REQUIRE(ns == 0 || slice_lengths, GKOC_E_INVALID, "null slice_lengths");                  \
        CV_LAUNCH(dense_slice_lengths_kernel<T>, ns + 1, rows, cols, in, ld, slice_size,               \
                  stride_factor, ns, slice_lengths, slice_sets);                                       \
        return device_exclusive_scan<uint64_t>(as_stream(s), slice_sets, ns + 1);                      \
    }
GKOC_DEF_CV_DENSE(double, f64)
GKOC_DEF_CV_DENSE(float, f32)
GKOC_DEF_CV_DENSE(gkoc_c128, c128)
GKOC_DEF_CV_DENSE(gkoc_c64, c64)

#define GKOC_DEF_CV(T, TN, I, IN)                                                                       \
    extern "C" int gkoc_dense_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols,           \
                                                 const T* in, int64_t ld, const I* row_ptrs,            \
                                                 I* out_cols, T* out_vals)                              \
    {                                                                                                   \
        GKOC_REQUIRE(rows >= 0 && cols >= 0 && (rows == 0 || row_ptrs), GKOC_E_INVALID, "bad argument"); \
        CV_LAUNCH((dense_to_rows_kernel<T, I, I>), rows, rows, cols, in, ld, row_ptrs,                  \
                  static_cast<I*>(nullptr), out_cols, out_vals);                                        \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_dense_to_coo_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols,           \
                                                 const T* in, int64_t ld, const int64_t* row_ptrs,      \
                                                 I* out_rows, I* out_cols, T* out_vals)                 \
    {                                                                                                   \
        GKOC_REQUIRE(rows >= 0 && cols >= 0 && (rows == 0 || row_ptrs), GKOC_E_INVALID, "bad argument"); \
        CV_LAUNCH((dense_to_rows_kernel<T, I, int64_t>), rows, rows, cols, in, ld, row_ptrs, out_rows,  \
                  out_cols, out_vals);                                                                  \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_dense_to_ell_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols,           \
                                                 const T* in, int64_t ld, int64_t ell_k,                \
                                                 int64_t ell_lim, int64_t stride, I* ell_cols,          \
                                                 T* ell_vals, const int64_t* coo_row_ptrs,              \
                                                 I* coo_rows, I* coo_cols, T* coo_vals)                 \
    {                                                                                                   \
        GKOC_REQUIRE(rows >= 0 && cols >= 0 && ell_k >= 0 && stride >= rows, GKOC_E_INVALID,            \
                     "bad argument");                                                                   \
        CV_LAUNCH((dense_to_ell_kernel<T, I>), stride, rows, cols, in, ld, ell_k, ell_lim, stride,      \
                  ell_cols, ell_vals, coo_row_ptrs, coo_rows, coo_cols, coo_vals);                      \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_dense_to_sellp_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols,         \
                                                   const T* in, int64_t ld, int64_t slice_size,         \
                                                   const uint64_t* slice_sets, I* out_cols,             \
                                                   T* out_vals)                                         \
    {                                                                                                   \
        GKOC_REQUIRE(rows >= 0 && cols >= 0 && slice_size > 0, GKOC_E_INVALID, "bad argument");         \
        CV_LAUNCH((dense_to_sellp_kernel<T, I>), rows, rows, cols, in, ld, slice_size, slice_sets,      \
                  out_cols, out_vals);                                                                  \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_csr_fill_in_dense_##TN##_##IN(gkoc_stream_t s, int64_t n_rows,                  \
                                                      const I* row_ptrs, const I* cols,                 \
                                                      const T* vals, T* out, int64_t ld)                \
    {                                                                                                   \
        CV_LAUNCH((csr_fill_in_dense_kernel<T, I>), n_rows, n_rows, row_ptrs, cols, vals, out, ld);     \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_coo_fill_in_dense_##TN##_##IN(gkoc_stream_t s, int64_t nnz, const I* rows,      \
                                                      const I* cols, const T* vals, T* out,             \
                                                      int64_t ld)                                       \
    {                                                                                                   \
        CV_LAUNCH((coo_fill_in_dense_kernel<T, I>), nnz, nnz, rows, cols, vals, out, ld);               \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_ell_fill_in_dense_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t ell_k,   \
                                                      int64_t stride, const I* cols, const T* vals,     \
                                                      T* out, int64_t ld)                               \
    {                                                                                                   \
        CV_LAUNCH((ell_fill_in_dense_kernel<T, I>), n_rows, n_rows, ell_k, stride, cols, vals, out, ld); \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_sellp_fill_in_dense_##TN##_##IN(gkoc_stream_t s, int64_t n_rows,                \
                                                        int64_t slice_size,                             \
                                                        const uint64_t* slice_sets, const I* cols,      \
                                                        const T* vals, T* out, int64_t ld)              \
    {                                                                                                   \
        GKOC_REQUIRE(slice_size > 0, GKOC_E_INVALID, "bad slice size");                                 \
        CV_LAUNCH((sellp_fill_in_dense_kernel<T, I>), n_rows, n_rows, slice_size, slice_sets, cols,     \
                  vals, out, ld);                                                                       \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_ell_extract_diagonal_##TN##_##IN(gkoc_stream_t s, int64_t n, int64_t ell_k,     \
                                                         int64_t stride, const I* cols,                 \
                                                         const T* vals, T* diag)                        \
    {                                                                                                   \
        CV_LAUNCH((ell_extract_diagonal_kernel<T, I>), n, n, ell_k, stride, cols, vals, diag);          \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_sellp_extract_diagonal_##TN##_##IN(gkoc_stream_t s, int64_t n,                  \
                                                           int64_t slice_size,                          \
                                                           const uint64_t* slice_sets, const I* cols,   \
                                                           const T* vals, T* diag)                      \
    {                                                                                                   \
        GKOC_REQUIRE(slice_size > 0, GKOC_E_INVALID, "bad slice size");                                 \
        CV_LAUNCH((sellp_extract_diagonal_kernel<T, I>), n, n, slice_size, slice_sets, cols, vals,      \
                  diag);                                                                                \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_coo_extract_diagonal_##TN##_##IN(gkoc_stream_t s, int64_t nnz, const I* rows,   \
                                                         const I* cols, const T* vals, T* diag)         \
    {                                                                                                   \
        CV_LAUNCH((coo_extract_diagonal_kernel<T, I>), nnz, nnz, rows, cols, vals, diag);               \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_csr_add_scaled_identity_##TN##_##IN(gkoc_stream_t s, int64_t n_rows,            \
                                                            const I* row_ptrs, const I* cols, T* vals,  \
                                                            const T* alpha, const T* beta)              \
    {                                                                                                   \
        GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null scalar");                                     \
        CV_LAUNCH((csr_add_scaled_identity_kernel<T, I>), n_rows, n_rows, row_ptrs, cols, vals, alpha,  \
                  beta);                                                                                \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_ell_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t ell_k,          \
                                               int64_t stride, const I* cols, const T* vals,            \
                                               const I* row_ptrs, I* out_cols, T* out_vals)             \
    {                                                                                                   \
        CV_LAUNCH((ell_to_csr_kernel<T, I>), n_rows, n_rows, ell_k, stride, cols, vals, row_ptrs,       \
                  out_cols, out_vals);                                                                  \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_sellp_to_csr_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, int64_t slice_size,   \
                                                 const uint64_t* slice_sets, const I* cols,             \
                                                 const T* vals, const I* row_ptrs, I* out_cols,         \
                                                 T* out_vals)                                           \
    {                                                                                                   \
        GKOC_REQUIRE(slice_size > 0, GKOC_E_INVALID, "bad slice size");                                 \
        CV_LAUNCH((sellp_to_csr_kernel<T, I>), n_rows, n_rows, slice_size, slice_sets, cols, vals,      \
                  row_ptrs, out_cols, out_vals);                                                        \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_hybrid_to_csr_##TN##_##IN(                                                      \
        gkoc_stream_t s, int64_t n_rows, int64_t ell_k, int64_t stride, const I* ell_cols,              \
        const T* ell_vals, const I* coo_cols, const T* coo_vals, const I* ell_row_ptrs,                 \
        const I* coo_row_ptrs, I* out_row_ptrs, I* out_cols, T* out_vals)                               \
    {                                                                                                   \
        GKOC_REQUIRE(n_rows >= 0 && ell_row_ptrs && coo_row_ptrs && out_row_ptrs, GKOC_E_INVALID,       \
                     "bad argument");                                                                   \
        CV_LAUNCH((hybrid_to_csr_kernel<T, I>), n_rows + 1, n_rows, ell_k, stride, ell_cols, ell_vals,  \
                  coo_cols, coo_vals, ell_row_ptrs, coo_row_ptrs, out_row_ptrs, out_cols, out_vals);    \
        return GKOC_OK;                                                                                 \
    }
GKOC_DEF_CV(double, f64, int32_t, i32)
GKOC_DEF_CV(double, f64, int64_t, i64)
GKOC_DEF_CV(float, f32, int32_t, i32)
GKOC_DEF_CV(float, f32, int64_t, i64)
GKOC_DEF_CV(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_CV(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_CV(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_CV(gkoc_c64, c64, int64_t, i64)

#define GKOC_DEF_CV_INDEX(I, IN)                                                                        \
    extern "C" int gkoc_ell_count_nonzeros_per_row_##IN(gkoc_stream_t s, int64_t n_rows,                \
                                                        int64_t ell_k, int64_t stride, const I* cols,   \
                                                        I* out)                                         \
    {                                                                                                   \
        CV_LAUNCH(ell_count_kernel<I>, n_rows, n_rows, ell_k, stride, cols, out);                       \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_sellp_count_nonzeros_per_row_##IN(gkoc_stream_t s, int64_t n_rows,              \
                                                          int64_t slice_size,                           \
                                                          const uint64_t* slice_sets, const I* cols,    \
                                                          I* out)                                       \
    {                                                                                                   \
        GKOC_REQUIRE(slice_size > 0, GKOC_E_INVALID, "bad slice size");                                 \
        CV_LAUNCH(sellp_count_kernel<I>, n_rows, n_rows, slice_size, slice_sets, cols, out);            \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    /* *missing_host = 1 if some row r < n has no entry (r, r) */                                       \
    extern "C" int gkoc_csr_missing_diagonal_##IN(gkoc_stream_t s, int64_t n, const I* row_ptrs,        \
                                                  const I* cols, int* missing_host)                     \
    {                                                                                                   \
        GKOC_REQUIRE(n >= 0 && missing_host, GKOC_E_INVALID, "bad argument");                           \
        *missing_host = 0;                                                                              \
        if (n == 0) return GKOC_OK;                                                                     \
        void* d = nullptr;                                                                              \
        GKOC_TRY(scratch_malloc(as_stream(s), &d, 4));                                                  \
        GKOC_HIP(hipMemsetAsync(d, 0, 4, as_stream(s)));                                                \
        CV_LAUNCH(csr_missing_diagonal_kernel<I>, n, n, row_ptrs, cols, static_cast<int*>(d));          \
        GKOC_HIP(hipMemcpyAsync(missing_host, d, 4, hipMemcpyDeviceToHost, as_stream(s)));              \
        GKOC_HIP(hipStreamSynchronize(as_stream(s)));                                                   \
        return scratch_free(as_stream(s), d);                                                           \
    }
GKOC_DEF_CV_INDEX(int32_t, i32)
GKOC_DEF_CV_INDEX(int64_t, i64)

#define GKOC_DEF_PERMUTE(T, TN, I, IN)                                                                  \
    extern "C" int gkoc_dense_permute_##TN##_##IN(gkoc_stream_t s, int64_t rows, int64_t cols,          \
                                                  const T* in, int64_t ldi, T* out, int64_t ldo,        \
                                                  const I* row_perm, const I* col_perm,                 \
                                                  const T* row_scale, const T* col_scale, int inverse)  \
    {                                                                                                   \
        GKOC_REQUIRE(rows >= 0 && cols >= 0 && (rows * cols == 0 || (in && out && in != out)),          \
                     GKOC_E_INVALID, "bad argument");                                                   \
        GKOC_REQUIRE((!row_scale || row_perm) && (!col_scale || col_perm), GKOC_E_INVALID,              \
                     "a scale needs its permutation");                                                  \
        if (inverse) {                                                                                  \
            CV_LAUNCH((dense_permute_kernel<T, I, true>), rows* cols, rows, cols, in, ldi, out, ldo,    \
                      row_perm, col_perm, row_scale, col_scale);                                        \
        } else {                                                                                        \
            CV_LAUNCH((dense_permute_kernel<T, I, false>), rows* cols, rows, cols, in, ldi, out, ldo,   \
                      row_perm, col_perm, row_scale, col_scale);                                        \
        }                                                                                               \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_dense_advanced_row_gather_##TN##_##IN(                                          \
        gkoc_stream_t s, int64_t n_gather, int64_t cols, const T* alpha, const I* rows_idx,             \
        const T* in, int64_t ldi, const T* beta, T* out, int64_t ldo)                                   \
    {                                                                                                   \
        GKOC_REQUIRE(n_gather >= 0 && cols >= 0 && alpha && beta, GKOC_E_INVALID, "bad argument");      \
        CV_LAUNCH((dense_advanced_row_gather_kernel<T, I>), n_gather* cols, n_gather, cols, alpha,      \
                  rows_idx, in, ldi, beta, out, ldo);                                                   \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_csr_permute_##TN##_##IN(                                                        \
        gkoc_stream_t s, int64_t n_rows, const I* in_rp, const I* in_ci, const T* in_v,                 \
        const I* row_perm, int row_inverse, const I* col_perm, const T* row_scale, const T* col_scale,  \
        int scale_mode, I* out_rp, I* out_ci, T* out_v)                                                 \
    {                                                                                                   \
        GKOC_REQUIRE(n_rows >= 0 && in_rp && out_rp, GKOC_E_INVALID, "bad argument");                   \
        GKOC_REQUIRE(scale_mode >= 0 && scale_mode <= 2 && (scale_mode != 1 || row_scale) &&            \
                         (scale_mode != 2 || row_scale || col_scale),                                   \
                     GKOC_E_INVALID, "bad scale arguments");                                            \
        CV_LAUNCH(csr_permute_lengths_kernel<I>, n_rows + 1, n_rows, in_rp, row_perm, row_inverse != 0, \
                  out_rp);                                                                              \
        GKOC_TRY(device_exclusive_scan<I>(as_stream(s), out_rp, n_rows + 1));                           \
        CV_LAUNCH((csr_permute_fill_kernel<T, I>), n_rows, n_rows, in_rp, in_ci, in_v, row_perm,        \
                  row_inverse != 0, col_perm, row_scale, col_scale, scale_mode, out_rp, out_ci, out_v); \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_scaled_permutation_invert_##TN##_##IN(gkoc_stream_t s, int64_t n,               \
                                                              const T* in_scale, const I* perm,         \
                                                              T* out_scale, I* out_perm)                \
    {                                                                                                   \
        GKOC_REQUIRE(n >= 0 && (n == 0 || (in_scale && perm && out_scale && out_perm)), GKOC_E_INVALID, \
                     "bad argument");                                                                   \
        CV_LAUNCH((permutation_invert_kernel<T, I>), n, n, in_scale, perm, out_scale, out_perm);        \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_scaled_permutation_compose_##TN##_##IN(                                         \
        gkoc_stream_t s, int64_t n, const T* first_scale, const I* first, const T* second_scale,        \
        const I* second, T* out_scale, I* out_perm)                                                     \
    {                                                                                                   \
        GKOC_REQUIRE(n >= 0 && (n == 0 || (first_scale && first && second_scale && second &&            \
                                           out_scale && out_perm)),                                     \
                     GKOC_E_INVALID, "bad argument");                                                   \
        CV_LAUNCH((permutation_compose_kernel<T, I>), n, n, first_scale, first, second_scale, second,   \
                  out_scale, out_perm);                                                                 \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_csr_count_in_span_##TN##_##IN(gkoc_stream_t s, int64_t n, int64_t row0,         \
                                                      int64_t col0, int64_t col1, const I* in_rp,       \
                                                      const I* in_ci, I* counts)                        \
    {                                                                                                   \
        CV_LAUNCH((csr_span_kernel<T, I, false>), n, n, row0, col0, col1, in_rp, in_ci,                 \
                  static_cast<const T*>(nullptr), counts, static_cast<const I*>(nullptr),               \
                  static_cast<I*>(nullptr), static_cast<T*>(nullptr));                                  \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_csr_submatrix_##TN##_##IN(gkoc_stream_t s, int64_t n, int64_t row0,             \
                                                  int64_t col0, int64_t col1, const I* in_rp,           \
                                                  const I* in_ci, const T* in_v, const I* out_rp,       \
                                                  I* out_ci, T* out_v)                                  \
    {                                                                                                   \
        CV_LAUNCH((csr_span_kernel<T, I, true>), n, n, row0, col0, col1, in_rp, in_ci, in_v,            \
                  static_cast<I*>(nullptr), out_rp, out_ci, out_v);                                     \
        return GKOC_OK;                                                                                 \
    }
#define GKOC_DEF_INDEX_SET(T, TN, I, IN)                                                                \
    extern "C" int gkoc_csr_count_in_index_set_##TN##_##IN(                                             \
        gkoc_stream_t s, int64_t n_result_rows, int64_t n_row_subsets, const I* row_begin,              \
        const I* row_superset, int64_t n_col_subsets, const I* col_begin, const I* col_end,             \
        int64_t col_set_size, const I* in_rp, const I* in_ci, I* counts)                                \
    {                                                                                                   \
        GKOC_REQUIRE(n_result_rows >= 0 && n_row_subsets >= 0 && n_col_subsets >= 0, GKOC_E_INVALID,    \
                     "negative size");                                                                  \
        GKOC_REQUIRE(n_result_rows == 0 || (n_row_subsets > 0 && n_col_subsets > 0 && row_begin &&      \
                                            row_superset && col_begin && col_end && in_rp && counts),   \
                     GKOC_E_INVALID, "empty index set or null pointer");                                \
        CV_LAUNCH((csr_index_set_kernel<T, I, false>), n_result_rows, n_result_rows, n_row_subsets,     \
                  row_begin, row_superset, n_col_subsets, col_begin, col_end,                           \
                  static_cast<const I*>(nullptr), col_set_size, in_rp, in_ci,                           \
                  static_cast<const T*>(nullptr), counts, static_cast<const I*>(nullptr),               \
                  static_cast<I*>(nullptr), static_cast<T*>(nullptr));                                  \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_csr_submatrix_from_index_set_##TN##_##IN(                                       \
        gkoc_stream_t s, int64_t n_result_rows, int64_t n_row_subsets, const I* row_begin,              \
        const I* row_superset, int64_t n_col_subsets, const I* col_begin, const I* col_end,             \
        const I* col_superset, int64_t col_set_size, const I* in_rp, const I* in_ci, const T* in_v,     \
        const I* out_rp, I* out_ci, T* out_v)                                                           \
    {                                                                                                   \
        GKOC_REQUIRE(n_result_rows >= 0 && n_row_subsets >= 0 && n_col_subsets >= 0, GKOC_E_INVALID,    \
                     "negative size");                                                                  \
        GKOC_REQUIRE(n_result_rows == 0 || (n_row_subsets > 0 && n_col_subsets > 0 && row_begin &&      \
                                            row_superset && col_begin && col_end && col_superset &&     \
                                            in_rp && out_rp),                                           \
                     GKOC_E_INVALID, "empty index set or null pointer");                                \
        CV_LAUNCH((csr_index_set_kernel<T, I, true>), n_result_rows, n_result_rows, n_row_subsets,      \
                  row_begin, row_superset, n_col_subsets, col_begin, col_end, col_superset,             \
                  col_set_size, in_rp, in_ci, in_v, static_cast<I*>(nullptr), out_rp, out_ci, out_v);   \
        return GKOC_OK;                                                                                 \
    }
GKOC_DEF_INDEX_SET(double, f64, int32_t, i32)
GKOC_DEF_INDEX_SET(double, f64, int64_t, i64)
GKOC_DEF_INDEX_SET(float, f32, int32_t, i32)
GKOC_DEF_INDEX_SET(float, f32, int64_t, i64)
GKOC_DEF_INDEX_SET(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_INDEX_SET(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_INDEX_SET(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_INDEX_SET(gkoc_c64, c64, int64_t, i64)
GKOC_DEF_PERMUTE(double, f64, int32_t, i32)
GKOC_DEF_PERMUTE(double, f64, int64_t, i64)
GKOC_DEF_PERMUTE(float, f32, int32_t, i32)
GKOC_DEF_PERMUTE(float, f32, int64_t, i64)
GKOC_DEF_PERMUTE(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_PERMUTE(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_PERMUTE(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_PERMUTE(gkoc_c64, c64, int64_t, i64)

#define GKOC_DEF_PERMUTATION(I, IN)                                                                     \
    extern "C" int gkoc_permutation_invert_##IN(gkoc_stream_t s, int64_t n, const I* perm, I* out)      \
    {                                                                                                   \
        GKOC_REQUIRE(n >= 0 && (n == 0 || (perm && out)), GKOC_E_INVALID, "bad argument");              \
        CV_LAUNCH((permutation_invert_kernel<double, I>), n, n, static_cast<const double*>(nullptr),    \
                  perm, static_cast<double*>(nullptr), out);                                            \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_permutation_compose_##IN(gkoc_stream_t s, int64_t n, const I* first,            \
                                                 const I* second, I* out)                               \
    {                                                                                                   \
        GKOC_REQUIRE(n >= 0 && (n == 0 || (first && second && out)), GKOC_E_INVALID, "bad argument");   \
        CV_LAUNCH((permutation_compose_kernel<double, I>), n, n, static_cast<const double*>(nullptr),   \
                  first, static_cast<const double*>(nullptr), second, static_cast<double*>(nullptr),    \
                  out);                                                                                 \
        return GKOC_OK;                                                                                 \
    }
GKOC_DEF_PERMUTATION(int32_t, i32)
GKOC_DEF_PERMUTATION(int64_t, i64)

// offsets[r] = where row r's contributions start (n_rows + 1 entries, int64); *total_host = their number
#define GKOC_DEF_SPGEMM_COUNT(I, IN)                                                                    \
    extern "C" int gkoc_csr_spgemm_count_##IN(gkoc_stream_t s, int64_t n_rows, const I* a_rp,           \
                                              const I* a_ci, const I* b_rp, const I* d_rp,              \
                                              int64_t* offsets, int64_t* total_host)                    \
    {                                                                                                   \
        GKOC_REQUIRE(n_rows >= 0 && a_rp && offsets && total_host, GKOC_E_INVALID, "bad argument");     \
        CV_LAUNCH(spgemm_count_kernel<I>, n_rows + 1, n_rows, a_rp, a_ci, b_rp, d_rp, offsets);         \
        GKOC_TRY(device_exclusive_scan<int64_t>(as_stream(s), offsets, n_rows + 1));                    \
        GKOC_HIP(hipMemcpyAsync(total_host, offsets + n_rows, 8, hipMemcpyDeviceToHost, as_stream(s))); \
        GKOC_HIP(hipStreamSynchronize(as_stream(s)));                                                   \
        return GKOC_OK;                                                                                 \
    }
GKOC_DEF_SPGEMM_COUNT(int32_t, i32)
GKOC_DEF_SPGEMM_COUNT(int64_t, i64)

#define GKOC_DEF_SPGEMM(T, TN, I, IN)                                                                   \
    extern "C" int gkoc_csr_spgemm_expand_##TN##_##IN(                                                  \
        gkoc_stream_t s, int64_t n_rows, const T* alpha, const I* a_rp, const I* a_ci, const T* a_v,    \
        const I* b_rp, const I* b_ci, const T* b_v, const T* beta, const I* d_rp, const I* d_ci,        \
        const T* d_v, const int64_t* offsets, I* t_rows, I* t_cols, T* t_vals)                          \
    {                                                                                                   \
        GKOC_REQUIRE(n_rows >= 0 && a_rp && offsets, GKOC_E_INVALID, "bad argument");                   \
        CV_LAUNCH((spgemm_expand_kernel<T, I>), n_rows, n_rows, alpha, a_rp, a_ci, a_v, b_rp, b_ci,     \
                  b_v, beta, d_rp, d_ci, d_v, offsets, t_rows, t_cols, t_vals);                         \
        return GKOC_OK;                                                                                 \
    }
GKOC_DEF_SPGEMM(double, f64, int32_t, i32)
GKOC_DEF_SPGEMM(double, f64, int64_t, i64)
GKOC_DEF_SPGEMM(float, f32, int32_t, i32)
GKOC_DEF_SPGEMM(float, f32, int64_t, i64)
GKOC_DEF_SPGEMM(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_SPGEMM(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_SPGEMM(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_SPGEMM(gkoc_c64, c64, int64_t, i64)

#define GKOC_DEF_L1(T, TN, I, IN)                                                                       \
    extern "C" int gkoc_jacobi_scalar_l1_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* rp,      \
                                                     const I* ci, const T* v, T* diag)                  \
    {                                                                                                   \
        CV_LAUNCH((jacobi_scalar_l1_kernel<T, I>), n_rows, n_rows, rp, ci, v, diag);                    \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_jacobi_block_l1_##TN##_##IN(gkoc_stream_t s, int64_t num_blocks,                \
                                                    const I* block_ptrs, const I* rp, const I* ci,      \
                                                    T* v)                                               \
    {                                                                                                   \
        CV_LAUNCH((jacobi_block_l1_kernel<T, I>), num_blocks, num_blocks, block_ptrs, rp, ci, v);       \
        return GKOC_OK;                                                                                 \
    }                                                                                                   \
    extern "C" int gkoc_csr_add_diagonal_fill_##TN##_##IN(gkoc_stream_t s, int64_t n_rows,              \
                                                          const I* rp, const I* ci, const T* v,         \
                                                          const I* shift, I* new_rp, I* new_ci,         \
                                                          T* new_v)                                     \
    {                                                                                                   \
        GKOC_REQUIRE(n_rows >= 0 && rp && shift && new_rp, GKOC_E_INVALID, "bad argument");             \
        CV_LAUNCH((add_diagonal_fill_kernel<T, I>), n_rows + 1, n_rows, rp, ci, v, shift, new_rp,       \
                  new_ci, new_v);                                                                       \
        return GKOC_OK;                                                                                 \
    }
GKOC_DEF_L1(double, f64, int32_t, i32)
GKOC_DEF_L1(double, f64, int64_t, i64)
GKOC_DEF_L1(float, f32, int32_t, i32)
GKOC_DEF_L1(float, f32, int64_t, i64)
GKOC_DEF_L1(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_L1(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_L1(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_L1(gkoc_c64, c64, int64_t, i64)

// shift (n_rows + 1): exclusive sums of "row r lacks its diagonal entry"; *missing_host = their number
#define GKOC_DEF_MISSING(I, IN)                                                                         \
    extern "C" int gkoc_csr_missing_diagonal_shift_##IN(gkoc_stream_t s, int64_t n_rows,                \
                                                        int64_t n_cols, const I* rp, const I* ci,       \
                                                        I* shift, int64_t* missing_host)                \
    {                                                                                                   \
        GKOC_REQUIRE(n_rows >= 0 && rp && shift && missing_host, GKOC_E_INVALID, "bad argument");       \
        CV_LAUNCH(missing_diagonal_count_kernel<I>, n_rows + 1, n_rows, n_cols, rp, ci, shift);         \
        GKOC_TRY(device_exclusive_scan<I>(as_stream(s), shift, n_rows + 1));                            \
        I total = 0;                                                                                    \
        GKOC_HIP(hipMemcpyAsync(&total, shift + n_rows, sizeof(I), hipMemcpyDeviceToHost,               \
                                as_stream(s)));                                                         \
        GKOC_HIP(hipStreamSynchronize(as_stream(s)));                                                   \
        *missing_host = int64_t(total);                                                                 \
        return GKOC_OK;                                                                                 \
    }
GKOC_DEF_MISSING(int32_t, i32)
GKOC_DEF_MISSING(int64_t, i64)
