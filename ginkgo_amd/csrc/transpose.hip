// csr::transpose / conj_transpose for real value types:
//   decl core/matrix/csr_kernels.hpp (GKO_DECLARE_CSR_TRANSPOSE_KERNEL);
//   reference/matrix/csr_kernels.cpp:693-731 (count the columns, prefix sum,
//   convert_csr_to_csc: rows in ascending order, entries in storage order).
// The reference's result is the STABLE sort of the entries by column index, so the
// device version is exactly that: entry positions are sorted by column with a stable
// LSD radix sort, then values and row indices are gathered through the permutation.
// Index arrays and values are bit-identical to the reference.
// This is set-up plumbing, not a hot kernel: the radix sort is rocPRIM's
// (rocprim::radix_sort_pairs, deterministic and stable); everything around it is
// hand-written.  Workspace: 4 index arrays of nnz entries + the sort's scratch
// (gkoc_csr_transpose_workspace_bytes).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace gkoc {
namespace {

inline size_t align_up(size_t v) { return (v + 255) / 256 * 256; }

template <typename I>
__global__ __launch_bounds__(256) void iota_kernel(int64_t n, I* out)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) out[i] = I(i);
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void gather_transposed_kernel(
    int64_t nnz, const I* __restrict__ perm, const I* __restrict__ row_of,
    const T* __restrict__ vals, I* __restrict__ out_cols, T* __restrict__ out_vals)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < nnz; i += stride) {
        const int64_t k = perm[i];
        out_cols[i] = row_of[k];
        out_vals[i] = vals[k];
    }
}

inline unsigned grid_for(int64_t n)
{
    int64_t b = ceildiv(n > 0 ? n : 1, 256);
    if (b > 4 * max_stream_blocks) b = 4 * max_stream_blocks;
    return unsigned(b);
}

inline int bits_for(int64_t n_cols)
{
    int bits = 1;
    while (bits < 63 && (int64_t(1) << bits) < n_cols) ++bits;
    return bits;
}

template <typename I>
size_t sort_scratch_bytes(int64_t nnz, int64_t n_cols)
{
    size_t bytes = 0;
    I* p = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, size_t(nnz), 0, bits_for(n_cols),
                                    hipStream_t(nullptr));
    return bytes;
}

template <typename I>
size_t transpose_work_bytes(int64_t nnz, int64_t n_cols)
{
    return 4 * align_up(size_t(nnz) * sizeof(I)) + align_up(sort_scratch_bytes<I>(nnz, n_cols)) + 256;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

extern "C" size_t gkoc_csr_transpose_workspace_bytes(int64_t nnz, int64_t n_cols, size_t index_size)
{
    if (nnz < 0) nnz = 0;
    return index_size == 8 ? transpose_work_bytes<int64_t>(nnz, n_cols)
                           : transpose_work_bytes<int32_t>(nnz, n_cols);
}

#define GKOC_DEF_TRANSPOSE(T, TN, I, IN)                                                    \
    extern "C" int gkoc_csr_transpose_##TN##_##IN(                                          \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const I* row_ptrs,                 \
        const I* col_idxs, const T* vals, int64_t nnz, I* t_row_ptrs, I* t_col_idxs,        \
        T* t_vals, void* work, size_t work_bytes)                                           \
    {                                                                                       \
        GKOC_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, GKOC_E_INVALID,                \
                     "negative dimension");                                                 \
        GKOC_REQUIRE(t_row_ptrs, GKOC_E_INVALID, "null pointer");                           \
        hipStream_t st = as_stream(s);                                                      \
        if (nnz == 0) {                                                                     \
            GKOC_HIP(hipMemsetAsync(t_row_ptrs, 0, size_t(n_cols + 1) * sizeof(I), st));    \
            return GKOC_OK;                                                                 \
        }                                                                                   \
        const size_t need = transpose_work_bytes<I>(nnz, n_cols);                           \
        GKOC_REQUIRE(work && work_bytes >= need, GKOC_E_WORKSPACE,                          \
                     "workspace too small (gkoc_csr_transpose_workspace_bytes)");           \
        char* w = static_cast<char*>(work);                                                 \
        const size_t seg = align_up(size_t(nnz) * sizeof(I));                               \
        I* row_of = reinterpret_cast<I*>(w);                                                \
        I* pos = reinterpret_cast<I*>(w + seg);                                             \
        I* keys_sorted = reinterpret_cast<I*>(w + 2 * seg);                                 \
        I* perm = reinterpret_cast<I*>(w + 3 * seg);                                        \
        void* scratch = w + 4 * seg;                                                        \
        size_t scratch_bytes = sort_scratch_bytes<I>(nnz, n_cols);                          \
        int rc = gkoc_convert_ptrs_to_idxs_##IN(s, row_ptrs, n_rows, row_of);               \
        if (rc != GKOC_OK) return rc;                                                       \
        iota_kernel<I><<<dim3(grid_for(nnz)), dim3(256), 0, st>>>(nnz, pos);                \
        GKOC_LAUNCH_OK();                                                                   \
        GKOC_HIP(rocprim::radix_sort_pairs(scratch, scratch_bytes, col_idxs, keys_sorted,   \
                                           pos, perm, size_t(nnz), 0, bits_for(n_cols),     \
                                           st));                                            \
        rc = gkoc_convert_idxs_to_ptrs_##IN(s, nnz, keys_sorted, n_cols, t_row_ptrs);       \
        if (rc != GKOC_OK) return rc;                                                       \
        gather_transposed_kernel<T, I><<<dim3(grid_for(nnz)), dim3(256), 0, st>>>(          \
            nnz, perm, row_of, vals, t_col_idxs, t_vals);                                   \
        GKOC_LAUNCH_OK();                                                                   \
        return GKOC_OK;                                                                     \
    }

GKOC_DEF_TRANSPOSE(double, f64, int32_t, i32)
GKOC_DEF_TRANSPOSE(double, f64, int64_t, i64)
GKOC_DEF_TRANSPOSE(float, f32, int32_t, i32)
GKOC_DEF_TRANSPOSE(float, f32, int64_t, i64)
// complex values move as pairs (conj_transpose: the binding conjugates the moved values)
GKOC_DEF_TRANSPOSE(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_TRANSPOSE(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_TRANSPOSE(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_TRANSPOSE(gkoc_c64, c64, int64_t, i64)
