// Device-side assembly of device_matrix_data: the kernels behind
// device_matrix_data::{sort_row_major, remove_zeros, sum_duplicates} and
// matrix_data <- device_matrix_data (soa_to_aos):
//   decl core/base/device_matrix_data_kernels.hpp:23-51;
//   reference/base/device_matrix_data_kernels.cpp:24-143.
// All of it is integer / copy work plus, in sum_duplicates, a left-to-right sum of
// the entries of one (row, column) run starting from 0 - reproduced exactly: the
// thread that owns the first entry of a run walks the run in storage order.
//   sort_row_major : std::stable_sort by (row, column) == two stable LSD radix
//                    sorts of the entry positions (by column, then by row) and one
//                    gather; the radix sort is rocPRIM's, the rest hand-written.
//   remove_zeros   : mark (value != 0; NaN stays, -0 goes) -> exclusive scan ->
//                    the count goes to the host (the caller allocates the compacted
//                    arrays, as the reference does) -> scatter.
//   sum_duplicates : mark run heads on sorted input -> scan -> count -> one thread
//                    per run sums it.
// Workspace layouts are private; sizes come from gkoc_*_workspace_bytes.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "scan.hpp"

namespace gkoc {
namespace {

inline size_t align_up(size_t v) { return (v + 255) / 256 * 256; }

inline unsigned grid_for(int64_t n)
{
    int64_t b = ceildiv(n > 0 ? n : 1, 256);
    if (b > 4 * max_stream_blocks) b = 4 * max_stream_blocks;
    return unsigned(b);
}

// Ginkgo's matrix_data_entry<T, I> (matrix_data.hpp:60): { I row; I column; T value; }
template <typename T, typename I>
struct md_entry {
    I row;
    I column;
    T value;
};

template <typename T, typename I>
__global__ __launch_bounds__(256) void soa_to_aos_kernel(
    int64_t nnz, const I* __restrict__ rows, const I* __restrict__ cols,
    const T* __restrict__ vals, md_entry<T, I>* __restrict__ out)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < nnz; i += stride) {
        md_entry<T, I> e;
        e.row = rows[i];
        e.column = cols[i];
        e.value = vals[i];
        out[i] = e;
    }
}

template <typename I>
__global__ __launch_bounds__(256) void iota_kernel(int64_t n, I* out)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) out[i] = I(i);
}

template <typename A, typename I>
__global__ __launch_bounds__(256) void gather_kernel(int64_t n, const I* __restrict__ perm,
                                                     const A* __restrict__ in,
                                                     A* __restrict__ out)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
        out[i] = in[int64_t(perm[i])];
    }
}

template <typename I>
size_t sort_scratch_bytes(int64_t nnz)
{
    size_t bytes = 0;
    I* p = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, size_t(nnz), 0,
                                    int(8 * sizeof(I)), hipStream_t(nullptr));
    return bytes;
}

template <typename T, typename I>
size_t sort_work_bytes(int64_t nnz)
{
    return 4 * align_up(size_t(nnz) * sizeof(I)) + align_up(size_t(nnz) * sizeof(T)) +
           align_up(sort_scratch_bytes<I>(nnz)) + 256;
}

// pos[0..nnz] holds the marks (pos[nnz] = 0) and, after the exclusive scan, the output
// position of every kept entry and the number of kept entries in pos[nnz]
inline size_t compact_work_bytes(int64_t nnz)
{
    return align_up(size_t(nnz + 1) * sizeof(int64_t)) +
           align_up(size_t(scan_scratch_count(nnz + 1)) * sizeof(int64_t)) + 256;
}

template <typename T>
__global__ __launch_bounds__(256) void mark_nonzeros_kernel(int64_t nnz,
                                                            const T* __restrict__ vals,
                                                            int64_t* __restrict__ pos)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i <= nnz; i += stride) {
        pos[i] = (i < nnz && vals[i] != zero_of<T>()) ? 1 : 0;
    }
}

template <typename I>
__global__ __launch_bounds__(256) void mark_run_heads_kernel(int64_t nnz,
                                                             const I* __restrict__ rows,
                                                             const I* __restrict__ cols,
                                                             int64_t* __restrict__ pos)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i <= nnz; i += stride) {
        // the reference starts from (invalid_index, invalid_index) = (-1, -1)
        const I pr = i > 0 ? rows[i - 1] : I(-1);
        const I pc = i > 0 ? cols[i - 1] : I(-1);
        pos[i] = (i < nnz && (rows[i] != pr || cols[i] != pc)) ? 1 : 0;
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void compact_kernel(
    int64_t nnz, const int64_t* __restrict__ pos, const I* __restrict__ rows,
    const I* __restrict__ cols, const T* __restrict__ vals, I* __restrict__ out_rows,
    I* __restrict__ out_cols, T* __restrict__ out_vals)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < nnz; i += stride) {
        const int64_t o = pos[i];
        if (pos[i + 1] != o) {
            out_rows[o] = rows[i];
            out_cols[o] = cols[i];
            out_vals[o] = vals[i];
        }
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void sum_runs_kernel(
    int64_t nnz, const int64_t* __restrict__ pos, const I* __restrict__ rows,
    const I* __restrict__ cols, const T* __restrict__ vals, I* __restrict__ out_rows,
    I* __restrict__ out_cols, T* __restrict__ out_vals)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < nnz; i += stride) {
        const int64_t o = pos[i];
        if (pos[i + 1] != o) {
            T sum = zero_of<T>();
            int64_t k = i;
            do {
                sum += vals[k];
                ++k;
            } while (k < nnz && pos[k + 1] == pos[k]);
            out_rows[o] = rows[i];
            out_cols[o] = cols[i];
            out_vals[o] = sum;
        }
    }
}

// scans the marks and brings the count to the host
inline int scan_and_count(hipStream_t st, int64_t nnz, void* work, int64_t* count_host)
{
    int64_t* pos = static_cast<int64_t*>(work);
    int64_t* scratch = reinterpret_cast<int64_t*>(static_cast<char*>(work) +
                                                  align_up(size_t(nnz + 1) * sizeof(int64_t)));
    int rc = device_exclusive_scan<int64_t>(st, pos, nnz + 1, scratch);
    if (rc != GKOC_OK) return rc;
    GKOC_HIP(hipMemcpyAsync(count_host, pos + nnz, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipStreamSynchronize(st));
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

extern "C" size_t gkoc_sort_row_major_workspace_bytes(int64_t nnz, size_t value_size,
                                                      size_t index_size)
{
    if (nnz < 0) nnz = 0;
    // value_size 4 / 8 / 16: float, double or complex<float>, complex<double>
    if (index_size == 8) {
        return value_size == 16  ? sort_work_bytes<gkoc_c128, int64_t>(nnz)
               : value_size == 8 ? sort_work_bytes<double, int64_t>(nnz)
                                 : sort_work_bytes<float, int64_t>(nnz);
    }
    return value_size == 16  ? sort_work_bytes<gkoc_c128, int32_t>(nnz)
           : value_size == 8 ? sort_work_bytes<double, int32_t>(nnz)
                             : sort_work_bytes<float, int32_t>(nnz);
}

extern "C" size_t gkoc_compact_workspace_bytes(int64_t nnz)
{
    return compact_work_bytes(nnz < 0 ? 0 : nnz);
}

#define GKOC_DEF_ASSEMBLY(T, TN, I, IN)                                                     \
    extern "C" int gkoc_soa_to_aos_##TN##_##IN(gkoc_stream_t s, int64_t nnz,                \
                                               const I* row_idxs, const I* col_idxs,        \
                                               const T* vals, void* entries)                \
    {                                                                                       \
        if (nnz <= 0) return GKOC_OK;                                                       \
        soa_to_aos_kernel<T, I><<<dim3(grid_for(nnz)), dim3(256), 0, as_stream(s)>>>(       \
            nnz, row_idxs, col_idxs, vals, static_cast<md_entry<T, I>*>(entries));          \
        GKOC_LAUNCH_OK();                                                                   \
        return GKOC_OK;                                                                     \
    }                                                                                       \
    extern "C" int gkoc_sort_row_major_##TN##_##IN(gkoc_stream_t s, int64_t nnz,            \
                                                   I* row_idxs, I* col_idxs, T* vals,       \
                                                   void* work, size_t work_bytes)           \
    {                                                                                       \
        GKOC_REQUIRE(nnz >= 0, GKOC_E_INVALID, "negative size");                            \
        if (nnz <= 1) return GKOC_OK;                                                       \
        const size_t need = sort_work_bytes<T, I>(nnz);                                     \
        GKOC_REQUIRE(work && work_bytes >= need, GKOC_E_WORKSPACE,                          \
                     "workspace too small (gkoc_sort_row_major_workspace_bytes)");          \
        hipStream_t st = as_stream(s);                                                      \
        char* w = static_cast<char*>(work);                                                 \
        const size_t seg = align_up(size_t(nnz) * sizeof(I));                               \
        I* a = reinterpret_cast<I*>(w);                                                     \
        I* b = reinterpret_cast<I*>(w + seg);                                               \
        I* c = reinterpret_cast<I*>(w + 2 * seg);                                           \
        I* d = reinterpret_cast<I*>(w + 3 * seg);                                           \
        T* e = reinterpret_cast<T*>(w + 4 * seg);                                           \
        void* scratch = w + 4 * seg + align_up(size_t(nnz) * sizeof(T));                    \
        size_t scratch_bytes = sort_scratch_bytes<I>(nnz);                                  \
        const dim3 grid(grid_for(nnz));                                                     \
        const int bits = int(8 * sizeof(I));                                                \
        iota_kernel<I><<<grid, dim3(256), 0, st>>>(nnz, a);                                 \
        GKOC_LAUNCH_OK();                                                                   \
        /* by column: b = sorted columns (unused), c = positions */                         \
        GKOC_HIP(rocprim::radix_sort_pairs(scratch, scratch_bytes, col_idxs, b, a, c,       \
                                           size_t(nnz), 0, bits, st));                      \
        gather_kernel<I, I><<<grid, dim3(256), 0, st>>>(nnz, c, row_idxs, d);               \
        GKOC_LAUNCH_OK();                                                                   \
        /* by row, stable: b = sorted rows, a = final positions */                          \
        GKOC_HIP(rocprim::radix_sort_pairs(scratch, scratch_bytes, d, b, c, a, size_t(nnz), \
                                           0, bits, st));                                   \
        gather_kernel<I, I><<<grid, dim3(256), 0, st>>>(nnz, a, col_idxs, d);               \
        GKOC_LAUNCH_OK();                                                                   \
        gather_kernel<T, I><<<grid, dim3(256), 0, st>>>(nnz, a, vals, e);                   \
        GKOC_LAUNCH_OK();                                                                   \
        GKOC_HIP(hipMemcpyAsync(row_idxs, b, size_t(nnz) * sizeof(I),                       \
                                hipMemcpyDeviceToDevice, st));                              \
        GKOC_HIP(hipMemcpyAsync(col_idxs, d, size_t(nnz) * sizeof(I),                       \
                                hipMemcpyDeviceToDevice, st));                              \
        GKOC_HIP(hipMemcpyAsync(vals, e, size_t(nnz) * sizeof(T), hipMemcpyDeviceToDevice,  \
                                st));                                                       \
        return GKOC_OK;                                                                     \
    }                                                                                       \
    extern "C" int gkoc_remove_zeros_fill_##TN##_##IN(                                      \
        gkoc_stream_t s, int64_t nnz, const I* row_idxs, const I* col_idxs, const T* vals,  \
        const void* work, I* out_rows, I* out_cols, T* out_vals)                            \
    {                                                                                       \
        if (nnz <= 0) return GKOC_OK;                                                       \
        GKOC_REQUIRE(work, GKOC_E_WORKSPACE, "null workspace");                             \
        compact_kernel<T, I><<<dim3(grid_for(nnz)), dim3(256), 0, as_stream(s)>>>(          \
            nnz, static_cast<const int64_t*>(work), row_idxs, col_idxs, vals, out_rows,     \
            out_cols, out_vals);                                                            \
        GKOC_LAUNCH_OK();                                                                   \
        return GKOC_OK;                                                                     \
    }                                                                                       \
    extern "C" int gkoc_sum_duplicates_fill_##TN##_##IN(                                    \
        gkoc_stream_t s, int64_t nnz, const I* row_idxs, const I* col_idxs, const T* vals,  \
        const void* work, I* out_rows, I* out_cols, T* out_vals)                            \
    {                                                                                       \
        if (nnz <= 0) return GKOC_OK;                                                       \
        GKOC_REQUIRE(work, GKOC_E_WORKSPACE, "null workspace");                             \
        sum_runs_kernel<T, I><<<dim3(grid_for(nnz)), dim3(256), 0, as_stream(s)>>>(         \
            nnz, static_cast<const int64_t*>(work), row_idxs, col_idxs, vals, out_rows,     \
            out_cols, out_vals);                                                            \
        GKOC_LAUNCH_OK();                                                                   \
        return GKOC_OK;                                                                     \
    }

GKOC_DEF_ASSEMBLY(double, f64, int32_t, i32)
GKOC_DEF_ASSEMBLY(double, f64, int64_t, i64)
GKOC_DEF_ASSEMBLY(float, f32, int32_t, i32)
GKOC_DEF_ASSEMBLY(float, f32, int64_t, i64)
// complex values: moved as pairs, zero if both parts are, summed component-wise
GKOC_DEF_ASSEMBLY(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_ASSEMBLY(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_ASSEMBLY(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_ASSEMBLY(gkoc_c64, c64, int64_t, i64)

#define GKOC_DEF_COUNT_NZ(T, TN)                                                            \
    extern "C" int gkoc_remove_zeros_count_##TN(gkoc_stream_t s, int64_t nnz,               \
                                                const T* vals, void* work,                  \
                                                size_t work_bytes, int64_t* count_host)     \
    {                                                                                       \
        GKOC_REQUIRE(count_host, GKOC_E_INVALID, "null result");                            \
        *count_host = 0;                                                                    \
        if (nnz <= 0) return GKOC_OK;                                                       \
        GKOC_REQUIRE(work && work_bytes >= compact_work_bytes(nnz), GKOC_E_WORKSPACE,       \
                     "workspace too small (gkoc_compact_workspace_bytes)");                 \
        mark_nonzeros_kernel<T><<<dim3(grid_for(nnz + 1)), dim3(256), 0, as_stream(s)>>>(   \
            nnz, vals, static_cast<int64_t*>(work));                                        \
        GKOC_LAUNCH_OK();                                                                   \
        return scan_and_count(as_stream(s), nnz, work, count_host);                         \
    }
GKOC_DEF_COUNT_NZ(double, f64)
GKOC_DEF_COUNT_NZ(float, f32)
GKOC_DEF_COUNT_NZ(gkoc_c128, c128)
GKOC_DEF_COUNT_NZ(gkoc_c64, c64)

#define GKOC_DEF_COUNT_RUNS(I, IN)                                                          \
    extern "C" int gkoc_sum_duplicates_count_##IN(gkoc_stream_t s, int64_t nnz,             \
                                                  const I* row_idxs, const I* col_idxs,     \
                                                  void* work, size_t work_bytes,            \
                                                  int64_t* count_host)                      \
    {                                                                                       \
        GKOC_REQUIRE(count_host, GKOC_E_INVALID, "null result");                            \
        *count_host = 0;                                                                    \
        if (nnz <= 0) return GKOC_OK;                                                       \
        GKOC_REQUIRE(work && work_bytes >= compact_work_bytes(nnz), GKOC_E_WORKSPACE,       \
                     "workspace too small (gkoc_compact_workspace_bytes)");                 \
        mark_run_heads_kernel<I><<<dim3(grid_for(nnz + 1)), dim3(256), 0, as_stream(s)>>>(  \
            nnz, row_idxs, col_idxs, static_cast<int64_t*>(work));                          \
        GKOC_LAUNCH_OK();                                                                   \
        return scan_and_count(as_stream(s), nnz, work, count_host);                         \
    }
GKOC_DEF_COUNT_RUNS(int32_t, i32)
GKOC_DEF_COUNT_RUNS(int64_t, i64)
