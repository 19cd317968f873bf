// Row-partitioned (multi-GPU) support kernels.
//
// Device-side equivalents of what Ginkgo's experimental::distributed::Matrix
// does when it is read (core/distributed/matrix.cpp:300-381:
// distributed_matrix::separate_local_nonlocal + index_map) and applied
// (matrix.cpp:450-509: local SpMV, halo gather, non-local SpMV):
//
//  * split: the rows a rank owns (CSR with GLOBAL column indices) are split into
//    `local` (columns in the rank's own range [col_lo, col_hi), re-based to 0)
//    and `non-local` (all other columns, compressed to a dense halo index space
//    = rank of the column among the sorted distinct non-local columns, which is
//    Ginkgo's index_map ordering for a contiguous partition).  The non-local
//    part is stored as a ROW LIST (only rows that have non-local entries), so
//    that applying it touches 2 boundary planes instead of re-reading and
//    re-writing the whole local vector as csr::advanced_spmv(1, A_nl, x, 1, y)
//    does in the stock path;
//  * row-list SpMV: y[rows[i]] += sum_k vals[k] * halo[cols[k]], sequential in
//    k starting from y (== advanced_spmv with alpha = beta = 1, bit-identical
//    to the reference's distributed apply on the same partition).
// Integer outputs are exact; everything runs on the caller's stream.
#include "common.hpp"
#include "scan.hpp"

namespace gkoc {
namespace {

template <typename I>
__global__ __launch_bounds__(256) void dist_mark_count_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    int64_t col_lo, int64_t col_hi, I* __restrict__ col_map,
    I* __restrict__ local_cnt, I* __restrict__ nl_cnt)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row > n_rows) return;
    if (row == n_rows) {
        local_cnt[row] = 0;
        nl_cnt[row] = 0;
        return;
    }
    I lc = 0, nc = 0;
    for (int64_t k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
        const int64_t c = cols[k];
        if (c >= col_lo && c < col_hi) {
            ++lc;
        } else {
            ++nc;
            col_map[c] = 1;  // benign race: every writer stores 1
        }
    }
    local_cnt[row] = lc;
    nl_cnt[row] = nc;
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void dist_fill_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, int64_t col_lo, int64_t col_hi,
    const I* __restrict__ col_map, const I* __restrict__ local_ptrs,
    const I* __restrict__ nl_ptrs_full, I* __restrict__ local_cols,
    T* __restrict__ local_vals, I* __restrict__ nl_cols, T* __restrict__ nl_vals)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n_rows) return;
    int64_t lp = local_ptrs[row], np = nl_ptrs_full[row];
    for (int64_t k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
        const int64_t c = cols[k];
        if (c >= col_lo && c < col_hi) {
            local_cols[lp] = I(c - col_lo);
            local_vals[lp] = vals[k];
            ++lp;
        } else {
            nl_cols[np] = col_map[c];
            nl_vals[np] = vals[k];
            ++np;
        }
    }
}

// recv_gidx[col_map[c]] = c for every marked column (col_map is the exclusive
// scan of the marks, so marked <=> col_map[c+1] - col_map[c] == 1)
template <typename I>
__global__ __launch_bounds__(256) void dist_halo_list_kernel(
    int64_t n_cols, const I* __restrict__ col_map, I* __restrict__ recv_gidx)
{
    const int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (c >= n_cols) return;
    if (col_map[c + 1] != col_map[c]) recv_gidx[int64_t(col_map[c])] = I(c);
}

template <typename I>
__global__ __launch_bounds__(256) void dist_row_flag_kernel(
    int64_t n_rows, const I* __restrict__ nl_ptrs_full, I* __restrict__ flag)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row > n_rows) return;
    flag[row] = row < n_rows ? I(nl_ptrs_full[row + 1] != nl_ptrs_full[row]) : I(0);
}

template <typename I>
__global__ __launch_bounds__(256) void dist_row_list_kernel(
    int64_t n_rows, const I* __restrict__ nl_ptrs_full,
    const I* __restrict__ pos, I* __restrict__ nl_rows, I* __restrict__ nl_ptrs)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row > n_rows) return;
    if (row == n_rows) {
        nl_ptrs[int64_t(pos[n_rows])] = nl_ptrs_full[n_rows];
        return;
    }
    if (nl_ptrs_full[row + 1] != nl_ptrs_full[row]) {
        nl_rows[int64_t(pos[row])] = I(row);
        nl_ptrs[int64_t(pos[row])] = nl_ptrs_full[row];
    }
}

// One wave per 64 list rows: their contiguous cols / vals range is staged in LDS
// with coalesced loads (up to rl_stage_cap entries), lane = list row then adds
// its products in k order; the y rows of a stencil halo are consecutive, so the
// read-modify-write of y is coalesced as well.
constexpr int rl_stage_cap = 2048;

// DOT: one column; additionally partial[block] = sum over the block's rows of xdot[row] * (what
// was added to y[row]) - the non-local part of <x, A x> next to the fused local SpMV + dot
template <typename T, typename I, bool DOT = false>
__global__ __launch_bounds__(64) void csr_rowlist_add_kernel(
    int64_t n_list, const I* __restrict__ rows, const I* __restrict__ ptrs,
    const I* __restrict__ cols, const T* __restrict__ vals,
    const T* __restrict__ halo, int64_t ld_halo, T* __restrict__ y, int64_t ldy,
    int nrhs, const T* __restrict__ xdot = nullptr, T* __restrict__ partial = nullptr)
{
    __shared__ T lv[rl_stage_cap];
    __shared__ I lc[rl_stage_cap];
    const int lane = threadIdx.x;
    const int64_t first = int64_t(blockIdx.x) * 64;
    const int64_t i = first + lane;
    const bool valid = i < n_list;
    const int64_t last = first + 64 < n_list ? first + 64 : n_list;
    const int64_t K0 = ptrs[first], K1 = ptrs[last];
    const int64_t ks = ptrs[valid ? i : last], ke = ptrs[valid ? i + 1 : last];
    const bool staged = K1 - K0 <= rl_stage_cap;
    if (staged) {
        for (int t = lane; t < int(K1 - K0); t += 64) {
            lv[t] = vals[K0 + t];
            lc[t] = cols[K0 + t];
        }
        wave_lds_sync();
    }
    T dot_acc = T(0);
    if (!valid && !DOT) return;
    const int64_t row = valid ? int64_t(rows[i]) : 0;
    for (int j = 0; valid && j < nrhs; ++j) {
        const T y0 = y[row * ldy + j];
        T sum = y0;
        int64_t k = ks;
        // eight gathers in flight, products added in k order
        for (; k + 8 <= ke; k += 8) {
            T v[8], h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = staged ? lv[k + u - K0] : vals[k + u];
                const I c = staged ? lc[k + u - K0] : cols[k + u];
                h[u] = halo[int64_t(c) * ld_halo + j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u] * h[u];
        }
        for (; k < ke; ++k) {
            const T v = staged ? lv[k - K0] : vals[k];
            const I c = staged ? lc[k - K0] : cols[k];
            sum += v * halo[int64_t(c) * ld_halo + j];
        }
        y[row * ldy + j] = sum;
        if (DOT) dot_acc += xdot[row] * (sum - y0);
    }
    if (DOT) {
        dot_acc = wave_sum(dot_acc);
        if (lane == 0) partial[blockIdx.x] = dot_acc;
    }
}

// ---- boundary rows as COMPLETE rows (slab partitions, one column) ------------------------
// The rows that have non-local entries, with ALL their entries in the original (global) column
// order; a column index below halo_base (>= n_local; the kernels' `n_local` argument) refers to
// the rank's own vector, halo_base + h to halo entry h.  y[rows[i]] = sum_k vals[k] * v(cols[k]) in k order: the single-domain row sum, bit
// for bit.  Such rows need nothing from the local SpMV, so that kernel can skip them (it is
// launched over the interior row range only) and this one can run on the exchange's stream as
// soon as the halo has arrived, next to the tail of the local SpMV.
template <typename I>
__global__ __launch_bounds__(256) void boundary_len_kernel(int64_t n_list, const I* __restrict__ rows,
                                                           const I* __restrict__ row_ptrs,
                                                           I* __restrict__ out)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i > n_list) return;
    out[i] = i < n_list ? I(row_ptrs[int64_t(rows[i]) + 1] - row_ptrs[rows[i]]) : I(0);
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void boundary_fill_kernel(
    int64_t n_list, const I* __restrict__ rows, const I* __restrict__ row_ptrs,
    const I* __restrict__ cols, const T* __restrict__ vals, int64_t col_lo, int64_t col_hi,
    int64_t halo_base, const I* __restrict__ col_map, const I* __restrict__ out_ptrs,
    I* __restrict__ out_cols, T* __restrict__ out_vals)
{
    // one wave per listed row
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t(blockIdx.x) * 256 + threadIdx.x) >> 6;
    if (i >= n_list) return;
    const int64_t r = rows[i];
    const int64_t k0 = row_ptrs[r], len = int64_t(row_ptrs[r + 1]) - k0, o0 = out_ptrs[i];
    for (int64_t t = lane; t < len; t += 64) {
        const int64_t c = cols[k0 + t];
        out_cols[o0 + t] = (c >= col_lo && c < col_hi) ? I(c - col_lo) : I(halo_base + int64_t(col_map[c]));
        out_vals[o0 + t] = vals[k0 + t];
    }
}

// This kernel runs NEXT TO the local SpMV, whose waves fill the LDS of every CU (8 KB each, 20 per
// CU): a workgroup that needs more LDS than one such wave frees is only dispatched once the SpMV
// has drained (a first version staged 24 KB per wave: 181 us serialised against 159 us), and one
// whose lanes walk their rows' entries straight from memory (64 lines per load instruction)
// slows the SpMV down by a quarter while it runs beside it.  So, like the SpMV kernel itself:
// one wave per 32 listed rows = one contiguous range of at most rl_full_cap entries; the lanes
// read values / columns with coalesced loads (entry = lane + 64 j), gather x / halo, and leave the
// PRODUCTS in LDS (8 B each, <= 7 KB per wave); lane = row then adds its products in k order.
// Longer ranges take the lane-per-row walk.
constexpr int rl_full_rows = 32;
constexpr int rl_full_cap = 896;

template <typename T, typename I>
__global__ __launch_bounds__(64) void csr_rowlist_full_kernel(
    int64_t n_list, const I* __restrict__ rows, const I* __restrict__ ptrs,
    const I* __restrict__ cols, const T* __restrict__ vals, int64_t n_local,
    const T* __restrict__ x, const T* __restrict__ halo, T* __restrict__ y)
{
    __shared__ T prod[rl_full_cap];
    const int lane = threadIdx.x;
    const int64_t first = int64_t(blockIdx.x) * rl_full_rows;
    const int64_t last = first + rl_full_rows < n_list ? first + rl_full_rows : n_list;
    const int64_t K0 = ptrs[first], K1 = ptrs[last];
    const int64_t i = first + lane;
    const bool valid = lane < rl_full_rows && i < last;
    const int64_t ks = valid ? int64_t(ptrs[i]) : K1, ke = valid ? int64_t(ptrs[i + 1]) : K1;
    if (K1 - K0 <= rl_full_cap) {
        for (int64_t t = K0 + lane; t < K1; t += 64) {
            const int64_t c = cols[t];
            prod[t - K0] = vals[t] * (c < n_local ? x[c] : halo[c - n_local]);
        }
        wave_lds_sync();
        if (!valid) return;
        T sum = T(0);
        for (int64_t k = ks; k < ke; ++k) sum += prod[k - K0];
        y[rows[i]] = sum;
        return;
    }
    if (!valid) return;
    T sum = T(0);
    for (int64_t k = ks; k < ke; ++k) {
        const int64_t c = cols[k];
        sum += vals[k] * (c < n_local ? x[c] : halo[c - n_local]);
    }
    y[rows[i]] = sum;
}

// inout[0] += sum of partial[0 .. count) (fixed tree)
template <typename T>
__global__ __launch_bounds__(1024) void add_partials_kernel(int64_t count, const T* __restrict__ partial,
                                                            T* __restrict__ inout)
{
    __shared__ T lds[1024 / 64];
    T acc = T(0);
    for (int64_t t = threadIdx.x; t < count; t += 1024) acc += partial[t];
    const T r = block_sum<1024>(acc, lds);
    if (threadIdx.x == 0) inout[0] += r;
}

inline dim3 grid_for(int64_t n) { return dim3(unsigned(ceildiv(n > 0 ? n : 1, 256))); }

template <typename I>
int split_count(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* cols,
                int64_t col_lo, int64_t col_hi, int64_t n_global_cols, I* col_map,
                I* local_ptrs, I* nl_ptrs_full, int64_t* n_halo, int64_t* nnz_local,
                int64_t* nnz_nl, int64_t* n_nl_rows)
{
    GKOC_REQUIRE(n_rows >= 0 && col_lo >= 0 && col_lo <= col_hi &&
                     col_hi <= n_global_cols,
                 GKOC_E_INVALID, "bad partition range");
    GKOC_REQUIRE(col_map && local_ptrs && nl_ptrs_full && n_halo && nnz_local &&
                     nnz_nl && n_nl_rows,
                 GKOC_E_INVALID, "null output");
    hipStream_t st = as_stream(s);
    GKOC_HIP(hipMemsetAsync(col_map, 0, sizeof(I) * (n_global_cols + 1), st));
    dist_mark_count_kernel<I><<<grid_for(n_rows + 1), dim3(256), 0, st>>>(
        n_rows, row_ptrs, cols, col_lo, col_hi, col_map, local_ptrs, nl_ptrs_full);
    GKOC_LAUNCH_OK();
    int rc = device_exclusive_scan<I>(st, col_map, n_global_cols + 1);
    if (rc) return rc;
    rc = device_exclusive_scan<I>(st, local_ptrs, n_rows + 1);
    if (rc) return rc;
    // number of rows with non-local entries: scan of flags in a temporary
    I* flag = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&flag), sizeof(I) * (n_rows + 1)));
    rc = device_exclusive_scan<I>(st, nl_ptrs_full, n_rows + 1);
    if (rc) return rc;
    dist_row_flag_kernel<I><<<grid_for(n_rows + 1), dim3(256), 0, st>>>(n_rows, nl_ptrs_full, flag);
    GKOC_LAUNCH_OK();
    rc = device_exclusive_scan<I>(st, flag, n_rows + 1);
    if (rc) return rc;
    I h[4] = {0, 0, 0, 0};
    GKOC_HIP(hipMemcpyAsync(&h[0], col_map + n_global_cols, sizeof(I), hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipMemcpyAsync(&h[1], local_ptrs + n_rows, sizeof(I), hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipMemcpyAsync(&h[2], nl_ptrs_full + n_rows, sizeof(I), hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipMemcpyAsync(&h[3], flag + n_rows, sizeof(I), hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipStreamSynchronize(st));
    GKOC_TRY(scratch_free(st, flag));
    *n_halo = int64_t(h[0]);
    *nnz_local = int64_t(h[1]);
    *nnz_nl = int64_t(h[2]);
    *n_nl_rows = int64_t(h[3]);
    return GKOC_OK;
}

template <typename T, typename I>
int split_fill(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* cols,
               const T* vals, int64_t col_lo, int64_t col_hi, int64_t n_global_cols,
               const I* col_map, const I* local_ptrs, const I* nl_ptrs_full,
               I* local_cols, T* local_vals, I* nl_rows, I* nl_ptrs, I* nl_cols,
               T* nl_vals, I* recv_gidx)
{
    hipStream_t st = as_stream(s);
    if (n_rows > 0) {
        dist_fill_kernel<T, I><<<grid_for(n_rows), dim3(256), 0, st>>>(
            n_rows, row_ptrs, cols, vals, col_lo, col_hi, col_map, local_ptrs,
            nl_ptrs_full, local_cols, local_vals, nl_cols, nl_vals);
        GKOC_LAUNCH_OK();
    }
    dist_halo_list_kernel<I><<<grid_for(n_global_cols), dim3(256), 0, st>>>(
        n_global_cols, col_map, recv_gidx);
    GKOC_LAUNCH_OK();
    I* pos = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&pos), sizeof(I) * (n_rows + 1)));
    dist_row_flag_kernel<I><<<grid_for(n_rows + 1), dim3(256), 0, st>>>(n_rows, nl_ptrs_full, pos);
    GKOC_LAUNCH_OK();
    int rc = device_exclusive_scan<I>(st, pos, n_rows + 1);
    if (rc) return rc;
    dist_row_list_kernel<I><<<grid_for(n_rows + 1), dim3(256), 0, st>>>(
        n_rows, nl_ptrs_full, pos, nl_rows, nl_ptrs);
    GKOC_LAUNCH_OK();
    GKOC_TRY(scratch_free(st, pos));
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_DIST_IDX(I, IN)                                               \
    extern "C" int gkoc_dist_split_count_##IN(                                 \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* cols,     \
        int64_t col_lo, int64_t col_hi, int64_t n_global_cols, I* col_map,     \
        I* local_row_ptrs, I* nl_row_ptrs_full, int64_t* n_halo_host,          \
        int64_t* nnz_local_host, int64_t* nnz_nl_host,                         \
        int64_t* n_nl_rows_host)                                               \
    {                                                                          \
        return split_count<I>(s, n_rows, row_ptrs, cols, col_lo, col_hi,       \
                              n_global_cols, col_map, local_row_ptrs,          \
                              nl_row_ptrs_full, n_halo_host, nnz_local_host,   \
                              nnz_nl_host, n_nl_rows_host);                    \
    }
GKOC_DEF_DIST_IDX(int32_t, i32)
GKOC_DEF_DIST_IDX(int64_t, i64)

#define GKOC_DEF_DIST_BND_IDX(I, IN)                                           \
    extern "C" int gkoc_dist_boundary_count_##IN(                              \
        gkoc_stream_t s, int64_t n_list, const I* rows, const I* row_ptrs,     \
        I* out_ptrs, int64_t* nnz_host)                                        \
    {                                                                          \
        GKOC_REQUIRE(out_ptrs && nnz_host && n_list >= 0, GKOC_E_INVALID,      \
                     "bad argument");                                          \
        hipStream_t st = as_stream(s);                                         \
        boundary_len_kernel<I><<<grid_for(n_list + 1), dim3(256), 0, st>>>(    \
            n_list, rows, row_ptrs, out_ptrs);                                 \
        GKOC_LAUNCH_OK();                                                      \
        int rc = device_exclusive_scan<I>(st, out_ptrs, n_list + 1);           \
        if (rc) return rc;                                                     \
        I h = 0;                                                               \
        GKOC_HIP(hipMemcpyAsync(&h, out_ptrs + n_list, sizeof(I),              \
                                hipMemcpyDeviceToHost, st));                   \
        GKOC_HIP(hipStreamSynchronize(st));                                    \
        *nnz_host = int64_t(h);                                                \
        return GKOC_OK;                                                        \
    }
GKOC_DEF_DIST_BND_IDX(int32_t, i32)
GKOC_DEF_DIST_BND_IDX(int64_t, i64)

#define GKOC_DEF_DIST(T, TN, I, IN)                                            \
    extern "C" int gkoc_dist_split_fill_##TN##_##IN(                           \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* cols,     \
        const T* vals, int64_t col_lo, int64_t col_hi, int64_t n_global_cols,  \
        const I* col_map, const I* local_row_ptrs, const I* nl_row_ptrs_full,  \
        I* local_cols, T* local_vals, I* nl_rows, I* nl_ptrs, I* nl_cols,      \
        T* nl_vals, I* recv_gidx)                                              \
    {                                                                          \
        return split_fill<T, I>(s, n_rows, row_ptrs, cols, vals, col_lo,       \
                                col_hi, n_global_cols, col_map,                \
                                local_row_ptrs, nl_row_ptrs_full, local_cols,  \
                                local_vals, nl_rows, nl_ptrs, nl_cols,         \
                                nl_vals, recv_gidx);                           \
    }                                                                          \
    extern "C" int gkoc_dist_boundary_fill_##TN##_##IN(                        \
        gkoc_stream_t s, int64_t n_list, const I* rows, const I* row_ptrs,     \
        const I* cols, const T* vals, int64_t col_lo, int64_t col_hi,          \
        int64_t halo_base, const I* col_map, const I* out_ptrs, I* out_cols,   \
        T* out_vals)                                                           \
    {                                                                          \
        if (n_list <= 0) return GKOC_OK;                                       \
        GKOC_REQUIRE(halo_base >= col_hi - col_lo, GKOC_E_INVALID,             \
                     "halo_base inside the local columns");                    \
        boundary_fill_kernel<T, I>                                             \
            <<<dim3(unsigned(ceildiv(n_list * 64, 256))), dim3(256), 0,        \
               as_stream(s)>>>(n_list, rows, row_ptrs, cols, vals, col_lo,     \
                               col_hi, halo_base, col_map, out_ptrs, out_cols, \
                               out_vals);                                      \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_csr_rowlist_spmv_full_##TN##_##IN(                     \
        gkoc_stream_t s, int64_t n_list, const I* rows, const I* ptrs,         \
        const I* cols, const T* vals, int64_t n_local, const T* x,             \
        const T* halo, T* y)                                                   \
    {                                                                          \
        if (n_list <= 0) return GKOC_OK;                                       \
        GKOC_REQUIRE(rows && ptrs && x && halo && y, GKOC_E_INVALID,           \
                     "null pointer");                                          \
        csr_rowlist_full_kernel<T, I>                                          \
            <<<dim3(unsigned(ceildiv(n_list, rl_full_rows))), dim3(64), 0,     \
               as_stream(s)>>>(n_list, rows, ptrs, cols, vals, n_local, x,     \
                               halo, y);                                       \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_x_csr_rowlist_spmv_add_dot_##TN##_##IN(                \
        gkoc_stream_t s, int64_t n_list, const I* rows, const I* ptrs,         \
        const I* cols, const T* vals, const T* halo, T* y, const T* x,         \
        T* dot_inout, void* work, size_t work_bytes)                           \
    {                                                                          \
        if (n_list <= 0) return GKOC_OK;                                       \
        GKOC_REQUIRE(rows && ptrs && halo && y && x && dot_inout && work,      \
                     GKOC_E_INVALID, "null pointer");                          \
        const int64_t nb = ceildiv(n_list, 64);                                \
        GKOC_REQUIRE(work_bytes >= size_t(nb) * sizeof(T), GKOC_E_WORKSPACE,   \
                     "workspace too small (one value per 64 listed rows)");    \
        csr_rowlist_add_kernel<T, I, true>                                     \
            <<<dim3(unsigned(nb)), dim3(64), 0, as_stream(s)>>>(               \
                n_list, rows, ptrs, cols, vals, halo, 1, y, 1, 1, x,           \
                static_cast<T*>(work));                                        \
        GKOC_LAUNCH_OK();                                                      \
        add_partials_kernel<T><<<dim3(1), dim3(1024), 0, as_stream(s)>>>(      \
            nb, static_cast<const T*>(work), dot_inout);                       \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_csr_rowlist_spmv_add_##TN##_##IN(                      \
        gkoc_stream_t s, int64_t n_list, const I* rows, const I* ptrs,         \
        const I* cols, const T* vals, const T* halo, int64_t ld_halo, T* y,    \
        int64_t ldy, int64_t nrhs)                                             \
    {                                                                          \
        if (n_list <= 0 || nrhs <= 0) return GKOC_OK;                          \
        csr_rowlist_add_kernel<T, I>                                           \
            <<<dim3(unsigned(ceildiv(n_list, 64))), dim3(64), 0,               \
               as_stream(s)>>>(                                                \
                n_list, rows, ptrs, cols, vals, halo, ld_halo, y, ldy,         \
                int(nrhs));                                                    \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }
GKOC_DEF_DIST(double, f64, int32_t, i32)
GKOC_DEF_DIST(double, f64, int64_t, i64)
GKOC_DEF_DIST(float, f32, int32_t, i32)
GKOC_DEF_DIST(float, f32, int64_t, i64)
