// Set-up kernels of Ginkgo's distributed classes on the device: what
// experimental::distributed::{Partition, index_map, Matrix::read_distributed,
// Vector::read_distributed, assemble_rows_from_neighbors} ask their executor for.
//   declarations: core/distributed/{partition,partition_helpers,index_map,matrix,vector,
//                 assembly}_kernels.hpp
//   semantics   : reference/distributed/*_kernels.cpp, reference/distributed/partition_helpers.hpp
//                 (find_range, map_to_local, find_local_range, map_to_global)
// Integer and copy work only (values are MOVED, never computed with: they travel as 4-, 8- or
// 16-byte words, so float / double / complex<float> / complex<double> share the code).  Every
// output is element-for-element the reference's: where the reference walks its input once and
// appends (separate_local_nonlocal, build_from_mapping) a mark + exclusive scan + scatter keeps
// the input order; where it calls std::stable_sort / std::sort + std::unique the same order
// comes from stable LSD radix sorts of positions (rocPRIM's radix sort, as in assembly.hip).
// Temporaries come from the arena's stream-ordered scratch.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "scan.hpp"

namespace gkoc {
namespace {

inline unsigned grid_of(int64_t n)
{
    int64_t b = ceildiv(n > 0 ? n : 1, 256);
    if (b > 4 * max_stream_blocks) b = 4 * max_stream_blocks;
    return unsigned(b);
}

#define GKOC_GRID_STRIDE(i, n)                                                         \
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x, i##_stride = int64_t(gridDim.x) * 256; \
         i < (n); i += i##_stride)

struct w16 {
    uint64_t a, b;
};

// RAII-less scratch helper: allocations are handed back in stream order by release()
struct scratch_set {
    hipStream_t st;
    void* ptrs[12];
    int n = 0;
    explicit scratch_set(hipStream_t s) : st(s) {}
    template <typename T>
    int get(T** p, int64_t count)
    {
        void* v = nullptr;
        int rc = scratch_malloc(st, &v, size_t(count > 0 ? count : 1) * sizeof(T));
        if (rc != GKOC_OK) return rc;
        ptrs[n++] = v;
        *p = static_cast<T*>(v);
        return GKOC_OK;
    }
    void release()
    {
        for (int i = 0; i < n; ++i) (void)scratch_free(st, ptrs[i]);
        n = 0;
    }
};

// a Partition as the kernels see it
template <typename L, typename G>
struct part_view {
    int64_t num_ranges;
    int32_t num_parts;
    const G* bounds;      // [num_ranges + 1]
    const int32_t* pids;  // [num_ranges]
    const L* starts;      // [num_ranges]
    const L* sizes;       // [num_parts]
};

template <typename L, typename G>
part_view<L, G> view_of(const gkoc_partition* p)
{
    return {p->num_ranges, p->num_parts, static_cast<const G*>(p->range_bounds), p->part_ids,
            static_cast<const L*>(p->range_starting_indices), static_cast<const L*>(p->part_sizes)};
}

// partition_helpers.hpp find_range: index of the range that holds idx = number of upper bounds
// (bounds[1..num_ranges]) that are <= idx
template <typename G>
__device__ __forceinline__ int64_t find_range(G idx, const G* __restrict__ bounds, int64_t num_ranges)
{
    int64_t lo = 0, hi = num_ranges;   // searching bounds[1 + lo .. 1 + hi)
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (bounds[1 + mid] <= idx) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    return lo;
}

template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(int64_t n, T* out, T v)
{
    GKOC_GRID_STRIDE(i, n) out[i] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void iota_kernel2(int64_t n, T* out)
{
    GKOC_GRID_STRIDE(i, n) out[i] = T(i);
}

template <typename K, typename V>
int stable_sort_pairs(hipStream_t st, int64_t n, const K* keys_in, K* keys_out, const V* vals_in,
                      V* vals_out, int end_bit = int(8 * sizeof(K)))
{
    if (n <= 0) return GKOC_OK;
    size_t bytes = 0;
    GKOC_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, size_t(n), 0,
                                       end_bit, st));
    void* tmp = nullptr;
    GKOC_TRY(scratch_malloc(st, &tmp, bytes ? bytes : 1));
    hipError_t e = rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, size_t(n), 0,
                                             end_bit, st);
    (void)scratch_free(st, tmp);
    GKOC_HIP(e);
    return GKOC_OK;
}

template <typename T>
int exclusive_scan_inplace(hipStream_t st, T* data, int64_t n)
{
    if (n <= 0) return GKOC_OK;
    T* scratch = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&scratch), size_t(scan_scratch_count(n)) * sizeof(T)));
    int rc = device_exclusive_scan<T>(st, data, n, scratch);
    (void)scratch_free(st, scratch);
    return rc;
}

int to_host(hipStream_t st, void* dst, const void* src, size_t bytes)
{
    GKOC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipStreamSynchronize(st));
    return GKOC_OK;
}

// ======================================================================= distributed_matrix
// class of an entry: 0 not ours, 1 local block, 2 non-local block; marks in two arrays of
// nnz + 1 entries that the scans turn into output positions
template <typename L, typename G>
__global__ __launch_bounds__(256) void separate_mark_kernel(int64_t nnz, const G* __restrict__ rows,
                                                           const G* __restrict__ cols,
                                                           part_view<L, G> rp, part_view<L, G> cp,
                                                           int32_t local_part, int64_t* __restrict__ pos_l,
                                                           int64_t* __restrict__ pos_n)
{
    GKOC_GRID_STRIDE(i, nnz + 1)
    {
        int64_t l = 0, n = 0;
        if (i < nnz) {
            const int64_t rr = find_range(rows[i], rp.bounds, rp.num_ranges);
            if (rp.pids[rr] == local_part) {
                const int64_t cr = find_range(cols[i], cp.bounds, cp.num_ranges);
                if (cp.pids[cr] == local_part) {
                    l = 1;
                } else {
                    n = 1;
                }
            }
        }
        pos_l[i] = l;
        pos_n[i] = n;
    }
}

template <typename L, typename G, typename W>
__global__ __launch_bounds__(256) void separate_fill_kernel(
    int64_t nnz, const G* __restrict__ rows, const G* __restrict__ cols, const W* __restrict__ vals,
    part_view<L, G> rp, part_view<L, G> cp, const int64_t* __restrict__ pos_l,
    const int64_t* __restrict__ pos_n, L* __restrict__ l_row, L* __restrict__ l_col, W* __restrict__ l_val,
    L* __restrict__ n_row, G* __restrict__ n_col, W* __restrict__ n_val)
{
    GKOC_GRID_STRIDE(i, nnz)
    {
        const int64_t ol = pos_l[i], on = pos_n[i];
        const bool is_l = pos_l[i + 1] != ol, is_n = pos_n[i + 1] != on;
        if (!is_l && !is_n) continue;
        const G gr = rows[i], gc = cols[i];
        const int64_t rr = find_range(gr, rp.bounds, rp.num_ranges);
        const L lr = L(gr - rp.bounds[rr]) + rp.starts[rr];
        if (is_l) {
            const int64_t cr = find_range(gc, cp.bounds, cp.num_ranges);
            l_row[ol] = lr;
            l_col[ol] = L(gc - cp.bounds[cr]) + cp.starts[cr];
            l_val[ol] = vals[i];
        } else {
            n_row[on] = lr;
            n_col[on] = gc;
            n_val[on] = vals[i];
        }
    }
}

template <typename L, typename G>
int separate_count(gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const gkoc_partition* rp,
                   const gkoc_partition* cp, int32_t local_part, void** state, int64_t* n_local,
                   int64_t* n_non_local)
{
    GKOC_REQUIRE(state && n_local && n_non_local && rp && cp && nnz >= 0, GKOC_E_INVALID, "bad argument");
    *state = nullptr;
    *n_local = *n_non_local = 0;
    if (nnz == 0) return GKOC_OK;
    hipStream_t st = as_stream(s);
    int64_t* pos = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&pos), size_t(2 * (nnz + 1)) * sizeof(int64_t)));
    separate_mark_kernel<L, G><<<dim3(grid_of(nnz + 1)), dim3(256), 0, st>>>(
        nnz, rows, cols, view_of<L, G>(rp), view_of<L, G>(cp), local_part, pos, pos + nnz + 1);
    GKOC_LAUNCH_OK();
    GKOC_TRY(exclusive_scan_inplace<int64_t>(st, pos, nnz + 1));
    GKOC_TRY(exclusive_scan_inplace<int64_t>(st, pos + nnz + 1, nnz + 1));
    int64_t h[2];
    GKOC_HIP(hipMemcpyAsync(&h[0], pos + nnz, 8, hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipMemcpyAsync(&h[1], pos + 2 * nnz + 1, 8, hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipStreamSynchronize(st));
    *n_local = h[0];
    *n_non_local = h[1];
    *state = pos;
    return GKOC_OK;
}

template <typename L, typename G, typename W>
int separate_fill_w(hipStream_t st, int64_t nnz, const G* rows, const G* cols, const void* vals,
                    const gkoc_partition* rp, const gkoc_partition* cp, const int64_t* pos, L* l_row, L* l_col,
                    void* l_val, L* n_row, G* n_col, void* n_val)
{
    separate_fill_kernel<L, G, W><<<dim3(grid_of(nnz)), dim3(256), 0, st>>>(
        nnz, rows, cols, static_cast<const W*>(vals), view_of<L, G>(rp), view_of<L, G>(cp), pos, pos + nnz + 1,
        l_row, l_col, static_cast<W*>(l_val), n_row, n_col, static_cast<W*>(n_val));
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename L, typename G>
int separate_fill(gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const void* vals,
                  size_t value_size, const gkoc_partition* rp, const gkoc_partition* cp, void* state, L* l_row,
                  L* l_col, void* l_val, L* n_row, G* n_col, void* n_val)
{
    if (nnz <= 0 || !state) return GKOC_OK;
    hipStream_t st = as_stream(s);
    const int64_t* pos = static_cast<const int64_t*>(state);
    int rc = GKOC_E_NOT_SUPPORTED;
    if (value_size == 4) {
        rc = separate_fill_w<L, G, uint32_t>(st, nnz, rows, cols, vals, rp, cp, pos, l_row, l_col, l_val, n_row,
                                             n_col, n_val);
    } else if (value_size == 8) {
        rc = separate_fill_w<L, G, uint64_t>(st, nnz, rows, cols, vals, rp, cp, pos, l_row, l_col, l_val, n_row,
                                             n_col, n_val);
    } else if (value_size == 16) {
        rc = separate_fill_w<L, G, w16>(st, nnz, rows, cols, vals, rp, cp, pos, l_row, l_col, l_val, n_row, n_col,
                                        n_val);
    } else {
        set_last_error("separate_local_nonlocal: value size %zu", value_size);
    }
    (void)scratch_free(st, state);
    return rc;
}

// ======================================================================= distributed_vector
template <typename L, typename G, typename W>
__global__ __launch_bounds__(256) void build_local_kernel(int64_t nnz, const G* __restrict__ rows,
                                                         const G* __restrict__ cols,
                                                         const W* __restrict__ vals, part_view<L, G> p,
                                                         int32_t local_part, W* __restrict__ out, int64_t ld)
{
    GKOC_GRID_STRIDE(i, nnz)
    {
        const G gr = rows[i];
        const int64_t rr = find_range(gr, p.bounds, p.num_ranges);
        if (p.pids[rr] == local_part) {
            const int64_t lr = int64_t(L(gr - p.bounds[rr]) + p.starts[rr]);
            out[lr * ld + int64_t(L(cols[i]))] = vals[i];
        }
    }
}

template <typename L, typename G>
int vector_build_local(gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const void* vals,
                       size_t value_size, const gkoc_partition* p, int32_t local_part, void* out, int64_t ld)
{
    GKOC_REQUIRE(p && nnz >= 0, GKOC_E_INVALID, "bad argument");
    if (nnz == 0) return GKOC_OK;
    hipStream_t st = as_stream(s);
    const dim3 g(grid_of(nnz));
    if (value_size == 4) {
        build_local_kernel<L, G, uint32_t><<<g, dim3(256), 0, st>>>(
            nnz, rows, cols, static_cast<const uint32_t*>(vals), view_of<L, G>(p), local_part,
            static_cast<uint32_t*>(out), ld);
    } else if (value_size == 8) {
        build_local_kernel<L, G, uint64_t><<<g, dim3(256), 0, st>>>(
            nnz, rows, cols, static_cast<const uint64_t*>(vals), view_of<L, G>(p), local_part,
            static_cast<uint64_t*>(out), ld);
    } else if (value_size == 16) {
        build_local_kernel<L, G, w16><<<g, dim3(256), 0, st>>>(nnz, rows, cols, static_cast<const w16*>(vals),
                                                               view_of<L, G>(p), local_part,
                                                               static_cast<w16*>(out), ld);
    } else {
        set_last_error("build_local: value size %zu", value_size);
        return GKOC_E_NOT_SUPPORTED;
    }
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// ======================================================================= index_map
template <typename G>
__global__ __launch_bounds__(256) void part_of_kernel(int64_t n, const G* __restrict__ gids,
                                                     const G* __restrict__ bounds, int64_t num_ranges,
                                                     const int32_t* __restrict__ pids,
                                                     int32_t* __restrict__ out)
{
    GKOC_GRID_STRIDE(i, n) out[i] = pids[find_range(gids[i], bounds, num_ranges)];
}

template <typename G>
__global__ __launch_bounds__(256) void gather_g_kernel(int64_t n, const int64_t* __restrict__ perm,
                                                      const G* __restrict__ in, G* __restrict__ out)
{
    GKOC_GRID_STRIDE(i, n) out[i] = in[perm[i]];
}

// heads of the runs of equal (part, gid) and of equal part, on the array sorted by (part, gid)
template <typename G>
__global__ __launch_bounds__(256) void mapping_mark_kernel(int64_t n, const int32_t* __restrict__ parts,
                                                          const G* __restrict__ gids,
                                                          int64_t* __restrict__ pos_u,
                                                          int64_t* __restrict__ pos_p)
{
    GKOC_GRID_STRIDE(i, n + 1)
    {
        int64_t u = 0, p = 0;
        if (i < n) {
            const bool newp = i == 0 || parts[i] != parts[i - 1];
            u = (newp || gids[i] != gids[i - 1]) ? 1 : 0;
            p = newp ? 1 : 0;
        }
        pos_u[i] = u;
        pos_p[i] = p;
    }
}

template <typename L, typename G>
__global__ __launch_bounds__(256) void mapping_fill_kernel(
    int64_t n, const int32_t* __restrict__ parts, const G* __restrict__ gids,
    const int64_t* __restrict__ pos_u, const int64_t* __restrict__ pos_p, part_view<L, G> p,
    int32_t* __restrict__ part_ids_out, L* __restrict__ remote_local, G* __restrict__ remote_global,
    int64_t* __restrict__ part_start)
{
    GKOC_GRID_STRIDE(i, n)
    {
        const int64_t ou = pos_u[i];
        if (pos_u[i + 1] != ou) {
            const G g = gids[i];
            const int64_t rr = find_range(g, p.bounds, p.num_ranges);
            remote_global[ou] = g;
            remote_local[ou] = L(g - p.bounds[rr]) + p.starts[rr];
        }
        const int64_t op = pos_p[i];
        if (pos_p[i + 1] != op) {
            part_ids_out[op] = parts[i];
            part_start[op] = ou;      // the first unique index of the op-th part that occurs
        }
    }
}

// number of unique indices per occurring part
__global__ __launch_bounds__(256) void mapping_sizes_kernel(int64_t n_part_unique, int64_t n_unique,
                                                           const int64_t* __restrict__ part_start,
                                                           int64_t* __restrict__ sizes)
{
    GKOC_GRID_STRIDE(k, n_part_unique)
    sizes[k] = (k + 1 < n_part_unique ? part_start[k + 1] : n_unique) - part_start[k];
}

struct mapping_state {
    int64_t n, n_unique, n_part_unique;
    int32_t* parts;     // sorted
    void* gids;         // sorted
    int64_t* pos_u;
    int64_t* pos_p;
};

template <typename L, typename G>
int build_mapping_count(gkoc_stream_t s, int64_t n, const G* recv, const gkoc_partition* part, void** state,
                        int64_t* n_unique, int64_t* n_part_unique)
{
    GKOC_REQUIRE(state && n_unique && n_part_unique && part && n >= 0, GKOC_E_INVALID, "bad argument");
    *state = nullptr;
    *n_unique = *n_part_unique = 0;
    if (n == 0) return GKOC_OK;
    hipStream_t st = as_stream(s);
    auto pv = view_of<L, G>(part);
    // 1. sort the indices; 2. their parts; 3. stable sort by part: order (part, gid)
    G *g1 = nullptr, *g2 = nullptr;
    int64_t *perm0 = nullptr, *perm1 = nullptr, *perm2 = nullptr;
    int32_t *p1 = nullptr, *p2 = nullptr;
    scratch_set tmp(st);
    GKOC_TRY(tmp.get(&g1, n));
    GKOC_TRY(tmp.get(&perm0, n));
    GKOC_TRY(tmp.get(&perm1, n));
    GKOC_TRY(tmp.get(&p1, n));
    GKOC_TRY(tmp.get(&perm2, n));
    // these two stay alive until the fill call
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&g2), size_t(n) * sizeof(G)));
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&p2), size_t(n) * sizeof(int32_t)));
    const dim3 g(grid_of(n));
    iota_kernel2<int64_t><<<g, dim3(256), 0, st>>>(n, perm0);
    GKOC_LAUNCH_OK();
    GKOC_TRY((stable_sort_pairs<G, int64_t>(st, n, recv, g1, perm0, perm1)));
    part_of_kernel<G><<<g, dim3(256), 0, st>>>(n, g1, pv.bounds, pv.num_ranges, pv.pids, p1);
    GKOC_LAUNCH_OK();
    GKOC_TRY((stable_sort_pairs<int32_t, int64_t>(st, n, p1, p2, perm0, perm2)));
    gather_g_kernel<G><<<g, dim3(256), 0, st>>>(n, perm2, g1, g2);
    GKOC_LAUNCH_OK();
    tmp.release();
    int64_t* pos = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&pos), size_t(2 * (n + 1)) * sizeof(int64_t)));
    mapping_mark_kernel<G><<<dim3(grid_of(n + 1)), dim3(256), 0, st>>>(n, p2, g2, pos, pos + n + 1);
    GKOC_LAUNCH_OK();
    GKOC_TRY(exclusive_scan_inplace<int64_t>(st, pos, n + 1));
    GKOC_TRY(exclusive_scan_inplace<int64_t>(st, pos + n + 1, n + 1));
    int64_t h[2];
    GKOC_HIP(hipMemcpyAsync(&h[0], pos + n, 8, hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipMemcpyAsync(&h[1], pos + 2 * n + 1, 8, hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipStreamSynchronize(st));
    auto* ms = new mapping_state{n, h[0], h[1], p2, g2, pos, pos + n + 1};
    *n_unique = h[0];
    *n_part_unique = h[1];
    *state = ms;
    return GKOC_OK;
}

template <typename L, typename G>
int build_mapping_fill(gkoc_stream_t s, const gkoc_partition* part, void* state, int32_t* part_ids_out,
                       L* remote_local, G* remote_global, int64_t* remote_sizes)
{
    if (!state) return GKOC_OK;
    hipStream_t st = as_stream(s);
    auto* ms = static_cast<mapping_state*>(state);
    int rc = GKOC_OK;
    int64_t* starts_next = nullptr;
    rc = scratch_malloc(st, reinterpret_cast<void**>(&starts_next), size_t(ms->n_part_unique + 1) * 8);
    if (rc == GKOC_OK) {
        mapping_fill_kernel<L, G><<<dim3(grid_of(ms->n)), dim3(256), 0, st>>>(
            ms->n, ms->parts, static_cast<const G*>(ms->gids), ms->pos_u, ms->pos_p, view_of<L, G>(part),
            part_ids_out, remote_local, remote_global, starts_next);
        mapping_sizes_kernel<<<dim3(grid_of(ms->n_part_unique)), dim3(256), 0, st>>>(
            ms->n_part_unique, ms->n_unique, starts_next, remote_sizes);
        if (hipGetLastError() != hipSuccess) rc = GKOC_E_INVALID;
        (void)scratch_free(st, starts_next);
    }
    (void)scratch_free(st, ms->parts);
    (void)scratch_free(st, ms->gids);
    (void)scratch_free(st, ms->pos_u);
    delete ms;
    return rc;
}

template <typename T>
__device__ __forceinline__ int64_t lower_bound_dev(const T* __restrict__ a, int64_t n, T v)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (a[mid] < v) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    return lo;
}

// index_space: 0 local, 1 non_local, 2 combined (include/ginkgo/core/distributed/index_map.hpp)
template <typename L, typename G>
__global__ __launch_bounds__(256) void map_to_local_kernel(
    int64_t n, const G* __restrict__ gids, part_view<L, G> p, int64_t n_targets,
    const int32_t* __restrict__ target_ids, const G* __restrict__ remote_flat,
    const int64_t* __restrict__ remote_offsets, int32_t rank, int is, L* __restrict__ out)
{
    const L invalid = L(-1);
    GKOC_GRID_STRIDE(i, n)
    {
        const G gid = gids[i];
        const int64_t rr = find_range(gid, p.bounds, p.num_ranges);
        const int32_t pid = p.pids[rr];
        L res = invalid;
        const bool want_local = is == 0 || (is == 2 && pid == rank);
        if (want_local) {
            if (pid == rank) res = L(gid - p.bounds[rr]) + p.starts[rr];
        } else {
            const int64_t set = lower_bound_dev<int32_t>(target_ids, n_targets, pid);
            if (set < n_targets) {
                // (as the reference: the segment of the first target id >= pid is searched)
                const int64_t b = remote_offsets[set], e = remote_offsets[set + 1];
                const int64_t k = b + lower_bound_dev<G>(remote_flat + b, e - b, gid);
                if (k < e && remote_flat[k] == gid) {
                    res = L(k);
                    if (is == 2) res += p.sizes[rank];
                }
            }
        }
        out[i] = res;
    }
}

template <typename L, typename G>
__global__ __launch_bounds__(256) void map_to_global_kernel(
    int64_t n, const L* __restrict__ lids, const G* __restrict__ bounds, const L* __restrict__ starts,
    L local_size, const uint64_t* __restrict__ local_ranges, int64_t n_local_ranges,
    const G* __restrict__ remote_flat, int64_t remote_size, int is, G* __restrict__ out)
{
    const G invalid = G(-1);
    GKOC_GRID_STRIDE(i, n)
    {
        L lid = lids[i];
        G res = invalid;
        bool local = is == 0;
        if (is == 2) {
            if (lid < local_size) {
                local = true;
            } else {
                lid -= local_size;
            }
        }
        if (local) {
            if (lid >= 0 && lid < local_size) {
                // find_local_range: the last local range whose starting index is <= lid
                int64_t lo = 0, hi = n_local_ranges;
                while (lo < hi) {   // upper_bound over starts[local_ranges[.]]
                    const int64_t mid = lo + (hi - lo) / 2;
                    if (lid < starts[local_ranges[mid]]) {
                        hi = mid;
                    } else {
                        lo = mid + 1;
                    }
                }
                const uint64_t rid = local_ranges[lo - 1];
                res = G(lid - starts[rid]) + bounds[rid];
            }
        } else {
            if (lid >= 0 && int64_t(lid) < remote_size) res = remote_flat[lid];
        }
        out[i] = res;
    }
}

// ======================================================================= partition
__global__ __launch_bounds__(256) void count_ranges_kernel(int64_t n, const int32_t* __restrict__ mapping,
                                                          unsigned long long* __restrict__ count)
{
    unsigned long long c = 0;
    GKOC_GRID_STRIDE(i, n) c += (i == 0 ? int32_t(-1) : mapping[i - 1]) != mapping[i] ? 1 : 0;
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

template <typename G>
__global__ __launch_bounds__(256) void from_contiguous_kernel(int64_t num_ranges, const G* __restrict__ ranges,
                                                             const int32_t* __restrict__ mapping,
                                                             G* __restrict__ bounds, int32_t* __restrict__ pids)
{
    GKOC_GRID_STRIDE(i, num_ranges)
    {
        if (i == 0) bounds[0] = 0;
        bounds[i + 1] = ranges[i + 1];
        pids[i] = mapping ? mapping[i] : int32_t(i);
    }
}

__global__ __launch_bounds__(256) void mapping_heads_kernel(int64_t n, const int32_t* __restrict__ mapping,
                                                           int64_t* __restrict__ pos)
{
    GKOC_GRID_STRIDE(i, n + 1)
    pos[i] = (i < n && (i == 0 ? int32_t(-1) : mapping[i - 1]) != mapping[i]) ? 1 : 0;
}

template <typename G>
__global__ __launch_bounds__(256) void from_mapping_fill_kernel(int64_t n, const int32_t* __restrict__ mapping,
                                                               const int64_t* __restrict__ pos,
                                                               G* __restrict__ bounds, int32_t* __restrict__ pids)
{
    GKOC_GRID_STRIDE(i, n + 1)
    {
        if (i == n) {
            bounds[pos[n]] = G(n);
        } else if (pos[i + 1] != pos[i]) {
            bounds[pos[i]] = G(i);
            pids[pos[i]] = mapping[i];
        }
    }
}

template <typename G>
__global__ __launch_bounds__(256) void ranges_from_size_kernel(int64_t num_parts, G global_size,
                                                              G* __restrict__ ranges)
{
    const G per = global_size / G(num_parts);
    const G rest = global_size - G(num_parts) * per;
    GKOC_GRID_STRIDE(i, num_parts + 1) ranges[i] = G(i) * per + (G(i) < rest ? G(i) : rest);
}

// ranks / sizes of build_starting_indices, on the ranges sorted (stably) by part
template <typename L, typename G>
__global__ __launch_bounds__(256) void range_sizes_kernel(int64_t num_ranges, const int64_t* __restrict__ order,
                                                         const G* __restrict__ offsets, int64_t* __restrict__ sz)
{
    GKOC_GRID_STRIDE(i, num_ranges + 1)
    sz[i] = i < num_ranges ? int64_t(offsets[order[i] + 1] - offsets[order[i]]) : 0;
}

__global__ __launch_bounds__(256) void part_first_kernel(int64_t num_ranges, const int32_t* __restrict__ parts_sorted,
                                                        const int64_t* __restrict__ scan, int64_t* __restrict__ first)
{
    GKOC_GRID_STRIDE(i, num_ranges)
    if (i == 0 || parts_sorted[i] != parts_sorted[i - 1]) first[parts_sorted[i]] = scan[i];
}

template <typename L>
__global__ __launch_bounds__(256) void starting_indices_kernel(
    int64_t num_ranges, const int32_t* __restrict__ parts_sorted, const int64_t* __restrict__ order,
    const int64_t* __restrict__ scan, const int64_t* __restrict__ first, L* __restrict__ ranks, L* __restrict__ sizes)
{
    GKOC_GRID_STRIDE(i, num_ranges)
    {
        const int32_t p = parts_sorted[i];
        ranks[order[i]] = L(scan[i] - first[p]);
        if (i + 1 == num_ranges || parts_sorted[i + 1] != p) sizes[p] = L(scan[i + 1] - first[p]);
    }
}

template <typename L>
__global__ __launch_bounds__(256) void count_zero_kernel(int64_t n, const L* __restrict__ v,
                                                        unsigned long long* __restrict__ count)
{
    unsigned long long c = 0;
    GKOC_GRID_STRIDE(i, n) c += v[i] == 0 ? 1 : 0;
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

__global__ __launch_bounds__(256) void histogram_kernel(int64_t n, const int32_t* __restrict__ keys,
                                                       unsigned long long* __restrict__ hist)
{
    GKOC_GRID_STRIDE(i, n) atomicAdd(&hist[keys[i]], 1ull);
}

__global__ __launch_bounds__(256) void unordered_kernel(int64_t n, const int32_t* __restrict__ v, int* __restrict__ flag)
{
    GKOC_GRID_STRIDE(i, n)
    if (i > 0 && v[i] < v[i - 1]) *flag = 1;
}

// ======================================================================= partition_helpers
template <typename G>
__global__ __launch_bounds__(256) void range_starts_kernel(int64_t n, const G* __restrict__ start_ends,
                                                          G* __restrict__ starts)
{
    GKOC_GRID_STRIDE(i, n) starts[i] = start_ends[2 * i];
}

template <typename G>
__global__ __launch_bounds__(256) void permute_ranges_kernel(int64_t n, const int64_t* __restrict__ perm,
                                                            const G* __restrict__ se_in,
                                                            const int32_t* __restrict__ pid_in,
                                                            G* __restrict__ se_out, int32_t* __restrict__ pid_out)
{
    GKOC_GRID_STRIDE(i, n)
    {
        const int64_t k = perm[i];
        se_out[2 * i] = se_in[2 * k];
        se_out[2 * i + 1] = se_in[2 * k + 1];
        pid_out[i] = pid_in[k];
    }
}

template <typename G>
__global__ __launch_bounds__(256) void not_consecutive_kernel(int64_t num_parts, const G* __restrict__ se,
                                                             int* __restrict__ flag)
{
    GKOC_GRID_STRIDE(i, num_parts - 1)
    if (se[2 * i + 2] != se[2 * i + 1]) *flag = 1;
}

template <typename G>
__global__ __launch_bounds__(256) void compress_ranges_kernel(int64_t n_offsets, const G* __restrict__ se,
                                                             G* __restrict__ offsets)
{
    GKOC_GRID_STRIDE(i, n_offsets) offsets[i] = i == 0 ? se[0] : se[2 * (i - 1) + 1];
}

// ======================================================================= assembly
template <typename L, typename G>
__global__ __launch_bounds__(256) void non_owning_mark_kernel(int64_t nnz, const G* __restrict__ rows,
                                                             part_view<L, G> p, int32_t local_part,
                                                             int32_t* __restrict__ key, G* __restrict__ orig,
                                                             int32_t* __restrict__ send_count)
{
    GKOC_GRID_STRIDE(i, nnz)
    {
        const int32_t pid = p.pids[find_range(rows[i], p.bounds, p.num_ranges)];
        if (pid != local_part) {
            atomicAdd(&send_count[pid], 1);
            orig[i] = G(i);
            key[i] = pid;
        } else {
            orig[i] = G(-1);
            key[i] = local_part;     // the comparison of the reference's stable_sort
        }
    }
}

template <typename G>
__global__ __launch_bounds__(256) void send_flags_kernel(int64_t nnz, const G* __restrict__ orig_sorted,
                                                        G* __restrict__ send_pos)
{
    GKOC_GRID_STRIDE(i, nnz) send_pos[i] = orig_sorted[i] == G(-1) ? G(0) : G(1);
}

template <typename G, typename W>
__global__ __launch_bounds__(256) void fill_send_kernel(int64_t nnz, const G* __restrict__ rows,
                                                       const G* __restrict__ cols, const W* __restrict__ vals,
                                                       const G* __restrict__ send_pos, const G* __restrict__ orig,
                                                       G* __restrict__ srow, G* __restrict__ scol, W* __restrict__ sval)
{
    GKOC_GRID_STRIDE(i, nnz)
    {
        const G in = orig[i];
        if (in >= 0) {
            const G o = send_pos[i];
            srow[o] = rows[in];
            scol[o] = cols[in];
            sval[o] = vals[in];
        }
    }
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

// ------------------------------------------------------------------ C ABI: (L, G) pairs
#define GKOC_DEF_DIST_LG(L, LN, G, GN)                                                                        \
    extern "C" int gkoc_dist_separate_local_nonlocal_count_##LN##_##GN(                                       \
        gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const gkoc_partition* row_part,           \
        const gkoc_partition* col_part, int32_t local_part, void** state, int64_t* n_local,                   \
        int64_t* n_non_local)                                                                                 \
    {                                                                                                         \
        return separate_count<L, G>(s, nnz, rows, cols, row_part, col_part, local_part, state, n_local,       \
                                    n_non_local);                                                             \
    }                                                                                                         \
    extern "C" int gkoc_dist_separate_local_nonlocal_fill_##LN##_##GN(                                        \
        gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const void* vals, size_t value_size,      \
        const gkoc_partition* row_part, const gkoc_partition* col_part, void* state, L* local_rows,           \
        L* local_cols, void* local_vals, L* non_local_rows, G* non_local_cols, void* non_local_vals)          \
    {                                                                                                         \
        return separate_fill<L, G>(s, nnz, rows, cols, vals, value_size, row_part, col_part, state,           \
                                   local_rows, local_cols, local_vals, non_local_rows, non_local_cols,        \
                                   non_local_vals);                                                           \
    }                                                                                                         \
    extern "C" int gkoc_dist_vector_build_local_##LN##_##GN(                                                  \
        gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const void* vals, size_t value_size,      \
        const gkoc_partition* part, int32_t local_part, void* local_values, int64_t ld)                       \
    {                                                                                                         \
        return vector_build_local<L, G>(s, nnz, rows, cols, vals, value_size, part, local_part,               \
                                        local_values, ld);                                                    \
    }                                                                                                         \
    extern "C" int gkoc_index_map_build_mapping_count_##LN##_##GN(                                            \
        gkoc_stream_t s, int64_t n, const G* recv_connections, const gkoc_partition* part, void** state,      \
        int64_t* n_unique, int64_t* n_part_unique)                                                            \
    {                                                                                                         \
        return build_mapping_count<L, G>(s, n, recv_connections, part, state, n_unique, n_part_unique);       \
    }                                                                                                         \
    extern "C" int gkoc_index_map_build_mapping_fill_##LN##_##GN(                                             \
        gkoc_stream_t s, const gkoc_partition* part, void* state, int32_t* part_ids, L* remote_local_idxs,    \
        G* remote_global_idxs, int64_t* remote_sizes)                                                         \
    {                                                                                                         \
        return build_mapping_fill<L, G>(s, part, state, part_ids, remote_local_idxs, remote_global_idxs,      \
                                        remote_sizes);                                                        \
    }                                                                                                         \
    extern "C" int gkoc_index_map_map_to_local_##LN##_##GN(                                                   \
        gkoc_stream_t s, int64_t n, const G* global_ids, const gkoc_partition* part, int64_t n_targets,       \
        const int32_t* remote_target_ids, const G* remote_global_flat, const int64_t* remote_offsets,         \
        int32_t rank, int index_space, L* local_ids)                                                          \
    {                                                                                                         \
        GKOC_REQUIRE(part && n >= 0 && index_space >= 0 && index_space <= 2, GKOC_E_INVALID, "bad argument"); \
        if (n == 0) return GKOC_OK;                                                                           \
        map_to_local_kernel<L, G><<<dim3(grid_of(n)), dim3(256), 0, as_stream(s)>>>(                          \
            n, global_ids, view_of<L, G>(part), n_targets, remote_target_ids, remote_global_flat,             \
            remote_offsets, rank, index_space, local_ids);                                                    \
        GKOC_LAUNCH_OK();                                                                                     \
        return GKOC_OK;                                                                                       \
    }                                                                                                         \
    extern "C" int gkoc_index_map_map_to_global_##LN##_##GN(                                                  \
        gkoc_stream_t s, int64_t n, const L* local_ids, const G* range_bounds, const L* starting_indices,     \
        int64_t local_size, const uint64_t* local_ranges, int64_t n_local_ranges, const G* remote_global_flat, \
        int64_t remote_size, int index_space, G* global_ids)                                                  \
    {                                                                                                         \
        GKOC_REQUIRE(n >= 0 && index_space >= 0 && index_space <= 2, GKOC_E_INVALID, "bad argument");         \
        if (n == 0) return GKOC_OK;                                                                           \
        map_to_global_kernel<L, G><<<dim3(grid_of(n)), dim3(256), 0, as_stream(s)>>>(                         \
            n, local_ids, range_bounds, starting_indices, L(local_size), local_ranges, n_local_ranges,        \
            remote_global_flat, remote_size, index_space, global_ids);                                        \
        GKOC_LAUNCH_OK();                                                                                     \
        return GKOC_OK;                                                                                       \
    }                                                                                                         \
    extern "C" int gkoc_partition_build_starting_indices_##LN##_##GN(                                         \
        gkoc_stream_t s, const G* range_offsets, const int32_t* range_parts, int64_t num_ranges,              \
        int32_t num_parts, int32_t* num_empty_parts, L* ranks, L* sizes)                                      \
    {                                                                                                         \
        GKOC_REQUIRE(num_empty_parts && num_ranges >= 0 && num_parts >= 0, GKOC_E_INVALID, "bad argument");   \
        hipStream_t st = as_stream(s);                                                                        \
        *num_empty_parts = num_parts;                                                                         \
        if (num_parts == 0) return GKOC_OK;                                                                   \
        GKOC_HIP(hipMemsetAsync(sizes, 0, size_t(num_parts) * sizeof(L), st));                                \
        if (num_ranges > 0) {                                                                                 \
            scratch_set tmp(st);                                                                              \
            int64_t *iota = nullptr, *order = nullptr, *sz = nullptr, *first = nullptr;                       \
            int32_t* ps = nullptr;                                                                            \
            GKOC_TRY(tmp.get(&iota, num_ranges));                                                             \
            GKOC_TRY(tmp.get(&order, num_ranges));                                                            \
            GKOC_TRY(tmp.get(&ps, num_ranges));                                                               \
            GKOC_TRY(tmp.get(&sz, num_ranges + 1));                                                           \
            GKOC_TRY(tmp.get(&first, num_parts));                                                             \
            const dim3 g(grid_of(num_ranges + 1));                                                            \
            iota_kernel2<int64_t><<<g, dim3(256), 0, st>>>(num_ranges, iota);                                 \
            GKOC_TRY((stable_sort_pairs<int32_t, int64_t>(st, num_ranges, range_parts, ps, iota, order)));    \
            range_sizes_kernel<L, G><<<g, dim3(256), 0, st>>>(num_ranges, order, range_offsets, sz);          \
            GKOC_TRY(exclusive_scan_inplace<int64_t>(st, sz, num_ranges + 1));                                \
            part_first_kernel<<<g, dim3(256), 0, st>>>(num_ranges, ps, sz, first);                            \
            starting_indices_kernel<L><<<g, dim3(256), 0, st>>>(num_ranges, ps, order, sz, first, ranks,      \
                                                                sizes);                                       \
            GKOC_LAUNCH_OK();                                                                                 \
            tmp.release();                                                                                    \
        }                                                                                                     \
        unsigned long long* cnt = nullptr;                                                                    \
        GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&cnt), 8));                                      \
        GKOC_HIP(hipMemsetAsync(cnt, 0, 8, st));                                                              \
        count_zero_kernel<L><<<dim3(grid_of(num_parts)), dim3(256), 0, st>>>(num_parts, sizes, cnt);          \
        unsigned long long h = 0;                                                                             \
        int rc = to_host(st, &h, cnt, 8);                                                                     \
        (void)scratch_free(st, cnt);                                                                          \
        *num_empty_parts = int32_t(h);                                                                        \
        return rc;                                                                                            \
    }                                                                                                         \
    extern "C" int gkoc_assembly_count_non_owning_entries_##LN##_##GN(                                        \
        gkoc_stream_t s, int64_t nnz, const G* rows, const gkoc_partition* part, int32_t local_part,          \
        int32_t* send_count, G* send_positions, G* original_positions)                                        \
    {                                                                                                         \
        GKOC_REQUIRE(part && nnz >= 0, GKOC_E_INVALID, "bad argument");                                       \
        if (nnz == 0) return GKOC_OK;                                                                         \
        hipStream_t st = as_stream(s);                                                                        \
        scratch_set tmp(st);                                                                                  \
        int32_t *key = nullptr, *key2 = nullptr;                                                              \
        G* orig = nullptr;                                                                                    \
        GKOC_TRY(tmp.get(&key, nnz));                                                                         \
        GKOC_TRY(tmp.get(&key2, nnz));                                                                        \
        GKOC_TRY(tmp.get(&orig, nnz));                                                                        \
        const dim3 g(grid_of(nnz));                                                                           \
        non_owning_mark_kernel<L, G><<<g, dim3(256), 0, st>>>(nnz, rows, view_of<L, G>(part), local_part,     \
                                                              key, orig, send_count);                         \
        GKOC_LAUNCH_OK();                                                                                     \
        GKOC_TRY((stable_sort_pairs<int32_t, G>(st, nnz, key, key2, orig, original_positions)));              \
        send_flags_kernel<G><<<g, dim3(256), 0, st>>>(nnz, original_positions, send_positions);               \
        GKOC_LAUNCH_OK();                                                                                     \
        GKOC_TRY(exclusive_scan_inplace<G>(st, send_positions, nnz));                                         \
        tmp.release();                                                                                        \
        return GKOC_OK;                                                                                       \
    }

GKOC_DEF_DIST_LG(int32_t, i32, int32_t, i32)
GKOC_DEF_DIST_LG(int32_t, i32, int64_t, i64)
GKOC_DEF_DIST_LG(int64_t, i64, int64_t, i64)

// ------------------------------------------------------------------ C ABI: global index type only
#define GKOC_DEF_DIST_G(G, GN)                                                                                \
    extern "C" int gkoc_partition_build_from_contiguous_##GN(                                                 \
        gkoc_stream_t s, int64_t num_ranges, const G* ranges, const int32_t* part_id_mapping,                 \
        G* range_bounds, int32_t* part_ids)                                                                   \
    {                                                                                                         \
        hipStream_t st = as_stream(s);                                                                        \
        if (num_ranges <= 0) {                                                                                \
            GKOC_HIP(hipMemsetAsync(range_bounds, 0, sizeof(G), st));                                         \
            return GKOC_OK;                                                                                   \
        }                                                                                                     \
        from_contiguous_kernel<G><<<dim3(grid_of(num_ranges)), dim3(256), 0, st>>>(                           \
            num_ranges, ranges, part_id_mapping, range_bounds, part_ids);                                     \
        GKOC_LAUNCH_OK();                                                                                     \
        return GKOC_OK;                                                                                       \
    }                                                                                                         \
    extern "C" int gkoc_partition_build_from_mapping_##GN(gkoc_stream_t s, int64_t n, const int32_t* mapping, \
                                                          G* range_bounds, int32_t* part_ids)                 \
    {                                                                                                         \
        GKOC_REQUIRE(n >= 0, GKOC_E_INVALID, "negative size");                                                \
        hipStream_t st = as_stream(s);                                                                        \
        if (n == 0) {                                                                                         \
            GKOC_HIP(hipMemsetAsync(range_bounds, 0, sizeof(G), st));                                         \
            return GKOC_OK;                                                                                   \
        }                                                                                                     \
        int64_t* pos = nullptr;                                                                               \
        GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&pos), size_t(n + 1) * 8));                      \
        mapping_heads_kernel<<<dim3(grid_of(n + 1)), dim3(256), 0, st>>>(n, mapping, pos);                    \
        GKOC_LAUNCH_OK();                                                                                     \
        GKOC_TRY(exclusive_scan_inplace<int64_t>(st, pos, n + 1));                                            \
        from_mapping_fill_kernel<G><<<dim3(grid_of(n + 1)), dim3(256), 0, st>>>(n, mapping, pos,              \
                                                                               range_bounds, part_ids);      \
        GKOC_LAUNCH_OK();                                                                                     \
        return scratch_free(st, pos);                                                                         \
    }                                                                                                         \
    extern "C" int gkoc_partition_build_ranges_from_global_size_##GN(gkoc_stream_t s, int32_t num_parts,      \
                                                                     G global_size, G* ranges)                \
    {                                                                                                         \
        GKOC_REQUIRE(num_parts > 0, GKOC_E_INVALID, "num_parts must be positive");                            \
        ranges_from_size_kernel<G><<<dim3(grid_of(num_parts + 1)), dim3(256), 0, as_stream(s)>>>(             \
            num_parts, global_size, ranges);                                                                  \
        GKOC_LAUNCH_OK();                                                                                     \
        return GKOC_OK;                                                                                       \
    }                                                                                                         \
    extern "C" int gkoc_partition_helpers_sort_by_range_start_##GN(gkoc_stream_t s, int64_t num_parts,        \
                                                                   G* range_start_ends, int32_t* part_ids)    \
    {                                                                                                         \
        if (num_parts <= 1) return GKOC_OK;                                                                   \
        hipStream_t st = as_stream(s);                                                                        \
        scratch_set tmp(st);                                                                                  \
        G *starts = nullptr, *starts2 = nullptr, *se = nullptr;                                               \
        int64_t *iota = nullptr, *perm = nullptr;                                                             \
        int32_t* pid = nullptr;                                                                               \
        GKOC_TRY(tmp.get(&starts, num_parts));                                                                \
        GKOC_TRY(tmp.get(&starts2, num_parts));                                                               \
        GKOC_TRY(tmp.get(&se, 2 * num_parts));                                                                \
        GKOC_TRY(tmp.get(&iota, num_parts));                                                                  \
        GKOC_TRY(tmp.get(&perm, num_parts));                                                                  \
        GKOC_TRY(tmp.get(&pid, num_parts));                                                                   \
        const dim3 g(grid_of(num_parts));                                                                     \
        range_starts_kernel<G><<<g, dim3(256), 0, st>>>(num_parts, range_start_ends, starts);                 \
        iota_kernel2<int64_t><<<g, dim3(256), 0, st>>>(num_parts, iota);                                      \
        GKOC_TRY((stable_sort_pairs<G, int64_t>(st, num_parts, starts, starts2, iota, perm)));                \
        GKOC_HIP(hipMemcpyAsync(se, range_start_ends, size_t(2 * num_parts) * sizeof(G),                      \
                                hipMemcpyDeviceToDevice, st));                                                \
        GKOC_HIP(hipMemcpyAsync(pid, part_ids, size_t(num_parts) * 4, hipMemcpyDeviceToDevice, st));          \
        permute_ranges_kernel<G><<<g, dim3(256), 0, st>>>(num_parts, perm, se, pid, range_start_ends,         \
                                                          part_ids);                                          \
        GKOC_LAUNCH_OK();                                                                                     \
        tmp.release();                                                                                        \
        return GKOC_OK;                                                                                       \
    }                                                                                                         \
    extern "C" int gkoc_partition_helpers_check_consecutive_ranges_##GN(gkoc_stream_t s, int64_t num_parts,   \
                                                                        const G* range_start_ends,            \
                                                                        int* result)                          \
    {                                                                                                         \
        GKOC_REQUIRE(result, GKOC_E_INVALID, "null result");                                                  \
        *result = 1;                                                                                          \
        if (num_parts <= 1) return GKOC_OK;                                                                   \
        hipStream_t st = as_stream(s);                                                                        \
        int* flag = nullptr;                                                                                  \
        GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&flag), 4));                                     \
        GKOC_HIP(hipMemsetAsync(flag, 0, 4, st));                                                             \
        not_consecutive_kernel<G><<<dim3(grid_of(num_parts)), dim3(256), 0, st>>>(num_parts,                  \
                                                                                 range_start_ends, flag);    \
        int h = 0;                                                                                            \
        int rc = to_host(st, &h, flag, 4);                                                                    \
        (void)scratch_free(st, flag);                                                                         \
        *result = h ? 0 : 1;                                                                                  \
        return rc;                                                                                            \
    }                                                                                                         \
    extern "C" int gkoc_partition_helpers_compress_ranges_##GN(gkoc_stream_t s, int64_t n_offsets,            \
                                                               const G* range_start_ends, G* range_offsets)   \
    {                                                                                                         \
        if (n_offsets <= 0) return GKOC_OK;                                                                   \
        compress_ranges_kernel<G><<<dim3(grid_of(n_offsets)), dim3(256), 0, as_stream(s)>>>(                  \
            n_offsets, range_start_ends, range_offsets);                                                      \
        GKOC_LAUNCH_OK();                                                                                     \
        return GKOC_OK;                                                                                       \
    }                                                                                                         \
    extern "C" int gkoc_assembly_fill_send_buffers_##GN(                                                      \
        gkoc_stream_t s, int64_t nnz, const G* rows, const G* cols, const void* vals, size_t value_size,      \
        const G* send_positions, const G* original_positions, G* send_rows, G* send_cols, void* send_vals)    \
    {                                                                                                         \
        if (nnz <= 0) return GKOC_OK;                                                                         \
        hipStream_t st = as_stream(s);                                                                        \
        const dim3 g(grid_of(nnz));                                                                           \
        if (value_size == 4) {                                                                                \
            fill_send_kernel<G, uint32_t><<<g, dim3(256), 0, st>>>(                                           \
                nnz, rows, cols, static_cast<const uint32_t*>(vals), send_positions, original_positions,      \
                send_rows, send_cols, static_cast<uint32_t*>(send_vals));                                     \
        } else if (value_size == 8) {                                                                         \
            fill_send_kernel<G, uint64_t><<<g, dim3(256), 0, st>>>(                                           \
                nnz, rows, cols, static_cast<const uint64_t*>(vals), send_positions, original_positions,      \
                send_rows, send_cols, static_cast<uint64_t*>(send_vals));                                     \
        } else if (value_size == 16) {                                                                        \
            fill_send_kernel<G, w16><<<g, dim3(256), 0, st>>>(nnz, rows, cols, static_cast<const w16*>(vals), \
                                                             send_positions, original_positions, send_rows,   \
                                                             send_cols, static_cast<w16*>(send_vals));        \
        } else {                                                                                              \
            set_last_error("fill_send_buffers: value size %zu", value_size);                                  \
            return GKOC_E_NOT_SUPPORTED;                                                                      \
        }                                                                                                     \
        GKOC_LAUNCH_OK();                                                                                     \
        return GKOC_OK;                                                                                       \
    }

GKOC_DEF_DIST_G(int32_t, i32)
GKOC_DEF_DIST_G(int64_t, i64)

extern "C" int gkoc_partition_count_ranges(gkoc_stream_t s, int64_t n, const int32_t* mapping, int64_t* num_ranges)
{
    GKOC_REQUIRE(num_ranges && n >= 0, GKOC_E_INVALID, "bad argument");
    *num_ranges = 0;
    if (n == 0) return GKOC_OK;
    hipStream_t st = as_stream(s);
    unsigned long long* cnt = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&cnt), 8));
    GKOC_HIP(hipMemsetAsync(cnt, 0, 8, st));
    count_ranges_kernel<<<dim3(grid_of(n)), dim3(256), 0, st>>>(n, mapping, cnt);
    unsigned long long h = 0;
    int rc = to_host(st, &h, cnt, 8);
    (void)scratch_free(st, cnt);
    *num_ranges = int64_t(h);
    return rc;
}

// range ids sorted by (part, range id) and the number of ranges of every part
extern "C" int gkoc_partition_build_ranges_by_part(gkoc_stream_t s, const int32_t* range_parts, int64_t num_ranges,
                                                   int32_t num_parts, uint64_t* range_ids, int64_t* sizes)
{
    GKOC_REQUIRE(num_ranges >= 0 && num_parts >= 0, GKOC_E_INVALID, "bad argument");
    hipStream_t st = as_stream(s);
    if (num_parts > 0) GKOC_HIP(hipMemsetAsync(sizes, 0, size_t(num_parts) * 8, st));
    if (num_ranges == 0) return GKOC_OK;
    scratch_set tmp(st);
    uint64_t* iota = nullptr;
    int32_t* ps = nullptr;
    GKOC_TRY(tmp.get(&iota, num_ranges));
    GKOC_TRY(tmp.get(&ps, num_ranges));
    iota_kernel2<uint64_t><<<dim3(grid_of(num_ranges)), dim3(256), 0, st>>>(num_ranges, iota);
    GKOC_TRY((stable_sort_pairs<int32_t, uint64_t>(st, num_ranges, range_parts, ps, iota, range_ids)));
    histogram_kernel<<<dim3(grid_of(num_ranges)), dim3(256), 0, st>>>(
        num_ranges, range_parts, reinterpret_cast<unsigned long long*>(sizes));
    GKOC_LAUNCH_OK();
    tmp.release();
    return GKOC_OK;
}

extern "C" int gkoc_partition_has_ordered_parts(gkoc_stream_t s, int64_t num_ranges, const int32_t* part_ids,
                                                int* result)
{
    GKOC_REQUIRE(result, GKOC_E_INVALID, "null result");
    *result = 1;
    if (num_ranges <= 1) return GKOC_OK;
    hipStream_t st = as_stream(s);
    int* flag = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&flag), 4));
    GKOC_HIP(hipMemsetAsync(flag, 0, 4, st));
    unordered_kernel<<<dim3(grid_of(num_ranges)), dim3(256), 0, st>>>(num_ranges, part_ids, flag);
    int h = 0;
    int rc = to_host(st, &h, flag, 4);
    (void)scratch_free(st, flag);
    *result = h ? 0 : 1;
    return rc;
}

// A count / fill pair that is abandoned between the two calls (the caller's resize threw): what the
// count call handed out as `state` goes back.  NULL is fine.
extern "C" int gkoc_dist_separate_state_free(gkoc_stream_t s, void* state)
{
    if (!state) return GKOC_OK;
    return gkoc::scratch_free(gkoc::as_stream(s), state);
}

extern "C" int gkoc_index_map_mapping_state_free(gkoc_stream_t s, void* state)
{
    if (!state) return GKOC_OK;
    auto* ms = static_cast<gkoc::mapping_state*>(state);
    hipStream_t st = gkoc::as_stream(s);
    (void)gkoc::scratch_free(st, ms->parts);
    (void)gkoc::scratch_free(st, ms->gids);
    (void)gkoc::scratch_free(st, ms->pos_u);
    delete ms;
    return GKOC_OK;
}
