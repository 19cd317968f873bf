// Rows far longer than the rest (hub rows of a power-law graph, a dense constraint row): csr::spmv,
// one right-hand side (round 5).
//
// The row-segment kernel (csr_spmv_pipe.hpp) gives 64 consecutive rows to ONE wave.  A row of 250 000
// entries in such a segment makes that wave stream the row twice, alone, while the other 5 000 resident
// waves have long finished: on the heavy-tailed stand-in of BASELINE configs[4] (ginkgo_amd/workloads.py
// irregular_rows: n = 4 M, 48 M entries, 16 rows beyond 4096, the longest 250 009) the product took 3.3 ms
// = 2.5 % of the HBM roofline - the time of its longest row.  The reference's answer is its load-balanced
// strategy (a host-built srow table that splits the ENTRY stream evenly over the warps, atomics on c:
// common/cuda_hip/matrix/csr_kernels.template.cpp:206-330, core/matrix/csr.cpp make_srow).  Here:
//
//   * the segments that contain a row longer than GKOC_CSR_LONG_ROW are FLAGGED (one bit per 64-row segment
//     + a list), once per matrix: csr_long_row_scan_kernel reads the row pointers (n x 4 bytes, 15 us at
//     256^3) the first time a (row_ptrs, n_rows) pair is seen; the answer - usually "none" - is cached by
//     the launcher (csr_spmv.hip), so a regular matrix pays one scan in its first product and nothing after;
//   * the row-segment kernel leaves flagged segments out (a wave owns one or two segments: it shortens
//     its range at its start - the only change to that kernel);
//   * csr_flagged_segments_kernel does them: PARTS = 64 workgroups per flagged segment.  Workgroup p takes the
//     segment's ORDINARY row p: its 256 threads form the products (coalesced loads of values and columns, the
//     gathers in flight together) into LDS and one thread adds them in entry order, separate multiply and
//     add - the reference's bits, as everywhere.  (Round 5 let one lane walk its row through global memory:
//     two dependent loads per entry at 1.3 us each - a 170-entry row next to a hub cost 220 us, as long as
//     the whole rest of the product: profiles/r06_irregular_pmc.txt.)  A long row is cut into PARTS equal
//     chunks, chunk p summed by workgroup p (per thread: entries first + t, + 256, ... in order; the 256
//     partial sums folded by a fixed tree);
//   * csr_long_rows_fold_kernel adds the PARTS chunk sums of every long row in chunk order.  No
//     floating-point atomics, no "last workgroup" tickets: the same bits every run, on every device (PARTS
//     is a constant), and the chunk sums live in a buffer of their own per (matrix, STREAM) - products of one
//     matrix in flight on two streams share nothing that is written (round 5 kept one
//     buffer and one set of tickets per matrix).
// A stale flag set (the arrays were rewritten in place under the same pointer) costs speed, never
// correctness: both kernels read the SAME flags, and each handles any row.
#pragma once
#include "common.hpp"

namespace gkoc {

#ifdef __HIPCC__

constexpr int LONG_PARTS = 64;        // workgroups per flagged segment (a constant: results do not depend on the device)
constexpr int LONG_WG = 256;
constexpr int LONG_MAX_PER_SEG = 8;   // long rows of one segment that are cut into chunks (more: summed by one workgroup)
constexpr int LONG_STAGE = 4096;      // products of an ordinary row that wait in LDS for their turn (>= GKOC_CSR_LONG_ROW)

// bit s of `bits` = segment s holds a row longer than GKOC_CSR_LONG_ROW; list[0] = how many, list[1 ..] = which
template <typename I>
__global__ __launch_bounds__(256) void csr_long_row_scan_kernel(int64_t n_rows, const I* __restrict__ row_ptrs,
                                                                uint32_t* __restrict__ bits,
                                                                unsigned long long* __restrict__ list,
                                                                int64_t list_cap)
{
    const int64_t seg = int64_t(blockIdx.x) * 4 + threadIdx.x / 64;      // one wave per 64-row segment
    const int lane = threadIdx.x % 64;
    const int64_t row = seg * 64 + lane;
    bool lng = false;
    if (row < n_rows) lng = int64_t(row_ptrs[row + 1]) - int64_t(row_ptrs[row]) > GKOC_CSR_LONG_ROW;
    if (__ballot(lng) != 0 && lane == 0) {
        atomicOr(bits + (seg >> 5), 1u << (seg & 31));
        const unsigned long long at = atomicAdd(list, 1ull);
        if (int64_t(at) < list_cap) list[1 + at] = static_cast<unsigned long long>(seg);
    }
}

// c[rows of the listed segments] = A b  (ADV: alpha A b + beta c), one right-hand side (column j of b / c);
// the long rows' chunk sums go to `partial`, csr_long_rows_fold_kernel finishes them
template <typename T, typename I, bool ADV, typename V = T>
__global__ __launch_bounds__(LONG_WG) void csr_flagged_segments_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols, const V* __restrict__ vals,
    const T* __restrict__ b, int64_t ldb, T* __restrict__ c, int64_t ldc, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p, const unsigned long long* __restrict__ list, T* __restrict__ partial)
{
    static_assert(LONG_PARTS == 64, "workgroup p of a segment takes its ordinary row p");
    const int64_t li = blockIdx.x / LONG_PARTS;          // which flagged segment
    const int part = blockIdx.x % LONG_PARTS;
    const int64_t seg = int64_t(list[1 + li]);
    const int tid = threadIdx.x;
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    __shared__ int64_t rp[65];
    __shared__ T red[LONG_WG];
    __shared__ T prod[LONG_STAGE];
    __shared__ int long_rows[LONG_MAX_PER_SEG];
    __shared__ int n_long;
    const int64_t row0 = seg * 64;
    if (tid <= 64) {
        const int64_t r = row0 + tid < n_rows ? row0 + tid : n_rows;
        rp[tid] = int64_t(row_ptrs[r]);
    }
    if (tid == 0) n_long = 0;
    __syncthreads();
    if (tid == 0) {
        // the long rows of the segment that are cut into chunks, in row order (every workgroup of the
        // segment finds the same list)
        for (int r = 0; r < 64; ++r) {
            if (rp[r + 1] - rp[r] > GKOC_CSR_LONG_ROW && n_long < LONG_MAX_PER_SEG) long_rows[n_long++] = r;
        }
    }
    __syncthreads();
    auto product = [&](int64_t k) {
        const T xb = b[int64_t(cols[k]) * ldb];
        return ADV ? (alpha * T(vals[k])) * xb : T(vals[k]) * xb;
    };
    // ---- row `part` of the segment, unless it is one of the chunked ones: products by everybody, sum by one
    {
        bool mine = row0 + part < n_rows;
        for (int q = 0; q < n_long; ++q) mine = mine && long_rows[q] != part;
        if (mine) {
            const int64_t a = rp[part], e = rp[part + 1];
            T sum = T(0);
            if (ADV && beta != T(0) && tid == 0) sum = c[(row0 + part) * ldc] * beta;
            // (a row beyond the stage - a ninth long row of one segment - goes through it in rounds)
            for (int64_t k0 = a; k0 < e; k0 += LONG_STAGE) {
                const int cnt = int(e - k0 < LONG_STAGE ? e - k0 : LONG_STAGE);
                for (int i = tid; i < cnt; i += LONG_WG) prod[i] = product(k0 + i);
                __syncthreads();
                if (tid == 0) {
                    int i = 0;
                    for (; i + 4 <= cnt; i += 4) {
                        const T p0 = prod[i], p1 = prod[i + 1], p2 = prod[i + 2], p3 = prod[i + 3];
                        sum += p0;
                        sum += p1;
                        sum += p2;
                        sum += p3;
                    }
                    for (; i < cnt; ++i) sum += prod[i];
                }
                __syncthreads();
            }
            if (tid == 0) c[(row0 + part) * ldc] = sum;
        }
    }
    // ---- the long rows: chunk `part` of each
    for (int q = 0; q < n_long; ++q) {
        const int r = long_rows[q];
        const int64_t a = rp[r], len = rp[r + 1] - rp[r];
        const int64_t chunk = (len + LONG_PARTS - 1) / LONG_PARTS;
        const int64_t first = a + chunk * part;
        const int64_t last = first + chunk < a + len ? first + chunk : a + len;
        T s = T(0);
        // four of the thread's entries under way at a time, added in entry order (the loop with one load per
        // pass waited for every gather in turn: 15 round trips for a 250 000-entry row)
        int64_t k = first + tid;
        for (; k + 3 * LONG_WG < last; k += 4 * LONG_WG) {
            const T p0 = product(k), p1 = product(k + LONG_WG), p2 = product(k + 2 * LONG_WG),
                    p3 = product(k + 3 * LONG_WG);
            s += p0;
            s += p1;
            s += p2;
            s += p3;
        }
        for (; k < last; k += LONG_WG) s += product(k);
        red[tid] = s;
        __syncthreads();
#pragma unroll
        for (int off = LONG_WG / 2; off > 0; off >>= 1) {
            if (tid < off) red[tid] = red[tid] + red[tid + off];
            __syncthreads();
        }
        if (tid == 0) partial[(li * LONG_MAX_PER_SEG + q) * LONG_PARTS + part] = red[0];
        __syncthreads();
    }
}

// one wave per flagged segment: lane q adds the chunk sums of the segment's q-th long row in chunk order
template <typename T, typename I, bool ADV>
__global__ __launch_bounds__(64) void csr_long_rows_fold_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, T* __restrict__ c, int64_t ldc,
    const T* __restrict__ beta_p, const unsigned long long* __restrict__ list, const T* __restrict__ partial)
{
    const int64_t li = blockIdx.x;
    const int64_t row0 = int64_t(list[1 + li]) * 64;
    const int lane = threadIdx.x;
    const int64_t r = row0 + lane;
    const bool lng = r < n_rows && int64_t(row_ptrs[r + 1]) - int64_t(row_ptrs[r]) > GKOC_CSR_LONG_ROW;
    // position of this row among the segment's long rows = long rows in front of it
    const unsigned long long m = __ballot(lng);
    const int q = __popcll(m & ((1ull << lane) - 1ull));
    if (!lng || q >= LONG_MAX_PER_SEG) return;
    const T* p = partial + (li * LONG_MAX_PER_SEG + q) * LONG_PARTS;
    T s = p[0];
    for (int i = 1; i < LONG_PARTS; ++i) s += p[i];
    T beta = T(0);
    if (ADV) beta = beta_p[0];
    c[r * ldc] = (ADV && beta != T(0)) ? c[r * ldc] * beta + s : s;
}

#endif  // __HIPCC__

// what the launcher remembers about a matrix (csr_spmv.hip)
struct csr_long_info {
    int64_t count = 0;                   // flagged segments (0: none - the common case)
    uint32_t* bits = nullptr;            // device: one bit per segment
    unsigned long long* list = nullptr;  // device: [count, segment indices ...]
    int64_t nnz = -1;                    // row_ptrs[n_rows] when the matrix was looked at (-1: unknown)
    uint64_t seq = 0;                    // order of arrival in the launcher's cache (the oldest entry makes room)
    void* partial = nullptr;             // device: chunk sums, THE CALLING STREAM's buffer (filled in per call)
};

}  // namespace gkoc
