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
//   * csr_flagged_segments_kernel does them: PARTS workgroups per flagged segment.  The ordinary rows of
//     the segment are summed by one lane each in entry order with separate multiply and add - the
//     reference's bits, as everywhere; a long row is cut into PARTS equal chunks, chunk p summed by
//     workgroup p (per thread: entries first + t, + 256, ... in order; the 256 partial sums folded by a
//     fixed tree), the PARTS chunk sums folded in chunk order by the workgroup that finishes last.  No
//     floating-point atomics: the same bits every run, on every device (PARTS is a constant).
// A stale flag set (the arrays were rewritten in place under the same pointer) costs speed, never
// correctness: both kernels read the SAME flags, and each handles any row.
#pragma once
#include "common.hpp"

namespace gkoc {

#ifdef __HIPCC__

constexpr int LONG_PARTS = 64;        // workgroups per flagged segment (a constant: results do not depend on the device)
constexpr int LONG_WG = 256;
constexpr int LONG_MAX_PER_SEG = 8;   // long rows of one segment that are cut into chunks (more: summed by one workgroup)

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

// c[rows of the listed segments] = A b  (ADV: alpha A b + beta c), one right-hand side (column j of b / c)
template <typename T, typename I, bool ADV, typename V = T>
__global__ __launch_bounds__(LONG_WG) void csr_flagged_segments_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols, const V* __restrict__ vals,
    const T* __restrict__ b, int64_t ldb, T* __restrict__ c, int64_t ldc, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p, const unsigned long long* __restrict__ list, T* __restrict__ partial,
    uint32_t* __restrict__ tickets)
{
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
    __shared__ int long_rows[LONG_MAX_PER_SEG];
    __shared__ int n_long, n_over;
    __shared__ uint32_t ticket;
    const int64_t row0 = seg * 64;
    if (tid <= 64) {
        const int64_t r = row0 + tid < n_rows ? row0 + tid : n_rows;
        rp[tid] = int64_t(row_ptrs[r]);
    }
    if (tid == 0) {
        n_long = 0;
        n_over = 0;
    }
    __syncthreads();
    if (tid == 0) {
        // the long rows of the segment, in row order (every workgroup of the segment finds the same list)
        for (int r = 0; r < 64; ++r) {
            if (rp[r + 1] - rp[r] > GKOC_CSR_LONG_ROW) {
                if (n_long < LONG_MAX_PER_SEG) {
                    long_rows[n_long++] = r;
                } else {
                    ++n_over;
                }
            }
        }
    }
    __syncthreads();
    auto product = [&](int64_t k) {
        const T xb = b[int64_t(cols[k]) * ldb];
        return ADV ? (alpha * T(vals[k])) * xb : T(vals[k]) * xb;
    };
    if (part == 0 && tid < 64 && row0 + tid < n_rows) {
        // the ordinary rows: lane = row, entry order, separate multiply and add (the reference's bits)
        const int64_t a = rp[tid], e = rp[tid + 1];
        const bool chunked = e - a > GKOC_CSR_LONG_ROW;
        bool mine = !chunked;
        if (chunked && n_over > 0) {
            // a long row beyond the LONG_MAX_PER_SEG that are cut into chunks: summed here, in order
            mine = true;
            for (int q = 0; q < n_long; ++q) mine = mine && long_rows[q] != tid;
        }
        if (mine) {
            T sum = T(0);
            if (ADV && beta != T(0)) sum = c[(row0 + tid) * ldc] * beta;
            for (int64_t k = a; k < e; ++k) sum += product(k);
            c[(row0 + tid) * ldc] = sum;
        }
    }
    // the long rows: chunk `part` of each
    for (int q = 0; q < n_long; ++q) {
        const int r = long_rows[q];
        const int64_t a = rp[r], len = rp[r + 1] - rp[r];
        const int64_t chunk = (len + LONG_PARTS - 1) / LONG_PARTS;
        const int64_t first = a + chunk * part;
        const int64_t last = first + chunk < a + len ? first + chunk : a + len;
        T s = T(0);
        for (int64_t k = first + tid; k < last; k += LONG_WG) s += product(k);
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
    if (n_long == 0) return;
    // the workgroup of the segment that finishes last adds the chunk sums in chunk order
    __threadfence();
    if (tid == 0) ticket = atomicAdd(tickets + li, 1u);
    __syncthreads();
    if (ticket != uint32_t(LONG_PARTS - 1)) return;
    __threadfence();
    if (tid < n_long) {
        const int r = long_rows[tid];
        const volatile T* p = partial + (li * LONG_MAX_PER_SEG + tid) * LONG_PARTS;
        T s = p[0];
        for (int i = 1; i < LONG_PARTS; ++i) s += p[i];
        const int64_t row = row0 + r;
        c[row * ldc] = (ADV && beta != T(0)) ? c[row * ldc] * beta + s : s;
    }
    if (tid == 0) tickets[li] = 0;      // ready for the next product
}

#endif  // __HIPCC__

// what the launcher remembers about a matrix (csr_spmv.hip)
struct csr_long_info {
    int64_t count = 0;                   // flagged segments (0: none - the common case)
    uint32_t* bits = nullptr;            // device: one bit per segment
    unsigned long long* list = nullptr;  // device: [count, segment indices ...]
    void* partial = nullptr;             // device: chunk sums
    uint32_t* tickets = nullptr;         // device: one per flagged segment, zero between launches
};

}  // namespace gkoc
