// COO SpMV and the CSR -> Hybrid conversion:
//   coo::{spmv, advanced_spmv, spmv2, advanced_spmv2}
//     (decl core/matrix/coo_kernels.hpp; reference/matrix/coo_kernels.cpp:33-100;
//      stock GPU version common/cuda_hip/matrix/coo_kernels.cpp: f64 atomics on c)
//   hybrid::compute_coo_row_ptrs (reference/matrix/hybrid_kernels.cpp:30-43),
//   csr::convert_to_hybrid      (reference/matrix/csr_kernels.cpp:911-955)
//
// COO in Ginkgo is row-major sorted (Coo::read sorts; conversions emit rows in
// order).  For such input the reference's entry-by-entry accumulation
//   c(row) = (...((c0 + p1) + p2) + ...)         c0 = 0, c, beta*c
// is the CSR row sum of the same entries started from c0 - exactly what
// csr::advanced_spmv computes (sum = beta*c, then += (alpha*val)*b in k order), and
// multiplying by a literal 1 is exact.  So:
//   c = A b, one column: ONE pass - the CSR kernel in its COO mode (csr_spmv_pipe.hpp) reads
//           values, columns and rows together, finds each wave's piece of the entry stream
//           through a pointer per 64-row segment (bisection of row_idxs) and derives the row
//           pointers in LDS while it streams; it also proves that the input is sorted;
//   the operations that read c (advanced_spmv, spmv2, advanced_spmv2) and several columns:
//   pass 1: row_idxs -> row_ptrs in the workspace, one kernel: one read of row_idxs
//           (4 B/nnz), every pointer written once by the entry that starts its run; the
//           same kernel notes whether the rows really are non-decreasing;
//   pass 2: the production CSR kernel (12 B/nnz) with (alpha, beta) in
//           {(alpha, beta), (1, 1), (alpha, 1)}.
// 16 B/nnz in total - the algorithmic traffic of COO - no atomics, results
// bit-identical to the reference.  Input whose rows are NOT sorted takes the
// reference's semantics through a fallback: pass 1 raises a device flag, the row
// pointers are cleared (the CSR pass then only applies beta), and an atomic kernel
// adds the products (sum order then differs: tolerance instead of bit-identity).
// Both extra kernels return at once when the flag is not set; no host round trip.
#include "common.hpp"
#include "csr_spmv_pipe.hpp"
#include "scan.hpp"

namespace gkoc {
namespace {

inline unsigned grid_for(int64_t n, int cap = 4 * max_stream_blocks)
{
    int64_t b = ceildiv(n > 0 ? n : 1, 256);
    if (b > cap) b = cap;
    return unsigned(b);
}

// workspace: [row_ptrs (n_rows + 1) of I | pad to 16 | flag int32, pad | one T, pad |
//             n_rows values: the copy of c that the one-pass forms of the operations reading c
//             keep until the input is known to be sorted]
inline size_t coo_ptr_bytes(int64_t n_rows, size_t index_size)
{
    return (size_t(n_rows + 1) * index_size + 15) / 16 * 16;
}

inline size_t coo_work_bytes(int64_t n_rows, size_t index_size, size_t value_size)
{
    return coo_ptr_bytes(n_rows, index_size) + 32 + (size_t(n_rows) * value_size + 15) / 16 * 16;
}

template <typename T, typename I>
struct coo_work {
    I* ptrs;
    int* flag;
    T* one;
    T* saved_c;
    static size_t bytes(int64_t n_rows) { return coo_work_bytes(n_rows, sizeof(I), sizeof(T)); }
    coo_work(void* w, int64_t n_rows)
    {
        char* c = static_cast<char*>(w);
        const size_t p = coo_ptr_bytes(n_rows, sizeof(I));
        ptrs = reinterpret_cast<I*>(c);
        flag = reinterpret_cast<int*>(c + p);
        one = reinterpret_cast<T*>(c + p + 16);
        saved_c = reinterpret_cast<T*>(c + p + 32);
    }
};

template <typename T>
__global__ void coo_init_kernel(int* flag, T* one)
{
    *flag = 0;
    *one = T(1);
}

// Row pointers of a row-sorted COO in ONE pass over row_idxs, no scan and no clearing pass:
// the entry k that starts a run of equal row indices knows the pointer of its own row and
// of every empty row in front of it,
//     row_ptrs[r] = k   for  row_idxs[k-1] < r <= row_idxs[k],
// the first entry additionally covers rows 0 .. row_idxs[0] and the last one the rows behind
// row_idxs[nnz-1] (pointer nnz).  Every pointer is written exactly once (plain stores, no
// atomics - device-scope atomics on a counter array were measured 2.7x slower).  Gaps of up
// to 8 empty rows are filled by the lane that found them, longer ones by the whole wave (a
// COO part with a few long rows, as in a Hybrid, has one long gap per stored row).  flag = 1
// if a row index is smaller than its predecessor or out of range (the pointers are then
// cleared and the atomic fallback adds the products).  Four entries per thread: one 16-byte
// (int32) / 32-byte (int64) load per lane, the neighbours' boundary entries come through
// wave shuffles (six scalar loads per lane, 16 bytes apart, kept the pipeline at 2 TB/s).
template <typename I>
struct alignas(16) idx4 {
    I v[4];
};

template <typename I>
__global__ __launch_bounds__(256) void coo_row_ptrs_kernel(int64_t nnz,
                                                           const I* __restrict__ rows,
                                                           int64_t n_rows, I* __restrict__ ptrs,
                                                           int* __restrict__ flag)
{
    constexpr int64_t inline_gap = 8;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t(blockIdx.x) * 256 + threadIdx.x) >> 6;
    const int64_t n_waves = (int64_t(gridDim.x) * 256) >> 6;
    const bool aligned = reinterpret_cast<uintptr_t>(rows) % 16 == 0;
    for (int64_t base = wave * 256; base < nnz; base += n_waves * 256) {
        const int64_t i0 = base + lane * 4;
        int64_t r[5];
        if (aligned && i0 + 4 <= nnz) {
            const idx4<I> q = *reinterpret_cast<const idx4<I>*>(rows + i0);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e + 1] = int64_t(q.v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e + 1] = i0 + e < nnz ? int64_t(rows[i0 + e]) : -2;
        }
        int64_t prev = __shfl_up(r[4], 1, 64);
        if (lane == 0) prev = base > 0 ? int64_t(rows[base - 1]) : -1;
        r[0] = prev;
        bool bad = false;
        // the (at most one per entry) range of pointers this lane has to write: (lo, hi] <- val;
        // long ranges are kept for the wave
        int64_t big_lo = 0, big_hi = -1, big_val = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t i = i0 + e;
            if (i >= nnz) break;
            const int64_t cur = r[e + 1], pv = r[e];
            // a predecessor outside [0, n_rows) makes this entry bad as well: its range of
            // pointers would start before (or reach behind) the array
            if (cur < 0 || cur >= n_rows || (i > 0 && (cur < pv || pv < 0 || pv >= n_rows))) {
                bad = true;
                continue;
            }
            if (i == 0 || cur != pv) {
                const int64_t lo = i == 0 ? -1 : pv;
                if (cur - lo <= inline_gap || big_hi >= big_lo + 1) {
                    for (int64_t t = lo + 1; t <= cur; ++t) ptrs[t] = I(i);
                } else {
                    big_lo = lo;
                    big_hi = cur;
                    big_val = i;
                }
            }
            if (i == nnz - 1) {
                for (int64_t t = cur + 1; t <= n_rows; ++t) {
                    if (t - cur > inline_gap) {
                        // the trailing rows: keep the rest for the wave unless a range is kept
                        if (big_hi < big_lo + 1) {
                            big_lo = t - 1;
                            big_hi = n_rows;
                            big_val = nnz;
                            break;
                        }
                    }
                    ptrs[t] = I(nnz);
                }
            }
        }
        if (bad) *flag = 1;
        unsigned long long m = __ballot(big_hi >= big_lo + 1);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const int64_t lo = __shfl(big_lo, src, 64);
            const int64_t hi = __shfl(big_hi, src, 64);
            const int64_t val = __shfl(big_val, src, 64);
            for (int64_t t = lo + 1 + lane; t <= hi; t += 64) ptrs[t] = I(val);
        }
    }
}

// seg_ptrs[s] = first entry whose row index is >= 64 s (bisection of the sorted row_idxs; on
// unsorted input: some index in [0, nnz], which the product kernel then detects), s = 0 .. n_seg
template <typename I>
__global__ __launch_bounds__(256) void coo_segment_ptrs_kernel(int64_t nnz,
                                                               const I* __restrict__ rows,
                                                               int64_t n_seg, I* __restrict__ seg_ptrs)
{
    const int64_t sgm = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (sgm > n_seg) return;
    const int64_t want = sgm * 64;
    int64_t lo = 0, hi = nnz;   // first k in [0, nnz] with rows[k] >= want
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (int64_t(rows[mid]) < want) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    seg_ptrs[sgm] = sgm == n_seg ? I(nnz) : I(lo);
}

template <typename T>
__global__ __launch_bounds__(256) void coo_save_kernel(int64_t rows, const T* __restrict__ c,
                                                       int64_t ldc, T* __restrict__ saved)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < rows; i += stride) saved[i] = c[i * ldc];
}

// unsorted input: c = beta * (what it was), ready for the atomic additions (beta == NULL: 1)
template <typename T>
__global__ __launch_bounds__(256) void coo_restore_if_unsorted_kernel(int64_t rows, T* __restrict__ c,
                                                                      int64_t ldc,
                                                                      const T* __restrict__ saved,
                                                                      const T* __restrict__ beta,
                                                                      const int* __restrict__ flag)
{
    if (*flag == 0) return;
    const T bt = beta ? beta[0] : T(1);
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < rows; i += stride) {
        c[i * ldc] = bt == T(0) ? T(0) : (beta ? bt * saved[i] : saved[i]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void coo_zero_if_unsorted_kernel(int64_t rows, T* __restrict__ c,
                                                                   int64_t ldc,
                                                                   const int* __restrict__ flag)
{
    if (*flag == 0) return;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < rows; i += stride) c[i * ldc] = T(0);
}

template <typename I>
__global__ __launch_bounds__(256) void coo_clear_ptrs_if_unsorted_kernel(
    int64_t n, I* __restrict__ ptrs, const int* __restrict__ flag)
{
    if (*flag == 0) return;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) ptrs[i] = I(0);
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void coo_atomic_if_unsorted_kernel(
    int64_t nnz, const I* __restrict__ rows, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ alpha, const T* __restrict__ b,
    int64_t ldb, T* __restrict__ c, int64_t ldc, int64_t nrhs, int64_t n_rows,
    const int* __restrict__ flag)
{
    if (*flag == 0) return;
    const T a = alpha ? alpha[0] : T(1);
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < nnz; i += stride) {
        const int64_t r = rows[i];
        if (r < 0 || r >= n_rows) continue;
        const T v = alpha ? a * vals[i] : vals[i];
        for (int64_t j = 0; j < nrhs; ++j) {
            atomicAdd(&c[r * ldc + j], v * b[int64_t(cols[i]) * ldb + j]);
        }
    }
}

template <typename T, typename I>
int csr_call(gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const T* alpha, const I* ptrs,
             const I* cols, const T* vals, const T* b, int64_t ldb, const T* beta, T* c,
             int64_t ldc, int64_t nrhs);

#define GKOC_COO_CSR(T, TN, I, IN)                                                          \
    template <>                                                                             \
    int csr_call<T, I>(gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const T* alpha,     \
                       const I* ptrs, const I* cols, const T* vals, const T* b,             \
                       int64_t ldb, const T* beta, T* c, int64_t ldc, int64_t nrhs)         \
    {                                                                                       \
        if (alpha == nullptr) {                                                             \
            return gkoc_csr_spmv_##TN##_##IN(s, n_rows, n_cols, ptrs, cols, vals, b, ldb,   \
                                             c, ldc, nrhs);                                 \
        }                                                                                   \
        return gkoc_csr_advanced_spmv_##TN##_##IN(s, n_rows, n_cols, alpha, ptrs, cols,     \
                                                  vals, b, ldb, beta, c, ldc, nrhs);        \
    }
GKOC_COO_CSR(double, f64, int32_t, i32)
GKOC_COO_CSR(double, f64, int64_t, i64)
GKOC_COO_CSR(float, f32, int32_t, i32)
GKOC_COO_CSR(float, f32, int64_t, i64)

// mode: 0 spmv, 1 advanced_spmv(alpha, beta), 2 spmv2, 3 advanced_spmv2(alpha)
template <typename T, typename I>
int launch_coo(gkoc_stream_t s, int mode, int64_t n_rows, int64_t n_cols, int64_t nnz,
               const T* alpha, const I* rows, const I* cols, const T* vals, const T* b,
               int64_t ldb, const T* beta, T* c, int64_t ldc, int64_t nrhs, void* work,
               size_t work_bytes)
{
    GKOC_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && nrhs >= 0, GKOC_E_INVALID,
                 "negative dimension");
    if (n_rows == 0 || nrhs == 0) return GKOC_OK;
    if (nnz == 0 && (mode == 2 || mode == 3)) return GKOC_OK;   // c += 0
    const size_t need = coo_work<T, I>::bytes(n_rows);
    GKOC_REQUIRE(work && work_bytes >= need, GKOC_E_WORKSPACE,
                 "workspace too small (gkoc_coo_workspace_bytes)");
    GKOC_REQUIRE(nnz == 0 || (rows && cols && vals), GKOC_E_INVALID, "null pointer");
    if (mode == 1) GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha / beta");
    if (mode == 3) GKOC_REQUIRE(alpha, GKOC_E_INVALID, "null alpha");
    coo_work<T, I> w(work, n_rows);
    hipStream_t st = as_stream(s);
    coo_init_kernel<T><<<dim3(1), dim3(1), 0, st>>>(w.flag, w.one);
    GKOC_LAUNCH_OK();
    {
        // One right-hand side: ONE pass over values, columns and rows (16 B per entry): the CSR
        // kernel in its COO mode (csr_spmv_pipe.hpp) finds each wave's piece of the entry stream
        // through a pointer per 64-row segment (bisection: n / 64 searches) and derives the row
        // pointers while it streams.  That the input is sorted is only known when the kernel has
        // run, i.e. when c has been overwritten: the operations that read c keep a copy of it in
        // the workspace (2 x 8 n bytes of traffic) for the atomic fallback to start from.
        constexpr int EV = 32 / sizeof(T);
        constexpr int RINGV = 8192 / sizeof(T);
        const bool vec_ok = reinterpret_cast<uintptr_t>(vals) % (EV * sizeof(T)) == 0 &&
                            reinterpret_cast<uintptr_t>(cols) % (EV * sizeof(I)) == 0 &&
                            reinterpret_cast<uintptr_t>(rows) % (EV * sizeof(I)) == 0;
        const int64_t n_seg = ceildiv(n_rows, 64);
        // the copy of c costs 16 B per row, the pointer pass it replaces 4 B per entry, and short
        // rows run the one-pass kernel below its streaming rate: measured on the COO part of the
        // 256^3 Hybrid (8.8 per row) 1.327 ms against 1.303 ms in two passes
        const bool pays = mode == 0 || nnz >= 16 * n_rows;
        if (nrhs == 1 && nnz > 0 && vec_ok && pays && n_seg < (int64_t(1) << 31) &&
            tune_value(GKOC_TUNE_COO_FUSED) != 0) {
            coo_segment_ptrs_kernel<I><<<dim3(unsigned(ceildiv(n_seg + 1, 256))), dim3(256), 0, st>>>(
                nnz, rows, n_seg, w.ptrs);
            GKOC_LAUNCH_OK();
            if (mode != 0) {
                coo_save_kernel<T><<<dim3(grid_for(n_rows)), dim3(256), 0, st>>>(n_rows, c, ldc, w.saved_c);
                GKOC_LAUNCH_OK();
            }
            const int segs_per_wave = n_seg < 65536 ? 1 : 2;
            const dim3 grid(static_cast<unsigned>(ceildiv(n_seg, segs_per_wave))), block(64);
            // (alpha, beta) of the CSR kernel: spmv -, advanced (alpha, beta), spmv2 (1, 1),
            // advanced_spmv2 (alpha, 1); a literal 1 multiplies exactly
            const T* ka = mode == 0 ? nullptr : (mode == 2 ? w.one : alpha);
            const T* kb = mode == 1 ? beta : w.one;
            // four entries per lane and load, one load group (the 2 x 3 layout of the CSR launcher costs
            // this mode - 99 VGPRs already - a wave per SIMD: L256 1.57 ms against 1.44)
            constexpr int CE = EV, CU = 1;
#define GKOC_LAUNCH_COO(ADV_, MODE_)                                                                    \
    csr_spmv_pipe3_kernel<T, I, ADV_, 64, CE, CU, RINGV, 1, MODE_ | 128><<<grid, block, 0, st>>>(        \
        n_rows, n_seg, segs_per_wave, w.ptrs, cols, vals, b, ldb, c, ldc, 1, ka, kb, nullptr, 0, rows,  \
        w.flag)
            if (mode == 0) {
                if (segs_per_wave == 2) {
                    GKOC_LAUNCH_COO(false, 0x2000);
                } else {
                    GKOC_LAUNCH_COO(false, 0x1000);
                }
            } else {
                if (segs_per_wave == 2) {
                    GKOC_LAUNCH_COO(true, 0x2000);
                } else {
                    GKOC_LAUNCH_COO(true, 0x1000);
                }
            }
#undef GKOC_LAUNCH_COO
            GKOC_LAUNCH_OK();
            // unsorted input (never produced by Ginkgo's own Coo): start again with atomics
            if (mode == 0) {
                coo_zero_if_unsorted_kernel<T><<<dim3(grid_for(n_rows)), dim3(256), 0, st>>>(n_rows, c, ldc,
                                                                                            w.flag);
            } else {
                coo_restore_if_unsorted_kernel<T><<<dim3(grid_for(n_rows)), dim3(256), 0, st>>>(
                    n_rows, c, ldc, w.saved_c, mode == 1 ? beta : nullptr, w.flag);
            }
            GKOC_LAUNCH_OK();
            coo_atomic_if_unsorted_kernel<T, I><<<dim3(grid_for(nnz)), dim3(256), 0, st>>>(
                nnz, rows, cols, vals, (mode == 1 || mode == 3) ? alpha : nullptr, b, ldb, c, ldc, 1, n_rows,
                w.flag);
            GKOC_LAUNCH_OK();
            return GKOC_OK;
        }
    }
    if (nnz > 0) {
        coo_row_ptrs_kernel<I><<<dim3(grid_for(ceildiv(nnz, 4), 8 * max_stream_blocks)), dim3(256),
                                 0, st>>>(nnz, rows, n_rows, w.ptrs, w.flag);
        GKOC_LAUNCH_OK();
    } else {
        GKOC_HIP(hipMemsetAsync(w.ptrs, 0, size_t(n_rows + 1) * sizeof(I), st));
    }
    coo_clear_ptrs_if_unsorted_kernel<I><<<dim3(grid_for(n_rows + 1)), dim3(256), 0, st>>>(
        n_rows + 1, w.ptrs, w.flag);
    GKOC_LAUNCH_OK();
    const T* a = mode == 0 ? nullptr : (mode == 2 ? w.one : alpha);
    const T* bt = mode == 1 ? beta : w.one;
    int rc = csr_call<T, I>(s, n_rows, n_cols, a, w.ptrs, cols, vals, b, ldb, bt, c, ldc, nrhs);
    if (rc != GKOC_OK) return rc;
    if (nnz > 0) {
        coo_atomic_if_unsorted_kernel<T, I><<<dim3(grid_for(nnz)), dim3(256), 0, st>>>(
            nnz, rows, cols, vals, (mode == 1 || mode == 3) ? alpha : nullptr, b, ldb, c, ldc,
            nrhs, n_rows, w.flag);
        GKOC_LAUNCH_OK();
    }
    return GKOC_OK;
}

// row_ptrs -> row index per stored entry (components::convert_ptrs_to_idxs,
// reference/components/format_conversion_kernels.cpp): one wavefront per 64 rows, the
// 65 pointers in LDS, lanes stride over the segment's entries (coalesced stores) and
// find their row by bisection.
template <typename I>
__global__ __launch_bounds__(64) void ptrs_to_idxs_kernel(int64_t n_rows,
                                                          const I* __restrict__ ptrs,
                                                          I* __restrict__ idxs)
{
    __shared__ int64_t lp[65];
    const int lane = threadIdx.x;
    const int64_t r0 = int64_t(blockIdx.x) * 64;
    const int64_t nr = n_rows - r0 < 64 ? n_rows - r0 : 64;
    if (lane < nr) lp[lane] = ptrs[r0 + lane];
    if (lane == 0) {
        for (int64_t k = nr; k <= 64; ++k) lp[k] = ptrs[r0 + nr];
    }
    __syncthreads();
    const int64_t K0 = lp[0], K1 = lp[nr];
    for (int64_t k = K0 + lane; k < K1; k += 64) {
        int lo = 0, hi = int(nr);          // largest r with lp[r] <= k
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (lp[mid] <= k) {
                lo = mid;
            } else {
                hi = mid;
            }
        }
        idxs[k] = I(r0 + lo);
    }
}

// ell::copy (reference/matrix/ell_kernels.cpp:195-206): same entries, other stride
template <typename T, typename I>
__global__ __launch_bounds__(256) void ell_copy_kernel(int64_t n_rows, int64_t k,
                                                       int64_t src_stride,
                                                       const I* __restrict__ src_cols,
                                                       const T* __restrict__ src_vals,
                                                       int64_t dst_stride, I* __restrict__ dst_cols,
                                                       T* __restrict__ dst_vals)
{
    const int64_t total = n_rows * k;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / n_rows, row = t - i * n_rows;
        dst_cols[row + i * dst_stride] = src_cols[row + i * src_stride];
        dst_vals[row + i * dst_stride] = src_vals[row + i * src_stride];
    }
}

// ------------------------------------------------------------------ hybrid
__global__ __launch_bounds__(256) void hybrid_overflow_counts_kernel(
    int64_t n_rows, const uint64_t* __restrict__ row_nnz, uint64_t ell_lim,
    int64_t* __restrict__ out)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i <= n_rows; i += stride) {
        out[i] = (i < n_rows && row_nnz[i] > ell_lim) ? int64_t(row_nnz[i] - ell_lim) : 0;
    }
}

// the entries of a row beyond the first ell_lim go to COO, in order
template <typename T, typename I>
__global__ __launch_bounds__(256) void hybrid_overflow_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, int64_t ell_lim, const int64_t* __restrict__ coo_row_ptrs,
    I* __restrict__ coo_rows, I* __restrict__ coo_cols, T* __restrict__ coo_vals)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x; row < n_rows; row += stride) {
        const int64_t rs = row_ptrs[row], re = row_ptrs[row + 1];
        int64_t pos = coo_row_ptrs[row];
        for (int64_t k = rs + ell_lim; k < re; ++k, ++pos) {
            coo_vals[pos] = vals[k];
            coo_cols[pos] = cols[k];
            coo_rows[pos] = I(row);
        }
    }
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

extern "C" size_t gkoc_coo_workspace_bytes(int64_t n_rows, size_t index_size, size_t value_size)
{
    if (n_rows < 0) n_rows = 0;
    return coo_work_bytes(n_rows, index_size, value_size ? value_size : 8);
}

extern "C" int gkoc_hybrid_compute_coo_row_ptrs(gkoc_stream_t s, int64_t n_rows,
                                                const uint64_t* row_nnz, uint64_t ell_lim,
                                                int64_t* coo_row_ptrs)
{
    GKOC_REQUIRE(n_rows >= 0 && coo_row_ptrs && (n_rows == 0 || row_nnz), GKOC_E_INVALID,
                 "bad argument");
    hybrid_overflow_counts_kernel<<<dim3(grid_for(n_rows + 1)), dim3(256), 0, as_stream(s)>>>(
        n_rows, row_nnz, ell_lim, coo_row_ptrs);
    GKOC_LAUNCH_OK();
    return device_exclusive_scan<int64_t>(as_stream(s), coo_row_ptrs, n_rows + 1);
}

#define GKOC_DEF_P2I(I, IN)                                                                 \
    extern "C" int gkoc_convert_ptrs_to_idxs_##IN(gkoc_stream_t s, const I* ptrs,           \
                                                  int64_t n_rows, I* idxs)                  \
    {                                                                                       \
        GKOC_REQUIRE(n_rows >= 0, GKOC_E_INVALID, "negative size");                         \
        if (n_rows == 0) return GKOC_OK;                                                    \
        ptrs_to_idxs_kernel<I><<<dim3(unsigned(ceildiv(n_rows, 64))), dim3(64), 0,          \
                                 as_stream(s)>>>(n_rows, ptrs, idxs);                       \
        GKOC_LAUNCH_OK();                                                                   \
        return GKOC_OK;                                                                     \
    }
GKOC_DEF_P2I(int32_t, i32)
GKOC_DEF_P2I(int64_t, i64)

#define GKOC_DEF_COO(T, TN, I, IN)                                                          \
    extern "C" int gkoc_coo_spmv_##TN##_##IN(                                               \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t nnz, const I* row_idxs,    \
        const I* col_idxs, const T* vals, const T* b, int64_t ldb, T* c, int64_t ldc,       \
        int64_t nrhs, void* work, size_t work_bytes)                                        \
    {                                                                                       \
        return launch_coo<T, I>(s, 0, n_rows, n_cols, nnz, nullptr, row_idxs, col_idxs,     \
                                vals, b, ldb, nullptr, c, ldc, nrhs, work, work_bytes);     \
    }                                                                                       \
    extern "C" int gkoc_coo_advanced_spmv_##TN##_##IN(                                      \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t nnz, const T* alpha,       \
        const I* row_idxs, const I* col_idxs, const T* vals, const T* b, int64_t ldb,       \
        const T* beta, T* c, int64_t ldc, int64_t nrhs, void* work, size_t work_bytes)      \
    {                                                                                       \
        return launch_coo<T, I>(s, 1, n_rows, n_cols, nnz, alpha, row_idxs, col_idxs, vals, \
                                b, ldb, beta, c, ldc, nrhs, work, work_bytes);              \
    }                                                                                       \
    extern "C" int gkoc_coo_spmv2_##TN##_##IN(                                              \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t nnz, const I* row_idxs,    \
        const I* col_idxs, const T* vals, const T* b, int64_t ldb, T* c, int64_t ldc,       \
        int64_t nrhs, void* work, size_t work_bytes)                                        \
    {                                                                                       \
        return launch_coo<T, I>(s, 2, n_rows, n_cols, nnz, nullptr, row_idxs, col_idxs,     \
                                vals, b, ldb, nullptr, c, ldc, nrhs, work, work_bytes);     \
    }                                                                                       \
    extern "C" int gkoc_coo_advanced_spmv2_##TN##_##IN(                                     \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t nnz, const T* alpha,       \
        const I* row_idxs, const I* col_idxs, const T* vals, const T* b, int64_t ldb, T* c, \
        int64_t ldc, int64_t nrhs, void* work, size_t work_bytes)                           \
    {                                                                                       \
        return launch_coo<T, I>(s, 3, n_rows, n_cols, nnz, alpha, row_idxs, col_idxs, vals, \
                                b, ldb, nullptr, c, ldc, nrhs, work, work_bytes);           \
    }
// (ell::copy and csr::convert_to_hybrid: for the complex types as well)
#define GKOC_DEF_COO_CONVERT(T, TN, I, IN)                                                  \
    extern "C" int gkoc_ell_copy_##TN##_##IN(                                               \
        gkoc_stream_t s, int64_t n_rows, int64_t k, int64_t src_stride, const I* src_cols,  \
        const T* src_vals, int64_t dst_stride, I* dst_cols, T* dst_vals)                    \
    {                                                                                       \
        GKOC_REQUIRE(n_rows >= 0 && k >= 0 && src_stride >= n_rows && dst_stride >= n_rows, \
                     GKOC_E_INVALID, "bad dimensions");                                     \
        if (n_rows == 0 || k == 0) return GKOC_OK;                                          \
        ell_copy_kernel<T, I><<<dim3(grid_for(n_rows * k)), dim3(256), 0, as_stream(s)>>>(  \
            n_rows, k, src_stride, src_cols, src_vals, dst_stride, dst_cols, dst_vals);     \
        GKOC_LAUNCH_OK();                                                                   \
        return GKOC_OK;                                                                     \
    }                                                                                       \
    extern "C" int gkoc_csr_convert_to_hybrid_##TN##_##IN(                                  \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* col_idxs,              \
        const T* vals, int64_t ell_lim, int64_t ell_stride, I* ell_cols, T* ell_vals,       \
        const int64_t* coo_row_ptrs, I* coo_rows, I* coo_cols, T* coo_vals)                 \
    {                                                                                       \
        GKOC_REQUIRE(n_rows >= 0 && ell_lim >= 0, GKOC_E_INVALID, "negative dimension");    \
        if (n_rows == 0) return GKOC_OK;                                                    \
        if (ell_lim > 0) {                                                                  \
            int rc = gkoc_csr_convert_to_ell_##TN##_##IN(s, n_rows, row_ptrs, col_idxs,     \
                                                         vals, ell_lim, ell_stride,         \
                                                         ell_cols, ell_vals);               \
            if (rc != GKOC_OK) return rc;                                                   \
        }                                                                                   \
        GKOC_REQUIRE(coo_row_ptrs, GKOC_E_INVALID, "null coo_row_ptrs");                    \
        hybrid_overflow_kernel<T, I>                                                        \
            <<<dim3(grid_for(n_rows)), dim3(256), 0, as_stream(s)>>>(                       \
                n_rows, row_ptrs, col_idxs, vals, ell_lim, coo_row_ptrs, coo_rows,          \
                coo_cols, coo_vals);                                                        \
        GKOC_LAUNCH_OK();                                                                   \
        return GKOC_OK;                                                                     \
    }

GKOC_DEF_COO(double, f64, int32_t, i32)
GKOC_DEF_COO(double, f64, int64_t, i64)
GKOC_DEF_COO(float, f32, int32_t, i32)
GKOC_DEF_COO(float, f32, int64_t, i64)
GKOC_DEF_COO_CONVERT(double, f64, int32_t, i32)
GKOC_DEF_COO_CONVERT(double, f64, int64_t, i64)
GKOC_DEF_COO_CONVERT(float, f32, int32_t, i32)
GKOC_DEF_COO_CONVERT(float, f32, int64_t, i64)
GKOC_DEF_COO_CONVERT(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_COO_CONVERT(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_COO_CONVERT(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_COO_CONVERT(gkoc_c64, c64, int64_t, i64)
