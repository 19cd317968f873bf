// ELL / SELL-P SpMV, CSR -> ELL / SELL-P conversions and the index utilities
// needed to build those formats on the device.
//
// Replaces gko::kernels::hip::ell::{spmv, advanced_spmv, compute_max_row_nnz},
// sellp::{spmv, advanced_spmv, compute_slice_sets},
// csr::{convert_to_ell, convert_to_sellp},
// components::{prefix_sum_nonnegative, convert_ptrs_to_sizes,
// convert_idxs_to_ptrs, fill_array, fill_seq_array}
// (decl core/matrix/{ell,sellp,csr}_kernels.hpp, core/components/*_kernels.hpp;
// semantics reference/matrix/ell_kernels.cpp:25-140,
// reference/matrix/sellp_kernels.cpp:25-130,
// reference/matrix/csr_kernels.cpp:529-600; stock GPU versions
// common/cuda_hip/matrix/ell_kernels.cpp:84-216, sellp_kernels.cpp:37-135).
//
// Both formats are lane-per-row with column-major (ELL) / slice-column-major
// (SELL-P, slice_size = 64 = one wavefront per slice) storage, so every
// val / col load of a wave is one contiguous 512 B / 256 B segment.  Loads are
// issued UNROLL columns ahead of the in-order accumulation, which keeps the
// reference's sequential summation order => bit-identical results.
// Algorithmic HBM bytes: stored_elements*(sizeof(T)+sizeof(I)) + 2 n sizeof(T).
#include <limits>

#include "common.hpp"
#include "csr_spmv_multi.hpp"
#include "scan.hpp"

namespace gkoc {
namespace {

constexpr int fmt_unroll = 8;

// One row of a column-major (ELL / SELL-P slice) layout: element i of the row
// lives at first + i * step.  Software-pipelined: while the b entries of chunk
// i are gathered, the val/col loads of chunk i+1 are already in flight; the
// products are added in column order (bit-identical to the reference loop).
// UNITB: b has unit row stride and j == 0 (the gather address is base + 8 col instead of a
// 64-bit multiply per entry)
// V: the type the matrix values are stored in (mixed precision: widened to T as they are used)
template <typename T, typename I, bool ADV, bool UNITB = false, typename V = T>
__device__ __forceinline__ T fmt_row_sum(T sum, int64_t len, int64_t first,
                                         int64_t step,
                                         const I* __restrict__ cols,
                                         const V* __restrict__ vals,
                                         const T* __restrict__ b, int64_t ldb,
                                         int j, T alpha)
{
    constexpr int U = fmt_unroll;
    const int64_t full = len / U * U;
    V v0[U], v1[U];
    I c0[U], c1[U];
    if (full > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v0[u] = vals[first + u * step];
            c0[u] = cols[first + u * step];
        }
    }
    int64_t i = 0;
    while (i < full) {
        T xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u] = c0[u] >= 0 ? (UNITB ? b[int64_t(c0[u])] : b[int64_t(c0[u]) * ldb + j]) : T(0);
        }
        const int64_t nx = i + U < full ? i + U : i;  // last chunk: harmless reload
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v1[u] = vals[first + (nx + u) * step];
            c1[u] = cols[first + (nx + u) * step];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const T t = ADV ? (alpha * T(v0[u])) * xv[u] : T(v0[u]) * xv[u];
            sum = c0[u] >= 0 ? sum + t : sum;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v0[u] = v1[u];
            c0[u] = c1[u];
        }
        i += U;
    }
    for (; i < len; ++i) {
        const I cc = cols[first + i * step];
        if (cc >= 0) {
            const T v = T(vals[first + i * step]);
            const T xv = UNITB ? b[int64_t(cc)] : b[int64_t(cc) * ldb + j];
            sum += ADV ? (alpha * v) * xv : v * xv;
        }
    }
    return sum;
}

// ELL / SELL-P SpMV with several right-hand sides: lane = row, one pass over the
// row's entries per chunk of NR columns (SELL: slice_sets != nullptr)
template <typename T, typename I, bool ADV, int NR, bool SELL>
__global__ __launch_bounds__(256) void fmt_spmv_multi_kernel(
    int64_t n_rows, int64_t k_per_row, int64_t stride, int64_t slice_size,
    const uint64_t* __restrict__ slice_sets, const uint64_t* __restrict__ slice_lengths,
    const I* __restrict__ cols, const T* __restrict__ vals, const T* __restrict__ b,
    int64_t ldb, T* __restrict__ c, int64_t ldc, int nrhs, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p, int64_t xcd_chunk = 0)
{
    const int64_t row = xcd_chunked_block(blockIdx.x, gridDim.x, xcd_chunk) * 256 + threadIdx.x;
    if (row >= n_rows) return;
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    int64_t len = k_per_row, first = row, step = stride;
    if (SELL) {
        const int64_t slice = row / slice_size;
        len = int64_t(slice_lengths[slice]);
        first = int64_t(slice_sets[slice]) * slice_size + (row - slice * slice_size);
        step = slice_size;
    }
    for (int j0 = 0; j0 < nrhs; j0 += NR) {
        int jcol[NR];
        T sum[NR];
#pragma unroll
        for (int jj = 0; jj < NR; ++jj) {
            jcol[jj] = j0 + jj < nrhs ? j0 + jj : nrhs - 1;
            sum[jj] = T(0);
            if (ADV && beta != T(0)) {
                sum[jj] = SELL ? c[row * ldc + jcol[jj]] * beta : beta * c[row * ldc + jcol[jj]];
            }
        }
        fmt_row_sum_multi<T, I, ADV, NR>(sum, len, first, step, cols, vals, b, ldb, jcol, alpha);
#pragma unroll
        for (int jj = 0; jj < NR; ++jj) {
            if (j0 + jj < nrhs) c[row * ldc + j0 + jj] = sum[jj];
        }
    }
}

// ELL / SELL-P SpMV with several right-hand sides, FRAGMENT layout: NR / 2 neighbouring lanes share a
// row, each owning two neighbouring columns of the chunk (one 16 B piece of the row-major b and c),
// and a wave walks its 64 rows in NR / 2 passes of 128 / NR rows that run side by side.  What this
// buys is the number of cache-line accesses of the vector L1: with lane = row (fmt_spmv_multi_kernel)
// every 16 B load of a lane is a line access of its own - 64 per instruction, 4 instructions per
// entry for eight columns - and the kernels ran AT the L1's rate of about one access per clock and
// CU (rocprofv3: 2.63 G accesses in 4.7 ms on L256 with eight columns; profiles/r03_multi_rhs_pmc.txt)
// while HBM idled at 2.9 TB/s.  Here the lanes of an instruction cover whole rows of b, 16 (eight
// columns) or 32 (four) contiguous 64 / 32 B runs.  The column index and value of an entry are
// loaded by all lanes of its row (same address: one access).  Every (row, column) sum is still
// formed by ONE lane in entry order, separate multiply and add: bit-identical to the reference.
template <typename T, typename I, bool ADV, int NR, bool SELL, bool IDX32>
__global__ __launch_bounds__(256) void fmt_spmv_frag_kernel(
    int64_t n_rows, int64_t k_per_row, int64_t stride, int64_t slice_size,
    const uint64_t* __restrict__ slice_sets, const uint64_t* __restrict__ slice_lengths,
    const I* __restrict__ cols, const T* __restrict__ vals, const T* __restrict__ b,
    int64_t ldb, T* __restrict__ c, int64_t ldc, int nrhs, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p, int64_t xcd_chunk)
{
    static_assert(NR == 4 || NR == 8, "chunks of 4 or 8 columns");
    constexpr int LPR = NR / 2;      // lanes per row
    constexpr int RPP = 64 / LPR;    // rows per pass
    constexpr int TT = LPR;          // passes side by side: 64 rows per wave
    using BV = vecT<T, 2>;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, rl = lane / LPR;
    const int64_t row_base =
        (xcd_chunked_block(blockIdx.x, gridDim.x, xcd_chunk) * 4 + (threadIdx.x >> 6)) * 64;
    if (row_base >= n_rows) return;
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    // lane = row view of the wave's 64 rows: the column index and value of entry k are loaded ONCE
    // per row (one coalesced instruction each) and handed to the row's NR / 2 lanes with
    // ds_bpermute; loading them from every lane of the row cost two of three vector-memory
    // instructions, and it is the instruction count the texture addresser is busy with
    // (rocprofv3: TA busy 90 % of the kernel's cycles; profiles/r03_multi_rhs_pmc.txt).
    const int64_t nrow = row_base + lane;
    const int64_t nr = nrow < n_rows ? nrow : n_rows - 1;
    int nlen = int(k_per_row);
    int64_t nfirst = nr, step = stride;
    if (SELL) {
        const int64_t slice = nr / slice_size;
        nlen = int(slice_lengths[slice]);
        nfirst = int64_t(slice_sets[slice]) * slice_size + (nr - slice * slice_size);
        step = slice_size;
    }
    if (nrow >= n_rows) nlen = 0;
    int maxlen = nlen;   // wave-wide maximum
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(maxlen, off, 64);
        maxlen = o > maxlen ? o : maxlen;
    }
    // The entry loop has NO branches: every lane always loads (an entry past the end of its row
    // re-reads the row's first entry and counts as padding, a padding entry gathers row 0 of b, a
    // lane whose columns lie past nrhs gathers columns 0 and 1) and the sums are selected -
    // conditional loads made the compiler wait for each of them in turn.
    for (int j0 = 0; j0 < nrhs; j0 += NR) {
        const int jc = j0 + 2 * sub;                         // the lane's first column
        const int ncol = nrhs - jc >= 2 ? 2 : nrhs - jc;     // its valid columns: 2, 1 or <= 0
        // the pair that is loaded: with one valid column its neighbour lies inside the row as well
        // (ldb is even where this kernel runs, so ldb > nrhs when nrhs is odd)
        const int jl = ncol >= 1 ? jc : 0;
        int64_t row[TT];
        T s0[TT], s1[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            row[t] = row_base + rl + RPP * t;
            const bool live = row[t] < n_rows && ncol > 0;
            s0[t] = s1[t] = T(0);
            if (ADV && beta != T(0) && live) {
                s0[t] = beta * c[row[t] * ldc + jc];
                if (ncol == 2) s1[t] = beta * c[row[t] * ldc + jc + 1];
            }
        }
        I ncc = I(-1);
        T nvv = T(0);
        if (maxlen > 0) {
            const I c_ = cols[nfirst];
            nvv = vals[nfirst];
            ncc = 0 < nlen ? c_ : I(-1);
        }
        for (int k = 0; k < maxlen; ++k) {
            I cc[TT];
            T vv[TT];
            BV x[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                cc[t] = __shfl(ncc, rl + RPP * t, 64);
                vv[t] = __shfl(nvv, rl + RPP * t, 64);
                const I ce = cc[t] >= 0 ? cc[t] : I(0);
                if (IDX32) {
                    const uint32_t off = uint32_t(ce) * uint32_t(ldb) + uint32_t(jl);
                    x[t] = *reinterpret_cast<const BV*>(b + off);
                } else {
                    x[t] = *reinterpret_cast<const BV*>(b + int64_t(ce) * ldb + jl);
                }
            }
            {
                const bool more = k + 1 < nlen;
                const int64_t at = nfirst + (more ? int64_t(k + 1) * step : int64_t(0));
                const I c_ = cols[at];
                nvv = vals[at];
                ncc = more ? c_ : I(-1);
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const T a = ADV ? alpha * vv[t] : vv[t];
                const T n0 = s0[t] + a * x[t].v[0];
                const T n1 = s1[t] + a * x[t].v[1];
                s0[t] = cc[t] >= 0 ? n0 : s0[t];
                s1[t] = cc[t] >= 0 ? n1 : s1[t];
            }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if (row[t] < n_rows && ncol > 0) {
                T* __restrict__ cp = c + row[t] * ldc + jc;
                if (ncol == 2) {
                    BV q;
                    q.v[0] = s0[t];
                    q.v[1] = s1[t];
                    *reinterpret_cast<BV*>(cp) = q;
                } else {
                    cp[0] = s0[t];
                }
            }
        }
    }
}

// the fragment layout serves three and more columns (two: the lane = row kernel is as fast, L256
// 1.30 ms both) and needs pairs of columns as aligned 2-element vectors of b and c
template <typename T>
inline bool frag_layout(int64_t nrhs, const T* b, int64_t ldb, const T* c, int64_t ldc)
{
    return nrhs >= 3 && reinterpret_cast<uintptr_t>(b) % (2 * sizeof(T)) == 0 && ldb % 2 == 0 &&
           reinterpret_cast<uintptr_t>(c) % (2 * sizeof(T)) == 0 && ldc % 2 == 0;
}

// ELL / SELL-P SpMV: lane = row, and every lane owns TWO rows, 64 apart, of the
// 128 consecutive rows of its wave, so that the wave ends with two back-to-back
// 512 B stores = one contiguous 1 KB burst.  (Sparse small writes between the
// read streams cost several times their byte share at the memory side, and
// only a burst of >= 1 KB issued by ONE wave avoids it - a barrier that lines
// up the 512 B stores of four waves does not)
template <typename T, typename I, bool ADV, typename V = T>
__global__ __launch_bounds__(256) void ell_spmv_kernel(
    int64_t n_rows, int64_t k_per_row, int64_t stride,
    const I* __restrict__ cols, const V* __restrict__ vals,
    const T* __restrict__ b, int64_t ldb, T* __restrict__ c, int64_t ldc,
    int nrhs, const T* __restrict__ alpha_p, const T* __restrict__ beta_p)
{
    const int64_t wave = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int64_t row0 = wave * 128 + (threadIdx.x & 63);
    const int64_t row1 = row0 + 64;
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    for (int j = 0; j < nrhs; ++j) {
        const bool unit_b = ldb == 1 && j == 0;   // uniform: one of the two instances runs
        T s0 = T(0), s1 = T(0);
        if (row0 < n_rows) {
            if (ADV && beta != T(0)) s0 = beta * c[row0 * ldc + j];
            s0 = unit_b ? fmt_row_sum<T, I, ADV, true, V>(s0, k_per_row, row0, stride, cols, vals, b, ldb, j, alpha)
                        : fmt_row_sum<T, I, ADV, false, V>(s0, k_per_row, row0, stride, cols, vals, b, ldb, j, alpha);
        }
        if (row1 < n_rows) {
            if (ADV && beta != T(0)) s1 = beta * c[row1 * ldc + j];
            s1 = unit_b ? fmt_row_sum<T, I, ADV, true, V>(s1, k_per_row, row1, stride, cols, vals, b, ldb, j, alpha)
                        : fmt_row_sum<T, I, ADV, false, V>(s1, k_per_row, row1, stride, cols, vals, b, ldb, j, alpha);
        }
        if (row0 < n_rows) c[row0 * ldc + j] = s0;
        if (row1 < n_rows) c[row1 * ldc + j] = s1;
    }
}

template <typename T, typename I, bool ADV>
__global__ __launch_bounds__(256) void sellp_spmv_kernel(
    int64_t n_rows, int64_t slice_size,
    const uint64_t* __restrict__ slice_sets,
    const uint64_t* __restrict__ slice_lengths, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ b, int64_t ldb,
    T* __restrict__ c, int64_t ldc, int nrhs, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p)
{
    const int64_t wave = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int64_t rows[2] = {wave * 128 + (threadIdx.x & 63), wave * 128 + 64 + (threadIdx.x & 63)};
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    for (int j = 0; j < nrhs; ++j) {
        const bool unit_b = ldb == 1 && j == 0;   // uniform: one of the two instances runs
        T sum[2] = {T(0), T(0)};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t row = rows[t];
            if (row < n_rows) {
                const int64_t slice = row / slice_size;
                const int64_t local = row - slice * slice_size;
                const int64_t len = int64_t(slice_lengths[slice]);
                const int64_t base = int64_t(slice_sets[slice]) * slice_size + local;
                if (ADV && beta != T(0)) sum[t] = c[row * ldc + j] * beta;
                sum[t] = unit_b ? fmt_row_sum<T, I, ADV, true>(sum[t], len, base, slice_size, cols, vals,
                                                               b, ldb, j, alpha)
                                : fmt_row_sum<T, I, ADV>(sum[t], len, base, slice_size, cols, vals, b,
                                                         ldb, j, alpha);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (rows[t] < n_rows) c[rows[t] * ldc + j] = sum[t];
        }
    }
}

// ------------------------------------------------------------- conversions
template <typename T, typename I>
__global__ __launch_bounds__(256) void csr_to_sellp_kernel(
    int64_t n_rows, int64_t slice_size, const I* __restrict__ row_ptrs,
    const I* __restrict__ cols, const T* __restrict__ vals,
    const uint64_t* __restrict__ slice_sets, I* __restrict__ s_cols,
    T* __restrict__ s_vals)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n_rows) return;
    const int64_t slice = row / slice_size;
    const int64_t local = row - slice * slice_size;
    const int64_t len_slice =
        int64_t(slice_sets[slice + 1]) - int64_t(slice_sets[slice]);
    const int64_t base = int64_t(slice_sets[slice]) * slice_size + local;
    const int64_t a = row_ptrs[row];
    const int64_t len = row_ptrs[row + 1] - a;
    for (int64_t i = 0; i < len_slice; ++i) {
        const bool in = i < len;
        s_vals[base + i * slice_size] = in ? vals[a + i] : T(0);
        s_cols[base + i * slice_size] = in ? cols[a + i] : I(-1);
    }
}

// Staged conversion of 64 consecutive rows per wave: the rows' contiguous CSR
// range goes through LDS with coalesced loads, then lane = row emits column j
// of all 64 rows per step, i.e. one contiguous 512 B / 256 B run per store -
// both sides of the copy are coalesced (the lane-walks-its-row kernels above
// reach 0.7 TB/s).  SELLP = false: ELL, element (row, j) at row + j * stride;
// SELLP = true (slice_size 64): (slice_sets[g] + j) * 64 + lane.  Segments with
// more than conv_stage_cap stored elements fall back to the direct loop.
constexpr int conv_stage_cap = 2048;

template <typename T, typename I, bool SELLP>
__global__ __launch_bounds__(64) void csr_to_colmajor_staged_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, int64_t k_per_row, int64_t stride,
    const uint64_t* __restrict__ slice_sets, I* __restrict__ out_cols,
    T* __restrict__ out_vals)
{
    __shared__ T lv[conv_stage_cap];
    __shared__ I lc[conv_stage_cap];
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x;
    const int64_t row = g * 64 + lane;
    const bool valid = row < n_rows;
    const int64_t last = (g + 1) * 64 < n_rows ? (g + 1) * 64 : n_rows;
    const int64_t rs = row_ptrs[valid ? row : last];
    const int64_t re = row_ptrs[valid ? row + 1 : last];
    const int64_t K0 = row_ptrs[g * 64];
    const int64_t K1 = row_ptrs[last];
    int64_t ncols, base, cstride;
    if (SELLP) {
        const int64_t s0 = int64_t(slice_sets[g]);
        ncols = int64_t(slice_sets[g + 1]) - s0;
        base = s0 * 64 + lane;
        cstride = 64;
    } else {
        ncols = k_per_row;
        base = row;
        cstride = stride;
    }
    const int64_t len = re - rs;
    if (K1 - K0 <= conv_stage_cap) {
        const int seg = int(K1 - K0);
        for (int i = lane; i < seg; i += 64) {
            lv[i] = vals[K0 + i];
            lc[i] = cols[K0 + i];
        }
        wave_lds_sync();
        const int off = int(rs - K0);
        if (valid) {
            for (int64_t j = 0; j < ncols; ++j) {
                const bool in = j < len;
                out_vals[base + j * cstride] = in ? lv[off + int(j)] : T(0);
                out_cols[base + j * cstride] = in ? lc[off + int(j)] : I(-1);
            }
        }
    } else if (valid) {
        for (int64_t j = 0; j < ncols; ++j) {
            const bool in = j < len;
            out_vals[base + j * cstride] = in ? vals[rs + j] : T(0);
            out_cols[base + j * cstride] = in ? cols[rs + j] : I(-1);
        }
    }
}

template <typename I>
__global__ __launch_bounds__(256) void max_row_nnz_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs,
    unsigned long long* __restrict__ result)
{
    __shared__ unsigned long long lds[4];
    unsigned long long m = 0;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n_rows;
         i += stride) {
        const unsigned long long len = (unsigned long long)(row_ptrs[i + 1] - row_ptrs[i]);
        m = len > m ? len : m;
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = lds[w] > m ? lds[w] : m;
        atomicMax(result, m);
    }
}

// one wave per slice when slice_size <= 64, else strided
template <typename I>
__global__ __launch_bounds__(256) void slice_lengths_kernel(
    int64_t n_rows, int64_t n_slices, int64_t slice_size,
    int64_t stride_factor, const I* __restrict__ row_ptrs,
    uint64_t* __restrict__ slice_lengths)
{
    const int lane = threadIdx.x & 63;
    const int64_t slice = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (slice >= n_slices) return;
    unsigned long long m = 0;
    for (int64_t l = lane; l < slice_size; l += 64) {
        const int64_t row = slice * slice_size + l;
        if (row < n_rows) {
            const unsigned long long len =
                (unsigned long long)(row_ptrs[row + 1] - row_ptrs[row]);
            const unsigned long long padded =
                (len + stride_factor - 1) / stride_factor * stride_factor;
            m = padded > m ? padded : m;
        }
    }
    m = wave_max(m);
    if (lane == 0) slice_lengths[slice] = m;
}

template <typename I>
__global__ __launch_bounds__(256) void ptrs_to_sizes_kernel(
    int64_t n, const I* __restrict__ ptrs, uint64_t* __restrict__ sizes)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) sizes[i] = uint64_t(ptrs[i + 1] - ptrs[i]);
}

// idxs sorted ascending; ptrs[r] = first position with idxs[pos] >= r
// (reference convert_idxs_to_ptrs, reference/components/format_conversion.hpp)
template <typename I, typename P = I>
__global__ __launch_bounds__(256) void idxs_to_ptrs_kernel(
    int64_t num_idxs, const I* __restrict__ idxs, int64_t n,
    P* __restrict__ ptrs)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i > num_idxs) return;
    const int64_t lo = i == 0 ? 0 : int64_t(idxs[i - 1]) + 1;
    const int64_t hi = i == num_idxs ? n : int64_t(idxs[i]);
    for (int64_t r = lo; r <= hi && r <= n; ++r) ptrs[r] = P(i);
}

template <typename I>
__global__ __launch_bounds__(256) void fill_idx_kernel(int64_t n, I* data,
                                                        I value, bool seq)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
        data[i] = seq ? I(i) : value;
    }
}

// Ginkgo's matrix_data_entry<T, I> (matrix_data.hpp:60): { I row; I column; T value; }
template <typename T, typename I>
struct md_entry {
    I row;
    I column;
    T value;
};

template <typename T, typename I>
__global__ __launch_bounds__(256) void aos_to_soa_kernel(
    int64_t nnz, const md_entry<T, I>* __restrict__ in, I* __restrict__ rows,
    I* __restrict__ cols, T* __restrict__ vals)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < nnz; i += stride) {
        const md_entry<T, I> e = in[i];
        rows[i] = e.row;
        cols[i] = e.column;
        vals[i] = e.value;
    }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void fill_in_md_kernel(
    int64_t nnz, const I* __restrict__ rows, const I* __restrict__ cols,
    const T* __restrict__ vals, T* __restrict__ out, int64_t ld)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < nnz; i += stride) {
        out[int64_t(rows[i]) * ld + cols[i]] = vals[i];
    }
}

inline unsigned blocks_for(int64_t n)
{
    return unsigned(ceildiv(n > 0 ? n : 1, 256));
}

template <typename T, typename I, bool ADV>
int launch_ell(gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t k,
               int64_t stride, const T* alpha, const I* cols, const T* vals,
               const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc,
               int64_t nrhs)
{
    (void)n_cols;
    GKOC_REQUIRE(n_rows >= 0 && k >= 0 && nrhs >= 0 && stride >= n_rows,
                 GKOC_E_INVALID, "bad ELL dimensions");
    if (n_rows == 0 || nrhs == 0) return GKOC_OK;
    if (ADV) GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha/beta");
    if (nrhs >= 2 && frag_layout(nrhs, b, ldb, c, ldc)) {
        const dim3 grid(unsigned(ceildiv(n_rows, 256)));
        const int64_t chunk = tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS) / 256;
        const bool idx32 = n_cols * ldb < (int64_t(1) << 32);
#define GKOC_LAUNCH_FRAG(NR_)                                                                    \
    do {                                                                                         \
        if (idx32) {                                                                             \
            fmt_spmv_frag_kernel<T, I, ADV, NR_, false, true><<<grid, dim3(256), 0, as_stream(s)>>>( \
                n_rows, k, stride, 0, nullptr, nullptr, cols, vals, b, ldb, c, ldc, int(nrhs),   \
                alpha, beta, chunk);                                                             \
        } else {                                                                                 \
            fmt_spmv_frag_kernel<T, I, ADV, NR_, false, false><<<grid, dim3(256), 0, as_stream(s)>>>( \
                n_rows, k, stride, 0, nullptr, nullptr, cols, vals, b, ldb, c, ldc, int(nrhs),   \
                alpha, beta, chunk);                                                             \
        }                                                                                        \
    } while (0)
        if (nrhs <= 4) {
            GKOC_LAUNCH_FRAG(4);
        } else {
            GKOC_LAUNCH_FRAG(8);
        }
#undef GKOC_LAUNCH_FRAG
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    if (nrhs >= 2) {
        const dim3 grid(unsigned(ceildiv(n_rows, 256)));
        if (nrhs == 2) {
            fmt_spmv_multi_kernel<T, I, ADV, 2, false><<<grid, dim3(256), 0, as_stream(s)>>>(
                n_rows, k, stride, 0, nullptr, nullptr, cols, vals, b, ldb, c, ldc, int(nrhs),
                alpha, beta, tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS) / 256);
        } else if (nrhs <= 4) {
            fmt_spmv_multi_kernel<T, I, ADV, 4, false><<<grid, dim3(256), 0, as_stream(s)>>>(
                n_rows, k, stride, 0, nullptr, nullptr, cols, vals, b, ldb, c, ldc, int(nrhs),
                alpha, beta, tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS) / 256);
        } else {
            // wider blocks: 8 columns per pass use every gathered b line in full
            fmt_spmv_multi_kernel<T, I, ADV, 8, false><<<grid, dim3(256), 0, as_stream(s)>>>(
                n_rows, k, stride, 0, nullptr, nullptr, cols, vals, b, ldb, c, ldc, int(nrhs),
                alpha, beta, tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS) / 256);
        }
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    ell_spmv_kernel<T, I, ADV><<<dim3(unsigned(ceildiv(n_rows, 512))), dim3(256), 0, as_stream(s)>>>(
        n_rows, k, stride, cols, vals, b, ldb, c, ldc, int(nrhs), alpha, beta);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// mixed precision (values float, vectors and arithmetic double): the one-column kernel, column by column
template <typename T, typename V, typename I, bool ADV>
int launch_ell_mixed(gkoc_stream_t s, int64_t n_rows, int64_t k, int64_t stride, const T* alpha,
                     const I* cols, const V* vals, const T* b, int64_t ldb, const T* beta, T* c,
                     int64_t ldc, int64_t nrhs)
{
    GKOC_REQUIRE(n_rows >= 0 && k >= 0 && nrhs >= 0 && stride >= n_rows, GKOC_E_INVALID,
                 "bad ELL dimensions");
    if (n_rows == 0 || nrhs == 0) return GKOC_OK;
    if (ADV) GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha/beta");
    ell_spmv_kernel<T, I, ADV, V><<<dim3(unsigned(ceildiv(n_rows, 512))), dim3(256), 0, as_stream(s)>>>(
        n_rows, k, stride, cols, vals, b, ldb, c, ldc, int(nrhs), alpha, beta);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename T, typename I, bool ADV>
int launch_sellp(gkoc_stream_t s, int64_t n_rows, int64_t n_cols,
                 int64_t slice_size, const T* alpha, const uint64_t* slice_sets,
                 const uint64_t* slice_lengths, const I* cols, const T* vals,
                 const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc,
                 int64_t nrhs)
{
    (void)n_cols;
    GKOC_REQUIRE(n_rows >= 0 && slice_size >= 1 && nrhs >= 0, GKOC_E_INVALID,
                 "bad SELL-P dimensions");
    if (n_rows == 0 || nrhs == 0) return GKOC_OK;
    if (ADV) GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha/beta");
    if (nrhs >= 2 && frag_layout(nrhs, b, ldb, c, ldc)) {
        const dim3 grid(unsigned(ceildiv(n_rows, 256)));
        const int64_t chunk = tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS) / 256;
        const bool idx32 = n_cols * ldb < (int64_t(1) << 32);
#define GKOC_LAUNCH_FRAG(NR_)                                                                    \
    do {                                                                                         \
        if (idx32) {                                                                             \
            fmt_spmv_frag_kernel<T, I, ADV, NR_, true, true><<<grid, dim3(256), 0, as_stream(s)>>>( \
                n_rows, 0, 0, slice_size, slice_sets, slice_lengths, cols, vals, b, ldb, c, ldc,     \
                int(nrhs), alpha, beta, chunk);                                                  \
        } else {                                                                                 \
            fmt_spmv_frag_kernel<T, I, ADV, NR_, true, false><<<grid, dim3(256), 0, as_stream(s)>>>( \
                n_rows, 0, 0, slice_size, slice_sets, slice_lengths, cols, vals, b, ldb, c, ldc,     \
                int(nrhs), alpha, beta, chunk);                                                  \
        }                                                                                        \
    } while (0)
        if (nrhs <= 4) {
            GKOC_LAUNCH_FRAG(4);
        } else {
            GKOC_LAUNCH_FRAG(8);
        }
#undef GKOC_LAUNCH_FRAG
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    if (nrhs >= 2) {
        const dim3 grid(unsigned(ceildiv(n_rows, 256)));
        if (nrhs == 2) {
            fmt_spmv_multi_kernel<T, I, ADV, 2, true><<<grid, dim3(256), 0, as_stream(s)>>>(
                n_rows, 0, 0, slice_size, slice_sets, slice_lengths, cols, vals, b, ldb, c, ldc,
                int(nrhs), alpha, beta,
                tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS) / 256);
        } else if (nrhs <= 4) {
            fmt_spmv_multi_kernel<T, I, ADV, 4, true><<<grid, dim3(256), 0, as_stream(s)>>>(
                n_rows, 0, 0, slice_size, slice_sets, slice_lengths, cols, vals, b, ldb, c, ldc,
                int(nrhs), alpha, beta,
                tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS) / 256);
        } else {
            fmt_spmv_multi_kernel<T, I, ADV, 8, true><<<grid, dim3(256), 0, as_stream(s)>>>(
                n_rows, 0, 0, slice_size, slice_sets, slice_lengths, cols, vals, b, ldb, c, ldc,
                int(nrhs), alpha, beta,
                tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS) / 256);
        }
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    sellp_spmv_kernel<T, I, ADV>
        <<<dim3(unsigned(ceildiv(n_rows, 512))), dim3(256), 0, as_stream(s)>>>(
            n_rows, slice_size, slice_sets, slice_lengths, cols, vals, b, ldb, c,
            ldc, int(nrhs), alpha, beta);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_ELL_MIXED(I, IN)                                                                  \
    extern "C" int gkoc_ell_spmv_f32_f64_##IN(gkoc_stream_t s, int64_t n_rows, int64_t n_cols,     \
                                              int64_t k, int64_t stride, const I* cols,            \
                                              const float* vals, const double* b, int64_t ldb,     \
                                              double* c, int64_t ldc, int64_t nrhs)                \
    {                                                                                              \
        (void)n_cols;                                                                              \
        return launch_ell_mixed<double, float, I, false>(s, n_rows, k, stride, nullptr, cols,      \
                                                         vals, b, ldb, nullptr, c, ldc, nrhs);     \
    }                                                                                              \
    extern "C" int gkoc_ell_advanced_spmv_f32_f64_##IN(                                            \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t k, int64_t stride,                \
        const double* alpha, const I* cols, const float* vals, const double* b, int64_t ldb,       \
        const double* beta, double* c, int64_t ldc, int64_t nrhs)                                  \
    {                                                                                              \
        (void)n_cols;                                                                              \
        return launch_ell_mixed<double, float, I, true>(s, n_rows, k, stride, alpha, cols, vals,   \
                                                        b, ldb, beta, c, ldc, nrhs);               \
    }
GKOC_DEF_ELL_MIXED(int32_t, i32)
GKOC_DEF_ELL_MIXED(int64_t, i64)

#define GKOC_DEF_FMT(T, TN, I, IN)                                             \
    extern "C" int gkoc_ell_spmv_##TN##_##IN(                                  \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t k,            \
        int64_t stride, const I* cols, const T* vals, const T* b,              \
        int64_t ldb, T* c, int64_t ldc, int64_t nrhs)                          \
    {                                                                          \
        return launch_ell<T, I, false>(s, n_rows, n_cols, k, stride, nullptr,  \
                                       cols, vals, b, ldb, nullptr, c, ldc,    \
                                       nrhs);                                  \
    }                                                                          \
    extern "C" int gkoc_ell_advanced_spmv_##TN##_##IN(                         \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t k,            \
        int64_t stride, const T* alpha, const I* cols, const T* vals,          \
        const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc,             \
        int64_t nrhs)                                                          \
    {                                                                          \
        return launch_ell<T, I, true>(s, n_rows, n_cols, k, stride, alpha,     \
                                      cols, vals, b, ldb, beta, c, ldc, nrhs); \
    }                                                                          \
    extern "C" int gkoc_sellp_spmv_##TN##_##IN(                                \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t slice_size,   \
        const uint64_t* slice_sets, const uint64_t* slice_lengths,             \
        const I* cols, const T* vals, const T* b, int64_t ldb, T* c,           \
        int64_t ldc, int64_t nrhs)                                             \
    {                                                                          \
        return launch_sellp<T, I, false>(s, n_rows, n_cols, slice_size,        \
                                         nullptr, slice_sets, slice_lengths,   \
                                         cols, vals, b, ldb, nullptr, c, ldc,  \
                                         nrhs);                                \
    }                                                                          \
    extern "C" int gkoc_sellp_advanced_spmv_##TN##_##IN(                       \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, int64_t slice_size,   \
        const T* alpha, const uint64_t* slice_sets,                            \
        const uint64_t* slice_lengths, const I* cols, const T* vals,           \
        const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc,             \
        int64_t nrhs)                                                          \
    {                                                                          \
        return launch_sellp<T, I, true>(s, n_rows, n_cols, slice_size, alpha,  \
                                        slice_sets, slice_lengths, cols, vals, \
                                        b, ldb, beta, c, ldc, nrhs);           \
    }
// (the conversions: also for the complex types - the complex products are in complex_formats.hip)
#define GKOC_DEF_FMT_CONVERT(T, TN, I, IN)                                     \
    extern "C" int gkoc_csr_convert_to_ell_##TN##_##IN(                        \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* cols,     \
        const T* vals, int64_t k, int64_t stride, I* ell_cols, T* ell_vals)    \
    {                                                                          \
        if (n_rows <= 0) return GKOC_OK;                                       \
        csr_to_colmajor_staged_kernel<T, I, false>                             \
            <<<dim3(unsigned(ceildiv(n_rows, 64))), dim3(64), 0,               \
               as_stream(s)>>>(n_rows, row_ptrs, cols, vals, k, stride,        \
                               nullptr, ell_cols, ell_vals);                   \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_csr_convert_to_sellp_##TN##_##IN(                      \
        gkoc_stream_t s, int64_t n_rows, int64_t slice_size,                   \
        const I* row_ptrs, const I* cols, const T* vals,                       \
        const uint64_t* slice_sets, I* s_cols, T* s_vals)                      \
    {                                                                          \
        if (n_rows <= 0) return GKOC_OK;                                       \
        if (slice_size == 64) {                                                \
            csr_to_colmajor_staged_kernel<T, I, true>                          \
                <<<dim3(unsigned(ceildiv(n_rows, 64))), dim3(64), 0,           \
                   as_stream(s)>>>(n_rows, row_ptrs, cols, vals, 0, 0,         \
                                   slice_sets, s_cols, s_vals);                \
            GKOC_LAUNCH_OK();                                                  \
            return GKOC_OK;                                                    \
        }                                                                      \
        csr_to_sellp_kernel<T, I>                                              \
            <<<dim3(blocks_for(n_rows)), dim3(256), 0, as_stream(s)>>>(        \
                n_rows, slice_size, row_ptrs, cols, vals, slice_sets, s_cols,  \
                s_vals);                                                       \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }

#define GKOC_DEF_MD(T, TN, I, IN)                                              \
    extern "C" int gkoc_aos_to_soa_##TN##_##IN(gkoc_stream_t s, int64_t nnz,   \
                                               const void* entries,            \
                                               I* row_idxs, I* col_idxs,       \
                                               T* vals)                        \
    {                                                                          \
        if (nnz <= 0) return GKOC_OK;                                          \
        int64_t nb = ceildiv(nnz, 256);                                        \
        if (nb > max_stream_blocks) nb = max_stream_blocks;                    \
        aos_to_soa_kernel<T, I>                                                \
            <<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(              \
                nnz, static_cast<const md_entry<T, I>*>(entries), row_idxs,    \
                col_idxs, vals);                                               \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_dense_fill_in_matrix_data_##TN##_##IN(                 \
        gkoc_stream_t s, int64_t nnz, const I* row_idxs, const I* col_idxs,    \
        const T* vals, T* out, int64_t ld_out)                                 \
    {                                                                          \
        if (nnz <= 0) return GKOC_OK;                                          \
        int64_t nb = ceildiv(nnz, 256);                                        \
        if (nb > max_stream_blocks) nb = max_stream_blocks;                    \
        fill_in_md_kernel<T, I>                                                \
            <<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(              \
                nnz, row_idxs, col_idxs, vals, out, ld_out);                   \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }
#define GKOC_DEF_COMPLEX_MD(T, TN, I, IN)                                       \
    extern "C" int gkoc_aos_to_soa_##TN##_##IN(gkoc_stream_t s, int64_t nnz,   \
                                               const void* entries,            \
                                               I* row_idxs, I* col_idxs,       \
                                               T* vals)                        \
    {                                                                          \
        if (nnz <= 0) return GKOC_OK;                                          \
        int64_t nb = ceildiv(nnz, 256);                                        \
        if (nb > max_stream_blocks) nb = max_stream_blocks;                    \
        aos_to_soa_kernel<T, I>                                                \
            <<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(              \
                nnz, static_cast<const md_entry<T, I>*>(entries), row_idxs,    \
                col_idxs, vals);                                               \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }
GKOC_DEF_COMPLEX_MD(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_COMPLEX_MD(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_COMPLEX_MD(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_COMPLEX_MD(gkoc_c64, c64, int64_t, i64)
GKOC_DEF_MD(double, f64, int32_t, i32)
GKOC_DEF_MD(double, f64, int64_t, i64)
GKOC_DEF_MD(float, f32, int32_t, i32)
GKOC_DEF_MD(float, f32, int64_t, i64)

GKOC_DEF_FMT(double, f64, int32_t, i32)
GKOC_DEF_FMT(double, f64, int64_t, i64)
GKOC_DEF_FMT(float, f32, int32_t, i32)
GKOC_DEF_FMT(float, f32, int64_t, i64)
GKOC_DEF_FMT_CONVERT(double, f64, int32_t, i32)
GKOC_DEF_FMT_CONVERT(double, f64, int64_t, i64)
GKOC_DEF_FMT_CONVERT(float, f32, int32_t, i32)
GKOC_DEF_FMT_CONVERT(float, f32, int64_t, i64)
GKOC_DEF_FMT_CONVERT(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_FMT_CONVERT(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_FMT_CONVERT(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_FMT_CONVERT(gkoc_c64, c64, int64_t, i64)

#define GKOC_DEF_IDX(I, IN)                                                    \
    extern "C" int gkoc_compute_max_row_nnz_##IN(                              \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, int64_t* max_host) \
    {                                                                          \
        GKOC_REQUIRE(max_host, GKOC_E_INVALID, "null result");                 \
        *max_host = 0;                                                         \
        if (n_rows <= 0) return GKOC_OK;                                       \
        unsigned long long* d = nullptr;                                       \
        GKOC_TRY(scratch_malloc(as_stream(s), reinterpret_cast<void**>(&d), 8)); \
        GKOC_HIP(hipMemsetAsync(d, 0, 8, as_stream(s)));                       \
        int64_t nb = ceildiv(n_rows, 256);                                     \
        if (nb > max_stream_blocks) nb = max_stream_blocks;                    \
        max_row_nnz_kernel<I><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>( \
            n_rows, row_ptrs, d);                                              \
        GKOC_LAUNCH_OK();                                                      \
        unsigned long long h = 0;                                              \
        GKOC_HIP(hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, as_stream(s))); \
        GKOC_HIP(hipStreamSynchronize(as_stream(s)));                          \
        GKOC_TRY(scratch_free(as_stream(s), d));                               \
        *max_host = int64_t(h);                                                \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_sellp_compute_slice_sets_##IN(                         \
        gkoc_stream_t s, int64_t n_rows, int64_t slice_size,                   \
        int64_t stride_factor, const I* row_ptrs, uint64_t* slice_sets,        \
        uint64_t* slice_lengths)                                               \
    {                                                                          \
        GKOC_REQUIRE(slice_size >= 1 && stride_factor >= 1, GKOC_E_INVALID,    \
                     "bad slice parameters");                                  \
        const int64_t n_slices = ceildiv(n_rows, slice_size);                  \
        if (n_slices > 0) {                                                    \
            slice_lengths_kernel<I>                                            \
                <<<dim3(unsigned(ceildiv(n_slices, 4))), dim3(256), 0,         \
                   as_stream(s)>>>(n_rows, n_slices, slice_size,               \
                                   stride_factor, row_ptrs, slice_lengths);    \
            GKOC_LAUNCH_OK();                                                  \
            GKOC_HIP(hipMemcpyAsync(slice_sets, slice_lengths,                 \
                                    sizeof(uint64_t) * n_slices,               \
                                    hipMemcpyDeviceToDevice, as_stream(s)));   \
        }                                                                      \
        GKOC_HIP(hipMemsetAsync(slice_sets + n_slices, 0, sizeof(uint64_t),    \
                                as_stream(s)));                                \
        return device_exclusive_scan<unsigned long long>(                      \
            as_stream(s), reinterpret_cast<unsigned long long*>(slice_sets),   \
            n_slices + 1);                                                     \
    }                                                                          \
    extern "C" int gkoc_convert_ptrs_to_sizes_##IN(                            \
        gkoc_stream_t s, int64_t n, const I* ptrs, uint64_t* sizes)            \
    {                                                                          \
        if (n <= 0) return GKOC_OK;                                            \
        ptrs_to_sizes_kernel<I>                                                \
            <<<dim3(blocks_for(n)), dim3(256), 0, as_stream(s)>>>(n, ptrs,     \
                                                                  sizes);      \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_convert_idxs_to_ptrs_##IN(                             \
        gkoc_stream_t s, int64_t num_idxs, const I* idxs, int64_t n, I* ptrs)  \
    {                                                                          \
        GKOC_REQUIRE(num_idxs >= 0 && n >= 0, GKOC_E_INVALID, "negative size"); \
        idxs_to_ptrs_kernel<I>                                                 \
            <<<dim3(blocks_for(num_idxs + 1)), dim3(256), 0, as_stream(s)>>>(  \
                num_idxs, idxs, n, ptrs);                                      \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_prefix_sum_nonnegative_##IN(gkoc_stream_t s,           \
                                                    I* counts, int64_t n)      \
    {                                                                          \
        return device_exclusive_scan<I>(as_stream(s), counts, n);              \
    }                                                                          \
    extern "C" int gkoc_prefix_sum_nonnegative_checked_##IN(                   \
        gkoc_stream_t s, I* counts, int64_t n)                                 \
    {                                                                          \
        int over = 0;                                                          \
        const int rc = scan_overflows(as_stream(s), counts, n - 1, sizeof(I),  \
                                      (unsigned long long)(std::numeric_limits<I>::max()), \
                                      &over);                                  \
        if (rc != GKOC_OK) return rc;                                          \
        GKOC_REQUIRE(!over, GKOC_E_OVERFLOW, "prefix sum overflows " #I);      \
        return device_exclusive_scan<I>(as_stream(s), counts, n);              \
    }                                                                          \
    extern "C" int gkoc_fill_array_##IN(gkoc_stream_t s, I* data, int64_t n,   \
                                        I value)                               \
    {                                                                          \
        if (n <= 0) return GKOC_OK;                                            \
        int64_t nb = ceildiv(n, 256);                                          \
        if (nb > max_stream_blocks) nb = max_stream_blocks;                    \
        fill_idx_kernel<I><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>( \
            n, data, value, false);                                            \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_fill_seq_array_##IN(gkoc_stream_t s, I* data,          \
                                            int64_t n)                         \
    {                                                                          \
        if (n <= 0) return GKOC_OK;                                            \
        int64_t nb = ceildiv(n, 256);                                          \
        if (nb > max_stream_blocks) nb = max_stream_blocks;                    \
        fill_idx_kernel<I><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>( \
            n, data, I(0), true);                                              \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }

GKOC_DEF_IDX(int32_t, i32)
GKOC_DEF_IDX(int64_t, i64)

// 64-bit row pointers (device_matrix_data readers) narrowed for matrices with 32-bit indices
__global__ __launch_bounds__(256) void narrow_i64_kernel(int64_t n, const int64_t* __restrict__ in,
                                                         int32_t* __restrict__ out)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
        out[i] = int32_t(in[i]);
    }
}

extern "C" int gkoc_narrow_i64_to_i32(gkoc_stream_t s, int64_t n, const int64_t* in, int32_t* out)
{
    if (n <= 0) return GKOC_OK;
    GKOC_REQUIRE(in && out, GKOC_E_INVALID, "null pointer");
    int64_t nb = ceildiv(n, 256);
    if (nb > max_stream_blocks) nb = max_stream_blocks;
    narrow_i64_kernel<<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(n, in, out);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// row pointers of another width than the indices (Ell / Sellp / Hybrid read their
// device_matrix_data with 64-bit row pointers: GKO_DECLARE_CONVERT_IDXS_TO_PTRS64)
extern "C" int gkoc_convert_idxs_to_ptrs_i32_i64(gkoc_stream_t s, int64_t num_idxs,
                                                 const int32_t* idxs, int64_t n, int64_t* ptrs)
{
    GKOC_REQUIRE(num_idxs >= 0 && n >= 0, GKOC_E_INVALID, "negative size");
    idxs_to_ptrs_kernel<int32_t, int64_t>
        <<<dim3(blocks_for(num_idxs + 1)), dim3(256), 0, as_stream(s)>>>(num_idxs, idxs, n, ptrs);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

extern "C" int gkoc_convert_idxs_to_ptrs_i64_i32(gkoc_stream_t s, int64_t num_idxs,
                                                 const int64_t* idxs, int64_t n, int32_t* ptrs)
{
    GKOC_REQUIRE(num_idxs >= 0 && n >= 0, GKOC_E_INVALID, "negative size");
    idxs_to_ptrs_kernel<int64_t, int32_t>
        <<<dim3(blocks_for(num_idxs + 1)), dim3(256), 0, as_stream(s)>>>(num_idxs, idxs, n, ptrs);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

extern "C" int gkoc_prefix_sum_nonnegative_checked_u64(gkoc_stream_t s, uint64_t* counts,
                                                       int64_t n)
{
    int over = 0;
    const int rc = scan_overflows(as_stream(s), counts, n - 1, 8, ~0ull, &over);
    if (rc != GKOC_OK) return rc;
    GKOC_REQUIRE(!over, GKOC_E_OVERFLOW, "prefix sum overflows uint64_t");
    return device_exclusive_scan<unsigned long long>(
        as_stream(s), reinterpret_cast<unsigned long long*>(counts), n);
}

extern "C" int gkoc_prefix_sum_nonnegative_u64(gkoc_stream_t s, uint64_t* counts,
                                               int64_t n)
{
    return device_exclusive_scan<unsigned long long>(
        as_stream(s), reinterpret_cast<unsigned long long*>(counts), n);
}
