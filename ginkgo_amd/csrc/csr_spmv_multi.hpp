// CSR SpMV for SEVERAL right-hand sides in one pass over the matrix (gfx950).
//
// The single-column kernel (csr_spmv_pipe.hpp) run once per column re-reads the
// whole matrix for every column and gathers b with a stride of ldb values
// (L256: 1.17 / 2.18 / 3.67 / 5.45 / 16.8 ms for 1 / 2 / 3 / 4 / 8 columns).
// Two kernels stream the matrix ONCE per chunk of columns instead:
//   * csr_spmv_multi_kernel, two columns: the row-segment-per-wavefront walk of
//     csr_spmv_pipe3_kernel (two register sets, 32 B vector loads of the matrix
//     stream); lane = E consecutive nonzeros gathers the two values b[col, j0..j0+2)
//     as one 16 B load, the two products of a nonzero sit next to each other in the
//     LDS ring, lane = row adds them in k order per column;
//   * csr_spmv_frag_kernel, three and more columns (chunks of 4 or 8): 16-row waves
//     with the segment staged in LDS and several lanes per row, so that the lanes of
//     one gather instruction cover whole rows of b (see there).
// In both, every (row, column) sum is formed in the reference's order with separate
// multiply and add => bit-identical per column to the sequential reference and to the
// single-column kernel.  Columns beyond nrhs in the last chunk are not stored.
// fmt_row_sum_multi below is the lane = row walk the ELL / SELL-P fallback kernels use.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "csr_spmv_pipe.hpp"

namespace gkoc {

#ifdef __HIPCC__

// The same for a chunk of NR right-hand sides: the row's entries are read ONCE, the NR
// values b[col, jcol[0..NR)) of an entry are neighbours in the row-major b (one cache
// line), and NR sums are carried; per column the products are added in column order,
// so every column is bit-identical to the single-column kernels.  jcol = column behind slot jj
// (slots past nrhs repeat the last column and are not stored).
template <typename T, typename I, bool ADV, int NR>
__device__ __forceinline__ void fmt_row_sum_multi(T (&sum)[NR], int64_t len, int64_t first,
                                                  int64_t step, const I* __restrict__ cols,
                                                  const T* __restrict__ vals,
                                                  const T* __restrict__ b, int64_t ldb,
                                                  const int (&jcol)[NR], T alpha)
{
    constexpr int U = 4;
    const int64_t full = len / U * U;
    T v0[U], v1[U];
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
        T xv[U][NR];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const T* __restrict__ brow = b + int64_t(c0[u] >= 0 ? c0[u] : I(0)) * ldb;
#pragma unroll
            for (int jj = 0; jj < NR; ++jj) xv[u][jj] = c0[u] >= 0 ? brow[jcol[jj]] : T(0);
        }
        const int64_t nx = i + U < full ? i + U : i;  // last chunk: harmless reload
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v1[u] = vals[first + (nx + u) * step];
            c1[u] = cols[first + (nx + u) * step];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int jj = 0; jj < NR; ++jj) {
                const T t = ADV ? (alpha * v0[u]) * xv[u][jj] : v0[u] * xv[u][jj];
                sum[jj] = c0[u] >= 0 ? sum[jj] + t : sum[jj];
            }
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
            const T v = vals[first + i * step];
            const T* __restrict__ brow = b + int64_t(cc) * ldb;
#pragma unroll
            for (int jj = 0; jj < NR; ++jj) {
                const T xv = brow[jcol[jj]];
                sum[jj] += ADV ? (alpha * v) * xv : v * xv;
            }
        }
    }
}

// CSR, several right-hand sides, FRAGMENT layout (the CSR form of fmt_spmv_frag_kernel in
// formats.hip): the wave's ROWS-row segment is staged in LDS with coalesced loads as above, then
// NR / 2 neighbouring lanes share a row, each owning two neighbouring columns, and the wave walks
// TT groups of 128 / NR rows side by side, KU entries per step.  The lanes of one gather
// instruction then cover whole rows of b - for banded / stencil matrices 32 (four columns) or 16
// (eight) consecutive rows, one contiguous run - where lane = row reads 16 B per lane from 64
// different rows, and it is the vector L1's line-access rate, not HBM, that bounds the lane = row
// kernels from four columns on (rocprofv3, profiles/r03_multi_rhs_pmc.txt).  The column index and
// value of an entry come from LDS (same address for the lanes of a row: a broadcast).  Every (row,
// column) sum is formed by ONE lane in entry order, separate multiply and add: bit-identical to
// the reference.  No branches in the entry loop: an entry past the end of a row reads entry 0 of
// the segment and row 0 of b, and its product is not added.
template <typename T, typename I, bool ADV, int NR, int CPL, int TT, int KU, bool IDX32>
__global__ __launch_bounds__(64) void csr_spmv_frag_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ b, int64_t ldb, T* __restrict__ c,
    int64_t ldc, int nrhs, const T* __restrict__ alpha_p, const T* __restrict__ beta_p,
    int64_t xcd_chunk)
{
    static_assert(NR == 2 || NR == 4 || NR == 8, "chunks of 2, 4 or 8 columns");
    static_assert(CPL == 1 || CPL == 2, "one or two columns per lane");
    constexpr int LPR = NR / CPL;     // lanes per row
    constexpr int RPP = 64 / LPR;     // rows per group
    constexpr int ROWS = RPP * TT;    // rows per wave
    constexpr int CAP = ROWS * 32;    // staged entries (12 B each)
    using BV = vecT<T, CPL>;
    __shared__ __attribute__((aligned(16))) T lv[CAP];
    __shared__ __attribute__((aligned(16))) I lc[CAP];
    const int lane = threadIdx.x;
    const int sub = lane % LPR, rl = lane / LPR;
    const int64_t g = xcd_chunked_block(blockIdx.x, gridDim.x, xcd_chunk);
    const int64_t row0 = g * ROWS;
    if (row0 >= n_rows) return;
    const int64_t last = row0 + ROWS < n_rows ? row0 + ROWS : n_rows;
    const int64_t K0 = row_ptrs[row0];
    const int64_t K1 = row_ptrs[last];
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    // a segment of up to CAP entries is staged as it lies in memory; a longer one in rounds of 32
    // entries of every row (slot = 32 * row + entry)
    const bool whole = K1 - K0 <= CAP;
    if (whole) {
        const int seg = int(K1 - K0);
        for (int i = lane; i < seg; i += 64) {
            lv[i] = vals[K0 + i];
            lc[i] = cols[K0 + i];
        }
        wave_lds_sync();
    }
    int64_t row[TT];
    int rs[TT], len[TT];
    int maxlen = 0;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        row[t] = row0 + rl + RPP * t;
        const int64_t r = row[t] < last ? row[t] : last;
        const int64_t a = row_ptrs[r];
        const int64_t e = row[t] < last ? int64_t(row_ptrs[r + 1]) : a;
        rs[t] = int(a - K0);
        len[t] = int(e - a);
        maxlen = len[t] > maxlen ? len[t] : maxlen;
    }
    int wave_maxlen = maxlen;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(wave_maxlen, off, 64);
        wave_maxlen = o > wave_maxlen ? o : wave_maxlen;
    }
    for (int j0 = 0; j0 < nrhs; j0 += NR) {
        const int jc = j0 + CPL * sub;                       // the lane's first column
        const int ncol = nrhs - jc >= CPL ? CPL : nrhs - jc; // its valid columns: CPL ... <= 0
        // what is loaded: with one valid column of a pair the neighbour lies inside the row as well
        // (ldb is even where pairs are used, so ldb > nrhs when nrhs is odd); a lane without
        // columns reads the first ones
        const int jl = ncol >= 1 ? jc : 0;
        T sum[TT][CPL];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                sum[t][q] = T(0);
                if (ADV && beta != T(0) && row[t] < last && q < ncol) {
                    sum[t][q] = c[row[t] * ldc + jc + q] * beta;
                }
            }
        }
        // entries [kb, ke) of every row, entry k of the lane's row t at LDS slot base[t] + k
        auto walk = [&](auto tiled_tag, int kb, int ke, const int (&base)[TT]) {
            constexpr bool TILED = decltype(tiled_tag)::value;
            const int kend = TILED && ke < maxlen ? ke : maxlen;
            for (int k = kb; k < kend; k += KU) {
                BV x[KU][TT];
                T vv[KU][TT];
                bool ok[KU][TT];
#pragma unroll
                for (int u = 0; u < KU; ++u) {
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        ok[u][t] = k + u < len[t] && (!TILED || k + u < ke);
                        const int at = ok[u][t] ? base[t] + k + u : 0;
                        const I cc = lc[at];
                        vv[u][t] = lv[at];
                        const I ce = ok[u][t] ? cc : I(0);
                        if (IDX32) {
                            const uint32_t off = uint32_t(ce) * uint32_t(ldb) + uint32_t(jl);
                            x[u][t] = *reinterpret_cast<const BV*>(b + off);
                        } else {
                            x[u][t] = *reinterpret_cast<const BV*>(b + int64_t(ce) * ldb + jl);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < KU; ++u) {
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        const T a = ADV ? alpha * vv[u][t] : vv[u][t];
#pragma unroll
                        for (int q = 0; q < CPL; ++q) {
                            const T nx = sum[t][q] + a * x[u][t].v[q];
                            sum[t][q] = ok[u][t] ? nx : sum[t][q];
                        }
                    }
                }
            }
        };
        if (whole) {
            if (K1 > K0) walk(std::false_type{}, 0, 0, rs);
        } else {
            for (int c0 = 0; c0 < wave_maxlen; c0 += 32) {
                wave_lds_sync();   // the previous round has been read
                for (int i = lane; i < CAP; i += 64) {
                    const int64_t r = row0 + (i >> 5);
                    if (r < last) {
                        const int64_t a = row_ptrs[r];
                        if (a + c0 + (i & 31) < int64_t(row_ptrs[r + 1])) {
                            lv[i] = vals[a + c0 + (i & 31)];
                            lc[i] = cols[a + c0 + (i & 31)];
                        }
                    }
                }
                wave_lds_sync();
                int base[TT];
#pragma unroll
                for (int t = 0; t < TT; ++t) base[t] = 32 * (rl + RPP * t) - c0;
                walk(std::true_type{}, c0, c0 + 32, base);
            }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if (row[t] < last && ncol > 0) {
                T* __restrict__ cp = c + row[t] * ldc + jc;
                if (ncol == CPL) {
                    BV q;
#pragma unroll
                    for (int i = 0; i < CPL; ++i) q.v[i] = sum[t][i];
                    *reinterpret_cast<BV*>(cp) = q;
                } else {
                    cp[0] = sum[t][0];
                }
            }
        }
    }
}

// The fragment layout as a PIPELINE (round 5).  Counters (profiles/r05_multi_rhs_pmc.txt) show the kernel
// above latency-bound: a wave lives 13.7 us for 432 entries, 84 % of it waiting, through a chain of dependent
// memory phases (row pointers -> the segment's values / indices -> 27 / KU rounds of gathers), with the
// number of resident waves capped by the LDS; neither the L1 access count nor the LDS is the wall.  Here a
// wave walks SEGS consecutive 16-row segments and keeps the next one's memory requests in flight while it
// works on the current one:
//   * the row pointers of segment s + 1 are requested when segment s begins (one load per lane, the lanes
//     hand them round by wave shuffle: no per-row pointer loads behind the staging);
//   * the values / indices of segment s + 1 travel into REGISTERS (PF entries per lane, coalesced) while
//     the gathers of segment s run, and go to the LDS when segment s is done: the matrix stream's HBM
//     latency is paid once per wave, not once per segment;
//   * two rounds of gathers are in flight: round k + 1 is requested before round k's products are added.
// Same LDS footprint per wave (one 6 KB buffer), same sums: every (row, column) sum by ONE lane in entry
// order, separate multiply and add - bit-identical to the kernel above and to the reference.  A segment
// with more than CAP entries takes the rounds of the kernel above (not pipelined).
//
// NB (round 6): a value of b that the row BELOW has just gathered is taken from that row's lanes instead of
// the memory system.  In a banded / stencil matrix entry k of row r and entry k - 1 of row r + 1 are the same
// column (r + 1 + dx - 1 = r + dx): the 16 rows of a wave ask for the same 15 rows of b again with every step
// through a run of neighbouring columns, and the in-order vector L1 stalls on every one of those hits while
// the line is still on its way (TCP_PENDING_STALL_CYCLES 44 %, 0.59 line accesses per cycle and CU where the
// single-column kernel sustains 0.99: profiles/r05_multi_rhs_pmc.txt).  The test is on the column INDICES
// (both in LDS), so it holds for any matrix: entry k of my row is taken from the lanes of the next row iff
// that row's entry k - 1 has the same column; everybody else gathers.  A lane that takes the neighbour's value
// still issues its load - branch-free, the compiler keeps all loads of a round in flight - but from row 0 of
// b, one line for all of them: a gather instruction touches 2 lines instead of 8.  The values are the same
// numbers whichever way they come: the sums keep the reference's bits.
template <typename T, typename I, bool ADV, int NR, int CPL, int KU, bool IDX32, int ST = 1, bool NB = false>
__global__ __launch_bounds__(64) void csr_spmv_frag_pipe_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ b, int64_t ldb, T* __restrict__ c,
    int64_t ldc, int nrhs, const T* __restrict__ alpha_p, const T* __restrict__ beta_p,
    int segs_per_wave, int64_t xcd_chunk)
{
    static_assert(NR == 4 || NR == 8, "chunks of 4 or 8 columns");
    static_assert(CPL == 1 || CPL == 2, "one or two columns per lane");
    constexpr int LPR = NR / CPL;     // lanes per row
    constexpr int ROWS = 64 / LPR;    // rows per segment (16)
    constexpr int CAP = ROWS * 32;    // staged entries (12 B each)
    constexpr int PF = CAP / 64;      // entries per lane that travel ahead in registers
    using BV = vecT<T, CPL>;
    __shared__ __attribute__((aligned(16))) T lv[CAP];
    __shared__ __attribute__((aligned(16))) I lc[CAP];
    const int lane = threadIdx.x;
    const int sub = lane % LPR, rl = lane / LPR;
    const int64_t n_seg = (n_rows + ROWS - 1) / ROWS;
    const int64_t w = xcd_chunked_block(blockIdx.x, gridDim.x, xcd_chunk);
    int64_t seg = w * segs_per_wave;
    const int64_t seg_end = seg + segs_per_wave < n_seg ? seg + segs_per_wave : n_seg;
    if (seg >= seg_end) return;
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    // row pointer `lane` of a segment (lanes 0 .. ROWS), clamped to the matrix
    auto load_rp = [&](int64_t sg) -> int64_t {
        const int64_t r = sg * ROWS + (lane <= ROWS ? lane : ROWS);
        return int64_t(row_ptrs[r < n_rows ? r : n_rows]);
    };
    T pv[PF];
    I pc[PF];
    auto prefetch = [&](int64_t k0, int count) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int i = lane + 64 * j;
            if (i < count) {
                pv[j] = vals[k0 + i];
                pc[j] = cols[k0 + i];
            }
        }
    };
    int64_t rp = load_rp(seg);
    int64_t K0 = __shfl(rp, 0, 64), K1 = __shfl(rp, ROWS, 64);
    bool staged_ahead = K1 - K0 <= CAP;
    if (staged_ahead) prefetch(K0, int(K1 - K0));
    int64_t nrp = seg + 1 < seg_end ? load_rp(seg + 1) : 0;

    while (seg < seg_end) {
        const int64_t row0 = seg * ROWS;
        const int64_t last = row0 + ROWS < n_rows ? row0 + ROWS : n_rows;
        const bool whole = staged_ahead;
        const int count = int(K1 - K0);
        if (whole) {
            wave_lds_sync();          // the previous segment has been read
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int i = lane + 64 * j;
                if (i < count) {
                    lv[i] = pv[j];
                    lc[i] = pc[j];
                }
            }
            wave_lds_sync();
        }
        // this segment's rows as this lane sees them
        const int64_t row = row0 + rl;
        const int64_t a = __shfl(rp, rl, 64), e = __shfl(rp, rl + 1, 64);
        const int rs = int(a - K0);
        const int len = row < last ? int(e - a) : 0;
        int maxlen = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(maxlen, off, 64);
            maxlen = o > maxlen ? o : maxlen;
        }
        // NB: the lanes of the row below (same columns), where that row starts in the LDS and how long it is
        const int nb_lane = lane + LPR < 64 ? lane + LPR : lane;
        const int nb_rs = __shfl(rs, nb_lane, 64);
        const int nb_len = (lane + LPR < 64) ? __shfl(len, nb_lane, 64) : 0;
        // the next segment: its pointers are here (requested one segment ago), its entries start to travel
        // now; the pointers of the one after are requested
        int64_t nK0 = 0, nK1 = 0;
        bool next_ahead = false;
        if (seg + 1 < seg_end) {
            nK0 = __shfl(nrp, 0, 64);
            nK1 = __shfl(nrp, ROWS, 64);
            next_ahead = nK1 - nK0 <= CAP;
            if (next_ahead) prefetch(nK0, int(nK1 - nK0));
        }
        const int64_t nnrp = seg + 2 < seg_end ? load_rp(seg + 2) : 0;

        for (int j0 = 0; j0 < nrhs; j0 += NR) {
            const int jc = j0 + CPL * sub;
            const int ncol = nrhs - jc >= CPL ? CPL : nrhs - jc;
            const int jl = ncol >= 1 ? jc : 0;
            T sum[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                sum[q] = T(0);
                if (ADV && beta != T(0) && row < last && q < ncol) sum[q] = c[row * ldc + jc + q] * beta;
            }
            // entries [kb, kend) of the lane's row, entry k at LDS slot base + k; two rounds of KU gathers in flight
            auto walk = [&](int kb, int kend, int klimit, int base, bool nb_on) {
                BV x0[KU], x1[KU];
                T v0[KU], v1[KU];
                bool ok0[KU], ok1[KU];
                bool nb0[KU], nb1[KU];       // NB: entry comes from the lanes of the row below
                BV xlast;                    // NB: this lane's value of the entry in front of the round being added
#pragma unroll
                for (int q = 0; q < CPL; ++q) xlast.v[q] = T(0);
                auto request = [&](int k, BV(&x)[KU], T(&vv)[KU], bool(&ok)[KU], bool(&nb)[KU]) {
#pragma unroll
                    for (int u = 0; u < KU; ++u) {
                        ok[u] = k + u < len && k + u < klimit;
                        const int at = ok[u] ? base + k + u : 0;
                        const I cc = lc[at];
                        vv[u] = lv[at];
                        const I ce = ok[u] ? cc : I(0);
                        nb[u] = false;
                        if constexpr (NB) {
                            // the row below, one entry back: the same column?  (k + u - 1 >= kb: that entry
                            // belongs to this walk, so its value is or will be in that row's registers)
                            const bool can = nb_on && ok[u] && k + u - 1 >= kb && k + u - 1 < nb_len;
                            const I nc = lc[can ? nb_rs + k + u - 1 : 0];
                            nb[u] = can && nc == cc;
                        }
                        if (IDX32) {
                            const uint32_t off = nb[u] ? uint32_t(jl) : uint32_t(ce) * uint32_t(ldb) + uint32_t(jl);
                            x[u] = *reinterpret_cast<const BV*>(b + off);
                        } else {
                            x[u] = *reinterpret_cast<const BV*>(b + (nb[u] ? int64_t(jl) : int64_t(ce) * ldb + jl));
                        }
                    }
                };
                auto add = [&](BV(&x)[KU], const T(&vv)[KU], const bool(&ok)[KU], const bool(&nb)[KU]) {
#pragma unroll
                    for (int u = 0; u < KU; ++u) {
                        if constexpr (NB) {
                            // in entry order: the row below has settled its entry k + u - 1 in the step before
                            // (u - 1 of this round, or the last one of the round before)
#pragma unroll
                            for (int q = 0; q < CPL; ++q) {
                                const T mine = u == 0 ? xlast.v[q] : x[u - 1].v[q];
                                const T theirs = __shfl(mine, nb_lane, 64);
                                x[u].v[q] = nb[u] ? theirs : x[u].v[q];
                            }
                        }
                        const T av = ADV ? alpha * vv[u] : vv[u];
#pragma unroll
                        for (int q = 0; q < CPL; ++q) {
                            const T nx = sum[q] + av * x[u].v[q];
                            sum[q] = ok[u] ? nx : sum[q];
                        }
                    }
                    if constexpr (NB) xlast = x[KU - 1];
                };
                if (kb >= kend) return;
                if constexpr (ST > 1) {
                    // STRIDED rounds.  Neighbouring entries of a row of a banded / stencil matrix are
                    // neighbouring rows of b, and the 16 rows of the wave make the gathers of entries k and
                    // k + 1 overlap in 15 of their 16 rows: the second gather hits lines that are still on
                    // their way, and the vector L1 - in order - stalls on such a hit until the data is back
                    // (TCP_PENDING_STALL_CYCLES = 44 % of the cycles, profiles/r05_multi_rhs_pmc.txt).  So
                    // a round asks for entries k, k + ST, k + 2 ST, ... (different lines), waits for them,
                    // and only then for k + 1, k + 1 + ST, ... - which now HIT.  The products are added in
                    // entry order afterwards: the same sums.
                    for (int k = kb; k < kend; k += KU * ST) {
                        BV x[ST][KU];
                        T vv[ST][KU];
                        bool ok[ST][KU];
#pragma unroll
                        for (int st = 0; st < ST; ++st) {
#pragma unroll
                            for (int u = 0; u < KU; ++u) {
                                const int kk = k + st + ST * u;
                                ok[st][u] = kk < len && kk < klimit;
                                const int at = ok[st][u] ? base + kk : 0;
                                const I cc = lc[at];
                                vv[st][u] = lv[at];
                                const I ce = ok[st][u] ? cc : I(0);
                                if (IDX32) {
                                    const uint32_t off = uint32_t(ce) * uint32_t(ldb) + uint32_t(jl);
                                    x[st][u] = *reinterpret_cast<const BV*>(b + off);
                                } else {
                                    x[st][u] = *reinterpret_cast<const BV*>(b + int64_t(ce) * ldb + jl);
                                }
                            }
                            if (st + 1 < ST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
#pragma unroll
                        for (int u = 0; u < KU; ++u) {
#pragma unroll
                            for (int st = 0; st < ST; ++st) {
                                const T av = ADV ? alpha * vv[st][u] : vv[st][u];
#pragma unroll
                                for (int q = 0; q < CPL; ++q) {
                                    const T nx = sum[q] + av * x[st][u].v[q];
                                    sum[q] = ok[st][u] ? nx : sum[q];
                                }
                            }
                        }
                    }
                    return;
                }
                request(kb, x0, v0, ok0, nb0);
                int k = kb;
                while (true) {
                    if (k + KU < kend) request(k + KU, x1, v1, ok1, nb1);
                    add(x0, v0, ok0, nb0);
                    k += KU;
                    if (k >= kend) break;
                    if (k + KU < kend) request(k + KU, x0, v0, ok0, nb0);
                    add(x1, v1, ok1, nb1);
                    k += KU;
                    if (k >= kend) break;
                }
            };
            if (whole) {
                if (count > 0) walk(0, maxlen, maxlen, rs, true);
            } else {
                for (int c0 = 0; c0 < maxlen; c0 += 32) {
                    wave_lds_sync();
                    for (int i = lane; i < CAP; i += 64) {
                        const int64_t r = row0 + (i >> 5);
                        if (r < last) {
                            const int64_t ra = row_ptrs[r];
                            if (ra + c0 + (i & 31) < int64_t(row_ptrs[r + 1])) {
                                lv[i] = vals[ra + c0 + (i & 31)];
                                lc[i] = cols[ra + c0 + (i & 31)];
                            }
                        }
                    }
                    wave_lds_sync();
                    const int ke = c0 + 32 < maxlen ? c0 + 32 : maxlen;
                    walk(c0, ke, c0 + 32, 32 * rl - c0, false);
                }
            }
            if (row < last && ncol > 0) {
                T* __restrict__ cp = c + row * ldc + jc;
                if (ncol == CPL) {
                    BV q;
#pragma unroll
                    for (int i = 0; i < CPL; ++i) q.v[i] = sum[i];
                    *reinterpret_cast<BV*>(cp) = q;
                } else {
                    cp[0] = sum[0];
                }
            }
        }
        ++seg;
        rp = nrp;
        nrp = nnrp;
        K0 = nK0;
        K1 = nK1;
        staged_ahead = next_ahead;
    }
}

template <typename T, typename I, bool ADV, int E, int U, int RING, int NR>
__global__ __launch_bounds__(64) void csr_spmv_multi_kernel(
    int64_t n_rows, int64_t n_segments, int64_t segs_per_wave,
    const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ b, int64_t ldb,
    T* __restrict__ c, int64_t ldc, int nrhs, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p, int b_vec_ok, int64_t xcd_chunk = 0)
{
    static_assert((RING & (RING - 1)) == 0, "RING must be a power of two");
    static_assert(NR == 2 || NR == 4, "chunks of 2 or 4 columns");
    constexpr int ROWS = 64;
    constexpr int G = 64 * E * U;
    static_assert(RING >= 2 * G, "ring too small for the group size");
    constexpr int MASK = RING - 1;
    // pairs of columns as one 2*sizeof(T) load
    using BV = vecT<T, 2>;
    __shared__ __attribute__((aligned(16))) T ring[RING * NR];

    const int lane = threadIdx.x;
    const int64_t sb = xcd_chunked_block(blockIdx.x, gridDim.x, xcd_chunk) * segs_per_wave;
    const int64_t se = sb + segs_per_wave < n_segments ? sb + segs_per_wave : n_segments;
    if (sb >= se) return;
    const int64_t row_e = se * ROWS < n_rows ? se * ROWS : n_rows;
    const int64_t K0 = row_ptrs[sb * ROWS];
    const int64_t K1 = row_ptrs[row_e];
    const int64_t NNZ = row_ptrs[n_rows];
    const int64_t K0a = K0 & ~int64_t(E - 1);
    const int k1o = int(K1 - K0a);
    const int nnzo = (NNZ - K0a) > int64_t(0x7fffff00) ? 0x7fffff00 : int(NNZ - K0a);
    const T* __restrict__ vals0 = vals + K0a;
    const I* __restrict__ cols0 = cols + K0a;

    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }

    using VT = vecT<T, E>;
    using VI = vecT<I, E>;

    auto load_group = [&](VT(&v)[U], VI(&ci)[U], int p) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = p + (u * 64 + lane) * E;
            if (k >= k1o) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    v[u].v[e] = T(0);
                    ci[u].v[e] = I(0);
                }
            } else if (k + E <= nnzo) {
                v[u] = *reinterpret_cast<const VT*>(vals0 + k);
                ci[u] = *reinterpret_cast<const VI*>(cols0 + k);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool in = k + e < nnzo;
                    v[u].v[e] = in ? vals0[k + e] : T(0);
                    ci[u].v[e] = in ? cols0[k + e] : I(0);
                }
            }
        }
    };

    for (int j0 = 0; j0 < nrhs; j0 += NR) {
        // column of b / c behind slot jj; slots past nrhs read the last column
        int jcol[NR];
        bool jok[NR];
#pragma unroll
        for (int jj = 0; jj < NR; ++jj) {
            jok[jj] = j0 + jj < nrhs;
            jcol[jj] = jok[jj] ? j0 + jj : nrhs - 1;
        }
        const bool full_vec = b_vec_ok && j0 + NR <= nrhs;

        auto gather = [&](I col, T(&x)[NR]) {
            const T* __restrict__ brow = b + int64_t(col) * ldb;
            if (full_vec) {
#pragma unroll
                for (int jj = 0; jj < NR; jj += 2) {
                    const BV t = *reinterpret_cast<const BV*>(brow + j0 + jj);
                    x[jj] = t.v[0];
                    x[jj + 1] = t.v[1];
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < NR; ++jj) x[jj] = brow[jcol[jj]];
            }
        };

        auto produce = [&](VT(&v)[U], VI(&ci)[U], int p) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                T xv[E][NR];
#pragma unroll
                for (int e = 0; e < E; ++e) gather(ci[u].v[e], xv[e]);
                const int k = p + (u * 64 + lane) * E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    // E divides RING and k is a multiple of E: no wrap inside a lane's run
                    T* dst = &ring[((k + e) & MASK) * NR];
#pragma unroll
                    for (int jj = 0; jj < NR; jj += 2) {
                        BV pr;
                        pr.v[0] = ADV ? (alpha * v[u].v[e]) * xv[e][jj] : v[u].v[e] * xv[e][jj];
                        pr.v[1] = ADV ? (alpha * v[u].v[e]) * xv[e][jj + 1]
                                      : v[u].v[e] * xv[e][jj + 1];
                        *reinterpret_cast<BV*>(dst + jj) = pr;
                    }
                }
            }
        };

        VT vA[U], vB[U];
        VI cA[U], cB[U];
        int p_load = 0;
        load_group(vA, cA, p_load);
        p_load += G;
        load_group(vB, cB, p_load);
        p_load += G;
        int produced = 0;          // offsets relative to K0a
        int cons = int(K0 - K0a);
        bool use_a = true;

        int64_t seg = sb;
        auto seg_rows = [&](int64_t s, int& rs, int& re, int& s_end) {
            const int64_t row = s * ROWS + lane;
            const int64_t last = (s + 1) * ROWS < n_rows ? (s + 1) * ROWS : n_rows;
            const bool valid = row < n_rows;
            rs = int(int64_t(row_ptrs[valid ? row : last]) - K0a);
            re = int(int64_t(row_ptrs[valid ? row + 1 : last]) - K0a);
            s_end = int(int64_t(row_ptrs[last]) - K0a);
        };
        auto init_sums = [&](int64_t s, T(&sum)[NR]) {
            const int64_t row = s * ROWS + lane;
#pragma unroll
            for (int jj = 0; jj < NR; ++jj) {
                sum[jj] = T(0);
                if (ADV && beta != T(0) && s < se && row < n_rows && jok[jj]) {
                    sum[jj] = c[row * ldc + jcol[jj]] * beta;
                }
            }
        };
        int rs, re, seg_end, nrs = 0, nre = 0, nseg_end = 0;
        seg_rows(seg, rs, re, seg_end);
        if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
        T sum[NR];
        init_sums(seg, sum);

        while (seg < se) {
            if (produced >= seg_end || produced + G - cons > RING) {
                const int upto = produced < seg_end ? produced : seg_end;
                wave_lds_sync();
                const bool is_long = (re - rs) > GKOC_CSR_LONG_ROW;
                if (!is_long) {
                    int k = rs > cons ? rs : cons;
                    const int e_ = re < upto ? re : upto;
                    for (; k + 2 <= e_; k += 2) {
                        T t0[NR], t1[NR];
#pragma unroll
                        for (int jj = 0; jj < NR; jj += 2) {
                            const BV a0 = *reinterpret_cast<const BV*>(&ring[(k & MASK) * NR + jj]);
                            const BV a1 =
                                *reinterpret_cast<const BV*>(&ring[((k + 1) & MASK) * NR + jj]);
                            t0[jj] = a0.v[0];
                            t0[jj + 1] = a0.v[1];
                            t1[jj] = a1.v[0];
                            t1[jj + 1] = a1.v[1];
                        }
#pragma unroll
                        for (int jj = 0; jj < NR; ++jj) {
                            sum[jj] += t0[jj];
                            sum[jj] += t1[jj];
                        }
                    }
                    for (; k < e_; ++k) {
#pragma unroll
                        for (int jj = 0; jj < NR; ++jj) sum[jj] += ring[(k & MASK) * NR + jj];
                    }
                }
                wave_lds_sync();
                cons = upto;
                if (cons >= seg_end) {
                    unsigned long long m = __ballot(is_long);
                    while (m) {
                        const int src = __builtin_ctzll(m);
                        m &= m - 1;
                        const int lrs = __shfl(rs, src, 64);
                        const int lre = __shfl(re, src, 64);
                        T part[NR];
#pragma unroll
                        for (int jj = 0; jj < NR; ++jj) part[jj] = T(0);
                        for (int k = lrs + lane; k < lre; k += 64) {
                            const T* __restrict__ brow = b + int64_t(cols0[k]) * ldb;
                            const T av = ADV ? alpha * vals0[k] : vals0[k];
#pragma unroll
                            for (int jj = 0; jj < NR; ++jj) part[jj] += av * brow[jcol[jj]];
                        }
#pragma unroll
                        for (int jj = 0; jj < NR; ++jj) {
                            const T tot = wave_sum(part[jj]);
                            if (lane == src) sum[jj] += tot;
                        }
                    }
                    const int64_t row = seg * ROWS + lane;
                    if (row < n_rows) {
#pragma unroll
                        for (int jj = 0; jj < NR; ++jj) {
                            if (jok[jj]) c[row * ldc + jcol[jj]] = sum[jj];
                        }
                    }
                    ++seg;
                    rs = nrs;
                    re = nre;
                    seg_end = nseg_end;
                    if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
                    init_sums(seg, sum);
                }
                continue;
            }
            if (use_a) {
                produce(vA, cA, produced);
                load_group(vA, cA, p_load);
            } else {
                produce(vB, cB, produced);
                load_group(vB, cB, p_load);
            }
            p_load += G;
            produced += G;
            use_a = !use_a;
        }
    }
}

#endif  // __HIPCC__

}  // namespace gkoc
