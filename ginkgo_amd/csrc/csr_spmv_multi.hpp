// CSR SpMV for SEVERAL right-hand sides in one pass over the matrix (gfx950).
//
// The single-column kernel (csr_spmv_pipe.hpp) run once per column re-reads the
// whole matrix for every column and gathers b with a stride of ldb values
// (L256: 1.17 / 2.18 / 3.67 / 5.45 / 16.8 ms for 1 / 2 / 3 / 4 / 8 columns).
// Here a wave streams its val / col range ONCE per chunk of NR columns:
//   * same row-segment-per-wavefront walk, two register sets, 16/32 B vector
//     loads of the matrix stream as in csr_spmv_pipe3_kernel;
//   * lane = E consecutive nonzeros: for every nonzero the NR values
//     b[col, j0 .. j0+NR) are one contiguous run of the row-major b (one or two
//     16 B loads when b allows it), so a chunk costs the b traffic of ONE gather
//     with NR times the payload per line;
//   * the NR products of a nonzero sit next to each other in the LDS ring
//     (ring[(k mod RING) * NR + jj]); lane = row then adds them in k order per
//     column - the reference's summation order, separate multiply and add =>
//     bit-identical per column to the sequential reference, and to the
//     single-column kernel;
//   * the results of a segment leave as 64 runs of NR contiguous values
//     (one contiguous block when ldc == NR).
// That kernel (csr_spmv_multi_kernel) serves 2-4 columns; for 5 and more,
// csr_spmv_rowmulti_kernel below stages the segment in LDS and gathers in ROW order,
// 8 columns per pass (the launcher in csr_spmv.hip holds the measured crossover).
// Columns beyond nrhs in the last chunk are computed on a clamped column index
// and not stored.  Rows longer than GKOC_CSR_LONG_ROW take the cooperative wave
// path of the single-column kernel (tolerance-compared, include/gko_cdna4.h).
#pragma once
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

// CSR, several right-hand sides, ROW-ORDERED gather: the wave's 64 rows own one
// contiguous val / col range, which is staged in LDS with coalesced loads; then
// lane = row walks its entries in k order (fmt_row_sum_multi on the staged copy) with
// NR sums in registers.  The 64 lanes of a gather instruction then read the b rows
// of entry i of 64 consecutive matrix rows - for banded / stencil matrices one
// contiguous run of b - instead of the scattered neighbours of a few rows.
// Segments above multi_stage_cap entries walk global memory directly (still exact).
constexpr int multi_stage_cap = 2048;

template <typename T, typename I, bool ADV, int NR>
__global__ __launch_bounds__(64) void csr_spmv_rowmulti_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ b, int64_t ldb, T* __restrict__ c,
    int64_t ldc, int nrhs, const T* __restrict__ alpha_p, const T* __restrict__ beta_p)
{
    __shared__ __attribute__((aligned(16))) T lv[multi_stage_cap];
    __shared__ __attribute__((aligned(16))) I lc[multi_stage_cap];
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x;
    const int64_t row = g * 64 + lane;
    const bool valid = row < n_rows;
    const int64_t last = (g + 1) * 64 < n_rows ? (g + 1) * 64 : n_rows;
    const int64_t rs = row_ptrs[valid ? row : last];
    const int64_t re = row_ptrs[valid ? row + 1 : last];
    const int64_t K0 = row_ptrs[g * 64];
    const int64_t K1 = row_ptrs[last];
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    const bool staged = K1 - K0 <= multi_stage_cap;
    if (staged) {
        const int seg = int(K1 - K0);
        for (int i = lane; i < seg; i += 64) {
            lv[i] = vals[K0 + i];
            lc[i] = cols[K0 + i];
        }
        wave_lds_sync();
    }
    const T* pv = staged ? lv : vals + K0;
    const I* pc = staged ? lc : cols + K0;
    for (int j0 = 0; j0 < nrhs; j0 += NR) {
        int jcol[NR];
        T sum[NR];
#pragma unroll
        for (int jj = 0; jj < NR; ++jj) {
            jcol[jj] = j0 + jj < nrhs ? j0 + jj : nrhs - 1;
            sum[jj] = T(0);
            if (ADV && beta != T(0) && valid) sum[jj] = c[row * ldc + jcol[jj]] * beta;
        }
        if (valid) {
            fmt_row_sum_multi<T, I, ADV, NR>(sum, re - rs, rs - K0, 1, pc, pv, b, ldb, jcol, alpha);
#pragma unroll
            for (int jj = 0; jj < NR; ++jj) {
                if (j0 + jj < nrhs) c[row * ldc + j0 + jj] = sum[jj];
            }
        }
    }
}

template <typename T, typename I, bool ADV, int E, int U, int RING, int NR>
__global__ __launch_bounds__(64) void csr_spmv_multi_kernel(
    int64_t n_rows, int64_t n_segments, int64_t segs_per_wave,
    const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ b, int64_t ldb,
    T* __restrict__ c, int64_t ldc, int nrhs, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p, int b_vec_ok)
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
    const int64_t sb = int64_t(blockIdx.x) * segs_per_wave;
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
