// Earlier CSR SpMV kernel generations, kept only for A/B measurements in
// tools/spmv_lab.hip (not part of the library).
#pragma once
#include "common.hpp"
#include "csr_spmv_pipe.hpp"

namespace gkoc {
namespace {

template <typename T>
struct tile_cap {
    // products per wave: 14 KB of LDS => 11 single-wave workgroups per CU
    static constexpr int value = 14336 / sizeof(T);
};

// bijective XCD-aware remap of the workgroup id: dispatcher places block b on
// XCD b % 8 (MI355X_MICROARCH.md); give every XCD a contiguous band of row
// segments so that the b-vector lines shared by neighbouring segments stay in
// one XCD's L2.
__device__ __forceinline__ int64_t xcd_band_remap(int64_t bid, int64_t n)
{
    constexpr int64_t nx = 8;
    const int64_t q = n / nx, r = n % nx;
    const int64_t xcd = bid % nx, idx = bid / nx;
    const int64_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <typename T, typename I, bool ADV, bool REMAP, int UNROLL>
__global__ __launch_bounds__(64) void csr_spmv_wave_kernel(
    int64_t n_rows, int64_t n_segments, const I* __restrict__ row_ptrs,
    const I* __restrict__ cols, const T* __restrict__ vals,
    const T* __restrict__ b, int64_t ldb, T* __restrict__ c, int64_t ldc,
    int nrhs, const T* __restrict__ alpha_p, const T* __restrict__ beta_p)
{
    constexpr int CAP = tile_cap<T>::value;
    __shared__ T prod[CAP];
    const int lane = threadIdx.x;
    const int64_t seg =
        REMAP ? xcd_band_remap(blockIdx.x, n_segments) : int64_t(blockIdx.x);
    const int64_t r0 = seg * 64;
    const int64_t row = r0 + lane;
    const bool valid = row < n_rows;
    const int64_t r_last = (r0 + 64 < n_rows) ? r0 + 64 : n_rows;
    const int64_t rs = row_ptrs[valid ? row : r_last];
    const int64_t re = row_ptrs[valid ? row + 1 : r_last];
    const int64_t k0 = __shfl(rs, 0, 64);
    const int64_t k1 = __shfl(re, int(r_last - r0 - 1), 64);
    const bool is_long = (re - rs) > GKOC_CSR_LONG_ROW;
    const unsigned long long long_mask = __ballot(is_long);

    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }

    for (int j = 0; j < nrhs; ++j) {
        T sum = T(0);
        if (ADV && valid && beta != T(0)) {
            sum = c[row * ldc + j] * beta;
        }
        for (int64_t t0 = k0; t0 < k1; t0 += CAP) {
            const int64_t t1 = (t0 + CAP < k1) ? t0 + CAP : k1;
            // phase 1: lane = nnz; coalesced stream of val/col, gather b
            for (int64_t base = t0; base < t1; base += 64 * UNROLL) {
                T v[UNROLL];
                I cc[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    int64_t k = base + u * 64 + lane;
                    k = k < t1 ? k : t1 - 1;
                    v[u] = vals[k];
                    cc[u] = cols[k];
                }
                T xv[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    xv[u] = b[int64_t(cc[u]) * ldb + j];
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int64_t k = base + u * 64 + lane;
                    if (k < t1) {
                        prod[k - t0] = ADV ? (alpha * v[u]) * xv[u]
                                           : v[u] * xv[u];
                    }
                }
            }
            wave_lds_sync();
            // phase 2: lane = row; sequential (reference-order) row sums
            if (!is_long) {
                const int64_t a = rs > t0 ? rs : t0;
                const int64_t e = re < t1 ? re : t1;
                for (int64_t k = a; k < e; ++k) {
                    sum += prod[k - t0];
                }
            }
            wave_lds_sync();
        }
        // long rows: whole-wave cooperative dot straight from global memory
        unsigned long long m = long_mask;
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const int64_t lrs = __shfl(rs, src, 64);
            const int64_t lre = __shfl(re, src, 64);
            T part = T(0);
            for (int64_t k = lrs + lane; k < lre; k += 64) {
                const T p = ADV ? (alpha * vals[k]) * b[int64_t(cols[k]) * ldb + j]
                                : vals[k] * b[int64_t(cols[k]) * ldb + j];
                part += p;
            }
            part = wave_sum(part);
            if (lane == src) sum += part;
        }
        if (valid) {
            c[row * ldc + j] = sum;
        }
    }
}


}  // namespace

// ABL: measurement-only switches: 1 = no b gather, 2 = no LDS row sums,
// 4 = non-temporal val/col stream loads, 8 = XCD-chunked wave order
template <typename T, typename I, bool ADV, int ROWS, int E, int U, int RING,
          int ABL = 0>
__global__ __launch_bounds__(64) void csr_spmv_pipe_kernel(
    int64_t n_rows, int64_t n_segments, int64_t segs_per_wave,
    const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ b, int64_t ldb,
    T* __restrict__ c, int64_t ldc, int nrhs, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p)
{
    static_assert((RING & (RING - 1)) == 0, "RING must be a power of two");
    constexpr int G = 64 * E * U;
    static_assert(RING >= 2 * G, "ring too small for the group size");
    static_assert(ROWS == 32 || ROWS == 64, "ROWS must be 32 or 64");
    constexpr int64_t MASK = RING - 1;
    __shared__ __attribute__((aligned(16))) T ring[RING];

    const int lane = threadIdx.x;
    int64_t wave_id = blockIdx.x;
    if (ABL & 8) {
        // hardware places block b on XCD b % 8: give each XCD chunks of 32
        // consecutive waves so that neighbouring row ranges share one L2
        constexpr int64_t C = 32;
        const int64_t nfull = (int64_t(gridDim.x) / (8 * C)) * (8 * C);
        if (wave_id < nfull) {
            const int64_t xcd = wave_id % 8, slot = wave_id / 8;
            wave_id = ((slot / C) * 8 + xcd) * C + (slot % C);
        }
    }
    const int64_t sb = wave_id * segs_per_wave;
    const int64_t se = sb + segs_per_wave < n_segments ? sb + segs_per_wave : n_segments;
    if (sb >= se) return;
    const int64_t row_e = se * ROWS < n_rows ? se * ROWS : n_rows;
    const int64_t K0 = row_ptrs[sb * ROWS];
    const int64_t K1 = row_ptrs[row_e];
    const int64_t NNZ = row_ptrs[n_rows];
    const int64_t K0a = K0 & ~int64_t(E - 1);

    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }

    using VT = vecT<T, E>;
    using VI = vecT<I, E>;

    auto load_group = [&](VT(&v)[U], VI(&ci)[U], int64_t p) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = p + (int64_t(u) * 64 + lane) * E;
            if (k >= K1) {
                // past this wave's range: nothing to fetch (values unused)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    v[u].v[e] = T(0);
                    ci[u].v[e] = I(0);
                }
            } else if (k + E <= NNZ) {
                if (ABL & 4) {
                    typedef T tvec __attribute__((ext_vector_type(E)));
                    typedef I ivec __attribute__((ext_vector_type(E)));
                    const tvec tv = __builtin_nontemporal_load(
                        reinterpret_cast<const tvec*>(vals + k));
                    const ivec iv = __builtin_nontemporal_load(
                        reinterpret_cast<const ivec*>(cols + k));
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        v[u].v[e] = tv[e];
                        ci[u].v[e] = iv[e];
                    }
                } else {
                    v[u] = *reinterpret_cast<const VT*>(vals + k);
                    ci[u] = *reinterpret_cast<const VI*>(cols + k);
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool in = k + e < NNZ;
                    v[u].v[e] = in ? vals[k + e] : T(0);
                    ci[u].v[e] = in ? cols[k + e] : I(0);
                }
            }
        }
    };

    for (int j = 0; j < nrhs; ++j) {
        auto produce = [&](VT(&v)[U], VI(&ci)[U], int64_t p) {
            VT xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (ABL & 16) {
                        // measurement only: perfectly coalesced "gather"
                        const int64_t kk = p + (int64_t(u) * 64 + lane) * E + e;
                        xv[u].v[e] = b[(kk % n_rows) * ldb + j + (ci[u].v[e] & 0)];
                    } else {
                        xv[u].v[e] = (ABL & 1) ? T(ci[u].v[e])
                                               : b[int64_t(ci[u].v[e]) * ldb + j];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                VT pr;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    pr.v[e] = ADV ? (alpha * v[u].v[e]) * xv[u].v[e]
                                  : v[u].v[e] * xv[u].v[e];
                }
                const int64_t k = p + (int64_t(u) * 64 + lane) * E;
                *reinterpret_cast<VT*>(&ring[k & MASK]) = pr;
            }
        };

        VT vA[U], vB[U];
        VI cA[U], cB[U];
        int64_t p_load = K0a;
        load_group(vA, cA, p_load);
        p_load += G;
        load_group(vB, cB, p_load);
        p_load += G;
        int64_t produced = K0a;  // products exist for stream indices < produced
        int64_t cons = K0;       // products below cons are consumed
        bool use_a = true;

        // segment state
        int64_t seg = sb;
        auto seg_rows = [&](int64_t s, int64_t& rs, int64_t& re, int64_t& s_end) {
            const int64_t row = s * ROWS + lane;
            const int64_t last = (s + 1) * ROWS < n_rows ? (s + 1) * ROWS : n_rows;
            const bool valid = lane < ROWS && row < n_rows;
            rs = row_ptrs[valid ? row : last];
            re = row_ptrs[valid ? row + 1 : last];
            s_end = row_ptrs[last];
        };
        int64_t rs, re, seg_end, nrs = 0, nre = 0, nseg_end = 0;
        seg_rows(seg, rs, re, seg_end);
        if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
        T sum = T(0);
        {
            const int64_t row = seg * ROWS + lane;
            if (ADV && beta != T(0) && lane < ROWS && row < n_rows) {
                sum = c[row * ldc + j] * beta;
            }
        }

        while (seg < se) {
            if (produced >= seg_end || produced + G - cons > RING) {
                // ---- consume: lane = row, reference-order partial sums
                const int64_t upto = produced < seg_end ? produced : seg_end;
                wave_lds_sync();
                const bool is_long = (re - rs) > GKOC_CSR_LONG_ROW;
                if (!is_long && !(ABL & 2)) {
                    int64_t k = rs > cons ? rs : cons;
                    const int64_t e_ = re < upto ? re : upto;
                    for (; k + 4 <= e_; k += 4) {
                        const T t0 = ring[k & MASK];
                        const T t1 = ring[(k + 1) & MASK];
                        const T t2 = ring[(k + 2) & MASK];
                        const T t3 = ring[(k + 3) & MASK];
                        sum += t0;
                        sum += t1;
                        sum += t2;
                        sum += t3;
                    }
                    for (; k < e_; ++k) sum += ring[k & MASK];
                }
                wave_lds_sync();
                cons = upto;
                if (cons >= seg_end) {
                    // rows longer than GKOC_CSR_LONG_ROW: cooperative wave dot
                    unsigned long long m = __ballot(is_long);
                    while (m) {
                        const int src = __builtin_ctzll(m);
                        m &= m - 1;
                        const int64_t lrs = __shfl(rs, src, 64);
                        const int64_t lre = __shfl(re, src, 64);
                        T part = T(0);
                        for (int64_t k = lrs + lane; k < lre; k += 64) {
                            const T xb = b[int64_t(cols[k]) * ldb + j];
                            part += ADV ? (alpha * vals[k]) * xb : vals[k] * xb;
                        }
                        part = wave_sum(part);
                        if (lane == src) sum += part;
                    }
                    const int64_t row = seg * ROWS + lane;
                    if (lane < ROWS && row < n_rows) c[row * ldc + j] = sum;
                    ++seg;
                    rs = nrs;
                    re = nre;
                    seg_end = nseg_end;
                    if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
                    sum = T(0);
                    const int64_t nrow = seg * ROWS + lane;
                    if (ADV && beta != T(0) && seg < se && lane < ROWS && nrow < n_rows) {
                        sum = c[nrow * ldc + j] * beta;
                    }
                }
                continue;
            }
            // ---- produce one group, then refill its register set
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


// ---------------------------------------------------------------------------
// Variant 2: the LDS ring holds the MATRIX stream (val + col, 12 B / nonzero)
// instead of the products, and the b-vector gather moves into the row phase:
// lane = row reads its (col, val) pairs from LDS in k order and gathers
// b[col].  For stencil-like matrices consecutive lanes then address
// consecutive b entries, so one 64-lane gather instruction touches 4-5 cache
// lines instead of up to 64 (the vector L1 processes one line per cycle; the
// nnz-ordered gather of variant 1 costs one L1 access per nonzero, measured
// with TCP_TOTAL_CACHE_ACCESSES).  The stream loads never wait for a gather:
// a produce step only waits for its own (old) loads, writes them to LDS and
// re-issues; the gather batches of a completed segment queue behind the
// in-flight stream loads, whose round trip they overlap.
template <typename T, typename I, bool ADV, int ROWS, int E, int U, int RING,
          int GB, int ABL = 0>
__global__ __launch_bounds__(64) void csr_spmv_pipe2_kernel(
    int64_t n_rows, int64_t n_segments, int64_t segs_per_wave,
    const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ b, int64_t ldb,
    T* __restrict__ c, int64_t ldc, int nrhs, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p)
{
    static_assert((RING & (RING - 1)) == 0, "RING must be a power of two");
    constexpr int G = 64 * E * U;
    static_assert(RING >= 2 * G, "ring too small for the group size");
    static_assert(ROWS == 32 || ROWS == 64, "ROWS must be 32 or 64");
    constexpr int64_t MASK = RING - 1;
    __shared__ __attribute__((aligned(16))) T ringv[RING];
    __shared__ __attribute__((aligned(16))) I ringc[RING];

    const int lane = threadIdx.x;
    const int64_t sb = int64_t(blockIdx.x) * segs_per_wave;
    const int64_t se = sb + segs_per_wave < n_segments ? sb + segs_per_wave : n_segments;
    if (sb >= se) return;
    const int64_t row_e = se * ROWS < n_rows ? se * ROWS : n_rows;
    const int64_t K0 = row_ptrs[sb * ROWS];
    const int64_t K1 = row_ptrs[row_e];
    const int64_t NNZ = row_ptrs[n_rows];
    const int64_t K0a = K0 & ~int64_t(E - 1);

    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }

    using VT = vecT<T, E>;
    using VI = vecT<I, E>;

    auto load_group = [&](VT(&v)[U], VI(&ci)[U], int64_t p) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = p + (int64_t(u) * 64 + lane) * E;
            if (k >= K1) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    v[u].v[e] = T(0);
                    ci[u].v[e] = I(0);
                }
            } else if (k + E <= NNZ) {
                v[u] = *reinterpret_cast<const VT*>(vals + k);
                ci[u] = *reinterpret_cast<const VI*>(cols + k);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool in = k + e < NNZ;
                    v[u].v[e] = in ? vals[k + e] : T(0);
                    ci[u].v[e] = in ? cols[k + e] : I(0);
                }
            }
        }
    };
    auto stage = [&](VT(&v)[U], VI(&ci)[U], int64_t p) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = p + (int64_t(u) * 64 + lane) * E;
            *reinterpret_cast<VT*>(&ringv[k & MASK]) = v[u];
            *reinterpret_cast<VI*>(&ringc[k & MASK]) = ci[u];
        }
    };

    for (int j = 0; j < nrhs; ++j) {
        VT vA[U], vB[U];
        VI cA[U], cB[U];
        int64_t p_load = K0a;
        load_group(vA, cA, p_load);
        p_load += G;
        load_group(vB, cB, p_load);
        p_load += G;
        int64_t produced = K0a;
        int64_t cons = K0;
        bool use_a = true;

        int64_t seg = sb;
        auto seg_rows = [&](int64_t s, int64_t& rs, int64_t& re, int64_t& s_end) {
            const int64_t row = s * ROWS + lane;
            const int64_t last = (s + 1) * ROWS < n_rows ? (s + 1) * ROWS : n_rows;
            const bool valid = lane < ROWS && row < n_rows;
            rs = row_ptrs[valid ? row : last];
            re = row_ptrs[valid ? row + 1 : last];
            s_end = row_ptrs[last];
        };
        int64_t rs, re, seg_end, nrs = 0, nre = 0, nseg_end = 0;
        seg_rows(seg, rs, re, seg_end);
        if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
        T sum = T(0);
        {
            const int64_t row = seg * ROWS + lane;
            if (ADV && beta != T(0) && lane < ROWS && row < n_rows) {
                sum = c[row * ldc + j] * beta;
            }
        }

        while (seg < se) {
            if (produced >= seg_end || produced + G - cons > RING) {
                // ---- row phase: lane = row, gather b and accumulate in k order
                const int64_t upto = produced < seg_end ? produced : seg_end;
                wave_lds_sync();
                const bool is_long = (re - rs) > GKOC_CSR_LONG_ROW;
                int64_t k = rs > cons ? rs : cons;
                const int64_t e_ = is_long ? k : (re < upto ? re : upto);
                while (__any(k < e_)) {
                    // all loads unconditional (inactive slots read column 0):
                    // a per-element branch around a load would serialise the
                    // gathers (hipcc waits vmcnt(0) per guarded load)
                    T xv[GB];
#pragma unroll
                    for (int g = 0; g < GB; ++g) {
                        I cc = ringc[(k + g) & MASK];
                        cc = (k + g < e_) ? cc : I(0);
                        xv[g] = (ABL & 1) ? T(cc) : b[int64_t(cc) * ldb + j];
                    }
#pragma unroll
                    for (int g = 0; g < GB; ++g) {
                        const T vv = ringv[(k + g) & MASK];
                        const T t = sum + (ADV ? (alpha * vv) * xv[g] : vv * xv[g]);
                        sum = (k + g < e_) ? t : sum;
                    }
                    k += GB;
                }
                wave_lds_sync();
                cons = upto;
                if (cons >= seg_end) {
                    unsigned long long m = __ballot(is_long);
                    while (m) {
                        const int src = __builtin_ctzll(m);
                        m &= m - 1;
                        const int64_t lrs = __shfl(rs, src, 64);
                        const int64_t lre = __shfl(re, src, 64);
                        T part = T(0);
                        for (int64_t kk = lrs + lane; kk < lre; kk += 64) {
                            const T xb = b[int64_t(cols[kk]) * ldb + j];
                            part += ADV ? (alpha * vals[kk]) * xb : vals[kk] * xb;
                        }
                        part = wave_sum(part);
                        if (lane == src) sum += part;
                    }
                    const int64_t row = seg * ROWS + lane;
                    if (lane < ROWS && row < n_rows) c[row * ldc + j] = sum;
                    ++seg;
                    rs = nrs;
                    re = nre;
                    seg_end = nseg_end;
                    if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
                    sum = T(0);
                    const int64_t nrow = seg * ROWS + lane;
                    if (ADV && beta != T(0) && seg < se && lane < ROWS && nrow < n_rows) {
                        sum = c[nrow * ldc + j] * beta;
                    }
                }
                continue;
            }
            // ---- stage one landed group into the ring, refill its registers
            if (use_a) {
                stage(vA, cA, produced);
                load_group(vA, cA, p_load);
            } else {
                stage(vB, cB, produced);
                load_group(vB, cB, p_load);
            }
            p_load += G;
            produced += G;
            use_a = !use_a;
        }
    }
}


}  // namespace gkoc

namespace gkoc {

// ---------------------------------------------------------------------------
// Variant 4 (lab): producer / consumer wave pair.  Workgroup = 2 waves sharing
// one LDS ring of (val, col): wave 0 only streams the matrix into the ring
// (register sets, deep queue; its vmcnt queue holds nothing but stream loads),
// wave 1 only does the row phase (lane = row, row-ordered b gather, k-ordered
// sums; its vmcnt queue holds only gathers / row pointers / the c store).
// Hand-off through two LDS words (produced / consumed offsets); a wave's LDS
// operations execute in order, so flag-after-data needs no hardware fence.
template <typename T, typename I, int ROWS, int E, int U, int NSETS, int RING,
          int GB, int ABL = 0>
__global__ __launch_bounds__(128) void csr_spmv_pair_kernel(
    int64_t n_rows, int64_t n_segments, int64_t segs_per_wg,
    const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ b, T* __restrict__ c)
{
    static_assert((RING & (RING - 1)) == 0, "RING must be a power of two");
    constexpr int G = 64 * E * U;
    static_assert(RING >= 2 * G, "ring too small");
    constexpr int MASK = RING - 1;
    __shared__ __attribute__((aligned(16))) T ringv[RING];
    __shared__ __attribute__((aligned(16))) I ringc[RING];
    __shared__ volatile int sh_prod;
    __shared__ volatile int sh_cons;

    const int lane = threadIdx.x & 63;
    const bool producer = threadIdx.x < 64;
    const int64_t sb = int64_t(blockIdx.x) * segs_per_wg;
    const int64_t se = sb + segs_per_wg < n_segments ? sb + segs_per_wg : n_segments;
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
    if (threadIdx.x == 0) {
        sh_prod = 0;
        sh_cons = int(K0 - K0a);
    }
    __syncthreads();

    using VT = vecT<T, E>;
    using VI = vecT<I, E>;

    if (producer) {
        VT v[NSETS][U];
        VI ci[NSETS][U];
        auto load_group = [&](VT(&vv)[U], VI(&cc)[U], int p) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = p + (u * 64 + lane) * E;
                if (k >= k1o) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        vv[u].v[e] = T(0);
                        cc[u].v[e] = I(0);
                    }
                } else if (k + E <= nnzo) {
                    vv[u] = *reinterpret_cast<const VT*>(vals0 + k);
                    cc[u] = *reinterpret_cast<const VI*>(cols0 + k);
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const bool in = k + e < nnzo;
                        vv[u].v[e] = in ? vals0[k + e] : T(0);
                        cc[u].v[e] = in ? cols0[k + e] : I(0);
                    }
                }
            }
        };
        int p_load = 0;
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            load_group(v[s], ci[s], p_load);
            p_load += G;
        }
        int produced = 0;
        while (produced < k1o) {
#pragma unroll
            for (int s = 0; s < NSETS; ++s) {
                if (produced < k1o) {
                    // wait for ring space
                    while (produced + G - sh_cons > RING) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int k = produced + (u * 64 + lane) * E;
                        *reinterpret_cast<VT*>(&ringv[k & MASK]) = v[s][u];
                        *reinterpret_cast<VI*>(&ringc[k & MASK]) = ci[s][u];
                    }
                    load_group(v[s], ci[s], p_load);
                    p_load += G;
                    produced += G;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    if (lane == 0) sh_prod = produced;
                }
            }
        }
        return;
    }

    // ------------------------------------------------------------- consumer
    int cons = int(K0 - K0a);
    auto seg_rows = [&](int64_t s, int& rs, int& re, int& s_end) {
        const int64_t row = s * ROWS + lane;
        const int64_t last = (s + 1) * ROWS < n_rows ? (s + 1) * ROWS : n_rows;
        const bool valid = lane < ROWS && row < n_rows;
        rs = int(int64_t(row_ptrs[valid ? row : last]) - K0a);
        re = int(int64_t(row_ptrs[valid ? row + 1 : last]) - K0a);
        s_end = int(int64_t(row_ptrs[last]) - K0a);
    };
    int rs, re, seg_end, nrs = 0, nre = 0, nseg_end = 0;
    int64_t seg = sb;
    seg_rows(seg, rs, re, seg_end);
    if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
    T sum = T(0);
    while (seg < se) {
        int P = sh_prod;
        while (!(P >= seg_end || P + G - cons > RING)) {
            __builtin_amdgcn_s_sleep(1);
            P = sh_prod;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int upto = P < seg_end ? P : seg_end;
        int k = rs > cons ? rs : cons;
        const int e_ = re < upto ? re : upto;
        while (__any(k < e_)) {
            T xv[GB];
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                I cc = ringc[(k + g) & MASK];
                cc = (k + g < e_) ? cc : I(0);
                xv[g] = (ABL & 1) ? T(cc) : b[cc];
            }
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const T vv = ringv[(k + g) & MASK];
                const T t = sum + vv * xv[g];
                sum = (k + g < e_) ? t : sum;
            }
            k += GB;
        }
        cons = upto;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane == 0) sh_cons = cons;
        if (cons >= seg_end) {
            const int64_t row = seg * ROWS + lane;
            if (lane < ROWS && row < n_rows) c[row] = sum;
            ++seg;
            rs = nrs;
            re = nre;
            seg_end = nseg_end;
            if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
            sum = T(0);
        }
    }
}


// ---------------------------------------------------------------------------
// Variant 5 (lab): pipe3's structure with THREE register sets and hand-counted
// waits.  hipcc guards the tail cases of pipe3's stream loads with branches,
// and its s_waitcnt pass then falls back to vmcnt(0) before the gather - the
// stream load issued one step earlier is waited for in full, every step
// (ISA of pipe3: "s_waitcnt vmcnt(0)" ahead of the four gather loads).  Here
// every VMEM operation of the steady state is an asm statement the compiler
// does not count, issued in a fixed order per step
//     gather(group i)  ->  stream load(group i+2)  ->  s_waitcnt vmcnt(3)
// so that the only thing a step waits for is its own gather (and the stream
// load issued a whole step earlier), while 6 KB of stream data stay in flight.
// fp64 / int32, one right-hand side, unit strides, no long rows, and the padded
// stream range of every wave must lie inside the arrays (lab only).
typedef double lab_d2 __attribute__((ext_vector_type(2)));
typedef int lab_i4 __attribute__((ext_vector_type(4)));

typedef int lab_i2 __attribute__((ext_vector_type(2)));
// element t of a 256-nonzero group: v0/c0 = nonzeros 2*lane, 2*lane+1;
// v1/c1 = nonzeros 128 + 2*lane, 129 + 2*lane  (every load instruction covers
// whole cache lines exactly once: 1 KB / 1 KB / 512 B / 512 B)
struct lab_set {
    lab_d2 v0, v1;
    lab_i2 c0, c1;
};

#define LAB_LD(MODSTR)                                                                        \
    asm volatile("global_load_dwordx4 %0, %1, off" MODSTR : "=v"(S.v0) : "v"(pv) : "memory");   \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" MODSTR : "=v"(S.v1) : "v"(pv) : "memory"); \
    asm volatile("global_load_dwordx2 %0, %1, off" MODSTR : "=v"(S.c0) : "v"(pc) : "memory");   \
    asm volatile("global_load_dwordx2 %0, %1, off offset:512" MODSTR : "=v"(S.c1) : "v"(pc) : "memory");

template <int RING, int ABL = 0>
__global__ __launch_bounds__(64) void csr_spmv_pipe5_kernel(
    int64_t n_rows, int64_t n_segments, int64_t segs_per_wave,
    const int* __restrict__ row_ptrs, const int* __restrict__ cols,
    const double* __restrict__ vals, const double* __restrict__ b,
    double* __restrict__ c)
{
    constexpr int ROWS = 64, E = 4, G = 256;
    constexpr int MASK = RING - 1;
    __shared__ __attribute__((aligned(16))) double ring[RING];
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
    const double* vals0 = vals + K0a;
    const int* cols0 = cols + K0a;

    // lane's stream position; lanes past the wave's range (or the arrays) keep
    // re-reading the wave's first vector - branch-free and always in bounds
    constexpr int MOD = (ABL >> 4) & 7;
    auto issue_load = [&](lab_set& S, int p) {
        // the whole 256-nonzero group must be inside the wave's padded range and
        // the arrays; otherwise re-read the wave's first group (lab simplification)
        const int pp = (p + G <= nnzo) ? p : 0;
        const double* pv = vals0 + pp + 2 * lane;
        const int* pc = cols0 + pp + 2 * lane;
        if (MOD == 0) { LAB_LD("") }
        else if (MOD == 1) { LAB_LD(" nt") }
        else if (MOD == 2) { LAB_LD(" sc1") }
        else if (MOD == 3) { LAB_LD(" sc0 sc1") }
        else if (MOD == 4) { LAB_LD(" sc1 nt") }
        else { LAB_LD(" sc0 sc1 nt") }
    };
    double x0, x1, x2, x3;
    auto issue_gather = [&](const lab_set& S) {
        if (ABL & 1) {
            x0 = S.c0.x; x1 = S.c0.y; x2 = S.c1.x; x3 = S.c1.y;
            return;
        }
        const unsigned o0 = unsigned(S.c0.x) * 8u, o1 = unsigned(S.c0.y) * 8u,
                       o2 = unsigned(S.c1.x) * 8u, o3 = unsigned(S.c1.y) * 8u;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(x0) : "v"(o0), "s"(b) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(x1) : "v"(o1), "s"(b) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(x2) : "v"(o2), "s"(b) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(x3) : "v"(o3), "s"(b) : "memory");
    };
    auto write_products = [&](const lab_set& S, int p) {
        const int k = p + 2 * lane;
        lab_d2 p0, p1;
        p0.x = S.v0.x * x0;
        p0.y = S.v0.y * x1;
        p1.x = S.v1.x * x2;
        p1.y = S.v1.y * x3;
        *reinterpret_cast<lab_d2*>(&ring[k & MASK]) = p0;
        *reinterpret_cast<lab_d2*>(&ring[(k + 128) & MASK]) = p1;
    };

    lab_set S0, S1, S2;
    issue_load(S0, 0);
    issue_load(S1, G);
    // everything outstanding so far must land before the loop (row_ptrs too)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(S0.v0), "+v"(S0.v1), "+v"(S0.c0), "+v"(S0.c1), "+v"(S1.v0), "+v"(S1.v1), "+v"(S1.c0), "+v"(S1.c1)::"memory");
    int p_load = 2 * G;
    int produced = 0;
    int cons = int(K0 - K0a);

    int64_t seg = sb;
    auto seg_rows = [&](int64_t s, int& rs, int& re, int& s_end) {
        const int64_t row = s * ROWS + lane;
        const int64_t last = (s + 1) * ROWS < n_rows ? (s + 1) * ROWS : n_rows;
        const bool valid = row < n_rows;
        rs = int(int64_t(row_ptrs[valid ? row : last]) - K0a);
        re = int(int64_t(row_ptrs[valid ? row + 1 : last]) - K0a);
        s_end = int(int64_t(row_ptrs[last]) - K0a);
    };
    int rs, re, seg_end, nrs = 0, nre = 0, nseg_end = 0;
    seg_rows(seg, rs, re, seg_end);
    if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
    double sum = 0;
    double ys0 = 0, ys1 = 0;

    // one produce step: X = set to turn into products, Y = next set (its loads
    // are older than X's gather), Z = free set that receives group i+2
#define LAB_STEP(X, Y, Z)                                                          \
    {                                                                              \
        issue_gather(X);                                                           \
        issue_load(Z, p_load);                                                     \
        if (ABL & 1) {                                                             \
            asm volatile("s_waitcnt vmcnt(4)"                                      \
                         : "+v"(X.v0), "+v"(X.v1), "+v"(Y.v0), "+v"(Y.v1), "+v"(Y.c0), "+v"(Y.c1)::"memory"); \
        } else {                                                                   \
            asm volatile("s_waitcnt vmcnt(4)"                                      \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(X.v0), "+v"(X.v1), \
                           "+v"(Y.v0), "+v"(Y.v1), "+v"(Y.c0), "+v"(Y.c1)::"memory"); \
        }                                                                          \
        write_products(X, produced);                                               \
        p_load += G;                                                               \
        produced += G;                                                             \
    }

    // row phases until the next produce step is possible (or the wave is done)
    auto row_phases = [&]() {
        while (seg < se && (produced >= seg_end || produced + G - cons > RING)) {
            const int upto = produced < seg_end ? produced : seg_end;
            wave_lds_sync();
            if (!(ABL & 2)) {
                int k = rs > cons ? rs : cons;
                const int e_ = re < upto ? re : upto;
                for (; k + 4 <= e_; k += 4) {
                    const double t0 = ring[k & MASK];
                    const double t1 = ring[(k + 1) & MASK];
                    const double t2 = ring[(k + 2) & MASK];
                    const double t3 = ring[(k + 3) & MASK];
                    sum += t0;
                    sum += t1;
                    sum += t2;
                    sum += t3;
                }
                for (; k < e_; ++k) sum += ring[k & MASK];
            }
            wave_lds_sync();
            cons = upto;
            if (cons >= seg_end) {
                if (seg == sb) ys0 = sum; else ys1 = sum;
                ++seg;
                rs = nrs;
                re = nre;
                seg_end = nseg_end;
                if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
                sum = 0;
            }
        }
    };
    // the three sets keep their registers: the rotation is unrolled, never
    // expressed as data movement (a copy of a set with a load in flight would
    // copy stale registers)
    for (;;) {
        row_phases();
        if (seg >= se) break;
        LAB_STEP(S0, S1, S2)
        row_phases();
        if (seg >= se) break;
        LAB_STEP(S1, S2, S0)
        row_phases();
        if (seg >= se) break;
        LAB_STEP(S2, S0, S1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const int64_t r0 = sb * ROWS + lane, r1 = (sb + 1) * ROWS + lane;
        if (r0 < n_rows) c[r0] = ys0;
        if (sb + 1 < se && r1 < n_rows) c[r1] = ys1;
    }
#undef LAB_STEP
}

}  // namespace gkoc
