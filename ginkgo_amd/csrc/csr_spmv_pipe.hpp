// Software-pipelined row-segment-per-wavefront CSR SpMV (gfx950).
//
// Each wavefront owns a CONTIGUOUS range of row segments, hence one contiguous
// range [K0, K1) of the val / col_idx streams, and walks it in groups of
// G = 64*E*U nonzeros:
//   * two register sets (A, B) hold the next two groups; the loads of group
//     j+2 are issued as soon as group j's products are written, so the HBM
//     stream round trip overlaps the b-vector gather round trip of the next
//     group instead of being serialised with it (vmcnt is in-order, which is
//     why the order "gather(j) -> wait -> write -> load(j+2)" is used);
//   * lane = E consecutive nonzeros per batch: val / col loads are 16-byte
//     vectors, every wave instruction reads one contiguous 64*E*8 B (val) or
//     64*E*4 B (col) run;
//   * products go to an LDS ring (index = nonzero index mod RING); whenever a
//     segment of ROWS rows is complete, lane = row adds its products from the
//     ring in k order (reference summation order, separate mul/add =>
//     bit-identical to the sequential reference) and the wave stores ROWS
//     contiguous results;
//   * a segment larger than the ring is consumed in several passes with the
//     partial sums carried in registers (order still sequential);
//   * the next segment's row pointers are prefetched one segment ahead.
// No atomics, no pre-zeroing of c, no host-side srow table.
#pragma once
#include "common.hpp"

namespace gkoc {

#ifdef __HIPCC__

template <typename T, int E>
struct alignas(sizeof(T) * E) vecT {
    T v[E];
};

// All stream positions are 32-bit offsets from the wave's aligned stream start
// K0a (keeps the address arithmetic in 32-bit registers: 72 VGPRs, 5 waves per
// SIMD with the 8 KB ring).  WPS = minimum waves per SIMD the register
// allocator must admit; ABL = mode bits: 64 = also emit this wave's part of
// <b, c> (square matrices, one right-hand side; used by gkoc_x_csr_spmv_dot);
// the others are measurement-only switches of tools/spmv_lab.hip (1: skip the
// b gather, 2: skip the LDS row sums, ...), 0 in the plain library kernel.
// (Earlier generations of this kernel - 64-bit indexing, LDS-staged matrix
// with row-ordered gather - are kept in tools/lab_kernels.hpp for A/B runs.)
template <typename T, typename I, bool ADV, int ROWS, int E, int U, int RING,
          int WPS, int ABL = 0>
__global__ __launch_bounds__(64, WPS) void csr_spmv_pipe3_kernel(
    int64_t n_rows, int64_t n_segments, int64_t segs_per_wave,
    const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, const T* __restrict__ b, int64_t ldb,
    T* __restrict__ c, int64_t ldc, int nrhs, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p, T* __restrict__ dot_partial = nullptr,
    int xcd_map = 0)
{
    static_assert((RING & (RING - 1)) == 0, "RING must be a power of two");
    constexpr int G = 64 * E * U;
    static_assert(RING >= 2 * G, "ring too small for the group size");
    static_assert(ROWS == 32 || ROWS == 64, "ROWS must be 32 or 64");
    constexpr int MASK = RING - 1;
    constexpr bool DOT = (ABL & 64) != 0;
    // DEFER = n > 0: the wave keeps the results of its (at most n) row segments
    // in registers and writes them in one burst at its end - >= 1 KB of
    // contiguous output per wave instead of one 256 B piece per segment
    // (sparse small writes between the read streams cost several times their
    // byte share at the memory side)
    constexpr int DEFER = (ABL >> 12) & 15;
    __shared__ __attribute__((aligned(64))) T ring[RING];

    const int lane = threadIdx.x;
    T dot_acc = T(0);
    int64_t wave_id = blockIdx.x;
    if (ABL >> 8) {
        // measurement only: workgroup b runs on XCD b % 8; hand every XCD
        // chunks of C consecutive waves instead of every 8th wave
        constexpr int64_t C = int64_t(1) << ((ABL >> 8) & 15);
        const int64_t nfull = (int64_t(gridDim.x) / (8 * C)) * (8 * C);
        if (wave_id < nfull) {
            const int64_t xcd = wave_id % 8, slot = wave_id / 8;
            wave_id = ((slot / C) * 8 + xcd) * C + (slot % C);
        }
    }
    if (xcd_map) {
        // Workgroup b runs on XCD b % 8 (observed dispatch rule; a wrong guess
        // costs speed, never correctness - the map is a bijection).  Give each
        // XCD ONE contiguous eighth of the waves, walked in order: the b lines
        // shared by neighbouring rows are then fetched by one L2 instead of by
        // all eight, and a CU only ever translates addresses of its own eighth.
        const int64_t nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int64_t xcd = wave_id & 7, slot = wave_id >> 3;
        wave_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int64_t sb = wave_id * segs_per_wave;
    const int64_t se = sb + segs_per_wave < n_segments ? sb + segs_per_wave : n_segments;
    if (sb >= se) {
        if (DOT && lane == 0) dot_partial[wave_id] = T(0);
        return;
    }
    const int64_t row_e = se * ROWS < n_rows ? se * ROWS : n_rows;
    const int64_t K0 = row_ptrs[sb * ROWS];
    const int64_t K1 = row_ptrs[row_e];
    const int64_t NNZ = row_ptrs[n_rows];
    const int64_t K0a = K0 & ~int64_t(E - 1);
    // wave-relative 32-bit offsets (a wave owns at most two 64-row segments, whose
    // entries beyond GKOC_CSR_LONG_ROW per row are summed by the whole wave; the
    // launcher refuses more than 2^31 segments)
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

    // b with unit row stride (every vector of a Krylov solver): the gather address is
    // base + 8 col; a run-time stride costs a 64-bit multiply per nonzero (three quarter-rate
    // integer multiplies in the ISA), more than the product itself
    const bool unit_b = ldb == 1;
    for (int j = 0; j < nrhs; ++j) {
        const T* __restrict__ bj = b + j;
        auto produce = [&](VT(&v)[U], VI(&ci)[U], int p) {
            VT xv[U];
            if (unit_b) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        xv[u].v[e] = (ABL & 1) ? T(ci[u].v[e]) : bj[int64_t(ci[u].v[e])];
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        xv[u].v[e] = (ABL & 1) ? T(ci[u].v[e])
                                               : bj[int64_t(ci[u].v[e]) * ldb];
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
                const int k = p + (u * 64 + lane) * E;
                *reinterpret_cast<VT*>(&ring[k & MASK]) = pr;
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
            const bool valid = lane < ROWS && row < n_rows;
            rs = int(int64_t(row_ptrs[valid ? row : last]) - K0a);
            re = int(int64_t(row_ptrs[valid ? row + 1 : last]) - K0a);
            s_end = int(int64_t(row_ptrs[last]) - K0a);
        };
        int rs, re, seg_end, nrs = 0, nre = 0, nseg_end = 0;
        seg_rows(seg, rs, re, seg_end);
        if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
        T sum = T(0);
        T ys[DEFER > 0 ? DEFER : 1];
        {
            const int64_t row = seg * ROWS + lane;
            if (ADV && beta != T(0) && lane < ROWS && row < n_rows) {
                sum = c[row * ldc + j] * beta;
            }
        }

        while (seg < se) {
            if (produced >= seg_end || produced + G - cons > RING) {
                const int upto = produced < seg_end ? produced : seg_end;
                wave_lds_sync();
                const bool is_long = (re - rs) > GKOC_CSR_LONG_ROW;
                if (!is_long && !(ABL & 2)) {
                    int k = rs > cons ? rs : cons;
                    const int e_ = re < upto ? re : upto;
                    // Four products per step, added in k order.  The range [k, e_) is at most
                    // one ring long, so it wraps at most once: split there and both pieces are
                    // contiguous in LDS - one pointer increment per step, the other three
                    // addresses are immediate offsets, and the loads of step i+1 are issued
                    // before the adds of step i.  With 81-nonzero rows this loop was 2/3 of the
                    // kernel's vector instructions (rocprofv3 SQ_INSTS_VALU, Flan-like matrix).
                    auto run = [&](int lo_, int hi_) {
                        const T* q4 = ring + (lo_ & MASK);
                        int n4 = (hi_ - lo_) >> 2;
                        if (n4 > 0) {
                            // two register sets, alternating: no copies between the steps
                            T a0 = q4[0], a1 = q4[1], a2 = q4[2], a3 = q4[3];
                            q4 += 4;
                            --n4;
                            while (n4 >= 2) {
                                const T b0 = q4[0], b1 = q4[1], b2 = q4[2], b3 = q4[3];
                                sum += a0;
                                sum += a1;
                                sum += a2;
                                sum += a3;
                                a0 = q4[4];
                                a1 = q4[5];
                                a2 = q4[6];
                                a3 = q4[7];
                                q4 += 8;
                                n4 -= 2;
                                sum += b0;
                                sum += b1;
                                sum += b2;
                                sum += b3;
                            }
                            sum += a0;
                            sum += a1;
                            sum += a2;
                            sum += a3;
                            if (n4 == 1) {
                                const T b0 = q4[0], b1 = q4[1], b2 = q4[2], b3 = q4[3];
                                q4 += 4;
                                sum += b0;
                                sum += b1;
                                sum += b2;
                                sum += b3;
                            }
                        }
                        const int rem = (hi_ - lo_) & 3;
                        if (rem > 0) sum += q4[0];
                        if (rem > 1) sum += q4[1];
                        if (rem > 2) sum += q4[2];
                    };
                    if (k < e_) {
                        const int wrap = (k | MASK) + 1;   // first index behind k that maps to ring[0]
                        if (wrap < e_) {
                            run(k, wrap);
                            run(wrap, e_);
                        } else {
                            run(k, e_);
                        }
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
                        T part = T(0);
                        for (int k = lrs + lane; k < lre; k += 64) {
                            const T xb = bj[int64_t(cols0[k]) * ldb];
                            part += ADV ? (alpha * vals0[k]) * xb : vals0[k] * xb;
                        }
                        part = wave_sum(part);
                        if (lane == src) sum += part;
                    }
                    const int64_t row = seg * ROWS + lane;
                    if (DOT && lane < ROWS && row < n_rows) {
                        dot_acc += bj[row * ldb] * sum;
                    }
                    if (DEFER > 0) {
                        const int kseg = int(seg - sb);
#pragma unroll
                        for (int t = 0; t < DEFER; ++t) ys[t] = t == kseg ? sum : ys[t];
                    } else if (lane < ROWS && row < n_rows) {
                        if (ABL & 32) {
                            __builtin_nontemporal_store(sum, &c[row * ldc + j]);
                        } else {
                            c[row * ldc + j] = sum;
                        }
                    }
#ifdef GKOC_LAB_TIMESTAMPS
                    if ((ABL & 16) && lane == 0 && seg + 1 == se) {
                        gkoc_lab_ts[blockIdx.x] = wall_clock64();
                    }
#endif
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
        if (DEFER > 0) {
#pragma unroll
            for (int t = 0; t < DEFER; ++t) {
                const int64_t row = (sb + t) * ROWS + lane;
                if (sb + t < se && lane < ROWS && row < n_rows) c[row * ldc + j] = ys[t];
            }
        }
    }
    if (DOT) {
        // fixed butterfly: the partial of a wave does not depend on timing
        dot_acc = wave_sum(dot_acc);
        if (lane == 0) dot_partial[wave_id] = dot_acc;
    }
}

#endif  // __HIPCC__

}  // namespace gkoc
