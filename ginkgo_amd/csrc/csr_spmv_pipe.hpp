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
// V = type the matrix values are STORED in (mixed precision: float values, double vectors and
// arithmetic - csr::spmv<MatrixValueType, InputValueType, OutputValueType> with
// arithmetic_type = highest_precision, common/cuda_hip/matrix/csr_kernels.template.cpp); every
// value is widened as it is loaded, nothing else changes.
template <typename T, typename I, bool ADV, int ROWS, int E, int U, int RING,
          int WPS, int ABL = 0, typename V = T>
__global__ __launch_bounds__(64, WPS) void csr_spmv_pipe3_kernel(
    int64_t n_rows_in, int64_t n_segments_in, int64_t segs_per_wave,
    const I* __restrict__ row_ptrs_in, const I* __restrict__ cols_in,
    const V* __restrict__ vals_in, const T* __restrict__ b, int64_t ldb,
    T* __restrict__ c_in, int64_t ldc, int nrhs, const T* __restrict__ alpha_p,
    const T* __restrict__ beta_p, T* __restrict__ dot_partial = nullptr,
    int xcd_map = 0, const I* __restrict__ row_idxs = nullptr,
    int* __restrict__ unsorted_flag = nullptr, int64_t head_rows = 0, int64_t tail_rows = 0,
    const uint32_t* __restrict__ gate = nullptr, uint32_t gate_epoch = 0,
    const I* __restrict__ bnd_ptrs = nullptr, const I* __restrict__ bnd_cols = nullptr,
    const V* __restrict__ bnd_vals = nullptr, uint32_t* __restrict__ fork_word = nullptr,
    uint32_t fork_number = 0, int gate_fence = 0, int64_t bnd_first = -1,
    const uint32_t* __restrict__ seg_skip = nullptr)
{
    // (GATE mode points a wave at one of two matrices; everywhere else these are the arguments)
    int64_t n_rows = n_rows_in, n_segments = n_segments_in;
    const I* __restrict__ row_ptrs = row_ptrs_in;
    const I* __restrict__ cols = cols_in;
    const V* __restrict__ vals = vals_in;
    T* __restrict__ c = c_in;
    int64_t out_jump_at = 0, out_jump = 0;      // GATE: output row = r < jump_at ? r : r + jump
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
    // COO = ABL & 128: the matrix is row-sorted COO (coo.hip).  `row_ptrs` then holds ONE pointer
    // per ROWS-row segment (n_segments + 1 entries, found by bisection of row_idxs); the row index
    // of every entry travels with its value and column through the same loads, and the entry that
    // starts a run of equal row indices notes, in an LDS table, where its row (and the empty rows
    // in front of it) begins - the pointers the row phase needs appear while the products are
    // written, no pass over row_idxs of its own, no row-pointer array in memory.  The same walk
    // proves that the wave's piece is sorted and belongs to its rows; otherwise unsorted_flag is
    // raised (the launcher then redoes the product with atomics).  Long rows are summed by their
    // lane like all others (the cooperative path needs the row length in advance).
    constexpr bool COO = (ABL & 128) != 0;
    constexpr int MAXSEG = DEFER > 0 ? DEFER : 2;
    constexpr int TAB_NONE = 0x7fffffff;
    __shared__ __attribute__((aligned(64))) T ring[RING];
    __shared__ int tab[COO ? ROWS * MAXSEG + 1 : 1];

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
    // GATE = ABL & 0x10000 (distributed product, one segment per wave).  Two matrices in one launch:
    //   * the rank's LOCAL block (row_ptrs / cols / vals, columns = local indices): its rows
    //     [head_rows, n_rows - tail_rows) read no halo entry and are complete - the interior waves
    //     compute them (row pointers and output shifted by head_rows);
    //   * the boundary rows - the first head_rows and the last tail_rows - as COMPLETE rows over
    //     [local columns | halo] (bnd_ptrs / bnd_cols / bnd_vals, head rows then tail rows; b is the
    //     local vector with the halo behind it).  They are given to the LAST waves of the grid, and
    //     those wait - every wave for itself - until the exchange, which travels on another stream
    //     while the interior rows are computed, has delivered the halo: *gate holds the number of
    //     the last exchange that has arrived (gkoc_gate_open on the exchange's stream), gate_epoch is
    //     the number of the exchange this product needs (the caller counts).
    // No second kernel beside this one, no event the stream waits for, no atomic read-modify-write
    // (2048 same-address atomics of the boundary waves cost 15 us when tried), no second copy of the
    // matrix (round 3 kept all rows again over [local | halo]).  By the time the last waves start the
    // halo has usually arrived; a wave that has to wait polls with s_sleep (after ~10 s it gives up,
    // sets gate[1] and goes on with whatever the halo holds: the caller checks).  The agent-scope acquire
    // behind the gate is paid only by a wave that HAD to wait (or by all of them with gate_fence = 1,
    // GKOC_TUNE_GATE_FENCE): see the argument at the fence below - it rests on the halo starting on a
    // 128-byte line of its own, which the launcher checks (launch_csr_gated: b and the halo offset
    // aligned, otherwise gate_fence is forced to 1).  Soaked: tests/test_gate_soak_gpu.py, 10^4 products
    // with a halo rewritten every iteration by a kernel on all XCDs, gate early and late.
    // Forward progress: the launcher admits at most 8 spinning waves per CU
    // (gkoc_csr_spmv_gated_fits), so the exchange's kernels and gkoc_gate_open always find room.
    constexpr bool GATE = (ABL & 0x10000) != 0;
    if constexpr (GATE) {
        // The fork of the exchange: "b is final" is what this kernel's START means on its stream, so
        // its first wave says so (the exchange's stream polls fork_word, gkoc_stream_fork_wait) - no
        // event record, no kernel of its own in front of this one on the main queue (either leaves
        // the device idle for 5-6 us: profiles/r04_dist_sim_timelines.txt).
        if (fork_word != nullptr && blockIdx.x == 0 && lane == 0) {
            __hip_atomic_store(fork_word, fork_number, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int64_t n_int_rows = n_rows_in - head_rows - tail_rows;
        const int64_t n_int = (n_int_rows + ROWS - 1) / ROWS;
        // the boundary waves are the waves [bnd_first, bnd_first + n_bnd) of the grid (bnd_first < 0:
        // the last ones): late enough for the halo to have arrived, early enough for the END of the
        // launch - where few waves are left and every one of them waits for memory alone - to consist
        // of interior rows, whose entries of b their neighbours have just read
        const int64_t n_bnd_w = n_segments_in - n_int;
        const int64_t first = bnd_first < 0 || bnd_first > n_int ? n_int : bnd_first;
        const bool is_bnd = wave_id >= first && wave_id < first + n_bnd_w;
        if (is_bnd) {
            wave_id -= first;
            row_ptrs = bnd_ptrs;
            cols = bnd_cols;
            vals = bnd_vals;
            n_rows = head_rows + tail_rows;
            n_segments = (n_rows + ROWS - 1) / ROWS;
            out_jump_at = head_rows;
            out_jump = n_rows_in - tail_rows - head_rows;
            bool waited = false;
            if (lane == 0) {
                long spins = 0;
                while (int32_t(__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) -
                               gate_epoch) < 0) {
                    __builtin_amdgcn_s_sleep(32);
                    waited = true;
                    if (++spins > (long(1) << 23)) {   // ~10 s: give up, say so, go on (the caller checks gate[1])
                        __hip_atomic_store(const_cast<uint32_t*>(gate) + 1, 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // Ordering of the halo reads behind the gate.  For the compiler: a workgroup-scope acquire
            // (no instruction).  For the caches: an agent-scope acquire invalidates this XCD's L2 for
            // EVERY wave that runs on it - 2048 of them cost the product 8 us (138 -> 146 us per rank
            // of 8 on 256^3, profiles/r04_dist_sim_variants.txt) - and is needed only if a line of the halo
            // can sit in a cache from BEFORE the exchange wrote it.  None can: the halo starts on a
            // 128-byte boundary behind the local vector, so interior waves never touch its lines;
            // boundary waves touch them only after they have seen the gate open (the spin loop's exit
            // is a branch on the loaded value: the loads behind it are issued after it), i.e. after
            // the exchange's kernel has ended and released its writes; and what the PREVIOUS product
            // left in the caches was dropped by the acquire at this kernel's start.  A wave that did
            // wait is the one case where its own CU may have re-fetched around it: it pays the fence.
            // gate_fence = 1 (GKOC_TUNE_GATE_FENCE) makes every boundary wave pay it.
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // gate_fence = 2 (a peer of the communicator sits on ANOTHER device, gate_fence_policy):
            // every boundary wave a SYSTEM-scope acquire - nothing is assumed about which agent wrote the
            // halo or about what this device's caches hold of lines another device has written.
            if (gate_fence >= 2) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            } else if (gate_fence != 0 || __builtin_amdgcn_readfirstlane(int(waited)) != 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else {
            if (wave_id >= first) wave_id -= n_bnd_w;
            row_ptrs = row_ptrs_in + head_rows;
            c = c_in + head_rows;
            n_rows = n_int_rows;
            n_segments = n_int;
        }
    }
    int64_t sb = wave_id * segs_per_wave;
    int64_t se = sb + segs_per_wave < n_segments ? sb + segs_per_wave : n_segments;
    if (seg_skip != nullptr) {
        // segments that hold a row far longer than the rest belong to csr_flagged_segments_kernel
        // (csr_long_rows.hpp): a wave owns one or two segments (the launcher sees to that when a matrix has
        // flagged segments), so what is left is one range.  (Round 6 walked the RUNS between flagged segments
        // in a loop, for waves of up to eight short-row segments: never faster, and the loop cost every
        // variant of this kernel 12-16 VGPRs - a wave per SIMD for most of them, L256 with the dot 985 ->
        // 1107 us; profiles/r06/r06_bench_kernel_stats_regression.csv.)
        while (sb < se && ((seg_skip[sb >> 5] >> (sb & 31)) & 1u) != 0) ++sb;
        while (se > sb && ((seg_skip[(se - 1) >> 5] >> ((se - 1) & 31)) & 1u) != 0) --se;
    }
    // (GATE renumbers the boundary waves: a wave's partial sum has the number of its workgroup)
    const int64_t dot_slot = GATE ? int64_t(blockIdx.x) : wave_id;
    if (sb >= se) {
        if (DOT && lane == 0) dot_partial[dot_slot] = T(0);
        return;
    }
    const int64_t row_e = se * ROWS < n_rows ? se * ROWS : n_rows;
    const int64_t K0 = COO ? row_ptrs[sb] : row_ptrs[sb * ROWS];
    const int64_t K1r = COO ? row_ptrs[se] : row_ptrs[row_e];
    const int64_t NNZ = COO ? row_ptrs[n_segments] : row_ptrs[n_rows];
    bool coo_bad = COO && K1r < K0;
    const int64_t K1 = coo_bad ? K0 : K1r;
    const int64_t K0a = K0 & ~int64_t(E - 1);
    const int64_t R0 = sb * ROWS;              // COO: first row of the wave
    const int nrw = int(row_e - R0);           //      its number of rows
    const int k0o = int(K0 - K0a);
    int carry_rel = -1;                        //      (row - R0) of the entry in front of the group
    const I* __restrict__ rows0 = COO ? row_idxs + K0a : nullptr;
    if constexpr (COO) {
        for (int t = lane; t <= nrw; t += 64) tab[t] = K0 == K1 ? k0o : TAB_NONE;
        wave_lds_sync();
    }
    // wave-relative 32-bit offsets (a wave owns at most two 64-row segments, whose
    // entries beyond GKOC_CSR_LONG_ROW per row are summed by the whole wave; the
    // launcher refuses more than 2^31 segments)
    const int k1o = int(K1 - K0a);
    const int nnzo = (NNZ - K0a) > int64_t(0x7fffff00) ? 0x7fffff00 : int(NNZ - K0a);
    const V* __restrict__ vals0 = vals + K0a;
    const I* __restrict__ cols0 = cols + K0a;

    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }

    using VT = vecT<T, E>;
    using VV = vecT<V, E>;
    using VI = vecT<I, E>;

    auto load_group = [&](VT(&v)[U], VI(&ci)[U], VI(&ri)[COO ? U : 1], int p) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = p + (u * 64 + lane) * E;
            if (k >= k1o) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    v[u].v[e] = T(0);
                    ci[u].v[e] = I(0);
                    if constexpr (COO) ri[u].v[e] = I(0);
                }
            } else if (k + E <= nnzo) {
                if constexpr (sizeof(V) == sizeof(T)) {
                    v[u] = *reinterpret_cast<const VT*>(vals0 + k);
                } else {
                    const VV raw = *reinterpret_cast<const VV*>(vals0 + k);
#pragma unroll
                    for (int e = 0; e < E; ++e) v[u].v[e] = T(raw.v[e]);
                }
                ci[u] = *reinterpret_cast<const VI*>(cols0 + k);
                if constexpr (COO) ri[u] = *reinterpret_cast<const VI*>(rows0 + k);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool in = k + e < nnzo;
                    v[u].v[e] = in ? T(vals0[k + e]) : T(0);
                    ci[u].v[e] = in ? cols0[k + e] : I(0);
                    if constexpr (COO) ri[u].v[e] = in ? rows0[k + e] : I(0);
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
        auto produce = [&](VT(&v)[U], VI(&ci)[U], VI(&ri)[COO ? U : 1], int p) {
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
            if constexpr (COO) {
                // rows relative to the wave's first row, in 32 bits (the range check is done on
                // the full index)
                auto rel = [&](I row) {
                    const int64_t d = int64_t(row) - R0;
                    return (d < -1 || d > int64_t(nrw)) ? (d < 0 ? -2 : nrw + 1) : int(d);
                };
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kb = p + (u * 64 + lane) * E;
                    // the entry in front of a lane's first one: in the lane to its left, in lane
                    // 63 of the load before, or in the group before
                    const int last_rel = rel(ri[u].v[E - 1]);
                    const int left = __shfl_up(last_rel, 1, 64);
                    const int before = u > 0 ? __shfl(rel(ri[u > 0 ? u - 1 : 0].v[E - 1]), 63, 64) : carry_rel;
                    int pv = lane == 0 ? before : left;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int k = kb + e;
                        const int cur = rel(ri[u].v[e]);
                        if (k < k0o || k >= k1o) {
                            pv = cur;
                            continue;
                        }
                        if (k == k0o) pv = -1;
                        if (cur < 0 || cur >= nrw || cur < pv) {
                            coo_bad = true;
                            pv = cur;
                            continue;
                        }
                        for (int t = pv + 1; t <= cur; ++t) tab[t] = k;
                        if (k == k1o - 1) {
                            for (int t = cur + 1; t <= nrw; ++t) tab[t] = k1o;
                        }
                        pv = cur;
                    }
                    if (u == U - 1) carry_rel = __shfl(last_rel, 63, 64);
                }
            }
        };

        VT vA[U], vB[U];
        VI cA[U], cB[U];
        VI rA[COO ? U : 1], rB[COO ? U : 1];
        int p_load = 0;
        load_group(vA, cA, rA, p_load);
        p_load += G;
        load_group(vB, cB, rB, p_load);
        p_load += G;
        int produced = 0;          // offsets relative to K0a
        int cons = int(K0 - K0a);
        bool use_a = true;

        int64_t seg = sb;
        auto seg_rows = [&](int64_t s, int& rs, int& re, int& s_end) {
            const int64_t row = s * ROWS + lane;
            const int64_t last = (s + 1) * ROWS < n_rows ? (s + 1) * ROWS : n_rows;
            const bool valid = lane < ROWS && row < n_rows;
            if constexpr (COO) {
                // read from the table at every hand-over (coo_rows below); nothing to prefetch
                rs = re = s_end = TAB_NONE;
            } else {
                rs = int(int64_t(row_ptrs[valid ? row : last]) - K0a);
                re = int(int64_t(row_ptrs[valid ? row + 1 : last]) - K0a);
                s_end = int(int64_t(row_ptrs[last]) - K0a);
            }
        };
        // COO: what the table knows about segment s after `produced` entries: a row whose first
        // entry has not been seen yet (or that is empty) starts where the next known row starts;
        // TAB_NONE = not known yet = behind everything produced so far
        auto coo_rows = [&](int64_t s, int& rs, int& re, int& s_end, int produced_) {
            const int64_t row = s * ROWS + lane;
            const int64_t last = (s + 1) * ROWS < n_rows ? (s + 1) * ROWS : n_rows;
            const bool valid = lane < ROWS && row < n_rows;
            rs = tab[int((valid ? row : last) - R0)];
            re = tab[int((valid ? row + 1 : last) - R0)];
            s_end = tab[int(last - R0)];
            if (produced_ >= k1o) {
                // the whole piece has been seen: whatever is still unknown lies behind it
                rs = rs < k1o ? rs : k1o;
                re = re < k1o ? re : k1o;
                s_end = s_end < k1o ? s_end : k1o;
            }
        };
        int rs, re, seg_end, nrs = 0, nre = 0, nseg_end = 0;
        seg_rows(seg, rs, re, seg_end);
        if (seg + 1 < se) seg_rows(seg + 1, nrs, nre, nseg_end);
        T sum = T(0);
        T ys[DEFER > 0 ? DEFER : 1];
        // DOT: this lane's entry of b for the row it sums, loaded when the segment begins (used when
        // it ends: a load issued there would be waited for by every wave - 8 us of a 140 us product)
        auto b_of_row = [&](int64_t s_) {
            const int64_t row = s_ * ROWS + lane;
            if (!(DOT && s_ < se && lane < ROWS && row < n_rows)) return T(0);
            const int64_t brow = GATE && row >= out_jump_at ? row + out_jump : row;
            return bj[(GATE ? brow + (c - c_in) : brow) * ldb];
        };
        T bdot = b_of_row(seg);
        {
            const int64_t row = seg * ROWS + lane;
            if (ADV && beta != T(0) && lane < ROWS && row < n_rows) {
                sum = c[row * ldc + j] * beta;
            }
        }

        while (seg < se) {
            if constexpr (COO) {
                wave_lds_sync();
                coo_rows(seg, rs, re, seg_end, produced);
            }
            if (produced >= seg_end || produced + G - cons > RING) {
                const int upto = produced < seg_end ? produced : seg_end;
                wave_lds_sync();
                if constexpr (COO) {
                    rs = rs < upto ? rs : upto;
                    re = re < upto ? re : upto;
                }
                const bool is_long = !COO && (re - rs) > GKOC_CSR_LONG_ROW;
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
                            part += ADV ? (alpha * T(vals0[k])) * xb : T(vals0[k]) * xb;
                        }
                        part = wave_sum(part);
                        if (lane == src) sum += part;
                    }
                    const int64_t row = seg * ROWS + lane;
                    if (DOT && lane < ROWS && row < n_rows) dot_acc += bdot * sum;
                    if (DEFER > 0) {
                        const int kseg = int(seg - sb);
#pragma unroll
                        for (int t = 0; t < DEFER; ++t) ys[t] = t == kseg ? sum : ys[t];
                    } else if (lane < ROWS && row < n_rows) {
                        const int64_t orow = GATE && row >= out_jump_at ? row + out_jump : row;
                        if constexpr ((ABL & 32) != 0) {
                            __builtin_nontemporal_store(sum, &c[orow * ldc + j]);
                        } else {
                            c[orow * ldc + j] = sum;
                        }
                    }
#ifdef GKOC_LAB_TIMESTAMPS
                    if ((ABL & 16) && lane == 0 && seg + 1 == se) {
                        gkoc_lab_ts[blockIdx.x] = wall_clock64();
                    }
#endif
                    ++seg;
                    if constexpr (DOT) bdot = b_of_row(seg);
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
                produce(vA, cA, rA, produced);
                load_group(vA, cA, rA, p_load);
            } else {
                produce(vB, cB, rB, produced);
                load_group(vB, cB, rB, p_load);
            }
            p_load += G;
            produced += G;
            use_a = !use_a;
        }
        if (DEFER > 0) {
#pragma unroll
            for (int t = 0; t < DEFER; ++t) {
                const int64_t row = (sb + t) * ROWS + lane;
                const int64_t orow = GATE && row >= out_jump_at ? row + out_jump : row;
                if (sb + t < se && lane < ROWS && row < n_rows) c[orow * ldc + j] = ys[t];
            }
        }
    }
    if constexpr (COO) {
        if (__ballot(coo_bad) && lane == 0) *unsorted_flag = 1;
    }
    if (DOT) {
        // fixed butterfly: the partial of a wave does not depend on timing
        dot_acc = wave_sum(dot_acc);
        if (lane == 0) dot_partial[dot_slot] = dot_acc;
    }
}

#endif  // __HIPCC__

}  // namespace gkoc
