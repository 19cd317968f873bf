// Block-Jacobi preconditioner on gfx950: block detection, inversion, apply.
//
// Replaces gko::kernels::hip::jacobi::{find_blocks, generate, simple_apply,
// apply} (decl core/preconditioner/jacobi_kernels.hpp:18-86; semantics
// reference/preconditioner/jacobi_kernels.cpp:47-118 (find_blocks), :125-411
// (extract / Gauss-Jordan / generate), :419-531 (apply); stock GPU versions
// common/cuda_hip/preconditioner/jacobi_kernels.cpp:58-270,
// jacobi_generate_kernels.instantiate.cpp, jacobi_simple_apply_kernels*.cpp).
//
// Storage is Ginkgo's block_interleaved_storage_scheme
// (include/ginkgo/core/preconditioner/jacobi.hpp:37-140): with
// max_block_stride = 64 on HIP a "group" of 2^group_power blocks is one
// 64-wide, column-major panel: lane l = block_offset*(b & mask) + r addresses
// row r of block b, column c sits `stride` elements further.
//  => apply: ONE wavefront per group, lane = (block, row); column c of all
//     blocks of the group is one coalesced 512 B load; b is read through L1
//     (8 lanes share an address); x is one coalesced store.  HBM-bound:
//     (bs + 2) * n values for uniform blocks of size bs.  No MFMA: at nrhs = 1
//     the arithmetic intensity is 0.2 flop/B.
//  => results are bit-identical to the reference (same k order, multiply and
//     add kept separate).
//  * find_blocks is fully parallel (the stock GPU backend runs two <<<1,1>>>
//    kernels over all rows): natural blocks and the greedy agglomeration are
//    both "follow next[] from row 0" chains, marked by pointer doubling in
//    ceil(log2 n) passes.  block_pointers are integer-exact.
//  * generate: one SUB-lane sub-wavefront per block (SUB = max_block_size
//    rounded up to a power of two), block staged in LDS, lane = row; pivoting
//    and operation order as in the reference => bit-identical inverse blocks.
#include "common.hpp"
#include "scan.hpp"
#include "fused.hpp"

namespace gkoc {
namespace {

// ------------------------------------------------------------- find_blocks
template <typename I>
__global__ __launch_bounds__(256) void same_pattern_kernel(
    int64_t n, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    uint8_t* __restrict__ same)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    bool s = false;
    if (i > 0) {
        const int64_t pa = row_ptrs[i - 1], ca = row_ptrs[i], na = row_ptrs[i + 1];
        s = (na - ca) == (ca - pa);
        for (int64_t k = 0; s && k < na - ca; ++k) {
            s = cols[pa + k] == cols[ca + k];
        }
    }
    same[i] = s;
}

// natural block starting at row s ends at min(s + max_bs, first t > s with
// !same[t], n)   (reference find_natural_blocks, :47-78)
template <typename I>
__global__ __launch_bounds__(256) void natural_next_kernel(
    int64_t n, int max_bs, const uint8_t* __restrict__ same,
    I* __restrict__ next)
{
    const int64_t s = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (s > n) return;
    if (s == n) {
        next[n] = I(n);
        return;
    }
    int64_t t = s + 1;
    const int64_t lim = (s + max_bs < n) ? s + max_bs : n;
    while (t < lim && same[t]) ++t;
    next[s] = I(t);
}

// agglomerated block starting at natural boundary s ends at the largest
// natural boundary t with s < t <= s + max_bs (reference
// agglomerate_supervariables, :81-103); boundary = marked row or n
template <typename I>
__global__ __launch_bounds__(256) void agglomerate_next_kernel(
    int64_t n, int max_bs, const uint8_t* __restrict__ nat_start,
    I* __restrict__ next)
{
    const int64_t s = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (s > n) return;
    if (s == n) {
        next[n] = I(n);
        return;
    }
    int64_t t = (s + max_bs < n) ? s + max_bs : n;
    while (t > s + 1 && !(t == n || nat_start[t])) --t;
    next[s] = I(t);
}

template <typename I>
__global__ __launch_bounds__(256) void chain_mark_kernel(
    int64_t n, const I* __restrict__ jump, uint8_t* __restrict__ marked)
{
    const int64_t s = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (s >= n) return;
    if (marked[s]) {
        const int64_t t = jump[s];
        if (t < n) marked[t] = 1;
    }
}

template <typename I>
__global__ __launch_bounds__(256) void chain_double_kernel(
    int64_t n, const I* __restrict__ jump, I* __restrict__ jump2)
{
    const int64_t s = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (s > n) return;
    jump2[s] = s == n ? I(n) : jump[int64_t(jump[s])];
}

// marks every row reachable from row 0 by following next[] (next[s] > s)
template <typename I>
int mark_chain(hipStream_t st, int64_t n, I* next, I* scratch, uint8_t* marked)
{
    GKOC_HIP(hipMemsetAsync(marked, 0, n, st));
    GKOC_HIP(hipMemsetAsync(marked, 1, 1, st));   // row 0 starts the chain
    const dim3 grid(unsigned(ceildiv(n + 1, 256))), block(256);
    I* a = next;
    I* b = scratch;
    for (int64_t reach = 1; reach < n; reach *= 2) {
        chain_mark_kernel<I><<<grid, block, 0, st>>>(n, a, marked);
        GKOC_LAUNCH_OK();
        chain_double_kernel<I><<<grid, block, 0, st>>>(n, a, b);
        GKOC_LAUNCH_OK();
        I* t = a;
        a = b;
        b = t;
    }
    chain_mark_kernel<I><<<grid, block, 0, st>>>(n, a, marked);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename I>
__global__ __launch_bounds__(256) void flags_to_index_kernel(
    int64_t n, const uint8_t* __restrict__ marked, I* __restrict__ pos)
{
    const int64_t s = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (s <= n) pos[s] = s < n ? I(marked[s]) : I(0);
}

template <typename I>
__global__ __launch_bounds__(256) void compact_kernel(
    int64_t n, const uint8_t* __restrict__ marked, const I* __restrict__ pos,
    I* __restrict__ block_ptrs)
{
    const int64_t s = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (s > n) return;
    if (s == n) {
        block_ptrs[int64_t(pos[n])] = I(n);
    } else if (marked[s]) {
        block_ptrs[int64_t(pos[s])] = I(s);
    }
}

template <typename I>
int find_blocks_impl(gkoc_stream_t s, int64_t n, const I* row_ptrs,
                     const I* cols, uint32_t max_bs, int64_t* num_blocks_host,
                     I* block_ptrs)
{
    GKOC_REQUIRE(num_blocks_host && block_ptrs, GKOC_E_INVALID, "null output");
    GKOC_REQUIRE(max_bs >= 1 && max_bs <= 64, GKOC_E_NOT_SUPPORTED,
                 "max_block_size must be in [1, 64]");
    hipStream_t st = as_stream(s);
    if (n == 0) {
        GKOC_HIP(hipMemsetAsync(block_ptrs, 0, sizeof(I), st));
        *num_blocks_host = 0;
        return GKOC_OK;
    }
    uint8_t *same = nullptr, *marked = nullptr;
    I *next = nullptr, *scratch = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&same), n));
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&marked), n));
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&next), sizeof(I) * (n + 1)));
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&scratch), sizeof(I) * (n + 1)));
    const dim3 grid(unsigned(ceildiv(n + 1, 256))), block(256);
    same_pattern_kernel<I><<<grid, block, 0, st>>>(n, row_ptrs, cols, same);
    GKOC_LAUNCH_OK();
    // 1) natural blocks
    natural_next_kernel<I><<<grid, block, 0, st>>>(n, int(max_bs), same, next);
    GKOC_LAUNCH_OK();
    int rc = mark_chain<I>(st, n, next, scratch, marked);
    if (rc != GKOC_OK) return rc;
    // 2) greedy agglomeration over the natural boundaries (`same` is reused to
    //    hold the natural-start flags)
    GKOC_HIP(hipMemcpyAsync(same, marked, n, hipMemcpyDeviceToDevice, st));
    agglomerate_next_kernel<I><<<grid, block, 0, st>>>(n, int(max_bs), same, next);
    GKOC_LAUNCH_OK();
    rc = mark_chain<I>(st, n, next, scratch, marked);
    if (rc != GKOC_OK) return rc;
    // 3) compact the marked rows into block_ptrs
    flags_to_index_kernel<I><<<grid, block, 0, st>>>(n, marked, next);
    GKOC_LAUNCH_OK();
    rc = device_exclusive_scan<I>(st, next, n + 1);
    if (rc != GKOC_OK) return rc;
    compact_kernel<I><<<grid, block, 0, st>>>(n, marked, next, block_ptrs);
    GKOC_LAUNCH_OK();
    I nb = 0;
    GKOC_HIP(hipMemcpyAsync(&nb, next + n, sizeof(I), hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipStreamSynchronize(st));
    *num_blocks_host = int64_t(nb);
    GKOC_TRY(scratch_free(st, same));
    GKOC_TRY(scratch_free(st, marked));
    GKOC_TRY(scratch_free(st, next));
    GKOC_TRY(scratch_free(st, scratch));
    return GKOC_OK;
}

// ---------------------------------------------------------------- generate
template <typename T>
__device__ __forceinline__ T gabs(T v)
{
    return v < T(0) ? -v : v;
}

// In-place Gauss-Jordan inversion with column pivoting of the bs x bs block held
// row-major (leading dimension ld) in LDS, one SUB-lane sub-wavefront per block, lane =
// row r; operation order of reference invert_block (:175-240) => bit-identical.  perm is
// the lane's entry of the row permutation.  Returns false when a pivot was zero (the
// reference stops transforming the block there).  All lanes of the wave must call it
// together (max_bs = the largest block size in the wave).
template <typename T>
__device__ __forceinline__ bool gauss_jordan_lds(T* Bm, int ld, int bs, int r, int g, int sub,
                                                 int max_bs, int& perm)
{
    bool dead = false;
    wave_lds_sync();
    for (int k = 0; k < max_bs; ++k) {
        const bool act = k < bs && !dead;
        // pivot = first row in [k, bs) with the largest |B(i,k)| (a real number for complex T too)
        real_t<T> pa = (act && r >= k && r < bs) ? abs_v(Bm[r * ld + k]) : real_t<T>(-1);
        int pi = r;
        for (int off = 1; off < sub; off <<= 1) {
            const real_t<T> oa = __shfl_xor(pa, off, 64);
            const int oi = __shfl_xor(pi, off, 64);
            if (oa > pa || (oa == pa && oi < pi)) {
                pa = oa;
                pi = oi;
            }
        }
        const int cp = pi;
        // swap rows k and cp (+ their perm entries)
        const int pk = __shfl(perm, g * sub + k, 64);
        const int pc = __shfl(perm, g * sub + cp, 64);
        if (act && cp != k) {
            if (r == k) {
                perm = pc;
                for (int j = 0; j < bs; ++j) {
                    const T t = Bm[k * ld + j];
                    Bm[k * ld + j] = Bm[cp * ld + j];
                    Bm[cp * ld + j] = t;
                }
            }
            if (r == cp) perm = pk;
        }
        wave_lds_sync();
        // Gauss-Jordan transform around (k,k) (reference :175-199)
        const T d = act ? Bm[k * ld + k] : T(1);
        if (act && d == T(0)) dead = true;
        const bool go = act && !dead;
        wave_lds_sync();
        if (go && r < bs) Bm[r * ld + k] = Bm[r * ld + k] / -d;
        wave_lds_sync();
        if (go && r == k) Bm[k * ld + k] = T(0);
        wave_lds_sync();
        if (go && r < bs) {
            const T f = Bm[r * ld + k];
            for (int j = 0; j < bs; ++j) {
                Bm[r * ld + j] = Bm[r * ld + j] + f * Bm[k * ld + j];
            }
        }
        wave_lds_sync();
        if (go && r == k) {
            for (int j = 0; j < bs; ++j) Bm[k * ld + j] = Bm[k * ld + j] / d;
            Bm[k * ld + k] = T(1) / d;
        }
        wave_lds_sync();
    }
    return !dead;
}

// one wave per 64/SUB blocks; dynamic LDS: (64/SUB) * SUB * (SUB+1) values
template <typename T, typename I>
__global__ __launch_bounds__(64) void jacobi_generate_kernel(
    const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, int64_t num_blocks, int sub,
    gkoc_jacobi_scheme scheme, const I* __restrict__ block_ptrs,
    T* __restrict__ blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);
    const int lane = threadIdx.x;
    const int per_wave = 64 / sub;
    const int g = lane / sub;
    const int r = lane % sub;
    const int ld = sub + 1;
    T* Bm = lds + int64_t(g) * sub * ld;
    const int64_t blk = int64_t(blockIdx.x) * per_wave + g;
    int64_t start = 0;
    int bs = 0;
    if (blk < num_blocks) {
        start = block_ptrs[blk];
        bs = int(block_ptrs[blk + 1] - start);
    }
    // extract the dense diagonal block (reference extract_block, :125-147)
    if (r < bs) {
        for (int j = 0; j < bs; ++j) Bm[r * ld + j] = T(0);
        const int64_t a = row_ptrs[start + r], e = row_ptrs[start + r + 1];
        for (int64_t k = a; k < e; ++k) {
            const int64_t c = int64_t(cols[k]) - start;
            if (c >= 0 && c < bs) Bm[r * ld + c] = vals[k];
        }
    }
    int perm = r;
    const int max_bs = wave_max(bs);
    gauss_jordan_lds<T>(Bm, ld, bs, r, g, sub, max_bs, perm);
    // store inverse, column-permuted, into the interleaved scheme
    // (reference permute_and_transpose_block, :242-258)
    const int64_t gsize = int64_t(1) << scheme.group_power;
    const int64_t stride = scheme.block_offset << scheme.group_power;
    const int64_t goff = scheme.group_offset * (blk >> scheme.group_power);
    const int64_t boff = scheme.block_offset * (blk & (gsize - 1));
    for (int j = 0; j < max_bs; ++j) {
        const int pj = __shfl(perm, g * sub + j, 64);
        if (r < bs && j < bs) {
            blocks[goff + boff + r + int64_t(pj) * stride] = Bm[r * ld + j];
        }
    }
}

// entry idx of a storage group held in reduced precision (defined with the storage types below)
__device__ double load_stored_any(int prec, const double* group, int64_t idx);
__device__ inline float load_stored_any(int, const float* group, int64_t idx) { return group[idx]; }

// ------------------------------------------------------------------- apply
// one wave per storage group; lane l < stride: block = l / block_offset,
// row = l % block_offset
template <typename T, typename I, bool ADV, bool STORED = false>
__global__ __launch_bounds__(256) void jacobi_apply_kernel(
    int64_t num_blocks, int64_t num_groups, gkoc_jacobi_scheme scheme,
    const I* __restrict__ block_ptrs, const T* __restrict__ blocks,
    const T* __restrict__ alpha_p, const T* __restrict__ b, int64_t ldb,
    const T* __restrict__ beta_p, T* __restrict__ x, int64_t ldx, int nrhs,
    const uint8_t* __restrict__ precisions = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int64_t group = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (group >= num_groups) return;
    const int64_t bo = scheme.block_offset;
    const int64_t stride = bo << scheme.group_power;
    if (lane >= stride) return;
    const int64_t blk = (group << scheme.group_power) + lane / bo;
    const int r = int(lane % bo);
    if (blk >= num_blocks) return;
    const int64_t start = block_ptrs[blk];
    const int bs = int(block_ptrs[blk + 1] - start);
    if (r >= bs) return;
    const T* gp = blocks + scheme.group_offset * group + lane;
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    for (int j = 0; j < nrhs; ++j) {
        // reference apply_block (:419-448): x = beta*x (or 0), then
        // x += (alpha * B(row,inner)) * b[inner] for inner = 0..bs-1
        T sum = T(0);
        if (ADV && beta != T(0)) sum = x[(start + r) * ldx + j] * beta;
        for (int c = 0; c < bs; ++c) {
            T m;
            if constexpr (STORED) {
                // block stored in the precision of its group (any layout; T = double)
                m = load_stored_any(int(precisions[blk]), blocks + scheme.group_offset * group,
                                    lane + c * stride);
            } else {
                m = gp[c * stride];
            }
            const T bv = b[(start + c) * ldb + j];
            sum += ADV ? (alpha * m) * bv : m * bv;
        }
        x[(start + r) * ldx + j] = sum;
    }
}

// Several right-hand sides on the matrix cores (f64, block_offset 8, 64-wide groups):
// X_blk (8 x nrhs) = Inv_blk (8 x 8) * B_blk (8 x nrhs) is the one GEMM-shaped piece of the Krylov
// hot path.  One wave per storage group (8 blocks).  v_mfma_f64_16x16x4_f64 computes
// D(16x16) += A(16x4) B(4x16); two blocks form a block-diagonal 16 x 16 operand (half of A is
// zero - the flops are free here, the kernel is bound by the b / x traffic), 16 right-hand
// sides are the 16 columns of B and D, four instructions walk the 16 inner indices.  What the
// matrix cores buy is the DATA LAYOUT: a lane of the B / D fragments holds column j of a row, so
// every b / x access is a run of 16 consecutive values of the row-major multi-vector, where the
// lane = row kernel above reads one value per row per instruction with a stride of ldb.
// The group's inverse blocks are read once (coalesced, masked to the block size) into LDS and
// re-read from there in fragment order for every chunk of 16 columns.
// Sums are fused multiply-adds in the unit's order: NOT bit-identical to the reference's
// separate multiply and add (relative difference ~5e-16), which everything else in this file
// is.  Used from four right-hand sides on (GKOC_TUNE_JACOBI_MFMA: 0 never, 1 from two, 2 = default).
// Fragment maps (cdna_hip_programming.md, f64 MFMA): A[i][k]: i = lane & 15, k = lane >> 4;
// B[k][j]: k = lane >> 4, j = lane & 15; D[i][j]: j = lane & 15, i = (lane >> 4) + 4 * reg.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

template <typename I, bool ADV>
__global__ __launch_bounds__(256) void jacobi_apply_mfma_kernel(
    int64_t num_blocks, int64_t num_groups, int64_t group_offset,
    const I* __restrict__ block_ptrs, const double* __restrict__ blocks,
    const double* __restrict__ alpha_p, const double* __restrict__ b, int64_t ldb,
    const double* __restrict__ beta_p, double* __restrict__ x, int64_t ldx, int nrhs)
{
    __shared__ double inv[4][8 * 64];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int64_t group = int64_t(blockIdx.x) * 4 + w;
    if (group >= num_groups) return;   // no block-wide barrier below
    double alpha = 1.0, beta = 0.0;
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    // block pointers of the group's 8 blocks (+ end) in lanes 0..8
    const int64_t blk0 = group << 3;
    int64_t pv = 0;
    {
        const int64_t bi = blk0 + lane < num_blocks ? blk0 + lane : num_blocks;
        if (lane <= 8) pv = block_ptrs[bi];
    }
    // stage the inverse blocks: lane = (block, row), masked to the block's size
    {
        const int bl = lane >> 3, r = lane & 7;
        const int64_t st = __shfl(pv, bl, 64), en = __shfl(pv, bl + 1, 64);
        const int bs = blk0 + bl < num_blocks ? int(en - st) : 0;
        const double* gp = blocks + group_offset * group + lane;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            inv[w][c * 64 + lane] = (r < bs && c < bs) ? gp[c * 64] : 0.0;
        }
    }
    wave_lds_sync();
    const int i16 = lane & 15, q = lane >> 4;
    for (int j0 = 0; j0 < nrhs; j0 += 16) {
        const int j = j0 + i16;
        const bool jok = j < nrhs;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                // A: row i16 of the block pair, inner index k = 4 kk + q
                const int k = 4 * kk + q;
                const int ca = k - 8 * (i16 >> 3);      // column inside the row's own block
                const double av = (ca >= 0 && ca < 8) ? inv[w][ca * 64 + 16 * p + i16] : 0.0;
                // B: inner index k -> block 2p + (k >> 3), row k & 7 of that block
                const int bl = 2 * p + (k >> 3), rb = k & 7;
                const int64_t st = __shfl(pv, bl, 64), en = __shfl(pv, bl + 1, 64);
                const bool rok = blk0 + bl < num_blocks && rb < int(en - st);
                const double bvv = (rok && jok) ? b[(st + rb) * ldb + j] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bvv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = q + 4 * reg;
                const int bl = 2 * p + (i >> 3), rd = i & 7;
                const int64_t st = __shfl(pv, bl, 64), en = __shfl(pv, bl + 1, 64);
                if (blk0 + bl < num_blocks && rd < int(en - st) && jok) {
                    double* xp = x + (st + rd) * ldx + j;
                    double v = acc[reg];
                    if (ADV) v = beta != 0.0 ? alpha * v + beta * xp[0] : alpha * v;
                    xp[0] = v;
                }
            }
        }
    }
}

// Fast path for the layout Ginkgo's compute_storage_scheme produces on a
// 64-wide device: block_offset BO (power of two), BO << group_power == 64,
// single right-hand side.  One wave handles GPW consecutive groups: per group
// all BO matrix columns (coalesced 512 B / 256 B runs) and the lane's own b
// value are requested before the first use; b[start + c] then comes from lane
// (block, c) through ds_bpermute instead of BO more gathers.  Accumulation
// order and rounding are those of the generic kernel (reference apply_block).
// ---- reduced-precision block storage (adaptive block-Jacobi with a fixed
// storage_optimization; include/ginkgo/core/preconditioner/jacobi.hpp, the types of
// core/preconditioner/jacobi_utils.hpp:15-37 for ValueType = double).  The stored
// blocks are widened to double on load and the product runs in double, exactly like
// reference apply_block with its default_converter.  PREC = the precision_reduction
// byte (preserving << 4 | nonpreserving):
//   0x01 float (round to nearest)            0x02 half (via float, round to nearest
//   0x10 upper 32 bits of the double              even; values below the smallest
//   0x11 upper 16 bits of the float               normal half become signed zero,
//   0x20 upper 16 bits of the double              like gko::half, half.hpp:405-433)
// A group keeps its place (group_offset is in doubles); its reduced entries sit at
// the same element index of the narrower type, i.e. in the first part of the group.
template <int PREC>
struct stored;
template <>
struct stored<0x01> {
    using type = float;
    __device__ static double load(type v) { return double(v); }
    __device__ static type store(double v) { return float(v); }
};
template <>
struct stored<0x10> {
    using type = uint32_t;
    __device__ static double load(type v) { return __longlong_as_double((long long)(uint64_t(v) << 32)); }
    __device__ static type store(double v) { return uint32_t(uint64_t(__double_as_longlong(v)) >> 32); }
};
template <>
struct stored<0x20> {
    using type = uint16_t;
    __device__ static double load(type v) { return __longlong_as_double((long long)(uint64_t(v) << 48)); }
    __device__ static type store(double v) { return uint16_t(uint64_t(__double_as_longlong(v)) >> 48); }
};
template <>
struct stored<0x11> {
    using type = uint16_t;
    __device__ static double load(type v) { return double(__uint_as_float(uint32_t(v) << 16)); }
    __device__ static type store(double v) { return uint16_t(__float_as_uint(float(v)) >> 16); }
};
template <>
struct stored<0x02> {
    using type = uint16_t;
    __device__ static double load(type h)
    {
        // the hardware conversion, except that subnormal halves read as signed zero
        // (gko::half does not decode them, half.hpp:444-446)
        _Float16 hv;
        __builtin_memcpy(&hv, &h, 2);
        const float f = (h & 0x7c00u) == 0 ? __uint_as_float(uint32_t(h & 0x8000u) << 16) : float(hv);
        return double(f);
    }
    __device__ static type store(double v)
    {
        const uint32_t f = __float_as_uint(float(v));
        const uint16_t sign = uint16_t((f >> 16) & 0x8000u);
        const uint32_t e = (f >> 23) & 0xffu, m = f & 0x007fffffu;
        if (e == 0xffu) return uint16_t(sign | 0x7c00u | (m ? 0x03ffu : 0u));
        if (e <= 112u) return sign;                      // below the normal half range
        if (e - 112u >= 31u) return uint16_t(sign | 0x7c00u);
        const uint16_t res = uint16_t(sign | ((e - 112u) << 10) | (m >> 13));
        const uint32_t tail = m & 0x1fffu;
        return uint16_t(res + ((tail > 0x1000u || (tail == 0x1000u && (res & 1u))) ? 1u : 0u));
    }
};

// runtime-selected storage type (one precision per storage group)
__device__ __forceinline__ double load_stored(int prec, const double* group, int64_t idx)
{
    switch (prec) {
    case 0x01: return stored<0x01>::load(reinterpret_cast<const float*>(group)[idx]);
    case 0x02: return stored<0x02>::load(reinterpret_cast<const uint16_t*>(group)[idx]);
    case 0x10: return stored<0x10>::load(reinterpret_cast<const uint32_t*>(group)[idx]);
    case 0x11: return stored<0x11>::load(reinterpret_cast<const uint16_t*>(group)[idx]);
    case 0x20: return stored<0x20>::load(reinterpret_cast<const uint16_t*>(group)[idx]);
    default: return group[idx];
    }
}

__device__ double load_stored_any(int prec, const double* group, int64_t idx)
{
    return load_stored(prec, group, idx);
}

__device__ __forceinline__ void store_stored(int prec, double* group, int64_t idx, double v)
{
    switch (prec) {
    case 0x01: reinterpret_cast<float*>(group)[idx] = stored<0x01>::store(v); break;
    case 0x02: reinterpret_cast<uint16_t*>(group)[idx] = stored<0x02>::store(v); break;
    case 0x10: reinterpret_cast<uint32_t*>(group)[idx] = stored<0x10>::store(v); break;
    case 0x11: reinterpret_cast<uint16_t*>(group)[idx] = stored<0x11>::store(v); break;
    case 0x20: reinterpret_cast<uint16_t*>(group)[idx] = stored<0x20>::store(v); break;
    default: group[idx] = v; break;
    }
}

// reference compute_inf_norm (reference/components/matrix_operations.hpp:22-37) applied to
// the ROW-major block as the reference applies it (element i + j * bs = row j, column i):
// lane r sums |B(j, r)| over j in order, the maximum over the lanes is order-free
__device__ __forceinline__ double block_norm_lds(const double* Bm, int ld, int bs, int r, int sub)
{
    double t = 0.0;
    if (r < bs) {
        for (int j = 0; j < bs; ++j) t += fabs(Bm[j * ld + r]);
    }
    for (int off = 1; off < sub; off <<= 1) {
        const double o = __shfl_xor(t, off, 64);
        t = o > t ? o : t;
    }
    return t;
}

// validate_precision_reduction_feasibility<ReducedType> (reference :280-307): round the
// inverse to the reduced type, invert that in double, condition number must be >= 1 and
// * eps(double) < 1e-3.  PREC 0x01 = float, 0x02 = half.  Tm: scratch block in LDS.
template <int PREC>
__device__ __forceinline__ bool feasible_lds(const double* Bm, double* Tm, int ld, int bs, int r,
                                             int g, int sub, int max_bs)
{
    if (r < bs) {
        for (int j = 0; j < bs; ++j) {
            Tm[r * ld + j] = stored<PREC>::load(stored<PREC>::store(Bm[r * ld + j]));
        }
    }
    wave_lds_sync();
    double cond = block_norm_lds(Tm, ld, bs, r, sub);
    int perm = r;
    const bool ok = gauss_jordan_lds<double>(Tm, ld, bs, r, g, sub, max_bs, perm);
    cond *= block_norm_lds(Tm, ld, bs, r, sub);
    wave_lds_sync();
    return ok && cond >= 1.0 && cond * (1.0 / 9007199254740992.0) < 1e-3;
}

// adaptive generate (reference/preconditioner/jacobi_kernels.cpp:313-411), value type
// double, one wave per storage group (64-wide groups: 64/SUB blocks).
// precisions[blk] in: requested precision_reduction byte (0xff = autodetect); out: the
// precision of the group = get_optimal_storage_reduction of the AND of the blocks'
// descriptors (core/preconditioner/jacobi_utils.hpp:104-176).  conditioning[blk] =
// norm(block) * norm(inverse).  dynamic LDS: 2 * (64/SUB) * SUB * (SUB+1) doubles.
template <typename I>
__global__ __launch_bounds__(64) void jacobi_generate_adaptive_kernel(
    const I* __restrict__ row_ptrs, const I* __restrict__ cols, const double* __restrict__ vals,
    int64_t num_blocks, int sub, gkoc_jacobi_scheme scheme, const I* __restrict__ block_ptrs,
    double accuracy, uint8_t* __restrict__ precisions, double* __restrict__ conditioning,
    double* __restrict__ blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* lds = reinterpret_cast<double*>(lds_raw);
    const int lane = threadIdx.x;
    const int per_wave = 64 / sub;
    const int g = lane / sub;
    const int r = lane % sub;
    const int ld = sub + 1;
    double* Bm = lds + int64_t(g) * sub * ld;
    double* Tm = lds + int64_t(per_wave) * sub * ld + int64_t(g) * sub * ld;
    const int64_t blk = int64_t(blockIdx.x) * per_wave + g;
    int64_t start = 0;
    int bs = 0;
    if (blk < num_blocks) {
        start = block_ptrs[blk];
        bs = int(block_ptrs[blk + 1] - start);
    }
    if (r < bs) {
        for (int j = 0; j < bs; ++j) Bm[r * ld + j] = 0.0;
        const int64_t a = row_ptrs[start + r], e = row_ptrs[start + r + 1];
        for (int64_t k = a; k < e; ++k) {
            const int64_t c = int64_t(cols[k]) - start;
            if (c >= 0 && c < bs) Bm[r * ld + c] = vals[k];
        }
    }
    const int max_bs = wave_max(bs);
    wave_lds_sync();
    double cond = block_norm_lds(Bm, ld, bs, r, sub);
    int perm = r;
    gauss_jordan_lds<double>(Bm, ld, bs, r, g, sub, max_bs, perm);
    cond *= block_norm_lds(Bm, ld, bs, r, sub);
    const int request = blk < num_blocks ? int(precisions[blk]) : -1;
    uint32_t desc = 0xffffffffu;           // blocks past the end do not restrict the group
    const bool any_auto = __ballot(request == 0xff) != 0;
    bool v1 = false, v2 = false;
    if (any_auto) {                        // wave-uniform: the inversions run in lock step
        v1 = feasible_lds<0x01>(Bm, Tm, ld, bs, r, g, sub, max_bs);
        v2 = feasible_lds<0x02>(Bm, Tm, ld, bs, r, g, sub, max_bs);
    }
    if (request == 0xff) {
        // get_supported_storage_reductions: eps of truncated<double,4>, truncated<float,2>,
        // half, truncated<double,2>, float; verificators are pure, so evaluating them
        // eagerly and replaying the short-circuit logic gives the same set
        int verified1 = 2;
        desc = 0;
        if (cond * (1.0 / 16) < accuracy) desc |= 0x04;
        if (cond * (1.0 / 128) < accuracy) {
            verified1 = v1 ? 1 : 0;
            if (v1) desc |= 0x02;
        }
        if (cond * (1.0 / 2048) < accuracy && verified1 != 0 && v2) desc |= 0x01;
        if (cond * (1.0 / 1048576) < accuracy) desc |= 0x10;
        if (cond * (1.0 / 16777216) < accuracy) {
            if (verified1 == 2) verified1 = v1 ? 1 : 0;
            if (verified1 == 1) desc |= 0x08;
        }
    } else if (request >= 0) {
        desc = request == 0x01 ? 0x08u : request == 0x02 ? 0x01u : request == 0x10 ? 0x10u
             : request == 0x11 ? 0x02u : request == 0x20 ? 0x04u : 0u;
    }
    for (int off = 1; off < 64; off <<= 1) desc &= __shfl_xor(desc, off, 64);
    const int p = (desc & 0x01) ? 0x02 : (desc & 0x02) ? 0x11 : (desc & 0x04) ? 0x20
                : (desc & 0x08) ? 0x01 : (desc & 0x10) ? 0x10 : 0x00;
    if (blk < num_blocks && r == 0) {
        precisions[blk] = uint8_t(p);
        if (conditioning) conditioning[blk] = cond;
    }
    const int64_t gsize = int64_t(1) << scheme.group_power;
    const int64_t stride = scheme.block_offset << scheme.group_power;
    double* group = blocks + scheme.group_offset * (blk >> scheme.group_power);
    const int64_t boff = scheme.block_offset * (blk & (gsize - 1));
    for (int j = 0; j < max_bs; ++j) {
        const int pj = __shfl(perm, g * sub + j, 64);
        if (r < bs && j < bs) store_stored(p, group, boff + r + int64_t(pj) * stride, Bm[r * ld + j]);
    }
}

// in place: the group's double entries become PREC entries at the same element index
template <int PREC, int BO>
__global__ __launch_bounds__(64) void jacobi_convert_storage_kernel(int64_t num_groups,
                                                                    int64_t group_offset,
                                                                    double* blocks)
{
    using S = typename stored<PREC>::type;
    const int64_t group = blockIdx.x;
    if (group >= num_groups) return;
    double* gp = blocks + group_offset * group;
    double v[BO];
#pragma unroll
    for (int c = 0; c < BO; ++c) v[c] = gp[c * 64 + threadIdx.x];
    // every lane has its loads back before any narrow entry is written
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    S* sp = reinterpret_cast<S*>(gp);
#pragma unroll
    for (int c = 0; c < BO; ++c) sp[c * 64 + threadIdx.x] = stored<PREC>::store(v[c]);
}

template <typename T, typename I, bool ADV, int BO, int GPW, bool DOT = false, int PREC = 0>
__global__ __launch_bounds__(256) void jacobi_apply_fixed_kernel(
    int64_t num_blocks, int64_t num_groups, int64_t group_offset,
    const I* __restrict__ block_ptrs, const T* __restrict__ blocks,
    const T* __restrict__ alpha_p, const T* __restrict__ b,
    const T* __restrict__ beta_p, T* __restrict__ x,
    T* __restrict__ dot_partial = nullptr, const uint8_t* __restrict__ precs = nullptr,
    int xcd_map = 0)
{
    __shared__ T dot_lds[4];
    T dot_acc = T(0);
    constexpr int LOG_BO = BO == 1 ? 0 : BO == 2 ? 1 : BO == 4 ? 2 : BO == 8 ? 3
                         : BO == 16 ? 4 : BO == 32 ? 5 : 6;
    constexpr int GP = 6 - LOG_BO;  // group_power
    const int lane = threadIdx.x & 63;
    const int r = lane & (BO - 1);
    const int lane0 = lane - r;
    int64_t wg = blockIdx.x;
    if (xcd_map) {
        // workgroup b runs on XCD b % 8: one contiguous eighth of the groups per
        // XCD (a CU then translates addresses of its own eighth only); bijective
        const int64_t nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7;
        const int64_t xcd = wg & 7, slot = wg >> 3;
        wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
    }
    const int64_t group0 = (wg * 4 + (threadIdx.x >> 6)) * GPW;
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    T m[GPW][BO];
    T bv[GPW];
    T xv[GPW];
    int64_t row[GPW];
    int bs[GPW];
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int64_t group = group0 + g;
        const int64_t blk = (group << GP) + (lane >> LOG_BO);
        const bool have = group < num_groups && blk < num_blocks;
        I start = 0, end = 0;
        if (have) {
            start = block_ptrs[blk];
            end = block_ptrs[blk + 1];
        }
        bs[g] = have && r < int(end - start) ? int(end - start) : 0;
        row[g] = int64_t(start) + r;
    }
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const T* gp = blocks + group_offset * (group0 + g) + lane;
        bv[g] = T(0);
        xv[g] = T(0);
        if (bs[g] > 0) {
            bv[g] = b[row[g]];
            if (ADV) xv[g] = x[row[g]];
            if constexpr (PREC == 0) {
#pragma unroll
                for (int c = 0; c < BO; ++c) m[g][c] = gp[c * 64];
            } else if constexpr (PREC == -1) {
                // one precision per group (adaptive block-Jacobi)
                const int64_t first = (group0 + g) << GP;
                const int pg = int(precs[first < num_blocks ? first : 0]);
                const double* gd = reinterpret_cast<const double*>(blocks + group_offset * (group0 + g));
                // the type is a property of the group: decide once, then BO typed loads
#define GKOC_LOAD_GROUP(P_)                                                             \
    {                                                                                   \
        const typename stored<P_>::type* sp =                                           \
            reinterpret_cast<const typename stored<P_>::type*>(gd) + lane;              \
        typename stored<P_>::type raw[BO];                                              \
        _Pragma("unroll") for (int c = 0; c < BO; ++c) raw[c] = sp[c * 64];             \
        _Pragma("unroll") for (int c = 0; c < BO; ++c) m[g][c] = T(stored<P_>::load(raw[c])); \
    }
                switch (pg) {
                case 0x01: GKOC_LOAD_GROUP(0x01) break;
                case 0x02: GKOC_LOAD_GROUP(0x02) break;
                case 0x10: GKOC_LOAD_GROUP(0x10) break;
                case 0x11: GKOC_LOAD_GROUP(0x11) break;
                case 0x20: GKOC_LOAD_GROUP(0x20) break;
                default:
#pragma unroll
                    for (int c = 0; c < BO; ++c) m[g][c] = T(gd[lane + c * 64]);
                    break;
                }
#undef GKOC_LOAD_GROUP
            } else {
                using S = typename stored<PREC>::type;
                const S* sp = reinterpret_cast<const S*>(blocks + group_offset * (group0 + g)) + lane;
#pragma unroll
                for (int c = 0; c < BO; ++c) m[g][c] = T(stored<PREC>::load(sp[c * 64]));
            }
        } else {
#pragma unroll
            for (int c = 0; c < BO; ++c) m[g][c] = T(0);
        }
    }
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        T sum = T(0);
        if (ADV && beta != T(0)) sum = xv[g] * beta;
#pragma unroll
        for (int c = 0; c < BO; ++c) {
            const T bc = __shfl(bv[g], lane0 + c, 64);
            const T t = ADV ? (alpha * m[g][c]) * bc : m[g][c] * bc;
            sum = c < bs[g] ? sum + t : sum;
        }
        if (bs[g] > 0) x[row[g]] = sum;
        if (DOT && bs[g] > 0) dot_acc += bv[g] * sum;   // <b, x> for this row
    }
    if (DOT) {
        const T r = block_sum<256>(dot_acc, dot_lds);
        if (threadIdx.x == 0) dot_partial[wg] = r;
    }
}

// Several right-hand sides, fast-path layout (one 64-lane group per wave): the single-column
// kernel's scheme - lane = row of its block, the block's rows in BO registers from coalesced loads,
// the block's b values by wave shuffle - with the matrix registers reused for every column: a lane
// loads ITS row's values of a chunk of four columns (one or two 16 B loads: neighbouring lanes,
// neighbouring rows - the wave's b tile is one contiguous piece when ldb == nrhs), the 8 x 4
// products per row come from shuffles, and the row's four results leave as one 32 B piece.  The
// general kernel (jacobi_apply_kernel) re-reads the block per column and gathers b with a stride of
// ldb values per lane; the matrix-core kernel reads b / x in 16-column fragments that are mostly
// empty below 16 columns.  Same operations per (row, column) as the single-column kernel: separate
// multiply and add in column order of the block => bit-identical to it and to the reference.
template <typename T, typename I, bool ADV, int BO, int PC>
__global__ __launch_bounds__(256) void jacobi_apply_fixed_multi_kernel(
    int64_t num_blocks, int64_t num_groups, int64_t group_offset, const I* __restrict__ block_ptrs,
    const T* __restrict__ blocks, const T* __restrict__ alpha_p, const T* __restrict__ b, int64_t ldb,
    const T* __restrict__ beta_p, T* __restrict__ x, int64_t ldx, int nrhs, int pairs_ok)
{
    static_assert(PC == 4 || PC == 8, "four or eight columns per pass");
    constexpr int LOG_BO = BO == 1 ? 0 : BO == 2 ? 1 : BO == 4 ? 2 : BO == 8 ? 3 : BO == 16 ? 4 : BO == 32 ? 5 : 6;
    constexpr int GP = 6 - LOG_BO;  // group_power
    struct alignas(2 * sizeof(T)) BV {
        T v[2];
    };
    const int lane = threadIdx.x & 63;
    const int r = lane & (BO - 1);
    const int lane0 = lane - r;
    const int64_t group = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (group >= num_groups) return;
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    const int64_t blk = (group << GP) + (lane >> LOG_BO);
    I start = 0, end = 0;
    if (blk < num_blocks) {
        start = block_ptrs[blk];
        end = block_ptrs[blk + 1];
    }
    const int bs = blk < num_blocks && r < int(end - start) ? int(end - start) : 0;
    const int64_t row = int64_t(start) + r;
    T m[BO];
    const T* gp = blocks + group_offset * group + lane;
#pragma unroll
    for (int c = 0; c < BO; ++c) m[c] = bs > 0 ? gp[c * 64] : T(0);
    // the lane's own row, columns [j0, j0 + PC): values of b, and beta x as the start of the sums
    auto load = [&](int j0, T(&bv)[PC], T(&sum)[PC]) {
        const int nc = nrhs - j0 < PC ? nrhs - j0 : PC;
#pragma unroll
        for (int q = 0; q < PC; ++q) {
            bv[q] = T(0);
            sum[q] = T(0);
        }
        if (bs > 0 && nc > 0) {
            const T* bp = b + row * ldb + j0;
            const T* xp = x + row * ldx + j0;
            if (pairs_ok && nc == PC) {
#pragma unroll
                for (int q = 0; q < PC; q += 2) {
                    const BV p = *reinterpret_cast<const BV*>(bp + q);
                    bv[q] = p.v[0];
                    bv[q + 1] = p.v[1];
                    if (ADV && beta != T(0)) {
                        const BV o = *reinterpret_cast<const BV*>(xp + q);
                        sum[q] = o.v[0] * beta;
                        sum[q + 1] = o.v[1] * beta;
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < PC; ++q) {
                    if (q < nc) {
                        bv[q] = bp[q];
                        if (ADV && beta != T(0)) sum[q] = xp[q] * beta;
                    }
                }
            }
        }
    };
    T bv[PC], sum[PC], bvn[PC], sumn[PC];
    load(0, bv, sum);
    for (int j0 = 0; j0 < nrhs; j0 += PC) {
        const int nc = nrhs - j0 < PC ? nrhs - j0 : PC;
        // the next pass's loads travel while this pass's products are formed.  (When b and x are
        // the same array the next pass reads columns this pass does not write.)
        load(j0 + PC, bvn, sumn);
#pragma unroll
        for (int c = 0; c < BO; ++c) {
#pragma unroll
            for (int q = 0; q < PC; ++q) {
                const T bc = __shfl(bv[q], lane0 + c, 64);
                const T t = ADV ? (alpha * m[c]) * bc : m[c] * bc;
                sum[q] = c < bs ? sum[q] + t : sum[q];
            }
        }
        if (bs > 0) {
            T* xo = x + row * ldx + j0;
            if (pairs_ok && nc == PC) {
#pragma unroll
                for (int q = 0; q < PC; q += 2) {
                    BV o;
                    o.v[0] = sum[q];
                    o.v[1] = sum[q + 1];
                    *reinterpret_cast<BV*>(xo + q) = o;
                }
            } else {
#pragma unroll
                for (int q = 0; q < PC; ++q) {
                    if (q < nc) xo[q] = sum[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PC; ++q) {
            bv[q] = bvn[q];
            sum[q] = sumn[q];
        }
    }
}

// cg::step_2 fused with the preconditioner application that follows it in the next iteration
// (cg.cpp:167-171, then :133-136; one column, unit strides, fast-path layout):
//   t = rho / beta ;  x += t p ;  r -= t q ;  z = M r ;  partials of <r, z> and <r, r>.
// The new residual goes from the registers that computed it straight into the block
// product - r is not read again, and one launch (plus one fold) disappears from the iteration.
// x, r, z are bit-identical to step_2 followed by simple_apply (same operations per element:
// separate multiply and add, the block row sum in column order).  A stopped column leaves x
// and r alone and recomputes the same z, as the two separate kernels do.
template <typename T, typename I, int BO, int GPW>
__global__ __launch_bounds__(256) void jacobi_step2_apply_kernel(
    int64_t num_blocks, int64_t num_groups, int64_t group_offset,
    const I* __restrict__ block_ptrs, const T* __restrict__ blocks, T* __restrict__ x,
    T* __restrict__ r, const T* __restrict__ p, const T* __restrict__ q,
    const T* __restrict__ beta_p, const T* __restrict__ rho_p, const uint8_t* __restrict__ stop,
    T* __restrict__ z, T* __restrict__ partial, int64_t pstride)
{
    __shared__ T lds[4];
    constexpr int LOG_BO = BO == 1 ? 0 : BO == 2 ? 1 : BO == 4 ? 2 : BO == 8 ? 3 : 4;
    constexpr int GP = 6 - LOG_BO;
    const int lane = threadIdx.x & 63;
    const int rr = lane & (BO - 1);
    const int lane0 = lane - rr;
    const int64_t wg = blockIdx.x;
    const int64_t group0 = (wg * 4 + (threadIdx.x >> 6)) * GPW;
    const T bt = beta_p[0];
    const bool noop = bt == T(0) || status_has_stopped(stop[0]);
    const T tmp = noop ? T(0) : rho_p[0] / bt;
    T m[GPW][BO];
    T rv[GPW], xv[GPW], pv[GPW], qv[GPW];
    int64_t row[GPW];
    int bs[GPW];
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int64_t group = group0 + g;
        const int64_t blk = (group << GP) + (lane >> LOG_BO);
        const bool have = group < num_groups && blk < num_blocks;
        I start = 0, end = 0;
        if (have) {
            start = block_ptrs[blk];
            end = block_ptrs[blk + 1];
        }
        bs[g] = have && rr < int(end - start) ? int(end - start) : 0;
        row[g] = int64_t(start) + rr;
    }
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const T* gp = blocks + group_offset * (group0 + g) + lane;
        rv[g] = xv[g] = pv[g] = qv[g] = T(0);
        if (bs[g] > 0) {
            rv[g] = r[row[g]];
            if (!noop) {
                xv[g] = x[row[g]];
                pv[g] = p[row[g]];
                qv[g] = q[row[g]];
            }
#pragma unroll
            for (int c = 0; c < BO; ++c) m[g][c] = gp[c * 64];
        } else {
#pragma unroll
            for (int c = 0; c < BO; ++c) m[g][c] = T(0);
        }
    }
    T dot_acc = T(0), nrm_acc = T(0);
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        if (!noop && bs[g] > 0) {
            const T tx = tmp * pv[g];
            const T tr = tmp * qv[g];
            xv[g] = xv[g] + tx;
            rv[g] = rv[g] - tr;
            x[row[g]] = xv[g];
            r[row[g]] = rv[g];
        }
        T sum = T(0);
#pragma unroll
        for (int c = 0; c < BO; ++c) {
            const T bc = __shfl(rv[g], lane0 + c, 64);
            const T t = m[g][c] * bc;
            sum = c < bs[g] ? sum + t : sum;
        }
        if (bs[g] > 0) {
            z[row[g]] = sum;
            dot_acc += rv[g] * sum;
            nrm_acc += rv[g] * rv[g];
        }
    }
    const T s0 = block_sum<256>(dot_acc, lds);
    __syncthreads();
    const T s1 = block_sum<256>(nrm_acc, lds);
    if (threadIdx.x == 0) {
        partial[wg] = s0;
        partial[pstride + wg] = s1;
    }
}

// PipeCg: pipe_cg::step_2 of iteration k, step_1 of iteration k + 1 (krylov_steps.hip,
// pipe_cg_step2_step1_dots_kernel: the same expressions in the same order, so p, q, f, g, x, r, z, w
// are bit-identical to the two kernels), the three partial sums <r,z>, <w,z>, <r,r> AND the
// preconditioner application m = M w of iteration k + 1 (pipe_cg.cpp:211), with the new w going
// from the registers that computed it into the block product (neighbours' values by shuffle), as
// in jacobi_step2_apply_kernel.  Lane = (block, row) of the 64-wide storage group.  m is read
// (step_2: f = m + t2 f) and then overwritten by the lane that owns the row.
template <typename T, typename I, int BO, int GPW>
__global__ __launch_bounds__(256) void jacobi_pipe_steps_kernel(
    int64_t num_blocks, int64_t num_groups, int64_t group_offset,
    const I* __restrict__ block_ptrs, const T* __restrict__ blocks, T* __restrict__ x,
    T* __restrict__ r, T* __restrict__ z, T* __restrict__ w, T* __restrict__ p, T* __restrict__ q,
    T* __restrict__ f, T* __restrict__ g, T* __restrict__ m, const T* __restrict__ nv,
    const T* __restrict__ prev_rho, const T* __restrict__ rho, const T* __restrict__ delta,
    const T* __restrict__ beta_in, T* __restrict__ beta_out, uint8_t* stop,
    T* __restrict__ partial, int64_t pstride, step_gate_dev<T> gate)
{
    __shared__ T lds[4];
    constexpr int LOG_BO = BO == 1 ? 0 : BO == 2 ? 1 : BO == 4 ? 2 : BO == 8 ? 3 : 4;
    constexpr int GP = 6 - LOG_BO;
    const int lane = threadIdx.x & 63;
    const int rr = lane & (BO - 1);
    const int lane0 = lane - rr;
    const int64_t wg = blockIdx.x;
    const int64_t group0 = (wg * 4 + (threadIdx.x >> 6)) * GPW;
    const bool stopped = step_gate_enter(gate, stop, wg == 0 && threadIdx.x == 0);
    const T pr = prev_rho[0];
    const bool plain = pr == T(0);
    const T t2 = plain ? T(0) : rho[0] / pr;
    T bnew = beta_in[0];
    if (!stopped) {
        if (plain) {
            bnew = delta[0];
        } else {
            const T a = fabs(t2);
            bnew = delta[0] - a * a * beta_in[0];
            if (bnew == T(0)) bnew = delta[0];
        }
    }
    const bool noop1 = bnew == T(0) || stopped;
    const T t1 = noop1 ? T(0) : rho[0] / bnew;
    if (wg == 0 && threadIdx.x == 0) beta_out[0] = bnew;
    T a_rz = T(0), a_wz = T(0), a_rr = T(0);
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
        const int64_t group = group0 + gi;
        const int64_t blk = (group << GP) + (lane >> LOG_BO);
        const bool have = group < num_groups && blk < num_blocks;
        I start = 0, end = 0;
        if (have) {
            start = block_ptrs[blk];
            end = block_ptrs[blk + 1];
        }
        const int bs = have && rr < int(end - start) ? int(end - start) : 0;
        const int64_t row = int64_t(start) + rr;
        T mc[BO];
        T rv = T(0), zv = T(0), wv = T(0);
        if (bs > 0) {
            const T* gp = blocks + group_offset * group + lane;
#pragma unroll
            for (int c = 0; c < BO; ++c) mc[c] = gp[c * 64];
            rv = r[row];
            zv = z[row];
            wv = w[row];
            if (!stopped) {
                const T pv = plain ? zv : zv + t2 * p[row];
                const T qv = plain ? wv : wv + t2 * q[row];
                const T fv = plain ? m[row] : m[row] + t2 * f[row];
                const T gv = plain ? nv[row] : nv[row] + t2 * g[row];
                p[row] = pv;
                q[row] = qv;
                f[row] = fv;
                g[row] = gv;
                if (!noop1) {
                    x[row] = x[row] + t1 * pv;
                    rv = rv - t1 * qv;
                    zv = zv - t1 * fv;
                    wv = wv - t1 * gv;
                    r[row] = rv;
                    z[row] = zv;
                    w[row] = wv;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < BO; ++c) mc[c] = T(0);
        }
        // m = M w on the new w: column c of the block times w of the block's row c
        T sum = T(0);
#pragma unroll
        for (int c = 0; c < BO; ++c) {
            const T bc = __shfl(wv, lane0 + c, 64);
            const T t = mc[c] * bc;
            sum = c < bs ? sum + t : sum;
        }
        if (bs > 0) {
            m[row] = sum;
            a_rz += rv * zv;
            a_wz += wv * zv;
            a_rr += rv * rv;
        }
    }
    const T s0 = block_sum<256>(a_rz, lds);
    __syncthreads();
    const T s1 = block_sum<256>(a_wz, lds);
    __syncthreads();
    const T s2 = block_sum<256>(a_rr, lds);
    if (threadIdx.x == 0) {
        partial[wg] = s0;
        partial[pstride + wg] = s1;
        partial[2 * pstride + wg] = s2;
    }
}

template <typename T, typename I>
int launch_pipe_steps(gkoc_stream_t s, int64_t num_blocks, int64_t n_rows, uint32_t max_bs,
                      gkoc_jacobi_scheme scheme, const I* block_ptrs, const T* blocks, T* x, T* r, T* z,
                      T* w, T* p, T* q, T* f, T* g, T* m, const T* nv, const T* prev_rho, const T* rho,
                      const T* delta, const T* beta_in, T* beta_out, const uint8_t* stop, T* out3, void* work,
                      size_t work_bytes, const gkoc_step_gate* gate_in)
{
    GKOC_REQUIRE(out3 && num_blocks > 0 && n_rows > 0, GKOC_E_INVALID, "bad argument");
    GKOC_REQUIRE(!gate_in || !gate_in->tau || (gate_in->orig_tau && gate_in->flags), GKOC_E_INVALID,
                 "gkoc_step_gate: a criterion needs orig_tau and flags");
    const step_gate_dev<T> gate = step_gate_of<T>(gate_in);
    GKOC_REQUIRE(block_ptrs && blocks && x && r && z && w && p && q && f && g && m && nv && prev_rho && rho &&
                     delta && stop && work,
                 GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(beta_in && beta_out && beta_in != beta_out, GKOC_E_INVALID,
                 "beta_in and beta_out must be two different scalars");
    const int64_t bo = scheme.block_offset;
    GKOC_REQUIRE(bo >= 1 && bo <= 16 && (bo << scheme.group_power) == 64 && (bo & (bo - 1)) == 0,
                 GKOC_E_NOT_SUPPORTED, "needs block_offset in {1,2,4,8,16} and a 64-wide group");
    GKOC_REQUIRE(max_bs <= uint64_t(bo), GKOC_E_INVALID, "max_block_size exceeds block_offset");
    GKOC_REQUIRE(work_bytes >= fused_workspace_bytes(n_rows, sizeof(T)), GKOC_E_WORKSPACE,
                 "workspace too small (gkoc_x_workspace_bytes)");
    const int64_t groups = ceildiv(num_blocks, int64_t(1) << scheme.group_power);
    const int gpw = (bo * int64_t(sizeof(T)) >= 128) ? 1 : 2;
    const int64_t nb = ceildiv(groups, 4 * gpw);
    const int64_t total = int64_t(fused_workspace_bytes(n_rows, sizeof(T)) / sizeof(T));
    GKOC_REQUIRE(3 * nb <= total, GKOC_E_WORKSPACE, "too many blocks for the workspace");
    T* partial = static_cast<T*>(work);
    const int64_t go = scheme.group_offset;
#define GKOC_JAC_PS(BO_)                                                                              \
    jacobi_pipe_steps_kernel<T, I, BO_, ((BO_ * sizeof(T) >= 128) ? 1 : 2)>                            \
        <<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(num_blocks, groups, go, block_ptrs, blocks, \
                                                             x, r, z, w, p, q, f, g, m, nv, prev_rho,  \
                                                             rho, delta, beta_in, beta_out,            \
                                                             const_cast<uint8_t*>(stop), partial, nb,  \
                                                             gate)
    switch (int(bo)) {
    case 1: GKOC_JAC_PS(1); break;
    case 2: GKOC_JAC_PS(2); break;
    case 4: GKOC_JAC_PS(4); break;
    case 8: GKOC_JAC_PS(8); break;
    default: GKOC_JAC_PS(16); break;
    }
#undef GKOC_JAC_PS
    GKOC_LAUNCH_OK();
    fold_rows_kernel<T><<<dim3(3), dim3(1024), 0, as_stream(s)>>>(nb, nb, partial, out3);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// does the fused step_2 + apply have room for its two rows of partial sums in the workspace of
// gkoc_x_workspace_bytes(n_rows, value_size)?  (blocks much smaller than max_block_size: no)
inline bool step2_apply_fits(int64_t num_blocks, int64_t n_rows, gkoc_jacobi_scheme scheme,
                             size_t value_size)
{
    const int64_t bo = scheme.block_offset;
    if (!(bo >= 1 && bo <= 16 && (bo << scheme.group_power) == 64 && (bo & (bo - 1)) == 0)) return false;
    if (num_blocks <= 0 || n_rows <= 0) return false;
    const int64_t groups = ceildiv(num_blocks, int64_t(1) << scheme.group_power);
    const int gpw = (bo * int64_t(value_size) >= 128) ? 1 : 2;
    const int64_t nb = ceildiv(groups, 4 * gpw);
    const int64_t total = int64_t(fused_workspace_bytes(n_rows, value_size) / value_size);
    const int64_t pstride = (total - 2 * fold_chunks) / 2;
    return nb <= pstride;
}

template <typename T, typename I>
int launch_step2_apply(gkoc_stream_t s, int64_t num_blocks, int64_t n_rows, uint32_t max_bs,
                       gkoc_jacobi_scheme scheme, const I* block_ptrs, const T* blocks, T* x, T* r,
                       const T* p, const T* q, const T* beta, const T* rho, const uint8_t* stop,
                       T* z, T* rho_out, T* norm_out, int take_sqrt, void* work, size_t work_bytes)
{
    GKOC_REQUIRE(rho_out && norm_out, GKOC_E_INVALID, "null result");
    GKOC_REQUIRE(num_blocks > 0 && n_rows > 0, GKOC_E_INVALID, "empty system");
    GKOC_REQUIRE(block_ptrs && blocks && x && r && p && q && beta && rho && stop && z && work,
                 GKOC_E_INVALID, "null pointer");
    const int64_t bo = scheme.block_offset;
    GKOC_REQUIRE(bo >= 1 && bo <= 16 && (bo << scheme.group_power) == 64 && (bo & (bo - 1)) == 0,
                 GKOC_E_NOT_SUPPORTED,
                 "fused step_2 + apply needs block_offset in {1,2,4,8,16} and a 64-wide group");
    GKOC_REQUIRE(max_bs <= uint64_t(bo), GKOC_E_INVALID, "max_block_size exceeds block_offset");
    GKOC_REQUIRE(work_bytes >= fused_workspace_bytes(n_rows, sizeof(T)), GKOC_E_WORKSPACE,
                 "workspace too small (gkoc_x_workspace_bytes)");
    const int64_t groups = ceildiv(num_blocks, int64_t(1) << scheme.group_power);
    // groups per wave as in the plain fused apply + dot: the partial sums then group the same
    // rows, and <r, z> comes out with the same bits
    const int gpw = (bo * int64_t(sizeof(T)) >= 128) ? 1 : 2;
    const int64_t nb = ceildiv(groups, 4 * gpw);
    // workspace of gkoc_x_workspace_bytes: [two rows of partials | 2 * fold_chunks of scratch]:
    // (n + 63) / 64 + 4096 + fold_chunks values; a row needs n / 64 at most (one partial per 8
    // groups of 8+ rows)
    const int64_t total = int64_t(fused_workspace_bytes(n_rows, sizeof(T)) / sizeof(T));
    const int64_t pstride = (total - 2 * fold_chunks) / 2;
    GKOC_REQUIRE(nb <= pstride, GKOC_E_WORKSPACE,
                 "too many blocks for the workspace (ask gkoc_x_cg_step_2_jacobi_apply_fits first)");
    T* partial = static_cast<T*>(work);
    T* scratch = partial + 2 * pstride;
    const int64_t go = scheme.group_offset;
#define GKOC_JAC_S2(BO_)                                                                          \
    jacobi_step2_apply_kernel<T, I, BO_, ((BO_ * sizeof(T) >= 128) ? 1 : 2)>                        \
        <<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(num_blocks, groups, go, block_ptrs,     \
                                                             blocks, x, r, p, q, beta, rho, stop, z, \
                                                             partial, pstride)
    switch (int(bo)) {
    case 1: GKOC_JAC_S2(1); break;
    case 2: GKOC_JAC_S2(2); break;
    case 4: GKOC_JAC_S2(4); break;
    case 8: GKOC_JAC_S2(8); break;
    default: GKOC_JAC_S2(16); break;
    }
#undef GKOC_JAC_S2
    GKOC_LAUNCH_OK();
    // <r, z> through the tree of gkoc_x_jacobi_simple_apply_dot: the same bits as the unfused pair
    return fold_partials2<T>(s, nb, pstride, partial, scratch, rho_out, norm_out, take_sqrt ? 1 : -1);
}

inline int jacobi_xcd_map(int64_t n_workgroups)
{
    return (tune_value(GKOC_TUNE_JACOBI_XCD_MAP) != 0 && n_workgroups >= 8 * 256) ? 1 : 0;
}

template <typename T, typename I, bool ADV, int BO>
void launch_apply_fixed(gkoc_stream_t s, int64_t num_blocks, int64_t groups,
                        int64_t group_offset, const I* block_ptrs,
                        const T* blocks, const T* alpha, const T* b,
                        const T* beta, T* x)
{
    constexpr int GPW = (BO * sizeof(T) >= 128) ? 1 : 2;
    jacobi_apply_fixed_kernel<T, I, ADV, BO, GPW>
        <<<dim3(unsigned(ceildiv(groups, 4 * GPW))), dim3(256), 0,
           as_stream(s)>>>(num_blocks, groups, group_offset, block_ptrs, blocks,
                           alpha, b, beta, x, nullptr, nullptr,
                           jacobi_xcd_map(ceildiv(groups, 4 * GPW)));
}

// x = M b and dot_out = <b, x> (fast-path layout only, one right-hand side)
template <typename T, typename I, int BO>
void launch_apply_dot_fixed(gkoc_stream_t s, int64_t num_blocks, int64_t groups,
                            int64_t group_offset, const I* block_ptrs,
                            const T* blocks, const T* b, T* x, T* partial,
                            int64_t* n_partials)
{
    constexpr int GPW = (BO * sizeof(T) >= 128) ? 1 : 2;
    const int64_t nb = ceildiv(groups, 4 * GPW);
    *n_partials = nb;
    jacobi_apply_fixed_kernel<T, I, false, BO, GPW, true>
        <<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(
            num_blocks, groups, group_offset, block_ptrs, blocks, nullptr, b,
            nullptr, x, partial, nullptr, jacobi_xcd_map(nb));
}

template <typename T, typename I>
int launch_apply_dot(gkoc_stream_t s, int64_t num_blocks, int64_t n_rows,
                     uint32_t max_bs, gkoc_jacobi_scheme scheme,
                     const I* block_ptrs, const T* blocks, const T* b, T* x,
                     T* dot_out, void* work, size_t work_bytes)
{
    GKOC_REQUIRE(dot_out, GKOC_E_INVALID, "null result");
    if (num_blocks <= 0) {
        GKOC_HIP(hipMemsetAsync(dot_out, 0, sizeof(T), as_stream(s)));
        return GKOC_OK;
    }
    GKOC_REQUIRE(block_ptrs && blocks && b && x && work, GKOC_E_INVALID, "null pointer");
    const int64_t bo = scheme.block_offset;
    GKOC_REQUIRE(bo >= 1 && bo <= 16 && (bo << scheme.group_power) == 64 &&
                     (bo & (bo - 1)) == 0,
                 GKOC_E_NOT_SUPPORTED,
                 "fused apply+dot needs block_offset in {1,2,4,8,16} and a 64-wide group");
    GKOC_REQUIRE(max_bs <= uint64_t(bo), GKOC_E_INVALID,
                 "max_block_size exceeds block_offset");
    GKOC_REQUIRE(work_bytes >= fused_workspace_bytes(n_rows, sizeof(T)), GKOC_E_WORKSPACE,
                 "workspace too small (gkoc_x_workspace_bytes)");
    const int64_t gsize = int64_t(1) << scheme.group_power;
    const int64_t groups = ceildiv(num_blocks, gsize);
    GKOC_REQUIRE(ceildiv(groups, 4) <= (n_rows + 63) / 64 + 4096, GKOC_E_INVALID,
                 "n_rows does not match the block count");
    T* partial = static_cast<T*>(work);
    T* scratch = partial + (fused_workspace_bytes(n_rows, sizeof(T)) / sizeof(T) - fold_chunks);
    int64_t np = 0;
    const int64_t go = scheme.group_offset;
    switch (int(bo)) {
    case 1: launch_apply_dot_fixed<T, I, 1>(s, num_blocks, groups, go, block_ptrs, blocks, b, x, partial, &np); break;
    case 2: launch_apply_dot_fixed<T, I, 2>(s, num_blocks, groups, go, block_ptrs, blocks, b, x, partial, &np); break;
    case 4: launch_apply_dot_fixed<T, I, 4>(s, num_blocks, groups, go, block_ptrs, blocks, b, x, partial, &np); break;
    case 8: launch_apply_dot_fixed<T, I, 8>(s, num_blocks, groups, go, block_ptrs, blocks, b, x, partial, &np); break;
    default: launch_apply_dot_fixed<T, I, 16>(s, num_blocks, groups, go, block_ptrs, blocks, b, x, partial, &np); break;
    }
    GKOC_LAUNCH_OK();
    return fold_partials<T>(s, np, partial, scratch, dot_out, false);
}

template <typename T, typename I, bool ADV>
int launch_apply(gkoc_stream_t s, int64_t num_blocks, uint32_t max_bs,
                 gkoc_jacobi_scheme scheme, const I* block_ptrs,
                 const T* blocks, const T* alpha, const T* b, int64_t ldb,
                 const T* beta, T* x, int64_t ldx, int64_t nrhs)
{
    if (num_blocks <= 0 || nrhs <= 0) return GKOC_OK;
    GKOC_REQUIRE(block_ptrs && blocks && b && x, GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(scheme.block_offset >= 1 &&
                     (scheme.block_offset << scheme.group_power) <= 64,
                 GKOC_E_NOT_SUPPORTED, "storage stride must be <= 64 (wave size)");
    GKOC_REQUIRE(max_bs <= uint64_t(scheme.block_offset), GKOC_E_INVALID,
                 "max_block_size exceeds block_offset");
    const int64_t gsize = int64_t(1) << scheme.group_power;
    const int64_t groups = ceildiv(num_blocks, gsize);
    const int64_t bo = scheme.block_offset;
    if (nrhs == 1 && ldb == 1 && ldx == 1 && (bo << scheme.group_power) == 64 &&
        bo <= 16) {
        const int64_t go = scheme.group_offset;
#define GKOC_JAC_FIXED(BO)                                                     \
    launch_apply_fixed<T, I, ADV, BO>(s, num_blocks, groups, go, block_ptrs,   \
                                      blocks, alpha, b, beta, x)
        switch (int(bo)) {
        case 1: GKOC_JAC_FIXED(1); break;
        case 2: GKOC_JAC_FIXED(2); break;
        case 4: GKOC_JAC_FIXED(4); break;
        case 8: GKOC_JAC_FIXED(8); break;
        default: GKOC_JAC_FIXED(16); break;
        }
#undef GKOC_JAC_FIXED
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    // GKOC_TUNE_JACOBI_MFMA: 0 never, 1 from two columns, 2 (default) from nine, 3 from four
    const int64_t mfma_mode = tune_value(GKOC_TUNE_JACOBI_MFMA);
    const int64_t mfma_from = mfma_mode == 1 ? 2 : mfma_mode == 3 ? 4 : 9;
    const bool mfma = sizeof(T) == 8 && mfma_mode != 0 && nrhs >= mfma_from && bo == 8 &&
                      scheme.group_power == 3 && b != x;
    if (nrhs >= 2 && (bo << scheme.group_power) == 64 && bo <= 16 && !mfma) {
        // several right-hand sides, fast-path layout: the blocks stay in registers for all columns
        // (L256, block size 8, profiles/r03_jacobi_multi_256.txt: 2 / 4 / 8 columns 279 / 375 / 630 us =
        // 72 / 72 / 64 % of 8 TB/s; round 2: 530 / 654 / 778 with the general and the matrix-core
        // kernel; 16 / 32 columns 1472 / 2974 us against 949 / 1787 on the matrix cores)
        const int pairs_ok = reinterpret_cast<uintptr_t>(b) % (2 * sizeof(T)) == 0 && ldb % 2 == 0 &&
                             reinterpret_cast<uintptr_t>(x) % (2 * sizeof(T)) == 0 && ldx % 2 == 0;
        const dim3 gm(unsigned(ceildiv(groups, 4)));
#define GKOC_JAC_MULTI(BO)                                                                        \
    do {                                                                                          \
        if (nrhs <= 4) {                                                                          \
            jacobi_apply_fixed_multi_kernel<T, I, ADV, BO, 4><<<gm, dim3(256), 0, as_stream(s)>>>( \
                num_blocks, groups, scheme.group_offset, block_ptrs, blocks, alpha, b, ldb, beta, \
                x, ldx, int(nrhs), pairs_ok);                                                     \
        } else {                                                                                  \
            jacobi_apply_fixed_multi_kernel<T, I, ADV, BO, 8><<<gm, dim3(256), 0, as_stream(s)>>>( \
                num_blocks, groups, scheme.group_offset, block_ptrs, blocks, alpha, b, ldb, beta, \
                x, ldx, int(nrhs), pairs_ok);                                                     \
        }                                                                                         \
    } while (0)
        switch (int(bo)) {
        case 1: GKOC_JAC_MULTI(1); break;
        case 2: GKOC_JAC_MULTI(2); break;
        case 4: GKOC_JAC_MULTI(4); break;
        case 8: GKOC_JAC_MULTI(8); break;
        default: GKOC_JAC_MULTI(16); break;
        }
#undef GKOC_JAC_MULTI
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    if constexpr (sizeof(T) == 8) {
        // matrix-core path (fused multiply-adds: ~5e-16 off the reference's bits).  Measured on
        // L256 (profiles/r02_jacobi_mfma_256.txt): 2 columns 603 us against 530 us for the lane =
        // row kernel, 4: 654 / 1251, 8: 778 / 3779, 16: 987 / 9680; since round 3 the exact
        // multi-column kernel above serves up to eight columns -> from nine columns on.
        if (mfma) {
            jacobi_apply_mfma_kernel<I, ADV>
                <<<dim3(unsigned(ceildiv(groups, 4))), dim3(256), 0, as_stream(s)>>>(
                    num_blocks, groups, scheme.group_offset, block_ptrs, blocks, alpha, b, ldb,
                    beta, x, ldx, int(nrhs));
            GKOC_LAUNCH_OK();
            return GKOC_OK;
        }
    }
    jacobi_apply_kernel<T, I, ADV>
        <<<dim3(unsigned(ceildiv(groups, 4))), dim3(256), 0, as_stream(s)>>>(
            num_blocks, groups, scheme, block_ptrs, blocks, alpha, b, ldb, beta,
            x, ldx, int(nrhs));
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename I, bool ADV, int BO, int PREC>
void launch_apply_stored_fixed(gkoc_stream_t s, int64_t num_blocks, int64_t groups,
                               int64_t group_offset, const I* block_ptrs, const double* blocks,
                               const double* alpha, const double* b, const double* beta, double* x)
{
    constexpr int GPW = 2;
    jacobi_apply_fixed_kernel<double, I, ADV, BO, GPW, false, PREC>
        <<<dim3(unsigned(ceildiv(groups, 4 * GPW))), dim3(256), 0, as_stream(s)>>>(
            num_blocks, groups, group_offset, block_ptrs, blocks, alpha, b, beta, x, nullptr,
            nullptr, jacobi_xcd_map(ceildiv(groups, 4 * GPW)));
}

inline bool known_precision(int prec)
{
    return prec == 0x01 || prec == 0x02 || prec == 0x10 || prec == 0x11 || prec == 0x20;
}

#define GKOC_FOR_PREC(M, ...)                 \
    switch (prec) {                           \
    case 0x01: M(0x01, __VA_ARGS__); break;   \
    case 0x02: M(0x02, __VA_ARGS__); break;   \
    case 0x10: M(0x10, __VA_ARGS__); break;   \
    case 0x11: M(0x11, __VA_ARGS__); break;   \
    default: M(0x20, __VA_ARGS__); break;     \
    }

template <typename I, bool ADV>
int launch_apply_stored(gkoc_stream_t s, int64_t num_blocks, uint32_t max_bs,
                        gkoc_jacobi_scheme scheme, const I* block_ptrs, const double* blocks,
                        int prec, const double* alpha, const double* b, int64_t ldb,
                        const double* beta, double* x, int64_t ldx, int64_t nrhs)
{
    if (prec == 0) {
        return launch_apply<double, I, ADV>(s, num_blocks, max_bs, scheme, block_ptrs, blocks,
                                            alpha, b, ldb, beta, x, ldx, nrhs);
    }
    if (num_blocks <= 0 || nrhs <= 0) return GKOC_OK;
    GKOC_REQUIRE(known_precision(prec), GKOC_E_NOT_SUPPORTED, "unknown storage precision");
    GKOC_REQUIRE(block_ptrs && blocks && b && x, GKOC_E_INVALID, "null pointer");
    const int64_t bo = scheme.block_offset;
    GKOC_REQUIRE(bo >= 1 && bo <= 16 && (bo & (bo - 1)) == 0 && (bo << scheme.group_power) == 64,
                 GKOC_E_NOT_SUPPORTED,
                 "reduced-precision storage needs block_offset in {1,2,4,8,16}, 64-wide groups");
    GKOC_REQUIRE(max_bs <= uint64_t(bo), GKOC_E_INVALID, "max_block_size exceeds block_offset");
    const int64_t groups = ceildiv(num_blocks, int64_t(1) << scheme.group_power);
    const int64_t go = scheme.group_offset;
    // one right-hand side per launch (reference: one apply_block per column as well)
    for (int64_t j = 0; j < nrhs; ++j) {
        GKOC_REQUIRE(ldb == 1 && ldx == 1, GKOC_E_NOT_SUPPORTED,
                     "reduced-precision storage: one right-hand side with unit strides");
#define GKOC_JAC_ST(PREC_, BO_)                                                                 \
    launch_apply_stored_fixed<I, ADV, BO_, PREC_>(s, num_blocks, groups, go, block_ptrs, blocks, \
                                                  alpha, b + j, beta, x + j)
#define GKOC_JAC_ST_BO(PREC_, dummy)            \
    switch (int(bo)) {                          \
    case 1: GKOC_JAC_ST(PREC_, 1); break;       \
    case 2: GKOC_JAC_ST(PREC_, 2); break;       \
    case 4: GKOC_JAC_ST(PREC_, 4); break;       \
    case 8: GKOC_JAC_ST(PREC_, 8); break;       \
    default: GKOC_JAC_ST(PREC_, 16); break;     \
    }
        GKOC_FOR_PREC(GKOC_JAC_ST_BO, 0)
#undef GKOC_JAC_ST_BO
#undef GKOC_JAC_ST
        GKOC_LAUNCH_OK();
    }
    return GKOC_OK;
}

int launch_convert_storage(gkoc_stream_t s, int64_t num_blocks, gkoc_jacobi_scheme scheme,
                           double* blocks, int prec)
{
    if (prec == 0 || num_blocks <= 0) return GKOC_OK;
    GKOC_REQUIRE(known_precision(prec), GKOC_E_NOT_SUPPORTED, "unknown storage precision");
    GKOC_REQUIRE(blocks, GKOC_E_INVALID, "null pointer");
    const int64_t bo = scheme.block_offset;
    GKOC_REQUIRE(bo >= 1 && bo <= 16 && (bo & (bo - 1)) == 0 && (bo << scheme.group_power) == 64,
                 GKOC_E_NOT_SUPPORTED,
                 "reduced-precision storage needs block_offset in {1,2,4,8,16}, 64-wide groups");
    const int64_t groups = ceildiv(num_blocks, int64_t(1) << scheme.group_power);
#define GKOC_JAC_CV(PREC_, BO_)                                                        \
    jacobi_convert_storage_kernel<PREC_, BO_>                                          \
        <<<dim3(unsigned(groups)), dim3(64), 0, as_stream(s)>>>(groups, scheme.group_offset, blocks)
#define GKOC_JAC_CV_BO(PREC_, dummy)            \
    switch (int(bo)) {                          \
    case 1: GKOC_JAC_CV(PREC_, 1); break;       \
    case 2: GKOC_JAC_CV(PREC_, 2); break;       \
    case 4: GKOC_JAC_CV(PREC_, 4); break;       \
    case 8: GKOC_JAC_CV(PREC_, 8); break;       \
    default: GKOC_JAC_CV(PREC_, 16); break;     \
    }
    GKOC_FOR_PREC(GKOC_JAC_CV_BO, 0)
#undef GKOC_JAC_CV_BO
#undef GKOC_JAC_CV
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename T, typename I>
int launch_generate(gkoc_stream_t s, const I* row_ptrs, const I* cols,
                    const T* vals, int64_t num_blocks, uint32_t max_bs,
                    gkoc_jacobi_scheme scheme, const I* block_ptrs, T* blocks)
{
    if (num_blocks <= 0) return GKOC_OK;
    GKOC_REQUIRE(max_bs >= 1 && max_bs <= 64, GKOC_E_NOT_SUPPORTED,
                 "max_block_size must be in [1, 64]");
    int sub = 1;
    while (sub < int(max_bs)) sub *= 2;
    const int per_wave = 64 / sub;
    const size_t lds = size_t(per_wave) * sub * (sub + 1) * sizeof(T);
    jacobi_generate_kernel<T, I>
        <<<dim3(unsigned(ceildiv(num_blocks, per_wave))), dim3(64), lds,
           as_stream(s)>>>(row_ptrs, cols, vals, num_blocks, sub, scheme,
                           block_ptrs, blocks);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// Block-Jacobi application for the value types without a tuned kernel (complex): one lane per
// (row, right-hand side) walks its row of the inverse block in the interleaved scheme - element
// (r, c) of block b at group_offset * (b >> gp) + block_offset * (b & mask) + r + c * stride
// (include/ginkgo/core/preconditioner/jacobi.hpp:37-140) - and adds the products in column order
// (reference apply_block, reference/preconditioner/jacobi_kernels.cpp:419-531).  row_block[row] = the
// block of a row (filled by jacobi_row_block_kernel).
// (defined with the adaptive kernels at the end of this file)
template <typename T, typename I, bool ADV>
bool launch_apply_lanes_any(hipStream_t st, int64_t num_blocks, gkoc_jacobi_scheme scheme, const I* block_ptrs,
                            const T* blocks, const uint8_t* precisions, const T* alpha, const T* b, int64_t ldb,
                            const T* beta, T* x, int64_t ldx, int64_t nrhs);

template <typename I>
__global__ __launch_bounds__(256) void jacobi_row_block_kernel(int64_t num_blocks,
                                                              const I* __restrict__ block_ptrs,
                                                              I* __restrict__ row_block)
{
    const int64_t b = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (b >= num_blocks) return;
    for (I r = block_ptrs[b]; r < block_ptrs[b + 1]; ++r) row_block[r] = I(b);
}

template <typename T, typename I, bool ADV>
__global__ __launch_bounds__(256) void jacobi_apply_simple_kernel(
    int64_t n_rows, int64_t nrhs, gkoc_jacobi_scheme scheme, const I* __restrict__ block_ptrs,
    const I* __restrict__ row_block, const T* __restrict__ blocks, const T* __restrict__ alpha_p,
    const T* __restrict__ b, int64_t ldb, const T* __restrict__ beta_p, T* __restrict__ x, int64_t ldx)
{
    const int64_t total = n_rows * nrhs, step = int64_t(gridDim.x) * 256;
    const int64_t stride = scheme.block_offset << scheme.group_power;
    const int64_t mask = (int64_t(1) << scheme.group_power) - 1;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += step) {
        const int64_t j = idx / n_rows, row = idx - j * n_rows;
        const int64_t blk = row_block[row];
        const int64_t start = block_ptrs[blk], bs = int64_t(block_ptrs[blk + 1]) - start;
        const T* __restrict__ m = blocks + scheme.group_offset * (blk >> scheme.group_power) +
                                  scheme.block_offset * (blk & mask) + (row - start);
        T sum = T(0);
        for (int64_t c = 0; c < bs; ++c) sum += m[c * stride] * b[(start + c) * ldb + j];
        if (ADV) {
            const T beta = beta_p[0];
            const T ax = alpha_p[0] * sum;
            x[row * ldx + j] = beta == T(0) ? ax : ax + beta * x[row * ldx + j];
        } else {
            x[row * ldx + j] = sum;
        }
    }
}

template <typename T, typename I, bool ADV>
int launch_apply_simple(gkoc_stream_t s, int64_t num_blocks, gkoc_jacobi_scheme scheme, const I* block_ptrs,
                        const T* blocks, const T* alpha, const T* b, int64_t ldb, const T* beta, T* x,
                        int64_t ldx, int64_t nrhs)
{
    if (num_blocks <= 0 || nrhs <= 0) return GKOC_OK;
    GKOC_REQUIRE(block_ptrs && blocks && b && x, GKOC_E_INVALID, "null pointer");
    hipStream_t st = as_stream(s);
    // complex values (the only users of this launcher): lane = (block, row), jacobi_apply_lanes_any_kernel
    if (tune_value(GKOC_TUNE_JACOBI_LANES) != 1 &&
        launch_apply_lanes_any<T, I, ADV>(st, num_blocks, scheme, block_ptrs, blocks, nullptr, alpha, b, ldb, beta,
                                          x, ldx, nrhs)) {
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    I last = 0;
    GKOC_HIP(hipMemcpyAsync(&last, block_ptrs + num_blocks, sizeof(I), hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipStreamSynchronize(st));
    const int64_t n_rows = int64_t(last);
    if (n_rows <= 0) return GKOC_OK;
    I* row_block = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&row_block), size_t(n_rows) * sizeof(I)));
    jacobi_row_block_kernel<I><<<dim3(unsigned(ceildiv(num_blocks, 256))), dim3(256), 0, st>>>(
        num_blocks, block_ptrs, row_block);
    int64_t nb = ceildiv(n_rows * nrhs, 256);
    if (nb > 8 * max_stream_blocks) nb = 8 * max_stream_blocks;
    jacobi_apply_simple_kernel<T, I, ADV><<<dim3(unsigned(nb)), dim3(256), 0, st>>>(
        n_rows, nrhs, scheme, block_ptrs, row_block, blocks, alpha, b, ldb, beta, x, ldx);
    const hipError_t e = hipGetLastError();
    (void)scratch_free(st, row_block);
    GKOC_HIP(e);
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

// complex block-Jacobi: find_blocks (indices only), generate (the Gauss-Jordan kernel above on
// gkoc_cplx: pivot by magnitude), simple_apply / apply (jacobi_apply_simple_kernel).  Uniform storage
// precision (block-wise / adaptive: the end of this file); agrees with the reference to rounding.
#define GKOC_DEF_CJACOBI(T, TN, I, IN)                                                                  \
    extern "C" int gkoc_jacobi_find_blocks_##TN##_##IN(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, \
                                                       const I* col_idxs, uint32_t max_block_size,      \
                                                       int64_t* num_blocks_host, I* block_ptrs)         \
    {                                                                                                   \
        return find_blocks_impl<I>(s, n_rows, row_ptrs, col_idxs, max_block_size, num_blocks_host,      \
                                   block_ptrs);                                                         \
    }                                                                                                   \
    extern "C" int gkoc_jacobi_generate_##TN##_##IN(                                                    \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* col_idxs, const T* vals,           \
        int64_t num_blocks, uint32_t max_block_size, gkoc_jacobi_scheme scheme, const I* block_ptrs,    \
        T* blocks, T* conditioning)                                                                     \
    {                                                                                                   \
        (void)n_rows;                                                                                   \
        GKOC_REQUIRE(conditioning == nullptr, GKOC_E_NOT_SUPPORTED,                                     \
                     "condition numbers: gkoc_jacobi_generate_adaptive_* computes them");               \
        GKOC_REQUIRE(max_block_size <= 32, GKOC_E_NOT_SUPPORTED, "complex blocks: max_block_size <= 32"); \
        return launch_generate<T, I>(s, row_ptrs, col_idxs, vals, num_blocks, max_block_size, scheme,   \
                                     block_ptrs, blocks);                                               \
    }                                                                                                   \
    extern "C" int gkoc_jacobi_simple_apply_##TN##_##IN(                                                \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size, gkoc_jacobi_scheme scheme,        \
        const I* block_ptrs, const T* blocks, const T* b, int64_t ldb, T* x, int64_t ldx, int64_t nrhs) \
    {                                                                                                   \
        (void)max_block_size;                                                                           \
        return launch_apply_simple<T, I, false>(s, num_blocks, scheme, block_ptrs, blocks, nullptr, b,  \
                                                ldb, nullptr, x, ldx, nrhs);                            \
    }                                                                                                   \
    extern "C" int gkoc_jacobi_apply_##TN##_##IN(                                                       \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size, gkoc_jacobi_scheme scheme,        \
        const I* block_ptrs, const T* blocks, const T* alpha, const T* b, int64_t ldb, const T* beta,   \
        T* x, int64_t ldx, int64_t nrhs)                                                                \
    {                                                                                                   \
        (void)max_block_size;                                                                           \
        GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha / beta");                               \
        return launch_apply_simple<T, I, true>(s, num_blocks, scheme, block_ptrs, blocks, alpha, b,     \
                                               ldb, beta, x, ldx, nrhs);                                \
    }
GKOC_DEF_CJACOBI(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_CJACOBI(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_CJACOBI(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_CJACOBI(gkoc_c64, c64, int64_t, i64)

#define GKOC_DEF_JACOBI(T, TN, I, IN)                                          \
    extern "C" int gkoc_jacobi_find_blocks_##TN##_##IN(                        \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                    \
        const I* col_idxs, uint32_t max_block_size,                            \
        int64_t* num_blocks_host, I* block_ptrs)                               \
    {                                                                          \
        return find_blocks_impl<I>(s, n_rows, row_ptrs, col_idxs,              \
                                   max_block_size, num_blocks_host,            \
                                   block_ptrs);                                \
    }                                                                          \
    extern "C" int gkoc_jacobi_generate_##TN##_##IN(                           \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                    \
        const I* col_idxs, const T* vals, int64_t num_blocks,                  \
        uint32_t max_block_size, gkoc_jacobi_scheme scheme,                    \
        const I* block_ptrs, T* blocks, T* conditioning)                       \
    {                                                                          \
        (void)n_rows;                                                          \
        GKOC_REQUIRE(conditioning == nullptr, GKOC_E_NOT_SUPPORTED,            \
                     "adaptive-precision block-Jacobi is not supported");      \
        return launch_generate<T, I>(s, row_ptrs, col_idxs, vals, num_blocks,  \
                                     max_block_size, scheme, block_ptrs,       \
                                     blocks);                                  \
    }                                                                          \
    extern "C" int gkoc_jacobi_simple_apply_##TN##_##IN(                       \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,          \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, const T* blocks,       \
        const T* b, int64_t ldb, T* x, int64_t ldx, int64_t nrhs)              \
    {                                                                          \
        return launch_apply<T, I, false>(s, num_blocks, max_block_size,        \
                                         scheme, block_ptrs, blocks, nullptr,  \
                                         b, ldb, nullptr, x, ldx, nrhs);       \
    }                                                                          \
    extern "C" int gkoc_x_jacobi_simple_apply_dot_##TN##_##IN(                 \
        gkoc_stream_t s, int64_t num_blocks, int64_t n_rows,                   \
        uint32_t max_block_size, gkoc_jacobi_scheme scheme,                    \
        const I* block_ptrs, const T* blocks, const T* b, T* x, T* dot_out,    \
        void* work, size_t work_bytes)                                         \
    {                                                                          \
        return launch_apply_dot<T, I>(s, num_blocks, n_rows, max_block_size,   \
                                      scheme, block_ptrs, blocks, b, x,        \
                                      dot_out, work, work_bytes);              \
    }                                                                          \
    extern "C" int gkoc_x_cg_step_2_jacobi_apply_##TN##_##IN(                  \
        gkoc_stream_t s, int64_t num_blocks, int64_t n_rows,                   \
        uint32_t max_block_size, gkoc_jacobi_scheme scheme,                    \
        const I* block_ptrs, const T* blocks, T* x, T* r, const T* p,          \
        const T* q, const T* beta, const T* rho, const uint8_t* stop_status,   \
        T* z, T* rho_out, T* norm_out, int take_sqrt, void* work,              \
        size_t work_bytes)                                                     \
    {                                                                          \
        return launch_step2_apply<T, I>(s, num_blocks, n_rows, max_block_size, \
                                        scheme, block_ptrs, blocks, x, r, p,   \
                                        q, beta, rho, stop_status, z, rho_out, \
                                        norm_out, take_sqrt, work, work_bytes); \
    }                                                                          \
    extern "C" int gkoc_x_pipe_cg_steps_jacobi_##TN##_##IN(                    \
        gkoc_stream_t s, int64_t num_blocks, int64_t n_rows,                   \
        uint32_t max_block_size, gkoc_jacobi_scheme scheme,                    \
        const I* block_ptrs, const T* blocks, T* x, T* r, T* z, T* w, T* p,    \
        T* q, T* f, T* g, T* m, const T* n, const T* prev_rho, const T* rho,   \
        const T* delta, const T* beta_in, T* beta_out,                         \
        const uint8_t* stop_status, T* out3, void* work, size_t work_bytes,    \
        const gkoc_step_gate* gate)                                            \
    {                                                                          \
        return launch_pipe_steps<T, I>(s, num_blocks, n_rows, max_block_size,  \
                                       scheme, block_ptrs, blocks, x, r, z, w, \
                                       p, q, f, g, m, n, prev_rho, rho, delta, \
                                       beta_in, beta_out, stop_status, out3,   \
                                       work, work_bytes, gate);                \
    }                                                                          \
    extern "C" int gkoc_jacobi_apply_##TN##_##IN(                              \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,          \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, const T* blocks,       \
        const T* alpha, const T* b, int64_t ldb, const T* beta, T* x,          \
        int64_t ldx, int64_t nrhs)                                             \
    {                                                                          \
        GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha/beta");        \
        return launch_apply<T, I, true>(s, num_blocks, max_block_size, scheme, \
                                        block_ptrs, blocks, alpha, b, ldb,     \
                                        beta, x, ldx, nrhs);                   \
    }

extern "C" int gkoc_x_cg_step_2_jacobi_apply_fits(int64_t num_blocks, int64_t n_rows,
                                                  gkoc_jacobi_scheme scheme, size_t value_size)
{
    return gkoc::step2_apply_fits(num_blocks, n_rows, scheme, value_size) ? 1 : 0;
}

namespace gkoc {
namespace {
__global__ void tile_bytes_kernel(int64_t n, const uint8_t* __restrict__ src, int64_t src_n,
                                  uint8_t* __restrict__ dst)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        dst[i] = src[i % src_n];
    }
}
}  // namespace
}  // namespace gkoc

namespace gkoc {
namespace {
// jacobi::transpose_jacobi / conj_transpose_jacobi for real types
// (reference/preconditioner/jacobi_kernels.cpp:597-627): every stored block is
// transposed in place of its group, in its own storage type (the entries are moved as
// raw words of the type's width, so no rounding takes place).
template <typename T, typename I>
__global__ __launch_bounds__(256) void jacobi_transpose_kernel(
    int64_t num_blocks, gkoc_jacobi_scheme scheme, const I* __restrict__ block_ptrs,
    const T* __restrict__ blocks, const uint8_t* __restrict__ precs, T* __restrict__ out)
{
    const int64_t bo = scheme.block_offset;
    const int64_t stride = bo << scheme.group_power;
    const int64_t gmask = (int64_t(1) << scheme.group_power) - 1;
    const int64_t total = num_blocks * bo;
    const int64_t step = int64_t(gridDim.x) * 256;
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < total; t += step) {
        const int64_t blk = t / bo;
        const int r = int(t - blk * bo);
        const int bs = int(block_ptrs[blk + 1] - block_ptrs[blk]);
        if (r >= bs) continue;
        const int64_t goff = scheme.group_offset * (blk >> scheme.group_power);
        const int64_t boff = bo * (blk & gmask);
        const int prec = precs ? int(precs[blk]) : 0;
        const int width = prec == 0 ? int(sizeof(T)) : (prec == 0x01 || prec == 0x10) ? 4 : 2;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(blocks + goff);
        unsigned char* dst = reinterpret_cast<unsigned char*>(out + goff);
        for (int c = 0; c < bs; ++c) {
            // out(r, c) = in(c, r)
            const int64_t from = (boff + c + int64_t(r) * stride) * width;
            const int64_t to = (boff + r + int64_t(c) * stride) * width;
            for (int k = 0; k < width; ++k) dst[to + k] = src[from + k];
        }
    }
}
}  // namespace
}  // namespace gkoc

#define GKOC_DEF_JACOBI_TRANSPOSE(T, TN, I, IN)                                             \
    extern "C" int gkoc_jacobi_transpose_##TN##_##IN(                                       \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,                       \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, const T* blocks,                    \
        const uint8_t* precisions, T* out_blocks)                                           \
    {                                                                                       \
        (void)max_block_size;                                                               \
        if (num_blocks <= 0) return GKOC_OK;                                                \
        GKOC_REQUIRE(block_ptrs && blocks && out_blocks, GKOC_E_INVALID, "null pointer");   \
        GKOC_REQUIRE(scheme.block_offset >= 1, GKOC_E_INVALID, "bad storage scheme");       \
        GKOC_REQUIRE(precisions == nullptr || sizeof(T) == 8, GKOC_E_NOT_SUPPORTED,         \
                     "reduced float blocks: gkoc_jacobi_transpose_adaptive_f32_*");         \
        int64_t nb = ceildiv(num_blocks * scheme.block_offset, 256);                        \
        if (nb > 4 * max_stream_blocks) nb = 4 * max_stream_blocks;                         \
        jacobi_transpose_kernel<T, I><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(  \
            num_blocks, scheme, block_ptrs, blocks, precisions, out_blocks);                \
        GKOC_LAUNCH_OK();                                                                   \
        return GKOC_OK;                                                                     \
    }
GKOC_DEF_JACOBI_TRANSPOSE(double, f64, int32_t, i32)
GKOC_DEF_JACOBI_TRANSPOSE(double, f64, int64_t, i64)
GKOC_DEF_JACOBI_TRANSPOSE(float, f32, int32_t, i32)
GKOC_DEF_JACOBI_TRANSPOSE(float, f32, int64_t, i64)

// ... and for complex values (uniform storage): conj != 0 conjugates the moved entries
// (conj_transpose_jacobi, reference/preconditioner/jacobi_kernels.cpp:613-627)
namespace gkoc {
namespace {
template <typename T, typename I>
__global__ __launch_bounds__(256) void cjacobi_transpose_kernel(int64_t num_blocks, gkoc_jacobi_scheme scheme,
                                                                const I* __restrict__ block_ptrs,
                                                                const T* __restrict__ blocks, int conj,
                                                                T* __restrict__ out)
{
    const int64_t bo = scheme.block_offset;
    const int64_t stride = bo << scheme.group_power;
    const int64_t gmask = (int64_t(1) << scheme.group_power) - 1;
    const int64_t total = num_blocks * bo;
    const int64_t step = int64_t(gridDim.x) * 256;
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < total; t += step) {
        const int64_t blk = t / bo;
        const int r = int(t - blk * bo);
        const int bs = int(block_ptrs[blk + 1] - block_ptrs[blk]);
        if (r >= bs) continue;
        const int64_t base = scheme.group_offset * (blk >> scheme.group_power) + bo * (blk & gmask);
        for (int c = 0; c < bs; ++c) {
            T v = blocks[base + c + int64_t(r) * stride];      // in(c, r)
            if (conj) v = conj_v(v);
            out[base + r + int64_t(c) * stride] = v;           // out(r, c)
        }
    }
}
}  // namespace
}  // namespace gkoc

#define GKOC_DEF_CJACOBI_TRANSPOSE(T, TN, I, IN)                                                        \
    extern "C" int gkoc_cjacobi_transpose_##TN##_##IN(gkoc_stream_t s, int64_t num_blocks,              \
                                                      gkoc_jacobi_scheme scheme, const I* block_ptrs,   \
                                                      const T* blocks, int conj, T* out_blocks)         \
    {                                                                                                   \
        if (num_blocks <= 0) return GKOC_OK;                                                            \
        GKOC_REQUIRE(block_ptrs && blocks && out_blocks, GKOC_E_INVALID, "null pointer");               \
        GKOC_REQUIRE(scheme.block_offset >= 1, GKOC_E_INVALID, "bad storage scheme");                   \
        int64_t nb = ceildiv(num_blocks * scheme.block_offset, 256);                                    \
        if (nb > 4 * max_stream_blocks) nb = 4 * max_stream_blocks;                                     \
        cjacobi_transpose_kernel<T, I><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(             \
            num_blocks, scheme, block_ptrs, blocks, conj, out_blocks);                                  \
        GKOC_LAUNCH_OK();                                                                               \
        return GKOC_OK;                                                                                 \
    }
GKOC_DEF_CJACOBI_TRANSPOSE(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_CJACOBI_TRANSPOSE(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_CJACOBI_TRANSPOSE(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_CJACOBI_TRANSPOSE(gkoc_c64, c64, int64_t, i64)

// jacobi::initialize_precisions (reference/preconditioner/jacobi_kernels.cpp:454-462)
extern "C" int gkoc_jacobi_initialize_precisions(gkoc_stream_t s, const uint8_t* source,
                                                 int64_t source_size, uint8_t* precisions,
                                                 int64_t n)
{
    GKOC_REQUIRE(n >= 0 && source_size >= 0, GKOC_E_INVALID, "negative size");
    if (n == 0) return GKOC_OK;
    GKOC_REQUIRE(source && precisions && source_size > 0, GKOC_E_INVALID, "bad argument");
    int64_t nb = ceildiv(n, 256);
    if (nb > max_stream_blocks) nb = max_stream_blocks;
    tile_bytes_kernel<<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(n, source, source_size,
                                                                         precisions);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

namespace gkoc {
namespace {

inline bool wide_group_layout(const gkoc_jacobi_scheme& sc)
{
    const int64_t bo = sc.block_offset;
    return bo >= 1 && bo <= 16 && (bo & (bo - 1)) == 0 && (bo << sc.group_power) == 64;
}

// lanes per block = the power of two at or above block_offset
// (Jacobi::compute_storage_scheme, include/ginkgo/core/preconditioner/jacobi.hpp: a group holds
// max_block_stride / that power blocks; max_block_stride = 64 on this device)
inline int subwarp_of(const gkoc_jacobi_scheme& sc)
{
    int sub = 1;
    while (sub < sc.block_offset) sub <<= 1;
    return sub;
}

inline bool wave_group_layout(const gkoc_jacobi_scheme& sc)
{
    return sc.block_offset >= 1 && sc.block_offset <= 32 &&
           (int64_t(subwarp_of(sc)) << sc.group_power) == 64;
}

template <typename I>
int launch_generate_adaptive(gkoc_stream_t s, const I* row_ptrs, const I* cols, const double* vals,
                             int64_t num_blocks, uint32_t max_bs, gkoc_jacobi_scheme scheme,
                             const I* block_ptrs, double accuracy, uint8_t* precisions,
                             double* conditioning, double* blocks)
{
    if (num_blocks <= 0) return GKOC_OK;
    GKOC_REQUIRE(row_ptrs && cols && vals && block_ptrs && precisions && blocks, GKOC_E_INVALID,
                 "null pointer");
    GKOC_REQUIRE(wave_group_layout(scheme) && max_bs >= 1 && max_bs <= uint64_t(scheme.block_offset),
                 GKOC_E_NOT_SUPPORTED,
                 "adaptive block-Jacobi needs max_block_size <= 32 and groups that fill a wavefront "
                 "(max_block_stride 64)");
    const int sub = subwarp_of(scheme);
    const int per_wave = 64 / sub;
    const size_t lds = 2 * size_t(per_wave) * sub * (sub + 1) * sizeof(double);
    jacobi_generate_adaptive_kernel<I>
        <<<dim3(unsigned(ceildiv(num_blocks, per_wave))), dim3(64), lds, as_stream(s)>>>(
            row_ptrs, cols, vals, num_blocks, sub, scheme, block_ptrs, accuracy, precisions,
            conditioning, blocks);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename I, bool ADV, int BO>
void launch_apply_adaptive_fixed(gkoc_stream_t s, int64_t num_blocks, int64_t groups,
                                 int64_t group_offset, const I* block_ptrs, const double* blocks,
                                 const uint8_t* precisions, const double* alpha, const double* b,
                                 const double* beta, double* x)
{
    constexpr int GPW = 2;
    jacobi_apply_fixed_kernel<double, I, ADV, BO, GPW, false, -1>
        <<<dim3(unsigned(ceildiv(groups, 4 * GPW))), dim3(256), 0, as_stream(s)>>>(
            num_blocks, groups, group_offset, block_ptrs, blocks, alpha, b, beta, x, nullptr,
            precisions, jacobi_xcd_map(ceildiv(groups, 4 * GPW)));
}

template <typename I, bool ADV>
int launch_apply_adaptive(gkoc_stream_t s, int64_t num_blocks, uint32_t max_bs,
                          gkoc_jacobi_scheme scheme, const I* block_ptrs, const double* blocks,
                          const uint8_t* precisions, const double* alpha, const double* b,
                          int64_t ldb, const double* beta, double* x, int64_t ldx, int64_t nrhs)
{
    if (num_blocks <= 0 || nrhs <= 0) return GKOC_OK;
    GKOC_REQUIRE(block_ptrs && blocks && precisions && b && x, GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(scheme.block_offset >= 1 && (scheme.block_offset << scheme.group_power) <= 64 &&
                     max_bs <= uint64_t(scheme.block_offset),
                 GKOC_E_NOT_SUPPORTED, "storage stride must be <= 64 (wave size)");
    const int64_t groups = ceildiv(num_blocks, int64_t(1) << scheme.group_power);
    if (!(wide_group_layout(scheme) && nrhs == 1 && ldb == 1 && ldx == 1)) {
        // any other layout (block_offset not a power of two, e.g. max_block_size 13), several
        // right-hand sides, strides: lane = row of a block, entries converted as they are read
        jacobi_apply_kernel<double, I, ADV, true>
            <<<dim3(unsigned(ceildiv(groups, 4))), dim3(256), 0, as_stream(s)>>>(
                num_blocks, groups, scheme, block_ptrs, blocks, alpha, b, ldb, beta, x, ldx, int(nrhs),
                precisions);
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    const int64_t go = scheme.group_offset;
#define GKOC_JAC_AD(BO_)                                                                    \
    launch_apply_adaptive_fixed<I, ADV, BO_>(s, num_blocks, groups, go, block_ptrs, blocks, \
                                             precisions, alpha, b, beta, x)
    switch (int(scheme.block_offset)) {
    case 1: GKOC_JAC_AD(1); break;
    case 2: GKOC_JAC_AD(2); break;
    case 4: GKOC_JAC_AD(4); break;
    case 8: GKOC_JAC_AD(8); break;
    default: GKOC_JAC_AD(16); break;
    }
#undef GKOC_JAC_AD
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

#define GKOC_DEF_JACOBI_ADAPTIVE(I, IN)                                                      \
    extern "C" int gkoc_jacobi_generate_adaptive_f64_##IN(                                   \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* col_idxs,               \
        const double* vals, int64_t num_blocks, uint32_t max_block_size,                     \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, double accuracy,                     \
        uint8_t* precisions, double* conditioning, double* blocks)                           \
    {                                                                                        \
        (void)n_rows;                                                                        \
        return launch_generate_adaptive<I>(s, row_ptrs, col_idxs, vals, num_blocks,          \
                                           max_block_size, scheme, block_ptrs, accuracy,     \
                                           precisions, conditioning, blocks);                \
    }                                                                                        \
    extern "C" int gkoc_jacobi_apply_adaptive_f64_##IN(                                      \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,                        \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, const double* blocks,                \
        const uint8_t* precisions, const double* alpha, const double* b, int64_t ldb,        \
        const double* beta, double* x, int64_t ldx, int64_t nrhs)                            \
    {                                                                                        \
        GKOC_REQUIRE((alpha == nullptr) == (beta == nullptr), GKOC_E_INVALID,                \
                     "pass alpha and beta, or neither");                                     \
        if (alpha) {                                                                         \
            return launch_apply_adaptive<I, true>(s, num_blocks, max_block_size, scheme,     \
                                                  block_ptrs, blocks, precisions, alpha, b,  \
                                                  ldb, beta, x, ldx, nrhs);                  \
        }                                                                                    \
        return launch_apply_adaptive<I, false>(s, num_blocks, max_block_size, scheme,        \
                                               block_ptrs, blocks, precisions, nullptr, b,   \
                                               ldb, nullptr, x, ldx, nrhs);                  \
    }
GKOC_DEF_JACOBI_ADAPTIVE(int32_t, i32)
GKOC_DEF_JACOBI_ADAPTIVE(int64_t, i64)

// reduced-precision storage: value type double only
extern "C" int gkoc_jacobi_convert_storage_f64(gkoc_stream_t s, int64_t num_blocks,
                                               gkoc_jacobi_scheme scheme, double* blocks,
                                               uint8_t precision)
{
    return launch_convert_storage(s, num_blocks, scheme, blocks, int(precision));
}

#define GKOC_DEF_JACOBI_STORED(I, IN)                                                        \
    extern "C" int gkoc_jacobi_apply_stored_f64_##IN(                                        \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size,                        \
        gkoc_jacobi_scheme scheme, const I* block_ptrs, const double* blocks,                \
        uint8_t precision, const double* alpha, const double* b, int64_t ldb,                \
        const double* beta, double* x, int64_t ldx, int64_t nrhs)                            \
    {                                                                                        \
        GKOC_REQUIRE((alpha == nullptr) == (beta == nullptr), GKOC_E_INVALID,                \
                     "pass alpha and beta, or neither");                                     \
        if (alpha) {                                                                         \
            return launch_apply_stored<I, true>(s, num_blocks, max_block_size, scheme,       \
                                                block_ptrs, blocks, int(precision), alpha,   \
                                                b, ldb, beta, x, ldx, nrhs);                 \
        }                                                                                    \
        return launch_apply_stored<I, false>(s, num_blocks, max_block_size, scheme,          \
                                             block_ptrs, blocks, int(precision), nullptr, b, \
                                             ldb, nullptr, x, ldx, nrhs);                    \
    }
GKOC_DEF_JACOBI_STORED(int32_t, i32)
GKOC_DEF_JACOBI_STORED(int64_t, i64)

GKOC_DEF_JACOBI(double, f64, int32_t, i32)
GKOC_DEF_JACOBI(double, f64, int64_t, i64)
GKOC_DEF_JACOBI(float, f32, int32_t, i32)
GKOC_DEF_JACOBI(float, f32, int64_t, i64)

// =========================================================================================
// Adaptive / block-wise storage precision for the value types float, complex<float> and
// complex<double> (VT instantiations of jacobi::generate / apply / transpose the reference
// compiles for every value type, core/preconditioner/jacobi_kernels.hpp:30-103).  A path of its
// own: the double kernels above stay as tuned.  The rules are the reference's, per component type
// R = remove_complex<T> (core/preconditioner/jacobi_utils.hpp:104-176, include/ginkgo/core/base/
// math.hpp:365-383, :546-582):
//   R = double: the five reduced types of the double path (float, half, the upper 32 / 16 bits of
//               the double, the upper 16 bits of the float);
//   R = float:  reduce_precision<float> = half and nothing below it, truncate_type<float> = the
//               upper 16 bits and nothing below them, so the five precision_reduction values fall
//               on two 16-bit types: (0,1) (0,2) (1,1) -> half, (1,0) (2,0) -> upper 16 bits.
// A complex value is stored as its two parts in the reduced type.  One wave per storage group;
// any group size of a 64-wide scheme (max_block_size <= 32).
namespace gkoc {
namespace {

template <typename T>
struct is_cplx {
    static constexpr bool value = false;
};
template <typename R>
struct is_cplx<gkoc_cplx<R>> {
    static constexpr bool value = true;
};

// precision_reduction byte -> storage kind in the codes of stored<> (0 = the value type itself)
template <typename R>
__host__ __device__ __forceinline__ int storage_kind(int p);
template <>
__host__ __device__ __forceinline__ int storage_kind<double>(int p)
{
    return (p == 0x01 || p == 0x02 || p == 0x10 || p == 0x11 || p == 0x20) ? p : 0;
}
template <>
__host__ __device__ __forceinline__ int storage_kind<float>(int p)
{
    return (p == 0x01 || p == 0x02 || p == 0x11) ? 0x02 : (p == 0x10 || p == 0x20) ? 0x11 : 0;
}

template <typename R>
__device__ __forceinline__ R load_part(int kind, const void* group, int64_t i)
{
    if (kind == 0) return reinterpret_cast<const R*>(group)[i];
    return R(load_stored(kind, reinterpret_cast<const double*>(group), i));
}
template <typename R>
__device__ __forceinline__ void store_part(int kind, void* group, int64_t i, R v)
{
    if (kind == 0) {
        reinterpret_cast<R*>(group)[i] = v;
    } else {
        store_stored(kind, reinterpret_cast<double*>(group), i, double(v));
    }
}
template <typename R>
__device__ __forceinline__ R round_part(int kind, R v)
{
    switch (kind) {
    case 0x01: return R(stored<0x01>::load(stored<0x01>::store(double(v))));
    case 0x02: return R(stored<0x02>::load(stored<0x02>::store(double(v))));
    case 0x10: return R(stored<0x10>::load(stored<0x10>::store(double(v))));
    case 0x11: return R(stored<0x11>::load(stored<0x11>::store(double(v))));
    case 0x20: return R(stored<0x20>::load(stored<0x20>::store(double(v))));
    default: return v;
    }
}

template <typename R>
__device__ __forceinline__ R load_value(int kind, const R* group, int64_t idx)
{
    return load_part<R>(kind, group, idx);
}
template <typename R>
__device__ __forceinline__ gkoc_cplx<R> load_value(int kind, const gkoc_cplx<R>* group, int64_t idx)
{
    return {load_part<R>(kind, group, 2 * idx), load_part<R>(kind, group, 2 * idx + 1)};
}
template <typename R>
__device__ __forceinline__ void store_value(int kind, R* group, int64_t idx, R v)
{
    store_part<R>(kind, group, idx, v);
}
template <typename R>
__device__ __forceinline__ void store_value(int kind, gkoc_cplx<R>* group, int64_t idx, gkoc_cplx<R> v)
{
    store_part<R>(kind, group, 2 * idx, v.re);
    store_part<R>(kind, group, 2 * idx + 1, v.im);
}
template <typename R>
__device__ __forceinline__ R round_value(int kind, R v)
{
    return round_part<R>(kind, v);
}
template <typename R>
__device__ __forceinline__ gkoc_cplx<R> round_value(int kind, gkoc_cplx<R> v)
{
    return {round_part<R>(kind, v.re), round_part<R>(kind, v.im)};
}

// the unit round-offs get_supported_storage_reductions compares with (jacobi_utils.hpp:118-146;
// float_traits<>::eps, core/base/extended_float.hpp): p2n0, p1n1, p0n2, p1n0, p0n1, the value
// type's own, and the storage kinds the two verificators round to
template <typename R>
struct reduction_rules;
template <>
struct reduction_rules<double> {
    static constexpr double p2n0 = 1.0 / 16, p1n1 = 1.0 / 128, p0n2 = 1.0 / 2048, p1n0 = 1.0 / 1048576,
                            p0n1 = 1.0 / 16777216, own = 1.0 / 9007199254740992.0;
    static constexpr int verify1 = 0x01, verify2 = 0x02;
};
template <>
struct reduction_rules<float> {
    static constexpr float p2n0 = 1.0f / 128, p1n1 = 1.0f / 2048, p0n2 = 1.0f / 2048, p1n0 = 1.0f / 128,
                           p0n1 = 1.0f / 2048, own = 1.0f / 16777216;
    static constexpr int verify1 = 0x02, verify2 = 0x02;
};

// compute_inf_norm as block_norm_lds above, any value type (|z| of a complex entry)
template <typename T>
__device__ __forceinline__ real_t<T> block_norm_any(const T* Bm, int ld, int bs, int r, int sub)
{
    using R = real_t<T>;
    R t = R(0);
    if (r < bs) {
        for (int j = 0; j < bs; ++j) t += abs_v(Bm[j * ld + r]);
    }
    for (int off = 1; off < sub; off <<= 1) {
        const R o = __shfl_xor(t, off, 64);
        t = o > t ? o : t;
    }
    return t;
}

// validate_precision_reduction_feasibility (reference :280-307) for a storage kind
template <typename T>
__device__ __forceinline__ bool feasible_any(int kind, const T* Bm, T* Tm, int ld, int bs, int r, int g,
                                             int sub, int max_bs)
{
    using R = real_t<T>;
    if (r < bs) {
        for (int j = 0; j < bs; ++j) Tm[r * ld + j] = round_value(kind, Bm[r * ld + j]);
    }
    wave_lds_sync();
    R cond = block_norm_any<T>(Tm, ld, bs, r, sub);
    int perm = r;
    const bool ok = gauss_jordan_lds<T>(Tm, ld, bs, r, g, sub, max_bs, perm);
    cond *= block_norm_any<T>(Tm, ld, bs, r, sub);
    wave_lds_sync();
    return ok && cond >= R(1) && cond * reduction_rules<R>::own < R(1e-3);
}

// dynamic LDS: 2 * (64/SUB) * SUB * ld values
template <typename T, typename I>
__global__ __launch_bounds__(64) void jacobi_generate_adaptive_any_kernel(
    const I* __restrict__ row_ptrs, const I* __restrict__ cols, const T* __restrict__ vals,
    int64_t num_blocks, int sub, int ld, gkoc_jacobi_scheme scheme, const I* __restrict__ block_ptrs,
    real_t<T> accuracy, uint8_t* __restrict__ precisions, real_t<T>* __restrict__ conditioning,
    T* __restrict__ blocks)
{
    using R = real_t<T>;
    using rules = reduction_rules<R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);
    const int lane = threadIdx.x;
    const int per_wave = 64 / sub;
    const int g = lane / sub;
    const int r = lane % sub;
    T* Bm = lds + int64_t(g) * sub * ld;
    T* Tm = lds + int64_t(per_wave) * sub * ld + int64_t(g) * sub * ld;
    const int64_t blk = int64_t(blockIdx.x) * per_wave + g;
    int64_t start = 0;
    int bs = 0;
    if (blk < num_blocks) {
        start = block_ptrs[blk];
        bs = int(block_ptrs[blk + 1] - start);
    }
    if (r < bs) {
        for (int j = 0; j < bs; ++j) Bm[r * ld + j] = T(0);
        const int64_t a = row_ptrs[start + r], e = row_ptrs[start + r + 1];
        for (int64_t k = a; k < e; ++k) {
            const int64_t c = int64_t(cols[k]) - start;
            if (c >= 0 && c < bs) Bm[r * ld + c] = vals[k];
        }
    }
    const int max_bs = wave_max(bs);
    wave_lds_sync();
    R cond = block_norm_any<T>(Bm, ld, bs, r, sub);
    int perm = r;
    gauss_jordan_lds<T>(Bm, ld, bs, r, g, sub, max_bs, perm);
    cond *= block_norm_any<T>(Bm, ld, bs, r, sub);
    // autodetect needs the condition numbers (reference :347: "... && cond"): without the array the
    // request reads as "keep the value type"
    int request = blk < num_blocks ? int(precisions[blk]) : -1;
    if (request == 0xff && conditioning == nullptr) request = 0;
    uint32_t desc = 0xffffffffu;
    const bool any_auto = __ballot(request == 0xff) != 0;
    bool v1 = false, v2 = false;
    if (any_auto) {
        v1 = feasible_any<T>(rules::verify1, Bm, Tm, ld, bs, r, g, sub, max_bs);
        v2 = rules::verify2 == rules::verify1
                 ? v1
                 : feasible_any<T>(rules::verify2, Bm, Tm, ld, bs, r, g, sub, max_bs);
    }
    if (request == 0xff) {
        int verified1 = 2;
        desc = 0;
        if (cond * rules::p2n0 < accuracy) desc |= 0x04;
        if (cond * rules::p1n1 < accuracy) {
            verified1 = v1 ? 1 : 0;
            if (v1) desc |= 0x02;
        }
        if (cond * rules::p0n2 < accuracy && verified1 != 0 && v2) desc |= 0x01;
        if (cond * rules::p1n0 < accuracy) desc |= 0x10;
        if (cond * rules::p0n1 < accuracy) {
            if (verified1 == 2) verified1 = v1 ? 1 : 0;
            if (verified1 == 1) desc |= 0x08;
        }
    } else if (request >= 0) {
        desc = request == 0x01 ? 0x08u : request == 0x02 ? 0x01u : request == 0x10 ? 0x10u
             : request == 0x11 ? 0x02u : request == 0x20 ? 0x04u : 0u;
    }
    for (int off = 1; off < 64; off <<= 1) desc &= __shfl_xor(desc, off, 64);
    const int p = (desc & 0x01) ? 0x02 : (desc & 0x02) ? 0x11 : (desc & 0x04) ? 0x20
                : (desc & 0x08) ? 0x01 : (desc & 0x10) ? 0x10 : 0x00;
    if (blk < num_blocks && r == 0) {
        precisions[blk] = uint8_t(p);
        if (conditioning) conditioning[blk] = cond;
    }
    const int kind = storage_kind<R>(p);
    const int64_t gsize = int64_t(1) << scheme.group_power;
    const int64_t stride = scheme.block_offset << scheme.group_power;
    T* group = blocks + scheme.group_offset * (blk >> scheme.group_power);
    const int64_t boff = scheme.block_offset * (blk & (gsize - 1));
    for (int j = 0; j < max_bs; ++j) {
        const int pj = __shfl(perm, g * sub + j, 64);
        if (r < bs && j < bs) store_value(kind, group, boff + r + int64_t(pj) * stride, Bm[r * ld + j]);
    }
}

template <typename T, typename I>
int launch_generate_adaptive_any(gkoc_stream_t s, const I* row_ptrs, const I* cols, const T* vals,
                                 int64_t num_blocks, uint32_t max_bs, gkoc_jacobi_scheme scheme,
                                 const I* block_ptrs, real_t<T> accuracy, uint8_t* precisions,
                                 real_t<T>* conditioning, T* blocks)
{
    if (num_blocks <= 0) return GKOC_OK;
    GKOC_REQUIRE(row_ptrs && cols && vals && block_ptrs && precisions && blocks, GKOC_E_INVALID,
                 "null pointer");
    GKOC_REQUIRE(wave_group_layout(scheme) && max_bs >= 1 && max_bs <= uint64_t(scheme.block_offset),
                 GKOC_E_NOT_SUPPORTED,
                 "adaptive block-Jacobi needs max_block_size <= 32 and groups that fill a wavefront "
                 "(max_block_stride 64)");
    const int sub = subwarp_of(scheme);
    const int per_wave = 64 / sub;
    // rows padded by one entry against bank conflicts where two blocks per lane group fit in 64 KB
    int ld = sub + 1;
    if (2 * size_t(per_wave) * sub * ld * sizeof(T) > 65536) ld = sub;
    const size_t lds = 2 * size_t(per_wave) * sub * ld * sizeof(T);
    jacobi_generate_adaptive_any_kernel<T, I>
        <<<dim3(unsigned(ceildiv(num_blocks, per_wave))), dim3(64), lds, as_stream(s)>>>(
            row_ptrs, cols, vals, num_blocks, sub, ld, scheme, block_ptrs, accuracy, precisions,
            conditioning, blocks);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// ---- round 6: lane = (block, row) for every value type and storage precision -------------------------
// The thread-per-row kernels above (jacobi_apply_simple_kernel, jacobi_apply_adaptive_any_kernel) need a
// row -> block table (a pass of its own plus a device-to-host copy of the row count) and read the block with
// one strided load per entry and lane: L256, block size 8: float adaptive 373 us = 13.5 % of 8 TB/s,
// complex<double> adaptive 582 us = 23 %, complex<double> full storage 586 us = 57 %
// (profiles/r06/r06_round5_additions.txt).  This is the scheme of jacobi_apply_fixed_kernel for them:
// one wave per storage group (GPW groups per wave), lane = (block, row) = the group's interleaved storage
// order, so column c of all blocks of the group is ONE contiguous run (lane + c * stride) in whatever type
// the group is stored in; the block's b values come from the block's other lanes by shuffle instead of
// `bs` gathers per lane; all loads of a group are requested before the first use.  SUB = the lanes of a
// block (the power of two at or above block_offset), so every 64-wide scheme is covered, block_offset 13
// included.  Per (row, column) the operations and their order are those of the thread-per-row kernels:
// sum = 0, sum += B(row, c) * b(c) for c = 0 .. bs-1, then alpha * sum (+ beta * x) - the same bits.
template <typename T, int KIND>
struct entry_loader {
    using R = real_t<T>;
    using S = typename stored<KIND>::type;
    __device__ static __forceinline__ T get(const T* group, int64_t idx)
    {
        if constexpr (is_cplx<T>::value) {
            const S* sp = reinterpret_cast<const S*>(group) + 2 * idx;
            const S re = sp[0], im = sp[1];
            return T(R(stored<KIND>::load(re)), R(stored<KIND>::load(im)));
        } else {
            return T(stored<KIND>::load(reinterpret_cast<const S*>(group)[idx]));
        }
    }
};

template <typename T, typename I, bool ADV, int SUB, int GPW>
__global__ __launch_bounds__(256) void jacobi_apply_lanes_any_kernel(
    int64_t num_blocks, int64_t num_groups, gkoc_jacobi_scheme scheme, const I* __restrict__ block_ptrs,
    const T* __restrict__ blocks, const uint8_t* __restrict__ precisions, const T* __restrict__ alpha_p,
    const T* __restrict__ b, int64_t ldb, const T* __restrict__ beta_p, T* __restrict__ x, int64_t ldx,
    int nrhs, int xcd_map)
{
    using R = real_t<T>;
    const int lane = threadIdx.x & 63;
    const int bo = int(scheme.block_offset);
    const int stride = bo << scheme.group_power;          // lanes of a group that hold a row
    const int big = lane / bo;                            // block within the group
    const int r = lane - big * bo;
    const int lane0 = lane - r;
    int64_t wg = blockIdx.x;
    if (xcd_map) {
        const int64_t nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7;
        const int64_t xcd = wg & 7, slot = wg >> 3;
        wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
    }
    const int64_t group0 = (wg * 4 + (threadIdx.x >> 6)) * GPW;
    T m[GPW][SUB];
    int64_t row[GPW];
    int bs[GPW];
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int64_t group = group0 + g;
        const int64_t blk = (group << scheme.group_power) + big;
        const bool have = lane < stride && group < num_groups && blk < num_blocks;
        I start = 0, end = 0;
        if (have) {
            start = block_ptrs[blk];
            end = block_ptrs[blk + 1];
        }
        bs[g] = have && r < int(end - start) ? int(end - start) : 0;
        row[g] = int64_t(start) + r;
    }
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int64_t group = group0 + g;
        const T* gp = blocks + scheme.group_offset * (group < num_groups ? group : num_groups - 1);
        const int ln = lane < stride ? lane : 0;
        int kind = 0;
        if (precisions != nullptr && bs[g] > 0) {
            kind = storage_kind<R>(int(precisions[(group << scheme.group_power) + big]));
        }
        // one storage type per group (generate gives a group ONE precision): decide once, typed loads;
        // a caller's array that mixes types inside a group is read entry by entry
        const unsigned long long act = __ballot(bs[g] > 0);
        const int k0 = act ? __shfl(kind, __builtin_ctzll(act), 64) : 0;
        const bool uniform = __ballot(bs[g] > 0 && kind != k0) == 0;
        // every load unconditional (a conditional load is waited for before the next one is issued) and
        // in bounds: a lane without a row reads lane 0's entries, a column past the block's last one reads the
        // last one again; what they deliver is never used
#define GKOC_LOAD_ENTRIES(K_)                                                                   \
    _Pragma("unroll") for (int c = 0; c < SUB; ++c)                                             \
    {                                                                                           \
        m[g][c] = entry_loader<T, K_>::get(gp, ln + int64_t(c < bo ? c : bo - 1) * stride);     \
    }
        if (!uniform) {
#pragma unroll
            for (int c = 0; c < SUB; ++c) {
                m[g][c] = load_value(kind, gp, ln + int64_t(c < bo ? c : bo - 1) * stride);
            }
        } else if (k0 == 0) {
#pragma unroll
            for (int c = 0; c < SUB; ++c) m[g][c] = gp[ln + int64_t(c < bo ? c : bo - 1) * stride];
        } else if (k0 == 0x02) {
            GKOC_LOAD_ENTRIES(0x02)
        } else if (k0 == 0x11) {
            GKOC_LOAD_ENTRIES(0x11)
        } else if constexpr (sizeof(R) == 8) {
            if (k0 == 0x01) {
                GKOC_LOAD_ENTRIES(0x01)
            } else if (k0 == 0x10) {
                GKOC_LOAD_ENTRIES(0x10)
            } else {
                GKOC_LOAD_ENTRIES(0x20)
            }
        }
#undef GKOC_LOAD_ENTRIES
    }
    T alpha = T(R(1)), beta = T(R(0));
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    for (int j = 0; j < nrhs; ++j) {
        T bv[GPW], xv[GPW];
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
            bv[g] = T(R(0));
            xv[g] = T(R(0));
            if (bs[g] > 0) {
                bv[g] = b[row[g] * ldb + j];
                if (ADV && beta != T(R(0))) xv[g] = x[row[g] * ldx + j];
            }
        }
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
            T sum = T(R(0));
#pragma unroll
            for (int c = 0; c < SUB; ++c) {
                const T bc = __shfl(bv[g], lane0 + c, 64);
                if (c < bs[g]) sum += m[g][c] * bc;
            }
            if (bs[g] > 0) {
                if (ADV) {
                    const T ax = alpha * sum;
                    x[row[g] * ldx + j] = beta == T(R(0)) ? ax : ax + beta * xv[g];
                } else {
                    x[row[g] * ldx + j] = sum;
                }
            }
        }
    }
}

// true: launched (the scheme fills a wavefront per group); false: the caller's thread-per-row path
template <typename T, typename I, bool ADV>
bool launch_apply_lanes_any(hipStream_t st, int64_t num_blocks, gkoc_jacobi_scheme scheme, const I* block_ptrs,
                            const T* blocks, const uint8_t* precisions, const T* alpha, const T* b, int64_t ldb,
                            const T* beta, T* x, int64_t ldx, int64_t nrhs)
{
    if (!wave_group_layout(scheme) || nrhs > 0x7fffffff || b == x) return false;
    const int64_t groups = ceildiv(num_blocks, int64_t(1) << scheme.group_power);
#define GKOC_JAC_LANES(SUB_, GPW_)                                                                       \
    {                                                                                                    \
        const int64_t nwg = ceildiv(groups, 4 * GPW_);                                                   \
        jacobi_apply_lanes_any_kernel<T, I, ADV, SUB_, GPW_><<<dim3(unsigned(nwg)), dim3(256), 0, st>>>( \
            num_blocks, groups, scheme, block_ptrs, blocks, precisions, alpha, b, ldb, beta, x, ldx,     \
            int(nrhs), jacobi_xcd_map(nwg));                                                             \
    }
    // groups per wave: what a wave keeps in flight is SUB loads of 64 stored entries per group - with half
    // storage 128 B each; four groups for the value types of up to eight bytes, two / one for complex<double>
    constexpr int G8 = sizeof(T) <= 8 ? 4 : 2, G16 = sizeof(T) <= 8 ? 2 : 1;
    switch (subwarp_of(scheme)) {
    case 1: GKOC_JAC_LANES(1, G8) break;
    case 2: GKOC_JAC_LANES(2, G8) break;
    case 4: GKOC_JAC_LANES(4, G8) break;
    case 8: GKOC_JAC_LANES(8, G8) break;
    case 16: GKOC_JAC_LANES(16, G16) break;
    default: GKOC_JAC_LANES(32, 1) break;
    }
#undef GKOC_JAC_LANES
    return true;
}

// x = M b / x = alpha M b + beta x, one thread per (row, right-hand side), blocks widened on load
// (reference apply_block with the resolved precision, :425-470)
template <typename T, typename I, bool ADV>
__global__ __launch_bounds__(256) void jacobi_apply_adaptive_any_kernel(
    int64_t n_rows, int64_t nrhs, gkoc_jacobi_scheme scheme, const I* __restrict__ block_ptrs,
    const I* __restrict__ row_block, const T* __restrict__ blocks,
    const uint8_t* __restrict__ precisions, const T* __restrict__ alpha_p, const T* __restrict__ b,
    int64_t ldb, const T* __restrict__ beta_p, T* __restrict__ x, int64_t ldx)
{
    using R = real_t<T>;
    const int64_t total = n_rows * nrhs, step = int64_t(gridDim.x) * 256;
    const int64_t stride = scheme.block_offset << scheme.group_power;
    const int64_t mask = (int64_t(1) << scheme.group_power) - 1;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += step) {
        const int64_t j = idx / n_rows, row = idx - j * n_rows;
        const int64_t blk = row_block[row];
        const int64_t start = block_ptrs[blk], bs = int64_t(block_ptrs[blk + 1]) - start;
        const int kind = precisions ? storage_kind<R>(int(precisions[blk])) : 0;
        const T* group = blocks + scheme.group_offset * (blk >> scheme.group_power);
        const int64_t first = scheme.block_offset * (blk & mask) + (row - start);
        T sum = T(0);
        for (int64_t c = 0; c < bs; ++c) {
            sum += load_value(kind, group, first + c * stride) * b[(start + c) * ldb + j];
        }
        if (ADV) {
            const T beta = beta_p[0];
            const T ax = alpha_p[0] * sum;
            x[row * ldx + j] = beta == T(0) ? ax : ax + beta * x[row * ldx + j];
        } else {
            x[row * ldx + j] = sum;
        }
    }
}

template <typename T, typename I, bool ADV>
int launch_apply_adaptive_any(gkoc_stream_t s, int64_t num_blocks, gkoc_jacobi_scheme scheme,
                              const I* block_ptrs, const T* blocks, const uint8_t* precisions,
                              const T* alpha, const T* b, int64_t ldb, const T* beta, T* x, int64_t ldx,
                              int64_t nrhs)
{
    if (num_blocks <= 0 || nrhs <= 0) return GKOC_OK;
    GKOC_REQUIRE(block_ptrs && blocks && b && x, GKOC_E_INVALID, "null pointer");
    hipStream_t st = as_stream(s);
    if (tune_value(GKOC_TUNE_JACOBI_LANES) != 1 &&
        launch_apply_lanes_any<T, I, ADV>(st, num_blocks, scheme, block_ptrs, blocks, precisions, alpha, b, ldb,
                                          beta, x, ldx, nrhs)) {
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    I last = 0;
    GKOC_HIP(hipMemcpyAsync(&last, block_ptrs + num_blocks, sizeof(I), hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipStreamSynchronize(st));
    const int64_t n_rows = int64_t(last);
    if (n_rows <= 0) return GKOC_OK;
    I* row_block = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&row_block), size_t(n_rows) * sizeof(I)));
    jacobi_row_block_kernel<I><<<dim3(unsigned(ceildiv(num_blocks, 256))), dim3(256), 0, st>>>(
        num_blocks, block_ptrs, row_block);
    int64_t nb = ceildiv(n_rows * nrhs, 256);
    if (nb > 8 * max_stream_blocks) nb = 8 * max_stream_blocks;
    jacobi_apply_adaptive_any_kernel<T, I, ADV><<<dim3(unsigned(nb)), dim3(256), 0, st>>>(
        n_rows, nrhs, scheme, block_ptrs, row_block, blocks, precisions, alpha, b, ldb, beta, x, ldx);
    const hipError_t e = hipGetLastError();
    (void)scratch_free(st, row_block);
    GKOC_HIP(e);
    return GKOC_OK;
}

// out block = (conjugate) transpose of the block, in the storage type of its group
// (reference transpose_jacobi / conj_transpose_jacobi, :528-600: the reduced values move as they are)
template <typename T, typename I>
__global__ __launch_bounds__(256) void jacobi_transpose_adaptive_any_kernel(
    int64_t num_blocks, gkoc_jacobi_scheme scheme, const I* __restrict__ block_ptrs,
    const T* __restrict__ blocks, const uint8_t* __restrict__ precisions, int conj, T* __restrict__ out)
{
    using R = real_t<T>;
    const int64_t bo = scheme.block_offset;
    const int64_t stride = bo << scheme.group_power;
    const int64_t gmask = (int64_t(1) << scheme.group_power) - 1;
    const int64_t total = num_blocks * bo;
    const int64_t step = int64_t(gridDim.x) * 256;
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < total; t += step) {
        const int64_t blk = t / bo;
        const int r = int(t - blk * bo);
        const int bs = int(block_ptrs[blk + 1] - block_ptrs[blk]);
        if (r >= bs) continue;
        const int kind = precisions ? storage_kind<R>(int(precisions[blk])) : 0;
        const int64_t goff = scheme.group_offset * (blk >> scheme.group_power);
        const int64_t base = bo * (blk & gmask);
        for (int c = 0; c < bs; ++c) {
            T v = load_value(kind, blocks + goff, base + c + int64_t(r) * stride);   // in(c, r)
            if (conj) v = conj_v(v);
            store_value(kind, out + goff, base + r + int64_t(c) * stride, v);        // out(r, c)
        }
    }
}

template <typename T, typename I>
int launch_transpose_adaptive_any(gkoc_stream_t s, int64_t num_blocks, gkoc_jacobi_scheme scheme,
                                  const I* block_ptrs, const T* blocks, const uint8_t* precisions,
                                  int conj, T* out)
{
    if (num_blocks <= 0) return GKOC_OK;
    GKOC_REQUIRE(block_ptrs && blocks && out, GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(scheme.block_offset >= 1, GKOC_E_INVALID, "storage scheme");
    int64_t nb = ceildiv(num_blocks * scheme.block_offset, 256);
    if (nb > 8 * max_stream_blocks) nb = 8 * max_stream_blocks;
    jacobi_transpose_adaptive_any_kernel<T, I><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(
        num_blocks, scheme, block_ptrs, blocks, precisions, conj, out);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

#define GKOC_DEF_JACOBI_ADAPTIVE_ANY(T, R, TN, I, IN)                                               \
    extern "C" int gkoc_jacobi_generate_adaptive_##TN##_##IN(                                       \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* col_idxs, const T* vals,       \
        int64_t num_blocks, uint32_t max_block_size, gkoc_jacobi_scheme scheme, const I* block_ptrs, \
        R accuracy, uint8_t* precisions, R* conditioning, T* blocks)                                \
    {                                                                                               \
        (void)n_rows;                                                                               \
        return launch_generate_adaptive_any<T, I>(s, row_ptrs, col_idxs, vals, num_blocks,          \
                                                  max_block_size, scheme, block_ptrs, accuracy,     \
                                                  precisions, conditioning, blocks);                \
    }                                                                                               \
    extern "C" int gkoc_jacobi_apply_adaptive_##TN##_##IN(                                          \
        gkoc_stream_t s, int64_t num_blocks, uint32_t max_block_size, gkoc_jacobi_scheme scheme,    \
        const I* block_ptrs, const T* blocks, const uint8_t* precisions, const T* alpha, const T* b, \
        int64_t ldb, const T* beta, T* x, int64_t ldx, int64_t nrhs)                                \
    {                                                                                               \
        (void)max_block_size;                                                                       \
        GKOC_REQUIRE((alpha == nullptr) == (beta == nullptr), GKOC_E_INVALID,                       \
                     "pass alpha and beta, or neither");                                            \
        if (alpha) {                                                                                \
            return launch_apply_adaptive_any<T, I, true>(s, num_blocks, scheme, block_ptrs, blocks, \
                                                         precisions, alpha, b, ldb, beta, x, ldx,   \
                                                         nrhs);                                     \
        }                                                                                           \
        return launch_apply_adaptive_any<T, I, false>(s, num_blocks, scheme, block_ptrs, blocks,    \
                                                      precisions, nullptr, b, ldb, nullptr, x, ldx, \
                                                      nrhs);                                        \
    }                                                                                               \
    extern "C" int gkoc_jacobi_transpose_adaptive_##TN##_##IN(                                      \
        gkoc_stream_t s, int64_t num_blocks, gkoc_jacobi_scheme scheme, const I* block_ptrs,        \
        const T* blocks, const uint8_t* precisions, int conj, T* out_blocks)                        \
    {                                                                                               \
        return launch_transpose_adaptive_any<T, I>(s, num_blocks, scheme, block_ptrs, blocks,       \
                                                   precisions, conj, out_blocks);                   \
    }
GKOC_DEF_JACOBI_ADAPTIVE_ANY(float, float, f32, int32_t, i32)
GKOC_DEF_JACOBI_ADAPTIVE_ANY(float, float, f32, int64_t, i64)
GKOC_DEF_JACOBI_ADAPTIVE_ANY(gkoc_c128, double, c128, int32_t, i32)
GKOC_DEF_JACOBI_ADAPTIVE_ANY(gkoc_c128, double, c128, int64_t, i64)
GKOC_DEF_JACOBI_ADAPTIVE_ANY(gkoc_c64, float, c64, int32_t, i32)
GKOC_DEF_JACOBI_ADAPTIVE_ANY(gkoc_c64, float, c64, int64_t, i64)
