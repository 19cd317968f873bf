// CSR SpMV for gfx950: software-pipelined row-segment-per-wavefront kernel.
//
// Replaces gko::kernels::hip::csr::{spmv, advanced_spmv}
// (decl core/matrix/csr_kernels.hpp:29-43; semantics
// reference/matrix/csr_kernels.cpp:49-118; stock GPU version
// common/cuda_hip/matrix/csr_kernels.template.cpp:206-586,2351-2468).
//
// The kernel lives in csr_spmv_pipe.hpp (variant 3).  Summary:
//  * one 64-lane wavefront owns 64 consecutive rows = one contiguous range of
//    the val / col_idx streams, read with 16-byte-per-lane vector loads (no
//    per-row alignment loss, no idle lanes on 27-nnz rows), double-buffered in
//    registers so the HBM round trip overlaps the b-vector gather;
//  * products val[k]*b[col[k]] go to an LDS ring (8 KB per wave, 20 waves/CU);
//  * lane = row then adds its products from LDS in k order => same summation
//    order and roundings as the sequential reference (this file is compiled
//    with -ffp-contract=off): BIT-IDENTICAL results, no atomics, no zeroing
//    pass over c, no host-built srow table, any strategy name accepted;
//  * rows longer than GKOC_CSR_LONG_ROW are summed cooperatively by the whole
//    wave (tolerance instead of bit-exactness for those rows only).
//
// Algorithmic HBM bytes: nnz*(sizeof(T)+sizeof(I)) + (n+1)*sizeof(I)
//                        + n_cols*sizeof(T) (b once) + n*sizeof(T) (c).
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.hpp"
#include "csr_long_rows.hpp"
#include "csr_spmv_multi.hpp"
#include "csr_spmv_pipe.hpp"
#include "fused.hpp"

namespace gkoc {
namespace {

// ---- which 64-row segments of a matrix hold a row beyond GKOC_CSR_LONG_ROW (csr_long_rows.hpp) --------
// Found by one scan of the row pointers the first time a (device, row_ptrs, n_rows) is multiplied, kept
// here (flags and list read-only from then on; the chunk sums of a product go to a buffer per (matrix, stream),
// so products of one matrix on several streams share nothing that is written); gkoc_free forgets the entries of
// a pointer that goes away (csr_long_rows_forget).  A few dozen matrices at most: the OLDEST entry makes room,
// after a device synchronisation (a product on any stream may still read its flags).  The first product of a
// matrix therefore synchronises its stream once (the scan's answer is needed on the host) - documented in
// gko_cdna4.h; inside a stream capture the matrix is multiplied by the row-segment kernel alone.
std::mutex g_long_mtx;
struct long_entry {
    csr_long_info info;
    // one buffer of chunk sums per stream that has multiplied this matrix (count x 8 x 64 values of 8 bytes)
    std::vector<std::pair<hipStream_t, void*>> partials;
};
std::map<std::tuple<int, const void*, int64_t>, long_entry> g_long_cache;
std::atomic<int> g_long_cached{0};
constexpr size_t long_cache_cap = 128;
constexpr int64_t long_list_cap = 4096;
constexpr int short_rows_default_layout = 0;      // (set by measurement: see launch_csr)      // more flagged segments than this: the matrix has no "few long rows"

thread_local bool t_long_releasing = false;      // gkoc_free below comes back through csr_long_rows_forget

void long_info_release(long_entry& en)
{
    csr_long_info& f = en.info;
    t_long_releasing = true;
    if (f.bits) (void)gkoc_free(f.bits);
    if (f.list) (void)gkoc_free(f.list);
    for (auto& sp : en.partials) (void)gkoc_free(sp.second);
    en.partials.clear();
    f = csr_long_info{};
    t_long_releasing = false;
}

bool stream_is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return true;
    }
    return false;
}

// the calling stream's buffer of chunk sums (allocated the first time the stream multiplies the matrix; not
// inside a stream capture: the product is then left to the row-segment kernel alone, which handles any row)
void long_partial_for(long_entry& en, hipStream_t st, csr_long_info* out)
{
    *out = en.info;
    if (en.info.count <= 0) return;
    for (auto& sp : en.partials) {
        if (sp.first == st) {
            out->partial = sp.second;
            return;
        }
    }
    void* p = nullptr;
    if (stream_is_capturing(st) ||
        gkoc_malloc(&p, size_t(en.info.count) * LONG_MAX_PER_SEG * LONG_PARTS * 8) != GKOC_OK) {
        (void)hipGetLastError();
        *out = csr_long_info{};
        out->nnz = en.info.nnz;
        return;
    }
    en.partials.emplace_back(st, p);
    out->partial = p;
}

template <typename T, typename I>
int long_info_of(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, csr_long_info* out)
{
    int dev = 0;
    GKOC_HIP(hipGetDevice(&dev));
    const auto key = std::make_tuple(dev, static_cast<const void*>(row_ptrs), n_rows);
    std::lock_guard<std::mutex> g(g_long_mtx);
    auto it = g_long_cache.find(key);
    if (it != g_long_cache.end()) {
        long_partial_for(it->second, as_stream(s), out);
        return GKOC_OK;
    }
    // a stream that is being captured into a hipGraph cannot be synchronised: a matrix first seen there is
    // multiplied by the row-segment kernel alone (correct for any row), and looked at on its next product
    if (stream_is_capturing(as_stream(s))) {
        *out = csr_long_info{};
        return GKOC_OK;
    }
    if (g_long_cache.size() >= long_cache_cap) {
        auto oldest = g_long_cache.begin();
        for (auto jt = g_long_cache.begin(); jt != g_long_cache.end(); ++jt) {
            if (jt->second.info.seq < oldest->second.info.seq) oldest = jt;
        }
        if (oldest->second.info.count > 0) GKOC_HIP(hipDeviceSynchronize());   // (a product in flight may use it)
        long_info_release(oldest->second);
        g_long_cache.erase(oldest);
    }
    csr_long_info f;
    static uint64_t arrivals = 0;
    f.seq = ++arrivals;
    const int64_t n_seg = ceildiv(n_rows, int64_t(64));
    const size_t bit_bytes = size_t(ceildiv(n_seg, int64_t(32))) * 4;
    const size_t list_bytes = size_t(1 + long_list_cap) * 8;
    void *bits = nullptr, *list = nullptr;
    GKOC_TRY(gkoc_malloc(&bits, bit_bytes));
    if (gkoc_malloc(&list, list_bytes) != GKOC_OK) {
        t_long_releasing = true;
        (void)gkoc_free(bits);
        t_long_releasing = false;
        return GKOC_E_INVALID;
    }
    hipStream_t st = as_stream(s);
    unsigned long long count = 0;
    I nnz_dev = I(0);
    bool ok = hipMemsetAsync(bits, 0, bit_bytes, st) == hipSuccess && hipMemsetAsync(list, 0, 8, st) == hipSuccess;
    if (ok) {
        csr_long_row_scan_kernel<I><<<dim3(unsigned(ceildiv(n_seg, int64_t(4)))), dim3(256), 0, st>>>(
            n_rows, row_ptrs, static_cast<uint32_t*>(bits), static_cast<unsigned long long*>(list), long_list_cap);
        // (the number of stored entries comes with the same wait: the launcher sizes a wave's share by it)
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(&count, list, 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipMemcpyAsync(&nnz_dev, row_ptrs + n_rows, sizeof(I), hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    if (ok) f.nnz = int64_t(nnz_dev);
    if (!ok || count == 0 || int64_t(count) > long_list_cap) {
        // none (the common case), or so many that "a few long rows" is not what this matrix has: the
        // row-segment kernel does everything, as before
        (void)hipGetLastError();
        t_long_releasing = true;
        (void)gkoc_free(bits);
        (void)gkoc_free(list);
        t_long_releasing = false;
    } else {
        f.count = int64_t(count);
        f.bits = static_cast<uint32_t*>(bits);
        f.list = static_cast<unsigned long long*>(list);
    }
    long_entry& en = g_long_cache[key];
    en.info = f;
    g_long_cached.store(int(g_long_cache.size()));
    long_partial_for(en, st, out);
    return GKOC_OK;
}

}  // namespace

void csr_long_rows_forget(const void* ptr)
{
    if (ptr == nullptr || t_long_releasing || g_long_cached.load(std::memory_order_relaxed) == 0) return;
    std::vector<long_entry> gone;
    {
        std::lock_guard<std::mutex> g(g_long_mtx);
        for (auto it = g_long_cache.begin(); it != g_long_cache.end();) {
            if (std::get<1>(it->first) == ptr) {
                gone.push_back(it->second);
                it = g_long_cache.erase(it);
            } else {
                ++it;
            }
        }
        g_long_cached.store(int(g_long_cache.size()));
    }
    for (auto& en : gone) {
        if (en.info.count > 0) (void)hipDeviceSynchronize();      // a product in flight may still use its buffers
        long_info_release(en);
    }
}

namespace {

template <typename T, typename I, bool ADV, int rows_per_seg = 64>
int launch_csr(gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const T* alpha,
               const I* row_ptrs, const I* col_idxs, const T* vals, const T* b,
               int64_t ldb, const T* beta, T* c, int64_t ldc, int64_t nrhs)
{
    GKOC_REQUIRE(n_rows >= 0 && n_cols >= 0 && nrhs >= 0, GKOC_E_INVALID,
                 "negative dimension");
    if (n_rows == 0 || nrhs == 0) return GKOC_OK;
    GKOC_REQUIRE(row_ptrs && c, GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(ldc >= nrhs && (n_cols == 0 || ldb >= nrhs), GKOC_E_INVALID,
                 "stride smaller than nrhs");
    if (ADV) GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha/beta");
    // 64-row segments (lane = row in the row phase).  Large matrices: 2 segments
    // = 128 rows per wavefront, whose 1 KB of results is written in one burst at
    // the end of the wave (mode 0x2000).  Below 65536 segments (4 M rows) the
    // grid is only a few rounds of the 5120 resident waves deep and the tail
    // dominates: one segment per wave then (+1 % at 2 M rows, +7 % at 0.9 M,
    // +15 % at 0.26 M rows).  In-order dispatch keeps the set of resident waves
    // on a compact window of rows, which is what lets the b-vector lines shared
    // by neighbouring rows hit in L2.
    const int64_t n_seg = ceildiv(n_rows, rows_per_seg);
    // (round 3, measured again on 1 / 2 / 4 / 16.7 M rows, profiles/r03_experiments.txt: forcing one or
    // two segments per wave or 32-row segments changes nothing or loses: 2.1 M rows 139.8 us with
    // this rule, 142-147 us with the alternatives)
    const int segs_per_wave = n_seg < 65536 ? 1 : 2;
    const int64_t n_waves = ceildiv(n_seg, segs_per_wave);
    GKOC_REQUIRE(n_waves < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED,
                 "more than 2^31 row segments");
    dim3 grid(static_cast<unsigned>(n_waves)), block(64);     // (the single-column path re-sizes it below)
    // 32 B of values per lane and load: 4 doubles or 8 floats (ring = 8 KB)
    constexpr int EV = 32 / sizeof(T);
    constexpr int RINGV = 8192 / sizeof(T);
    // EV-element vector loads need the array bases aligned like the vector types
    // (vecT<T,EV> / vecT<I,EV>: true for whole allocations; sub-views that are not
    // take the scalar-load instantiation E = 1)
    const bool vec_ok =
        reinterpret_cast<uintptr_t>(vals) % (EV * sizeof(T)) == 0 &&
        reinterpret_cast<uintptr_t>(col_idxs) % (EV * sizeof(I)) == 0;
    if (nrhs >= 3) {
        // several right-hand sides, fragment layout (csr_spmv_frag_kernel in csr_spmv_multi.hpp): the
        // lanes of a gather cover whole rows of b.  Small waves - 16 rows, 6 KB of LDS - so that 20+
        // of them are resident per CU.  L256 (profiles/r03_multi_rhs_256.txt): 3 / 4 / 8 columns
        // 2.17 / 2.23 / 5.08 ms with the ring and row-ordered kernels of round 2 -> 1.77 / 1.77 /
        // 1.97 ms.  Measured variants: 32 / 64 rows per wave (12 / 24 KB: 2.2 - 4.6 ms), 2 - 8 entries
        // per step (+- 3 %; 3 for eight columns, 4 for four); two columns stay with the ring kernel
        // below (1.46 ms, this layout 2.36 ms with 32-row waves).
        const bool idx32 = n_cols * ldb < (int64_t(1) << 32);
        const bool pairs_ok = reinterpret_cast<uintptr_t>(b) % (2 * sizeof(T)) == 0 && ldb % 2 == 0 &&
                              reinterpret_cast<uintptr_t>(c) % (2 * sizeof(T)) == 0 && ldc % 2 == 0;
        const int64_t chunk_rows = tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS);
#define GKOC_LAUNCH_CSR_FRAG(NR_, CPL_, KU_) GKOC_LAUNCH_CSR_FRAG_T(NR_, CPL_, 1, KU_)
#define GKOC_LAUNCH_CSR_FRAG_T(NR_, CPL_, TT_, KU_)                                              \
    do {                                                                                         \
        constexpr int rows_ = 64 * CPL_ / NR_ * TT_;                                             \
        const int64_t nwg = ceildiv(n_rows, rows_);                                              \
        GKOC_REQUIRE(nwg < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED, "more than 2^31 waves");    \
        const dim3 gf(static_cast<unsigned>(nwg));                                               \
        if (idx32) {                                                                             \
            csr_spmv_frag_kernel<T, I, ADV, NR_, CPL_, TT_, KU_, true>                           \
                <<<gf, block, 0, as_stream(s)>>>(n_rows, row_ptrs, col_idxs, vals, b, ldb, c,    \
                                                 ldc, static_cast<int>(nrhs), alpha, beta,       \
                                                 chunk_rows / rows_);                            \
        } else {                                                                                 \
            csr_spmv_frag_kernel<T, I, ADV, NR_, CPL_, TT_, KU_, false>                          \
                <<<gf, block, 0, as_stream(s)>>>(n_rows, row_ptrs, col_idxs, vals, b, ldb, c,    \
                                                 ldc, static_cast<int>(nrhs), alpha, beta,       \
                                                 chunk_rows / rows_);                            \
        }                                                                                        \
    } while (0)
        const int64_t variant = tune_value(GKOC_TUNE_CSR_MULTI_VARIANT);
#define GKOC_LAUNCH_CSR_FRAG_PIPE(NR_, CPL_, KU_, SEGS_) GKOC_LAUNCH_CSR_FRAG_PIPE_ST(NR_, CPL_, KU_, SEGS_, 1)
#define GKOC_LAUNCH_CSR_FRAG_PIPE_ST(NR_, CPL_, KU_, SEGS_, ST_) GKOC_LAUNCH_CSR_FRAG_PIPE_NB(NR_, CPL_, KU_, SEGS_, ST_, false)
#define GKOC_LAUNCH_CSR_FRAG_PIPE_NB(NR_, CPL_, KU_, SEGS_, ST_, NB_)                            \
    do {                                                                                         \
        constexpr int rows_ = 64 * CPL_ / NR_;                                                   \
        const int64_t nwg = ceildiv(ceildiv(n_rows, rows_), SEGS_);                              \
        GKOC_REQUIRE(nwg < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED, "more than 2^31 waves");    \
        const dim3 gf(static_cast<unsigned>(nwg));                                               \
        if (idx32) {                                                                             \
            csr_spmv_frag_pipe_kernel<T, I, ADV, NR_, CPL_, KU_, true, ST_, NB_>                 \
                <<<gf, block, 0, as_stream(s)>>>(n_rows, row_ptrs, col_idxs, vals, b, ldb, c,    \
                                                 ldc, static_cast<int>(nrhs), alpha, beta,       \
                                                 SEGS_, chunk_rows / (rows_ * SEGS_));           \
        } else {                                                                                 \
            csr_spmv_frag_pipe_kernel<T, I, ADV, NR_, CPL_, KU_, false, ST_, NB_>                \
                <<<gf, block, 0, as_stream(s)>>>(n_rows, row_ptrs, col_idxs, vals, b, ldb, c,    \
                                                 ldc, static_cast<int>(nrhs), alpha, beta,       \
                                                 SEGS_, chunk_rows / (rows_ * SEGS_));           \
        }                                                                                        \
    } while (0)
        // default (round 5): the pipelined form, four segments per wave - L256 3 / 4 / 8 columns 1.87 / 1.90 /
        // 2.10 -> 1.76 / 1.79 / 2.00 ms (profiles/r05_multi_rhs_pmc.txt); GKOC_TUNE_CSR_MULTI_VARIANT: 9 = the
        // kernel of rounds 3-4, 1 .. 3 its layout variants, >= 10 the pipeline's own (segments per wave x 10
        // + 1000 x entries-per-step choice)
        const int segs = variant >= 10 ? int(variant % 1000 / 10) : 4;
        const bool pipe = variant == 0 || variant >= 10;
        if (pipe && nrhs <= 4) {
            if (variant / 1000 == 5) GKOC_LAUNCH_CSR_FRAG_PIPE_NB(4, 1, 4, segs, 1, true);
            else if (variant / 1000 == 6) GKOC_LAUNCH_CSR_FRAG_PIPE_NB(4, 1, 3, segs, 1, true);
            else if (variant / 1000 == 7) GKOC_LAUNCH_CSR_FRAG_PIPE_NB(4, 1, 2, segs, 1, true);
            else if (variant / 1000 == 1) GKOC_LAUNCH_CSR_FRAG_PIPE(4, 1, 2, segs);
            else if (variant / 1000 == 3) GKOC_LAUNCH_CSR_FRAG_PIPE_ST(4, 1, 3, segs, 3);
            else if (variant / 1000 == 4) GKOC_LAUNCH_CSR_FRAG_PIPE_ST(4, 1, 2, segs, 3);
            else if (variant / 1000 == 2) GKOC_LAUNCH_CSR_FRAG_PIPE(4, 1, 3, segs);
            else GKOC_LAUNCH_CSR_FRAG_PIPE(4, 1, 4, segs);
            GKOC_LAUNCH_OK();
            return GKOC_OK;
        }
        if (pipe && pairs_ok) {
            if (variant / 1000 == 5) GKOC_LAUNCH_CSR_FRAG_PIPE_NB(8, 2, 3, segs, 1, true);
            else if (variant / 1000 == 6) GKOC_LAUNCH_CSR_FRAG_PIPE_NB(8, 2, 4, segs, 1, true);
            else if (variant / 1000 == 7) GKOC_LAUNCH_CSR_FRAG_PIPE_NB(8, 2, 2, segs, 1, true);
            else if (variant / 1000 == 1) GKOC_LAUNCH_CSR_FRAG_PIPE(8, 2, 2, segs);
            else if (variant / 1000 == 3) GKOC_LAUNCH_CSR_FRAG_PIPE_ST(8, 2, 3, segs, 3);
            else if (variant / 1000 == 4) GKOC_LAUNCH_CSR_FRAG_PIPE_ST(8, 2, 2, segs, 3);
            else if (variant / 1000 == 2) GKOC_LAUNCH_CSR_FRAG_PIPE(8, 2, 4, segs);
            else GKOC_LAUNCH_CSR_FRAG_PIPE(8, 2, 3, segs);
            GKOC_LAUNCH_OK();
            return GKOC_OK;
        }
        if (nrhs <= 4 && pairs_ok && variant == 1) {
            GKOC_LAUNCH_CSR_FRAG(4, 2, 3);          // two lanes per row, a pair of columns each: 32-row waves
        } else if (nrhs <= 4 && pairs_ok && variant == 2) {
            GKOC_LAUNCH_CSR_FRAG(4, 2, 4);
        } else if (nrhs <= 4 && pairs_ok && variant == 3) {
            GKOC_LAUNCH_CSR_FRAG(4, 2, 2);
        } else if (nrhs <= 4) {
            GKOC_LAUNCH_CSR_FRAG(4, 1, 4);          // four lanes per row, one column each
        } else if (pairs_ok && variant == 1) {
            GKOC_LAUNCH_CSR_FRAG(8, 2, 4);
        } else if (pairs_ok && variant == 2) {
            GKOC_LAUNCH_CSR_FRAG_T(8, 2, 2, 2);     // 32-row waves, two groups side by side
        } else if (pairs_ok && variant == 3) {
            GKOC_LAUNCH_CSR_FRAG(8, 2, 5);
        } else if (pairs_ok) {
            GKOC_LAUNCH_CSR_FRAG(8, 2, 3);          // four lanes per row, a pair of columns each
        } else {
            GKOC_LAUNCH_CSR_FRAG(8, 1, 4);
        }
#undef GKOC_LAUNCH_CSR_FRAG
#undef GKOC_LAUNCH_CSR_FRAG_T
#undef GKOC_LAUNCH_CSR_FRAG_PIPE
#undef GKOC_LAUNCH_CSR_FRAG_PIPE_ST
#undef GKOC_LAUNCH_CSR_FRAG_PIPE_NB
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    if (nrhs == 2 && vec_ok) {
        // two right-hand sides: the row-segment walk of the single-column kernel with the two
        // products of an entry side by side in the LDS ring (csr_spmv_multi_kernel; 8 KB ring of 512
        // entries).  L256: 1.46 ms against 2.18 ms for one pass per column.
        const int b_vec_ok = reinterpret_cast<uintptr_t>(b) % (2 * sizeof(T)) == 0 && ldb % 2 == 0;
        // two entries per lane and load, two load groups in flight (more, smaller loads under way: the
        // single-column kernel's lesson of round 3): L256 1.457 -> 1.38-1.40 ms = 55 % of 8 TB/s
        // (profiles/r04_experiments.txt); GKOC_TUNE_CSR_LOAD_GROUPS = 1: four entries, one group
        if (tune_value(GKOC_TUNE_CSR_LOAD_GROUPS) == 1) {
            csr_spmv_multi_kernel<T, I, ADV, EV, 1, RINGV / 2, 2><<<grid, block, 0, as_stream(s)>>>(
                n_rows, n_seg, segs_per_wave, row_ptrs, col_idxs, vals, b, ldb, c, ldc,
                static_cast<int>(nrhs), alpha, beta, b_vec_ok,
                tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS) / (64 * segs_per_wave));
        } else {
            csr_spmv_multi_kernel<T, I, ADV, EV / 2, 2, RINGV / 2, 2><<<grid, block, 0, as_stream(s)>>>(
                n_rows, n_seg, segs_per_wave, row_ptrs, col_idxs, vals, b, ldb, c, ldc,
                static_cast<int>(nrhs), alpha, beta, b_vec_ok,
                tune_value(GKOC_TUNE_MULTI_XCD_CHUNK_ROWS) / (64 * segs_per_wave));
        }
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    // rows far longer than the rest: their segments are left out here and done by many workgroups
    // (csr_long_rows.hpp); one right-hand side, 64-row segments
    csr_long_info lng;
    if (nrhs == 1 && rows_per_seg == 64 && tune_value(GKOC_TUNE_CSR_LONG_ROWS) != 0) {
        GKOC_TRY((long_info_of<T, I>(s, n_rows, row_ptrs, &lng)));
    }
    const uint32_t* seg_skip = lng.count > 0 ? lng.bits : nullptr;
    // the hub rows' kernels run BEHIND the row-segment kernel on the caller's stream.  (Round 6 also ran them
    // BESIDE it - a side stream per calling stream, forked in front of the product and joined behind it by two
    // events; the two kernels write disjoint rows of c.  The heavy-tailed stand-in took 245.7 us that way
    // against 231.5 us in sequence, profiles/r06/r06_hub_rows_beside.txt: gone again.)
    auto hub_kernels = [&](hipStream_t st) {
        // (the chunk sums go to the CALLING stream's buffer: long_partial_for)
        csr_flagged_segments_kernel<T, I, ADV><<<dim3(unsigned(lng.count * LONG_PARTS)), dim3(LONG_WG), 0, st>>>(
            n_rows, row_ptrs, col_idxs, vals, b, ldb, c, ldc, alpha, beta, lng.list, static_cast<T*>(lng.partial));
        GKOC_LAUNCH_OK();
        csr_long_rows_fold_kernel<T, I, ADV><<<dim3(unsigned(lng.count)), dim3(64), 0, st>>>(
            n_rows, row_ptrs, c, ldc, beta, lng.list, static_cast<const T*>(lng.partial));
        GKOC_LAUNCH_OK();
        return int(GKOC_OK);
    };
    // GKOC_TUNE_CSR_SEGS_PER_WAVE forces 1 or 2 segments per wave (round 6 tried "up to eight for matrices
    // with short rows" - the heavy-tailed stand-in 256 us with one, 261 with two, 276 with four, 320 with eight
    // segments; 5-pt 4096^2 325 / 289 / 325 / 315; profiles/r06/r06_segments_per_wave.txt: the size rule above
    // stays, and the four / eight-segment variants - whose loop over runs of unflagged segments cost every
    // variant of the kernel 12-16 VGPRs - are gone again)
    int spw = segs_per_wave;
    const int64_t forced_spw = tune_value(GKOC_TUNE_CSR_SEGS_PER_WAVE);
    if (forced_spw == 1 || forced_spw == 2) spw = int(forced_spw);
    const int64_t n_waves_1 = ceildiv(n_seg, spw);
    GKOC_REQUIRE(n_waves_1 < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED, "more than 2^31 row segments");
    grid = dim3(static_cast<unsigned>(n_waves_1));
#define GKOC_LAUNCH_PIPE3(E_, U_, MODE_)                                       \
    csr_spmv_pipe3_kernel<T, I, ADV, rows_per_seg, E_, U_, RINGV, 1, MODE_>    \
        <<<grid, block, 0, as_stream(s)>>>(                                    \
            n_rows, n_seg, spw, row_ptrs, col_idxs, vals, b, ldb, c,           \
            ldc, static_cast<int>(nrhs), alpha, beta, nullptr, xcd_map,        \
            static_cast<const I*>(nullptr), static_cast<int*>(nullptr),        \
            int64_t(0), int64_t(0), static_cast<const uint32_t*>(nullptr),     \
            uint32_t(0), static_cast<const I*>(nullptr),                       \
            static_cast<const I*>(nullptr), static_cast<const T*>(nullptr),    \
            static_cast<uint32_t*>(nullptr), uint32_t(0), 0, int64_t(-1),      \
            seg_skip)
    // XCD-contiguous wave order needs enough waves per XCD to keep the in-order
    // window argument valid; below that the plain order is used
    // ... and is the default for a matrix WITH HUB ROWS (flagged segments): its gathers go all over b, and with
    // one contiguous eighth of the rows per XCD an L2 holds the lines of its own eighth instead of a share of
    // everybody's.  Heavy-tailed stand-in 246.6 -> 226.0 us; the stencils lose 0.5-3 % with it (27-pt 256^3
    // 982 -> 1003, 5-pt 4096^2 274 -> 282) and keep the plain order (profiles/r06/r06_waves_per_workgroup.txt).
    // GKOC_TUNE_CSR_XCD_MAP: 1 = always, 2 = never.
    const int64_t xcd_choice = tune_value(GKOC_TUNE_CSR_XCD_MAP);
    const int xcd_map =
        ((xcd_choice == 1 || (xcd_choice == 0 && lng.count > 0)) && n_waves_1 >= 8 * 1024) ? 1 : 0;
    // Measured and rejected on the Flan-like matrix and on L256 (profiles/r02_experiments,
    // profiles/r02_flan_pmc): 16 / 32 KB rings with 2-4 load groups (fewer resident waves: 299-475 us
    // against 288), 32-row segments (294), one or two entries per lane and load so that neighbouring
    // lanes gather neighbouring columns (301-305).  What helped was the row-phase loop
    // (csr_spmv_pipe.hpp): 295 -> 277 us.
    // Entries per lane and load x load groups in flight (ring 8 KB): 2 x 3 for double, 4 x 2 for
    // float - 16 B of values per lane and load, 48 / 32 B in flight.  Measured in one process on L256
    // (tools/f32_variants.py, profiles/r03_experiments.txt): double 4 x 1 (rounds 1-2) 985 us, 2 x 3
    // 954, 2 x 4 960, 1 x 6 968, 4 x 2 980, 1 x 8 978, 1 x 4 986; float 8 x 1 (rounds 1-2) 847 us,
    // 4 x 2 793, 4 x 3 823, 8 x 2 / 4 x 4 / 2 x 8 slower.  By size (tools/csr_layout_ab.py, double): 4.1 M rows
    // 251 -> 246 us, 8.4 M 510 -> 495, 16.8 M 985 -> 953; the Flan-like matrix (81 per row) 272 both;
    // 5-pt 4096^2 397 -> 405.
    constexpr int PE = sizeof(T) == 8 ? 2 : 4, PU = sizeof(T) == 8 ? 3 : 2;
    // (round 6 also tried two / four / eight WAVES per workgroup, every wave with its own segments and ring - a
    // grid of one-wave workgroups of short-row segments is paced by the dispatcher: 3.5 ns per wave chip-wide,
    // profiles/r06_irregular_pmc.txt.  The pace is per wave, not per workgroup: heavy-tailed stand-in 247 / 241 /
    // 246 / 299 us, every stencil 10-40 % slower; profiles/r06/r06_waves_per_workgroup.txt.)
    if (vec_ok) {
        // below 2 M rows the grid is a few rounds deep and the wider layout of rounds 1-2 is as fast or
        // faster (64^3: 18.4 against 20.7 us; 1 - 2 M rows: equal); key 1 forces it for A/B runs
        // float values and LONG rows (40 and more entries on average; the count comes with the first product's
        // look at the row pointers): ONE entry per lane and load, eight load groups - neighbouring lanes then
        // gather neighbouring entries of a row, whose columns come in runs (a 27-pt stencil with 2 / 3 / 5
        // unknowns per node, 53 / 79 / 130 per row: 172 -> 165, 200 -> 182, 231 -> 207 us against four entries
        // per lane; 7 and 27 per row lose 5-6 % with it, and for double all layouts are within 1.5 %:
        // profiles/r06/r06_load_layouts.txt).  GKOC_TUNE_CSR_LOAD_GROUPS: 3 forces it, 4 = two entries x four
        // groups, 1 = the wide loads.
        const int64_t lay = tune_value(GKOC_TUNE_CSR_LOAD_GROUPS);
        const bool long_float_rows = sizeof(T) == 4 && lng.nnz > 0 && lng.nnz >= 40 * n_rows;
        if (lay == 3 || (lay == 0 && long_float_rows)) {
            if (spw == 2) GKOC_LAUNCH_PIPE3(1, 8, 0x2000);
            else GKOC_LAUNCH_PIPE3(1, 8, 0x1000);
        } else if (lay == 4) {
            if (spw == 2) GKOC_LAUNCH_PIPE3(2, 4, 0x2000);
            else GKOC_LAUNCH_PIPE3(2, 4, 0x1000);
        } else if (tune_value(GKOC_TUNE_CSR_LOAD_GROUPS) == 1 || n_seg < 32768) {
            if (spw == 2) {
                GKOC_LAUNCH_PIPE3(EV, 1, 0x2000);
            } else {
                GKOC_LAUNCH_PIPE3(EV, 1, 0x1000);
            }
        } else if (spw == 2) {
            GKOC_LAUNCH_PIPE3(PE, PU, 0x2000);
        } else {
            GKOC_LAUNCH_PIPE3(PE, PU, 0x1000);
        }
    } else {
        if (spw == 2) {
            GKOC_LAUNCH_PIPE3(1, 4, 0x2000);
        } else {
            GKOC_LAUNCH_PIPE3(1, 4, 0x1000);
        }
    }
#undef GKOC_LAUNCH_PIPE3
#undef GKOC_LAUNCH_PIPE3X
    GKOC_LAUNCH_OK();
    if (lng.count > 0) GKOC_TRY(hub_kernels(as_stream(s)));
    return GKOC_OK;
}

// The distributed product in ONE kernel (csr_spmv_pipe.hpp, GATE): the interior rows of the rank's
// local block and, on the last waves of the grid, the boundary rows as complete rows over [local
// columns | halo]; b = [local vector | halo] (unit stride); gate = two counters in device memory
// (zero at the start).
__global__ void gate_open_kernel(uint32_t* gate, uint32_t epoch)
{
    __hip_atomic_store(gate, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Boundary waves may SPIN (their halo travels on another stream), and a spinning wave keeps its
// slot and its 8 KB of LDS until the gate opens.  The kernels that open it - the exchange's
// (RCCL's send / recv workgroups) and gkoc_gate_open - need room on the device to run at all, and
// stream priority does not preempt resident waves.  So at most 8 boundary waves per CU are ever
// launched (2048 on an MI355X, of about 5120 that are resident at a time with 8 KB of LDS each):
// more than half of every resource stays with waves that finish by themselves.  Above that bound
// the caller takes the join-based product (local rows || exchange, stream-ordered boundary rows),
// whose waiting is the stream's, as in the reference (core/distributed/matrix.cpp:450-509 req.wait()).
int64_t gated_wave_cap() { return int64_t(current_device_props().num_cu) * 8; }

bool gated_fits(int64_t n_rows, int64_t head_rows, int64_t tail_rows)
{
    if (n_rows <= 0 || head_rows < 0 || tail_rows < 0 || head_rows + tail_rows > n_rows ||
        head_rows + tail_rows == 0) {
        return false;
    }
    return ceildiv(head_rows + tail_rows, 64) <= gated_wave_cap();
}

template <typename T, typename I>
int launch_csr_gated(gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* col_idxs, const T* vals,
                     const I* bnd_ptrs, const I* bnd_cols, const T* bnd_vals, const T* b, T* c,
                     int64_t head_rows, int64_t tail_rows, const uint32_t* gate, uint32_t epoch,
                     uint32_t* fork_word, uint32_t fork_number, T* dot_out = nullptr, void* work = nullptr,
                     size_t work_bytes = 0)
{
    GKOC_REQUIRE(n_rows > 0 && head_rows >= 0 && tail_rows >= 0 && head_rows + tail_rows <= n_rows,
                 GKOC_E_INVALID, "bad dimensions");
    GKOC_REQUIRE(head_rows + tail_rows > 0, GKOC_E_INVALID, "no boundary rows: use gkoc_csr_spmv_*");
    GKOC_REQUIRE(row_ptrs && col_idxs && vals && bnd_ptrs && bnd_cols && bnd_vals && b && c && gate,
                 GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(gated_fits(n_rows, head_rows, tail_rows), GKOC_E_NOT_SUPPORTED,
                 "more boundary rows than may wait on the device at once (gkoc_csr_spmv_gated_fits): "
                 "use the join-based product");
    constexpr int EV = 32 / sizeof(T);
    constexpr int RINGV = 8192 / sizeof(T);
    auto aligned = [](const void* v, const void* i) {
        return reinterpret_cast<uintptr_t>(v) % (EV * sizeof(T)) == 0 &&
               reinterpret_cast<uintptr_t>(i) % (EV * sizeof(I)) == 0;
    };
    GKOC_REQUIRE(aligned(vals, col_idxs) && aligned(bnd_vals, bnd_cols), GKOC_E_NOT_SUPPORTED,
                 "values / column indices not aligned for vector loads");
    const int64_t n_int = ceildiv(n_rows - head_rows - tail_rows, 64);
    const int64_t n_bnd = ceildiv(head_rows + tail_rows, 64);
    GKOC_REQUIRE(n_int + n_bnd < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED, "more than 2^31 row segments");
    T* partial = nullptr;
    if (dot_out) {
        // one partial sum per wave (interior waves first, then the boundary waves)
        GKOC_REQUIRE(work && work_bytes >= fused_workspace_bytes(n_rows + 128, sizeof(T)), GKOC_E_WORKSPACE,
                     "workspace too small (gkoc_x_workspace_bytes(n_rows + 128))");
        partial = static_cast<T*>(work);
    }
    const dim3 grid(static_cast<unsigned>(n_int + n_bnd)), block(64);
    constexpr int PE = sizeof(T) == 8 ? 2 : 4, PU = sizeof(T) == 8 ? 3 : 2;
#define GKOC_LAUNCH_GATED(E_, U_, MODE_)                                                           \
    csr_spmv_pipe3_kernel<T, I, false, 64, E_, U_, RINGV, 1, MODE_><<<grid, block, 0, as_stream(s)>>>( \
        n_rows, n_int + n_bnd, 1, row_ptrs, col_idxs, vals, b, 1, c, 1, 1, nullptr, nullptr, partial, 0, \
        nullptr, nullptr, head_rows, tail_rows, gate, epoch, bnd_ptrs, bnd_cols, bnd_vals, fork_word,    \
        fork_number, gate_fence, bnd_first)
    // the cheap gate (no agent-scope acquire for a wave that did not wait) rests on the halo lying on
    // 128-byte lines of its own: b on a line boundary and the halo at a multiple of 128 bytes behind it
    // (the documented layout: n_rows rounded up to 32 entries).  A b that is not aligned cannot have that.
    // ... and on the halo's writer being THIS device: with a peer on another device the policy is 2 until the
    // caller's self-check on that communicator has passed (common.hpp gate_fence_policy)
    const int gate_fence = std::max(
        gate_fence_policy(),
        (tune_value(GKOC_TUNE_GATE_FENCE) != 0 || reinterpret_cast<uintptr_t>(b) % 128 != 0) ? 1 : 0);
    // where in the grid the boundary waves sit: GKOC_TUNE_GATE_POS per cent of the interior waves
    // in front of them (100 = they are the last waves)
    int64_t pos = tune_value(GKOC_TUNE_GATE_POS);
    if (pos < 0 || pos > 100) pos = 100;
    const int64_t bnd_first = pos >= 100 ? -1 : n_int * pos / 100;
    // load layout: the two-entry loads with three groups in flight of the full-size kernel also for a
    // rank's share of a strong-scaling run (30 000 waves, 130 us: 2 us per product faster than one
    // 32-byte load per lane and stream, profiles/r04_dist_sim_variants.txt); GKOC_TUNE_CSR_LOAD_GROUPS = 1
    // selects the other
    const bool wide = tune_value(GKOC_TUNE_CSR_LOAD_GROUPS) == 1;
    if (dot_out) {
        if (wide) {
            GKOC_LAUNCH_GATED(EV, 1, 0x11040);
        } else {
            GKOC_LAUNCH_GATED(PE, PU, 0x11040);
        }
    } else if (wide) {
        GKOC_LAUNCH_GATED(EV, 1, 0x11000);
    } else {
        GKOC_LAUNCH_GATED(PE, PU, 0x11000);
    }
#undef GKOC_LAUNCH_GATED
    GKOC_LAUNCH_OK();
    if (dot_out) {
        // ONE fold launch whatever the count (<= 2^31 / 64 partial sums would still be one block's
        // loop; a rank's share has some 30 000): fixed tree, the value does not depend on timing
        fold_partials_wide_kernel<T><<<dim3(1), dim3(fold_block), 0, as_stream(s)>>>(n_int + n_bnd, partial,
                                                                                      dot_out);
        GKOC_LAUNCH_OK();
    }
    return GKOC_OK;
}

// Diagnostic: `blocks` workgroups of `threads` threads and `lds_bytes` of LDS each that stay on
// the device for `usec` microseconds (tests: a late exchange, an RCCL-sized kernel that must find
// room next to waiting waves)
__global__ void delay_kernel(long long ticks)
{
    extern __shared__ char delay_lds[];
    if (threadIdx.x == 0) delay_lds[0] = 0;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// Mixed precision: values stored as V (float), vectors and arithmetic T (double).  8 instead of
// 12 bytes per stored entry; the result has the bits of the T kernel on the widened values.
// Columns one after the other (the matrix is streamed once per column).
template <typename T, typename V, typename I, bool ADV>
int launch_csr_mixed(gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const T* alpha,
                     const I* row_ptrs, const I* col_idxs, const V* vals, const T* b, int64_t ldb,
                     const T* beta, T* c, int64_t ldc, int64_t nrhs)
{
    GKOC_REQUIRE(n_rows >= 0 && n_cols >= 0 && nrhs >= 0, GKOC_E_INVALID, "negative dimension");
    if (n_rows == 0 || nrhs == 0) return GKOC_OK;
    GKOC_REQUIRE(row_ptrs && c, GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(ldc >= nrhs && (n_cols == 0 || ldb >= nrhs), GKOC_E_INVALID,
                 "stride smaller than nrhs");
    if (ADV) GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha/beta");
    const int64_t n_seg = ceildiv(n_rows, 64);
    const int segs_per_wave = n_seg < 65536 ? 1 : 2;
    const int64_t n_waves = ceildiv(n_seg, segs_per_wave);
    GKOC_REQUIRE(n_waves < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED, "more than 2^31 row segments");
    const dim3 grid(static_cast<unsigned>(n_waves)), block(64);
    // four entries per lane and load (16 B of values, 16 B of columns); eight need 129 VGPRs
    // (three waves per SIMD)
    constexpr int EV = 32 / sizeof(T);
    constexpr int RINGV = 8192 / sizeof(T);
    const bool vec_ok = reinterpret_cast<uintptr_t>(vals) % (EV * sizeof(V)) == 0 &&
                        reinterpret_cast<uintptr_t>(col_idxs) % (EV * sizeof(I)) == 0;
#define GKOC_LAUNCH_MIXED(E_, U_, MODE_)                                                        \
    csr_spmv_pipe3_kernel<T, I, ADV, 64, E_, U_, RINGV, 1, MODE_, V><<<grid, block, 0, as_stream(s)>>>( \
        n_rows, n_seg, segs_per_wave, row_ptrs, col_idxs, vals, b, ldb, c, ldc, static_cast<int>(nrhs), \
        alpha, beta, nullptr, 0)
    if (vec_ok) {
        if (segs_per_wave == 2) {
            // (round 3: 2 x 1, 2 x 2, 1 x 2, 1 x 4, 4 x 2 entries x groups all lose, 1000 - 1440 us
            // against 879)
            GKOC_LAUNCH_MIXED(EV, 1, 0x2000);
        } else {
            GKOC_LAUNCH_MIXED(EV, 1, 0x1000);
        }
    } else {
        if (segs_per_wave == 2) {
            GKOC_LAUNCH_MIXED(1, 4, 0x2000);
        } else {
            GKOC_LAUNCH_MIXED(1, 4, 0x1000);
        }
    }
#undef GKOC_LAUNCH_MIXED
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// c = A b and dot_out = <b, c> in one pass over the matrix (square A, one
// right-hand side, unit strides): every wave also emits its part of the dot
// product, a fixed two-level fold adds them up
template <typename T, typename I>
int launch_csr_dot(gkoc_stream_t s, int64_t n, const I* row_ptrs,
                   const I* col_idxs, const T* vals, const T* b, T* c,
                   T* dot_out, void* work, size_t work_bytes)
{
    GKOC_REQUIRE(n >= 0, GKOC_E_INVALID, "negative dimension");
    GKOC_REQUIRE(dot_out, GKOC_E_INVALID, "null result");
    if (n == 0) {
        GKOC_HIP(hipMemsetAsync(dot_out, 0, sizeof(T), as_stream(s)));
        return GKOC_OK;
    }
    GKOC_REQUIRE(row_ptrs && b && c && work, GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(work_bytes >= fused_workspace_bytes(n, sizeof(T)), GKOC_E_WORKSPACE,
                 "workspace too small (gkoc_x_workspace_bytes)");
    // a matrix with rows far beyond the rest: the product with its long rows spread over many workgroups
    // (csr_long_rows.hpp), then the dot product as a pass of its own - the fusion saves one pass over two
    // vectors, a hub row summed by one wave costs milliseconds
    if (tune_value(GKOC_TUNE_CSR_LONG_ROWS) != 0) {
        csr_long_info lng;
        GKOC_TRY((long_info_of<T, I>(s, n, row_ptrs, &lng)));
        if (lng.count > 0) {
            GKOC_TRY((launch_csr<T, I, false>(s, n, n, nullptr, row_ptrs, col_idxs, vals, b, 1, nullptr, c, 1, 1)));
            if constexpr (sizeof(T) == 8) {
                return gkoc_dense_compute_dot_f64(s, n, 1, reinterpret_cast<const double*>(b), 1,
                                                  reinterpret_cast<const double*>(c), 1,
                                                  reinterpret_cast<double*>(dot_out), work, work_bytes);
            } else {
                return gkoc_dense_compute_dot_f32(s, n, 1, reinterpret_cast<const float*>(b), 1,
                                                  reinterpret_cast<const float*>(c), 1,
                                                  reinterpret_cast<float*>(dot_out), work, work_bytes);
            }
        }
    }
    constexpr int rows_per_seg = 64;
    const int64_t n_seg = ceildiv(n, rows_per_seg);
    // one segment per wave below 4 M rows, as in the plain kernel (a rank's share of a
    // strong-scaling run is that small: 2.1 M rows at 8 ranks, 255 -> 24x us per CG iteration)
    const int segs_per_wave = n_seg < 65536 ? 1 : 2;
    const int64_t n_waves = ceildiv(n_seg, segs_per_wave);
    GKOC_REQUIRE(n_waves < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED,
                 "more than 2^31 row segments");
    T* partial = static_cast<T*>(work);
    T* scratch = partial + (fused_workspace_bytes(n, sizeof(T)) / sizeof(T) - fold_chunks);
    dim3 grid(static_cast<unsigned>(n_waves)), block(64);
    const bool vec_ok =
        reinterpret_cast<uintptr_t>(vals) % (4 * sizeof(T)) == 0 &&
        reinterpret_cast<uintptr_t>(col_idxs) % (4 * sizeof(I)) == 0;  // E = 4 below
    const int xcd_map = (tune_value(GKOC_TUNE_CSR_XCD_MAP) == 1 && n_waves >= 8 * 1024) ? 1 : 0;
    constexpr int PE = sizeof(T) == 8 ? 2 : 4, PU = sizeof(T) == 8 ? 3 : 2;   // as in launch_csr
    // four waves per SIMD, like the plain product: the dot's registers (and, since round 6, the loop over runs
    // of unflagged segments) had taken the double / int32 kernel to 133 VGPRs = three waves - 985 -> 1107 us on
    // L256 (profiles/r06/r06_bench_kernel_stats_regression.csv); with the bound the compiler stays at 128
    constexpr int DOT_WPS = sizeof(I) == 4 ? 4 : 1;      // (64-bit indices: 161 VGPRs, the bound would spill)
#define GKOC_LAUNCH_DOT(E_, U_, MODE_)                                               \
    csr_spmv_pipe3_kernel<T, I, false, rows_per_seg, E_, U_, 1024, DOT_WPS, MODE_>   \
        <<<grid, block, 0, as_stream(s)>>>(n, n_seg, segs_per_wave, row_ptrs, col_idxs, \
                                           vals, b, 1, c, 1, 1, nullptr, nullptr,    \
                                           partial, xcd_map)
    if (vec_ok) {
        if (segs_per_wave == 2) {
            GKOC_LAUNCH_DOT(PE, PU, 0x2040);
        } else {
            GKOC_LAUNCH_DOT(PE, PU, 0x1040);
        }
    } else {
        if (segs_per_wave == 2) {
            GKOC_LAUNCH_DOT(1, 4, 0x2040);
        } else {
            GKOC_LAUNCH_DOT(1, 4, 0x1040);
        }
    }
#undef GKOC_LAUNCH_DOT
    GKOC_LAUNCH_OK();
    return fold_partials<T>(s, n_waves, partial, scratch, dot_out, false);
}

// ---- diagonal extraction / sortedness / per-row sort ---------------------

// One wave per 64 rows: the rows' contiguous column-index (and value) range is
// staged in LDS with coalesced loads, lane = row then works on its row there.
// Segments longer than the stage fall back to walking global memory per lane.
constexpr int row_stage_cap = 2048;

template <typename I>
struct row_segment {
    int64_t row, rs, re, K0, K1;
    bool valid;
};

template <typename I>
__device__ __forceinline__ row_segment<I> load_row_segment(int64_t n_rows,
                                                           const I* row_ptrs)
{
    row_segment<I> g;
    const int64_t first = int64_t(blockIdx.x) * 64;
    g.row = first + threadIdx.x;
    g.valid = g.row < n_rows;
    const int64_t last = first + 64 < n_rows ? first + 64 : n_rows;
    g.rs = row_ptrs[g.valid ? g.row : last];
    g.re = row_ptrs[g.valid ? g.row + 1 : last];
    g.K0 = row_ptrs[first];
    g.K1 = row_ptrs[last];
    return g;
}

template <typename T, typename I>
__global__ __launch_bounds__(64) void extract_diag_kernel(
    int64_t n_diag, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    const T* __restrict__ vals, T* __restrict__ diag)
{
    __shared__ I lc[row_stage_cap];
    const row_segment<I> g = load_row_segment<I>(n_diag, row_ptrs);
    const bool staged = g.K1 - g.K0 <= row_stage_cap;
    if (staged) {
        for (int i = threadIdx.x; i < int(g.K1 - g.K0); i += 64) lc[i] = cols[g.K0 + i];
        wave_lds_sync();
    }
    if (!g.valid) return;
    T d = T(0);
    for (int64_t k = g.rs; k < g.re; ++k) {
        const I c = staged ? lc[k - g.K0] : cols[k];
        if (int64_t(c) == g.row) {
            d = vals[k];
            break;
        }
    }
    diag[g.row] = d;
}

template <typename I>
__global__ __launch_bounds__(64) void is_sorted_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, const I* __restrict__ cols,
    int* __restrict__ flag)
{
    __shared__ I lc[row_stage_cap];
    const row_segment<I> g = load_row_segment<I>(n_rows, row_ptrs);
    const bool staged = g.K1 - g.K0 <= row_stage_cap;
    if (staged) {
        for (int i = threadIdx.x; i < int(g.K1 - g.K0); i += 64) lc[i] = cols[g.K0 + i];
        wave_lds_sync();
    }
    bool ok = true;
    if (g.valid) {
        for (int64_t k = g.rs + 1; k < g.re; ++k) {
            const I a = staged ? lc[k - 1 - g.K0] : cols[k - 1];
            const I b = staged ? lc[k - g.K0] : cols[k];
            if (a > b) {
                ok = false;
                break;
            }
        }
    }
    if (__any(!ok) && threadIdx.x == 0) atomicAnd(flag, 0);
}

// stable insertion sort per row (rows are short on this path; Ginkgo only
// calls it when is_sorted_by_column_index returned false)
template <typename T, typename I>
__global__ __launch_bounds__(64) void sort_rows_kernel(
    int64_t n_rows, const I* __restrict__ row_ptrs, I* __restrict__ cols,
    T* __restrict__ vals)
{
    __shared__ I lc[row_stage_cap];
    __shared__ T lv[row_stage_cap];
    const row_segment<I> g = load_row_segment<I>(n_rows, row_ptrs);
    const bool staged = g.K1 - g.K0 <= row_stage_cap;
    if (staged) {
        const int seg = int(g.K1 - g.K0);
        for (int i = threadIdx.x; i < seg; i += 64) {
            lc[i] = cols[g.K0 + i];
            lv[i] = vals[g.K0 + i];
        }
        wave_lds_sync();
        if (g.valid) {
            const int a = int(g.rs - g.K0), e = int(g.re - g.K0);
            for (int i = a + 1; i < e; ++i) {
                const I ci = lc[i];
                const T vi = lv[i];
                int k = i - 1;
                while (k >= a && lc[k] > ci) {
                    lc[k + 1] = lc[k];
                    lv[k + 1] = lv[k];
                    --k;
                }
                lc[k + 1] = ci;
                lv[k + 1] = vi;
            }
        }
        wave_lds_sync();
        for (int i = threadIdx.x; i < seg; i += 64) {
            cols[g.K0 + i] = lc[i];
            vals[g.K0 + i] = lv[i];
        }
        return;
    }
    if (!g.valid) return;
    for (int64_t i = g.rs + 1; i < g.re; ++i) {
        const I ci = cols[i];
        const T vi = vals[i];
        int64_t k = i - 1;
        while (k >= g.rs && cols[k] > ci) {
            cols[k + 1] = cols[k];
            vals[k + 1] = vals[k];
            --k;
        }
        cols[k + 1] = ci;
        vals[k + 1] = vi;
    }
}

}  // namespace

// Complex values (round 6).  The product of round 5 was one thread per row walking global memory: 15.0 ms on
// the 27-pt 256^3 matrix with complex<double> values = 0.63 TB/s, 78 % of a CbGmres<complex<double>> iteration
// (profiles/r06/r06_cb_gmres_complex_kernel_stats.csv).  The row-segment kernel is a template on the value
// type and needs nothing a complex value does not have: products (textbook expression) go to the LDS ring as
// 16 / 8-byte entries, lane = row adds them in k order - y(row) = sum_k a(row, k) b(col_k) in the reference's
// order, (alpha a) b on top of beta y for the advanced form (reference/matrix/csr_kernels.cpp:53-112).
// One entry per lane and load for complex<double> (16-byte value loads, four load groups in flight), two for
// complex<float>; 8 KB ring.  Several right-hand sides: one pass per column (the kernel's nrhs loop).
template <typename T, typename I>
int csr_spmv_complex(gkoc_stream_t s, int64_t n_rows, int64_t nrhs, const I* row_ptrs, const I* col_idxs,
                     const T* vals, const T* alpha, const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc)
{
    if (n_rows <= 0 || nrhs <= 0) return GKOC_OK;
    GKOC_REQUIRE(row_ptrs && c, GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(nrhs <= 0x7fffffff, GKOC_E_NOT_SUPPORTED, "more than 2^31 right-hand sides");
    // a matrix without entries or without columns (the non-local part of a rank that has no neighbours) comes
    // with null arrays / a null b: the thread-per-row kernel never touches them, this one's idle lanes read b[0]
    if (col_idxs == nullptr || vals == nullptr || b == nullptr) return GKOC_E_NOT_SUPPORTED;
    constexpr int E = sizeof(T) == 16 ? 1 : 2, U = sizeof(T) == 16 ? 4 : 2;
    constexpr int RINGV = 8192 / sizeof(T);
    if (reinterpret_cast<uintptr_t>(vals) % (E * sizeof(T)) != 0 ||
        reinterpret_cast<uintptr_t>(col_idxs) % (E * sizeof(I)) != 0 ||
        reinterpret_cast<uintptr_t>(b) % sizeof(T) != 0 || reinterpret_cast<uintptr_t>(c) % sizeof(T) != 0) {
        return GKOC_E_NOT_SUPPORTED;
    }
    const int64_t n_seg = ceildiv(n_rows, 64);
    const int spw = n_seg >= 65536 ? 2 : 1;
    const int64_t n_waves = ceildiv(n_seg, spw);
    GKOC_REQUIRE(n_waves < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED, "more than 2^31 row segments");
    const dim3 grid(static_cast<unsigned>(n_waves)), block(64);
    // four waves per SIMD where the kernel is within a few registers of it (complex<double>, 32-bit indices: 132)
    constexpr int CX_WPS = sizeof(I) == 4 ? 4 : 1;
#define GKOC_LAUNCH_CX(ADV_, MODE_)                                                                 \
    csr_spmv_pipe3_kernel<T, I, ADV_, 64, E, U, RINGV, (ADV_ ? 1 : CX_WPS), MODE_><<<grid, block, 0, as_stream(s)>>>( \
        n_rows, n_seg, spw, row_ptrs, col_idxs, vals, b, ldb, c, ldc, static_cast<int>(nrhs), alpha, beta)
    if (alpha != nullptr) {
        GKOC_REQUIRE(beta != nullptr, GKOC_E_INVALID, "alpha and beta go together");
        if (spw == 2) {
            GKOC_LAUNCH_CX(true, 0x2000);
        } else {
            GKOC_LAUNCH_CX(true, 0x1000);
        }
    } else if (spw == 2) {
        GKOC_LAUNCH_CX(false, 0x2000);
    } else {
        GKOC_LAUNCH_CX(false, 0x1000);
    }
#undef GKOC_LAUNCH_CX
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}
#define GKOC_INST_CX(T, I)                                                                                       \
    template int csr_spmv_complex<T, I>(gkoc_stream_t, int64_t, int64_t, const I*, const I*, const T*, const T*, \
                                        const T*, int64_t, const T*, T*, int64_t);
GKOC_INST_CX(gkoc_c128, int32_t)
GKOC_INST_CX(gkoc_c128, int64_t)
GKOC_INST_CX(gkoc_c64, int32_t)
GKOC_INST_CX(gkoc_c64, int64_t)
#undef GKOC_INST_CX

}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_CSR(T, TN, I, IN)                                             \
    extern "C" int gkoc_csr_spmv_##TN##_##IN(                                  \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const I* row_ptrs,    \
        const I* col_idxs, const T* vals, const T* b, int64_t ldb, T* c,       \
        int64_t ldc, int64_t nrhs)                                             \
    {                                                                          \
        return launch_csr<T, I, false>(s, n_rows, n_cols, nullptr, row_ptrs,   \
                                       col_idxs, vals, b, ldb, nullptr, c,     \
                                       ldc, nrhs);                             \
    }                                                                          \
    extern "C" int gkoc_csr_advanced_spmv_##TN##_##IN(                         \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const T* alpha,       \
        const I* row_ptrs, const I* col_idxs, const T* vals, const T* b,       \
        int64_t ldb, const T* beta, T* c, int64_t ldc, int64_t nrhs)           \
    {                                                                          \
        return launch_csr<T, I, true>(s, n_rows, n_cols, alpha, row_ptrs,      \
                                      col_idxs, vals, b, ldb, beta, c, ldc,    \
                                      nrhs);                                   \
    }                                                                          \
    extern "C" int gkoc_x_csr_spmv_dot_##TN##_##IN(                            \
        gkoc_stream_t s, int64_t n, const I* row_ptrs, const I* col_idxs,      \
        const T* vals, const T* b, T* c, T* dot_out, void* work,               \
        size_t work_bytes)                                                     \
    {                                                                          \
        return launch_csr_dot<T, I>(s, n, row_ptrs, col_idxs, vals, b, c,      \
                                    dot_out, work, work_bytes);                \
    }                                                                          \
    extern "C" int gkoc_csr_extract_diagonal_##TN##_##IN(                      \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const I* row_ptrs,    \
        const I* col_idxs, const T* vals, T* diag)                             \
    {                                                                          \
        const int64_t nd = n_rows < n_cols ? n_rows : n_cols;                  \
        if (nd <= 0) return GKOC_OK;                                           \
        extract_diag_kernel<T, I>                                              \
            <<<dim3(unsigned(ceildiv(nd, 64))), dim3(64), 0, as_stream(s)>>>(  \
                nd, row_ptrs, col_idxs, vals, diag);                           \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_csr_is_sorted_by_column_index_##TN##_##IN(             \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs,                    \
        const I* col_idxs, int* is_sorted_host)                                \
    {                                                                          \
        GKOC_REQUIRE(is_sorted_host, GKOC_E_INVALID, "null result");           \
        *is_sorted_host = 1;                                                   \
        if (n_rows <= 0) return GKOC_OK;                                       \
        int* flag = nullptr;                                                   \
        GKOC_TRY(scratch_malloc(as_stream(s), reinterpret_cast<void**>(&flag), \
                                sizeof(int)));                                 \
        /* any non-zero pattern = sorted; set on the device (an asynchronous */ \
        /* copy from pageable host memory is not ordered with the kernel)    */ \
        GKOC_HIP(hipMemsetAsync(flag, 1, sizeof(int), as_stream(s)));          \
        is_sorted_kernel<I>                                                    \
            <<<dim3(unsigned(ceildiv(n_rows, 64))), dim3(64), 0,               \
               as_stream(s)>>>(n_rows, row_ptrs, col_idxs, flag);              \
        GKOC_LAUNCH_OK();                                                      \
        GKOC_HIP(hipMemcpyAsync(is_sorted_host, flag, sizeof(int),             \
                                hipMemcpyDeviceToHost, as_stream(s)));         \
        GKOC_HIP(hipStreamSynchronize(as_stream(s)));                          \
        GKOC_TRY(scratch_free(as_stream(s), flag));                            \
        *is_sorted_host = *is_sorted_host != 0;                                \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_csr_sort_by_column_index_##TN##_##IN(                  \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, I* col_idxs,       \
        T* vals)                                                               \
    {                                                                          \
        if (n_rows <= 0) return GKOC_OK;                                       \
        sort_rows_kernel<T, I>                                                 \
            <<<dim3(unsigned(ceildiv(n_rows, 64))), dim3(64), 0,               \
               as_stream(s)>>>(n_rows, row_ptrs, col_idxs, vals);              \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }

GKOC_DEF_CSR(double, f64, int32_t, i32)
GKOC_DEF_CSR(double, f64, int64_t, i64)
GKOC_DEF_CSR(float, f32, int32_t, i32)
GKOC_DEF_CSR(float, f32, int64_t, i64)

// csr::sort_by_column_index for complex values (pairs; the kernel only moves them)
#define GKOC_DEF_CSR_SORT(T, TN, I, IN)                                        \
    extern "C" int gkoc_csr_sort_by_column_index_##TN##_##IN(                  \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, I* col_idxs,       \
        T* vals)                                                               \
    {                                                                          \
        if (n_rows <= 0) return GKOC_OK;                                       \
        sort_rows_kernel<T, I>                                                 \
            <<<dim3(unsigned(ceildiv(n_rows, 64))), dim3(64), 0,               \
               as_stream(s)>>>(n_rows, row_ptrs, col_idxs, vals);              \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }
GKOC_DEF_CSR_SORT(gkoc_c128, c128, int32_t, i32)
GKOC_DEF_CSR_SORT(gkoc_c128, c128, int64_t, i64)
GKOC_DEF_CSR_SORT(gkoc_c64, c64, int32_t, i32)
GKOC_DEF_CSR_SORT(gkoc_c64, c64, int64_t, i64)

#define GKOC_DEF_CSR_GATED(T, TN, I, IN)                                                             \
    extern "C" int gkoc_csr_spmv_gated_##TN##_##IN(                                                   \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* col_idxs, const T* vals,         \
        const I* bnd_ptrs, const I* bnd_cols, const T* bnd_vals, const T* b, T* c, int64_t head_rows, \
        int64_t tail_rows, const uint32_t* gate, uint32_t epoch, uint32_t* fork_word,                 \
        uint32_t fork_number)                                                                         \
    {                                                                                                 \
        return launch_csr_gated<T, I>(s, n_rows, row_ptrs, col_idxs, vals, bnd_ptrs, bnd_cols,        \
                                      bnd_vals, b, c, head_rows, tail_rows, gate, epoch, fork_word,   \
                                      fork_number);                                                   \
    }
GKOC_DEF_CSR_GATED(double, f64, int32_t, i32)
GKOC_DEF_CSR_GATED(double, f64, int64_t, i64)
GKOC_DEF_CSR_GATED(float, f32, int32_t, i32)
GKOC_DEF_CSR_GATED(float, f32, int64_t, i64)

#define GKOC_DEF_CSR_GATED_DOT(T, TN, I, IN)                                                         \
    extern "C" int gkoc_x_csr_spmv_gated_dot_##TN##_##IN(                                             \
        gkoc_stream_t s, int64_t n_rows, const I* row_ptrs, const I* col_idxs, const T* vals,         \
        const I* bnd_ptrs, const I* bnd_cols, const T* bnd_vals, const T* b, T* c, int64_t head_rows, \
        int64_t tail_rows, const uint32_t* gate, uint32_t epoch, uint32_t* fork_word,                 \
        uint32_t fork_number, T* dot_out, void* work, size_t work_bytes)                              \
    {                                                                                                 \
        GKOC_REQUIRE(dot_out, GKOC_E_INVALID, "null result");                                         \
        return launch_csr_gated<T, I>(s, n_rows, row_ptrs, col_idxs, vals, bnd_ptrs, bnd_cols,        \
                                      bnd_vals, b, c, head_rows, tail_rows, gate, epoch, fork_word,   \
                                      fork_number, dot_out, work, work_bytes);                        \
    }
GKOC_DEF_CSR_GATED_DOT(double, f64, int32_t, i32)
GKOC_DEF_CSR_GATED_DOT(double, f64, int64_t, i64)
GKOC_DEF_CSR_GATED_DOT(float, f32, int32_t, i32)
GKOC_DEF_CSR_GATED_DOT(float, f32, int64_t, i64)

extern "C" int gkoc_csr_spmv_gated_fits(int64_t n_rows, int64_t head_rows, int64_t tail_rows)
{
    return gkoc::gated_fits(n_rows, head_rows, tail_rows) ? 1 : 0;
}

extern "C" int gkoc_gate_open(gkoc_stream_t s, uint32_t* gate, uint32_t epoch)
{
    GKOC_REQUIRE(gate, GKOC_E_INVALID, "gate == NULL");
    gkoc::gate_open_kernel<<<dim3(1), dim3(1), 0, as_stream(s)>>>(gate, epoch);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

extern "C" int gkoc_debug_delay(gkoc_stream_t s, int64_t usec, int blocks, int threads, int lds_bytes)
{
    GKOC_REQUIRE(usec >= 0 && blocks > 0 && threads > 0 && threads <= 1024 && lds_bytes >= 0 &&
                     lds_bytes <= 64 * 1024,
                 GKOC_E_INVALID, "bad delay arguments");
    /* wall_clock64 counts at 100 MHz on gfx950 */                                                    
    gkoc::delay_kernel<<<dim3(unsigned(blocks)), dim3(unsigned(threads)), size_t(lds_bytes), as_stream(s)>>>(
        (long long)usec * 100);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

#define GKOC_DEF_CSR_MIXED(I, IN)                                                                  \
    extern "C" int gkoc_csr_spmv_f32_f64_##IN(gkoc_stream_t s, int64_t n_rows, int64_t n_cols,     \
                                              const I* row_ptrs, const I* col_idxs,                \
                                              const float* vals, const double* b, int64_t ldb,     \
                                              double* c, int64_t ldc, int64_t nrhs)                \
    {                                                                                              \
        return launch_csr_mixed<double, float, I, false>(s, n_rows, n_cols, nullptr, row_ptrs,     \
                                                         col_idxs, vals, b, ldb, nullptr, c, ldc,  \
                                                         nrhs);                                    \
    }                                                                                              \
    extern "C" int gkoc_csr_advanced_spmv_f32_f64_##IN(                                            \
        gkoc_stream_t s, int64_t n_rows, int64_t n_cols, const double* alpha, const I* row_ptrs,   \
        const I* col_idxs, const float* vals, const double* b, int64_t ldb, const double* beta,    \
        double* c, int64_t ldc, int64_t nrhs)                                                      \
    {                                                                                              \
        return launch_csr_mixed<double, float, I, true>(s, n_rows, n_cols, alpha, row_ptrs,        \
                                                        col_idxs, vals, b, ldb, beta, c, ldc,      \
                                                        nrhs);                                     \
    }
GKOC_DEF_CSR_MIXED(int32_t, i32)
GKOC_DEF_CSR_MIXED(int64_t, i64)

extern "C" size_t gkoc_x_workspace_bytes(int64_t n, size_t value_size)
{
    return gkoc::fused_workspace_bytes(n < 0 ? 0 : n, value_size);
}
