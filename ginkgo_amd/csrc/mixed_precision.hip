// csr::spmv / advanced_spmv and ell::spmv / advanced_spmv for every (matrix, input, output)
// value-type triple of a Ginkgo core built with GINKGO_MIXED_PRECISION
// (core/base/mixed_precision_types.hpp:15-120: {float, double}^3 and {complex<float>,
// complex<double>}^3; reference/matrix/csr_kernels.cpp:45-118, ell_kernels.cpp:27-125).
// arithmetic_type = the widest of the three (include/ginkgo/core/base/math.hpp:419-438); every
// value is widened as it is loaded, products and sums are arithmetic_type in the reference's order
// (k ascending within a row, multiply and add rounded separately), the result is narrowed once on
// the store.  Real triples: bit-identical to the reference; complex: to rounding (complex_type.hpp).
//
// The uniform triples and (float matrix, double vectors) are routed to the tuned kernels
// (csr_spmv.hip, formats.hip); the other triples use the two plain kernels below:
//   CSR: one wave per 64 rows; the wave loads its rows' entries in storage order (coalesced),
//        writes the products into an LDS chunk, then lane r adds the products of row r in k order.
//   ELL: one lane per row over the column-major slices (coalesced as stored).
// Both are HBM-stream kernels without the software pipeline of the tuned ones.
//
// dense::row_gather / advanced_row_gather<ValueType, OutputType, IndexType> with two different
// precisions (reference/matrix/dense_kernels.cpp:915-950) are at the end of the file.
#include <type_traits>

#include "common.hpp"

namespace gkoc {
namespace {

template <typename A, typename B>
struct widest2 {
    using type = decltype(A{} + B{});
};
template <typename A, typename B>
struct widest2<gkoc_cplx<A>, gkoc_cplx<B>> {
    using type = gkoc_cplx<decltype(A{} + B{})>;
};
template <typename A, typename B, typename C>
using widest = typename widest2<A, typename widest2<B, C>::type>::type;

template <typename To, typename From>
__device__ __forceinline__ To convert_to(From x)
{
    return static_cast<To>(x);
}
template <typename To, typename R>
__device__ __forceinline__ To convert_to(gkoc_cplx<R> x)
{
    using RT = decltype(To{}.re);
    return To{static_cast<RT>(x.re), static_cast<RT>(x.im)};
}

constexpr int mixed_chunk = 1024;   // products per LDS chunk (16 KB for complex<double>)

template <typename MT, typename IT, typename OT, typename I, bool ADV>
__global__ __launch_bounds__(64) void csr_spmv_mixed_kernel(int64_t n_rows, const I* __restrict__ row_ptrs,
                                                            const I* __restrict__ col_idxs,
                                                            const MT* __restrict__ vals,
                                                            const IT* __restrict__ b, int64_t ldb,
                                                            OT* __restrict__ c, int64_t ldc, int nrhs,
                                                            const MT* __restrict__ alpha,
                                                            const OT* __restrict__ beta)
{
    using AT = widest<MT, IT, OT>;
    __shared__ AT prod[mixed_chunk];
    const int lane = threadIdx.x;
    const int64_t row0 = int64_t(blockIdx.x) * 64;
    const int64_t rows_here = n_rows - row0 < 64 ? n_rows - row0 : 64;
    const int64_t row = row0 + lane;
    const int64_t seg_begin = row_ptrs[row0], seg_end = row_ptrs[row0 + rows_here];
    const int64_t rb = lane < rows_here ? int64_t(row_ptrs[row]) : seg_end;
    const int64_t re = lane < rows_here ? int64_t(row_ptrs[row + 1]) : seg_end;
    AT valpha = zero_of<AT>(), vbeta = zero_of<AT>();
    if (ADV) {
        valpha = convert_to<AT>(alpha[0]);
        vbeta = convert_to<AT>(beta[0]);
    }
    for (int j = 0; j < nrhs; ++j) {
        AT sum = zero_of<AT>();
        if (ADV && lane < rows_here && !(vbeta == zero_of<AT>())) {
            sum = convert_to<AT>(c[row * ldc + j]) * vbeta;
        }
        for (int64_t cb = seg_begin; cb < seg_end; cb += mixed_chunk) {
            const int64_t left = seg_end - cb;
            const int here = left < mixed_chunk ? int(left) : mixed_chunk;
            for (int t = lane; t < here; t += 64) {
                AT v = convert_to<AT>(vals[cb + t]);
                if (ADV) v = valpha * v;
                prod[t] = v * convert_to<AT>(b[int64_t(col_idxs[cb + t]) * ldb + j]);
            }
            wave_lds_sync();
            const int64_t lo = rb > cb ? rb : cb;
            const int64_t hi = re < cb + here ? re : cb + here;
            for (int64_t k = lo; k < hi; ++k) sum += prod[k - cb];
            wave_lds_sync();
        }
        if (lane < rows_here) c[row * ldc + j] = convert_to<OT>(sum);
    }
}

template <typename MT, typename IT, typename OT, typename I, bool ADV>
__global__ __launch_bounds__(256) void ell_spmv_mixed_kernel(int64_t n_rows, int64_t k_per_row, int64_t stride,
                                                             const I* __restrict__ cols,
                                                             const MT* __restrict__ vals,
                                                             const IT* __restrict__ b, int64_t ldb,
                                                             OT* __restrict__ c, int64_t ldc, int nrhs,
                                                             const MT* __restrict__ alpha,
                                                             const OT* __restrict__ beta)
{
    using AT = widest<MT, IT, OT>;
    const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    AT valpha = zero_of<AT>(), vbeta = zero_of<AT>();
    if (ADV) {
        valpha = convert_to<AT>(alpha[0]);
        vbeta = convert_to<AT>(beta[0]);
    }
    for (int j = 0; j < nrhs; ++j) {
        AT result = zero_of<AT>();
        if (ADV && !(vbeta == zero_of<AT>())) result = vbeta * convert_to<AT>(c[row * ldc + j]);
        for (int64_t i = 0; i < k_per_row; ++i) {
            const I col = cols[row + i * stride];
            if (col != I(-1)) {
                AT v = convert_to<AT>(vals[row + i * stride]);
                if (ADV) v = valpha * v;
                const AT p = v * convert_to<AT>(b[int64_t(col) * ldb + j]);
                result += p;
            }
        }
        c[row * ldc + j] = convert_to<OT>(result);
    }
}

template <typename From, typename To>
__global__ void widen_scalar_kernel(const From* in, To* out)
{
    out[0] = convert_to<To>(in[0]);
}

struct spmv_args {
    gkoc_stream_t s;
    int64_t n_rows, n_cols, k, stride;   // k, stride: ELL only
    const void* alpha;
    const void* ptrs;                    // CSR row_ptrs (ELL: unused)
    const void* cols;
    const void* vals;
    const void* b;
    int64_t ldb;
    const void* beta;
    void* c;
    int64_t ldc, nrhs;
};

template <typename MT, typename IT, typename OT, typename I, bool ADV>
int launch_csr_plain(const spmv_args& a)
{
    const int64_t n_blocks = ceildiv(a.n_rows, 64);
    GKOC_REQUIRE(n_blocks < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED, "more than 2^31 row segments");
    csr_spmv_mixed_kernel<MT, IT, OT, I, ADV><<<dim3(unsigned(n_blocks)), dim3(64), 0, as_stream(a.s)>>>(
        a.n_rows, static_cast<const I*>(a.ptrs), static_cast<const I*>(a.cols), static_cast<const MT*>(a.vals),
        static_cast<const IT*>(a.b), a.ldb, static_cast<OT*>(a.c), a.ldc, int(a.nrhs),
        static_cast<const MT*>(a.alpha), static_cast<const OT*>(a.beta));
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename MT, typename IT, typename OT, typename I, bool ADV>
int launch_ell_plain(const spmv_args& a)
{
    const int64_t n_blocks = ceildiv(a.n_rows, 256);
    GKOC_REQUIRE(n_blocks < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED, "more than 2^39 rows");
    ell_spmv_mixed_kernel<MT, IT, OT, I, ADV><<<dim3(unsigned(n_blocks)), dim3(256), 0, as_stream(a.s)>>>(
        a.n_rows, a.k, a.stride, static_cast<const I*>(a.cols), static_cast<const MT*>(a.vals),
        static_cast<const IT*>(a.b), a.ldb, static_cast<OT*>(a.c), a.ldc, int(a.nrhs),
        static_cast<const MT*>(a.alpha), static_cast<const OT*>(a.beta));
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// the tuned kernels behind their C entries (same library)
template <typename I>
struct tuned;
#define GKOC_TUNED(I, IN)                                                                              \
    template <>                                                                                        \
    struct tuned<I> {                                                                                  \
        static int csr(int vt, bool adv, const spmv_args& a)                                           \
        {                                                                                              \
            const I* rp = static_cast<const I*>(a.ptrs);                                               \
            const I* ci = static_cast<const I*>(a.cols);                                               \
            switch (vt) {                                                                              \
            case GKOC_VT_F64:                                                                          \
                return adv ? gkoc_csr_advanced_spmv_f64_##IN(a.s, a.n_rows, a.n_cols, (const double*)a.alpha, rp, \
                                                             ci, (const double*)a.vals, (const double*)a.b, \
                                                             a.ldb, (const double*)a.beta, (double*)a.c, \
                                                             a.ldc, a.nrhs)                            \
                           : gkoc_csr_spmv_f64_##IN(a.s, a.n_rows, a.n_cols, rp, ci, (const double*)a.vals, \
                                                    (const double*)a.b, a.ldb, (double*)a.c, a.ldc, a.nrhs); \
            case GKOC_VT_F32:                                                                          \
                return adv ? gkoc_csr_advanced_spmv_f32_##IN(a.s, a.n_rows, a.n_cols, (const float*)a.alpha, rp, \
                                                             ci, (const float*)a.vals, (const float*)a.b, \
                                                             a.ldb, (const float*)a.beta, (float*)a.c, a.ldc, \
                                                             a.nrhs)                                   \
                           : gkoc_csr_spmv_f32_##IN(a.s, a.n_rows, a.n_cols, rp, ci, (const float*)a.vals, \
                                                    (const float*)a.b, a.ldb, (float*)a.c, a.ldc, a.nrhs); \
            case GKOC_VT_C128:                                                                         \
                return gkoc_ccsr_spmv_c128_##IN(a.s, a.n_rows, a.nrhs, rp, ci, (const gkoc_c128*)a.vals, \
                                                (const gkoc_c128*)a.alpha, (const gkoc_c128*)a.b, a.ldb, \
                                                (const gkoc_c128*)a.beta, (gkoc_c128*)a.c, a.ldc);     \
            default:                                                                                   \
                return gkoc_ccsr_spmv_c64_##IN(a.s, a.n_rows, a.nrhs, rp, ci, (const gkoc_c64*)a.vals, \
                                               (const gkoc_c64*)a.alpha, (const gkoc_c64*)a.b, a.ldb,  \
                                               (const gkoc_c64*)a.beta, (gkoc_c64*)a.c, a.ldc);        \
            }                                                                                          \
        }                                                                                              \
        static int csr_f32_f64(bool adv, const spmv_args& a, const double* alpha, const double* beta)  \
        {                                                                                              \
            const I* rp = static_cast<const I*>(a.ptrs);                                               \
            const I* ci = static_cast<const I*>(a.cols);                                               \
            return adv ? gkoc_csr_advanced_spmv_f32_f64_##IN(a.s, a.n_rows, a.n_cols, alpha, rp, ci,   \
                                                             (const float*)a.vals, (const double*)a.b, \
                                                             a.ldb, beta, (double*)a.c, a.ldc, a.nrhs) \
                       : gkoc_csr_spmv_f32_f64_##IN(a.s, a.n_rows, a.n_cols, rp, ci, (const float*)a.vals, \
                                                    (const double*)a.b, a.ldb, (double*)a.c, a.ldc, a.nrhs); \
        }                                                                                              \
        static int ell(int vt, bool adv, const spmv_args& a)                                           \
        {                                                                                              \
            const I* ci = static_cast<const I*>(a.cols);                                               \
            switch (vt) {                                                                              \
            case GKOC_VT_F64:                                                                          \
                return adv ? gkoc_ell_advanced_spmv_f64_##IN(a.s, a.n_rows, a.n_cols, a.k, a.stride,   \
                                                             (const double*)a.alpha, ci, (const double*)a.vals, \
                                                             (const double*)a.b, a.ldb, (const double*)a.beta, \
                                                             (double*)a.c, a.ldc, a.nrhs)              \
                           : gkoc_ell_spmv_f64_##IN(a.s, a.n_rows, a.n_cols, a.k, a.stride, ci,        \
                                                    (const double*)a.vals, (const double*)a.b, a.ldb,  \
                                                    (double*)a.c, a.ldc, a.nrhs);                      \
            case GKOC_VT_F32:                                                                          \
                return adv ? gkoc_ell_advanced_spmv_f32_##IN(a.s, a.n_rows, a.n_cols, a.k, a.stride,   \
                                                             (const float*)a.alpha, ci, (const float*)a.vals, \
                                                             (const float*)a.b, a.ldb, (const float*)a.beta, \
                                                             (float*)a.c, a.ldc, a.nrhs)               \
                           : gkoc_ell_spmv_f32_##IN(a.s, a.n_rows, a.n_cols, a.k, a.stride, ci,        \
                                                    (const float*)a.vals, (const float*)a.b, a.ldb,    \
                                                    (float*)a.c, a.ldc, a.nrhs);                       \
            case GKOC_VT_C128:                                                                         \
                return adv ? gkoc_ell_advanced_spmv_c128_##IN(a.s, a.n_rows, a.n_cols, a.k, a.stride,  \
                                                              (const gkoc_c128*)a.alpha, ci,           \
                                                              (const gkoc_c128*)a.vals, (const gkoc_c128*)a.b, \
                                                              a.ldb, (const gkoc_c128*)a.beta,         \
                                                              (gkoc_c128*)a.c, a.ldc, a.nrhs)          \
                           : gkoc_ell_spmv_c128_##IN(a.s, a.n_rows, a.n_cols, a.k, a.stride, ci,       \
                                                     (const gkoc_c128*)a.vals, (const gkoc_c128*)a.b, a.ldb, \
                                                     (gkoc_c128*)a.c, a.ldc, a.nrhs);                  \
            default:                                                                                   \
                return adv ? gkoc_ell_advanced_spmv_c64_##IN(a.s, a.n_rows, a.n_cols, a.k, a.stride,   \
                                                             (const gkoc_c64*)a.alpha, ci,             \
                                                             (const gkoc_c64*)a.vals, (const gkoc_c64*)a.b, \
                                                             a.ldb, (const gkoc_c64*)a.beta, (gkoc_c64*)a.c, \
                                                             a.ldc, a.nrhs)                            \
                           : gkoc_ell_spmv_c64_##IN(a.s, a.n_rows, a.n_cols, a.k, a.stride, ci,        \
                                                    (const gkoc_c64*)a.vals, (const gkoc_c64*)a.b, a.ldb, \
                                                    (gkoc_c64*)a.c, a.ldc, a.nrhs);                    \
            }                                                                                          \
        }                                                                                              \
        static int ell_f32_f64(bool adv, const spmv_args& a, const double* alpha, const double* beta)  \
        {                                                                                              \
            const I* ci = static_cast<const I*>(a.cols);                                               \
            return adv ? gkoc_ell_advanced_spmv_f32_f64_##IN(a.s, a.n_rows, a.n_cols, a.k, a.stride, alpha, ci, \
                                                             (const float*)a.vals, (const double*)a.b, \
                                                             a.ldb, beta, (double*)a.c, a.ldc, a.nrhs) \
                       : gkoc_ell_spmv_f32_f64_##IN(a.s, a.n_rows, a.n_cols, a.k, a.stride, ci,        \
                                                    (const float*)a.vals, (const double*)a.b, a.ldb,   \
                                                    (double*)a.c, a.ldc, a.nrhs);                      \
        }                                                                                              \
    };
GKOC_TUNED(int32_t, i32)
GKOC_TUNED(int64_t, i64)
#undef GKOC_TUNED

template <bool ELL, typename MT, typename IT, typename OT, typename I>
int launch_plain(bool adv, const spmv_args& a)
{
    if (ELL) {
        return adv ? launch_ell_plain<MT, IT, OT, I, true>(a) : launch_ell_plain<MT, IT, OT, I, false>(a);
    }
    return adv ? launch_csr_plain<MT, IT, OT, I, true>(a) : launch_csr_plain<MT, IT, OT, I, false>(a);
}

template <bool ELL, typename F, typename D, typename I>
int dispatch_pair(int m, int i, int o, bool adv, const spmv_args& a)
{
    // m, i, o in {0 = the wide type D, 1 = the narrow type F}
    switch (m * 4 + i * 2 + o) {
    case 1: return launch_plain<ELL, D, D, F, I>(adv, a);
    case 2: return launch_plain<ELL, D, F, D, I>(adv, a);
    case 3: return launch_plain<ELL, D, F, F, I>(adv, a);
    case 4: return launch_plain<ELL, F, D, D, I>(adv, a);
    case 5: return launch_plain<ELL, F, D, F, I>(adv, a);
    case 6: return launch_plain<ELL, F, F, D, I>(adv, a);
    default: break;
    }
    set_last_error("mixed spmv: uniform triple reached the plain dispatch");
    return GKOC_E_INVALID;
}

template <bool ELL, typename I>
int spmv_mixed(int mt, int it, int ot, spmv_args a)
{
    GKOC_REQUIRE(mt >= 0 && mt <= 3 && it >= 0 && it <= 3 && ot >= 0 && ot <= 3, GKOC_E_INVALID,
                 "unknown value type code");
    GKOC_REQUIRE((mt >> 1) == (it >> 1) && (mt >> 1) == (ot >> 1), GKOC_E_NOT_SUPPORTED,
                 "real and complex value types in one product");
    GKOC_REQUIRE((a.alpha == nullptr) == (a.beta == nullptr), GKOC_E_INVALID, "alpha and beta: both or none");
    GKOC_REQUIRE(a.n_rows >= 0 && a.n_cols >= 0 && a.nrhs >= 0, GKOC_E_INVALID, "negative dimension");
    const bool adv = a.alpha != nullptr;
    if (mt == it && it == ot) return ELL ? tuned<I>::ell(mt, adv, a) : tuned<I>::csr(mt, adv, a);
    if (a.n_rows == 0 || a.nrhs == 0) return GKOC_OK;
    GKOC_REQUIRE(a.cols || (ELL ? a.k == 0 : true), GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(a.c && (ELL || a.ptrs), GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(a.ldc >= a.nrhs && (a.n_cols == 0 || a.ldb >= a.nrhs), GKOC_E_INVALID,
                 "stride smaller than nrhs");
    GKOC_REQUIRE(a.nrhs < (int64_t(1) << 31), GKOC_E_NOT_SUPPORTED, "more than 2^31 columns");
    if (ELL) GKOC_REQUIRE(a.k >= 0 && a.stride >= a.n_rows, GKOC_E_INVALID, "bad ELL shape");
    if (mt == GKOC_VT_F32 && it == GKOC_VT_F64 && ot == GKOC_VT_F64) {
        // the tuned (float values, double vectors) kernels take alpha as a double: widen it (exact)
        double* wide = nullptr;
        if (adv) {
            GKOC_TRY(scratch_malloc(as_stream(a.s), reinterpret_cast<void**>(&wide), sizeof(double)));
            widen_scalar_kernel<<<dim3(1), dim3(1), 0, as_stream(a.s)>>>(static_cast<const float*>(a.alpha), wide);
        }
        const int rc = ELL ? tuned<I>::ell_f32_f64(adv, a, wide, static_cast<const double*>(a.beta))
                           : tuned<I>::csr_f32_f64(adv, a, wide, static_cast<const double*>(a.beta));
        if (wide) GKOC_TRY(scratch_free(as_stream(a.s), wide));
        return rc;
    }
    if (mt >> 1) return dispatch_pair<ELL, gkoc_c64, gkoc_c128, I>(mt & 1, it & 1, ot & 1, adv, a);
    return dispatch_pair<ELL, float, double, I>(mt & 1, it & 1, ot & 1, adv, a);
}


// reference/matrix/dense_kernels.cpp:915-925 / :931-950: out(i, j) = orig(rows[i], j), or
// out(i, j) = type(alpha * orig(rows[i], j)) + type(beta) * type(out(i, j)), type = the wider one
template <typename VT, typename OT, typename I, bool ADV>
__global__ __launch_bounds__(256) void row_gather_mixed_kernel(int64_t n_gather, int64_t cols,
                                                               const I* __restrict__ rows,
                                                               const VT* __restrict__ orig, int64_t ldo,
                                                               OT* __restrict__ out, int64_t ld_out,
                                                               const VT* __restrict__ alpha,
                                                               const VT* __restrict__ beta)
{
    using AT = typename widest2<VT, OT>::type;
    const int64_t total = n_gather * cols;
    for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
         t += int64_t(gridDim.x) * blockDim.x) {
        const int64_t i = t / cols, j = t - i * cols;
        const VT v = orig[int64_t(rows[i]) * ldo + j];
        if (ADV) {
            const AT scaled = convert_to<AT>(alpha[0] * v);
            const AT kept = convert_to<AT>(beta[0]) * convert_to<AT>(out[i * ld_out + j]);
            out[i * ld_out + j] = convert_to<OT>(scaled + kept);
        } else {
            out[i * ld_out + j] = convert_to<OT>(v);
        }
    }
}

template <typename VT, typename OT, typename I>
int launch_row_gather_mixed(gkoc_stream_t s, int64_t n_gather, int64_t cols, const void* alpha, const I* rows,
                            const void* orig, int64_t ldo, const void* beta, void* out, int64_t ld_out)
{
    const int64_t total = n_gather * cols;
    const int64_t want = ceildiv(total, 256);
    const dim3 grid(unsigned(want < max_stream_blocks ? want : max_stream_blocks)), block(256);
    if (alpha) {
        row_gather_mixed_kernel<VT, OT, I, true><<<grid, block, 0, as_stream(s)>>>(
            n_gather, cols, rows, static_cast<const VT*>(orig), ldo, static_cast<OT*>(out), ld_out,
            static_cast<const VT*>(alpha), static_cast<const VT*>(beta));
    } else {
        row_gather_mixed_kernel<VT, OT, I, false><<<grid, block, 0, as_stream(s)>>>(
            n_gather, cols, rows, static_cast<const VT*>(orig), ldo, static_cast<OT*>(out), ld_out, nullptr,
            nullptr);
    }
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename I>
int row_gather_mixed(gkoc_stream_t s, int vt, int ot, int64_t n_gather, int64_t cols, const void* alpha,
                     const I* rows, const void* orig, int64_t ldo, const void* beta, void* out, int64_t ld_out)
{
    GKOC_REQUIRE(vt >= 0 && vt <= 3 && ot >= 0 && ot <= 3, GKOC_E_INVALID, "unknown value type code");
    GKOC_REQUIRE((vt >> 1) == (ot >> 1), GKOC_E_NOT_SUPPORTED, "real and complex value types in one gather");
    GKOC_REQUIRE(vt != ot, GKOC_E_INVALID, "one precision: use gkoc_dense_row_gather_* / advanced_row_gather_*");
    GKOC_REQUIRE((alpha == nullptr) == (beta == nullptr), GKOC_E_INVALID, "alpha and beta: both or none");
    GKOC_REQUIRE(n_gather >= 0 && cols >= 0, GKOC_E_INVALID, "negative dimension");
    if (n_gather == 0 || cols == 0) return GKOC_OK;
    GKOC_REQUIRE(rows && orig && out && ldo >= cols && ld_out >= cols, GKOC_E_INVALID, "null pointer or stride");
    switch (vt * 4 + ot) {
    case GKOC_VT_F64 * 4 + GKOC_VT_F32:
        return launch_row_gather_mixed<double, float, I>(s, n_gather, cols, alpha, rows, orig, ldo, beta, out, ld_out);
    case GKOC_VT_F32 * 4 + GKOC_VT_F64:
        return launch_row_gather_mixed<float, double, I>(s, n_gather, cols, alpha, rows, orig, ldo, beta, out, ld_out);
    case GKOC_VT_C128 * 4 + GKOC_VT_C64:
        return launch_row_gather_mixed<gkoc_c128, gkoc_c64, I>(s, n_gather, cols, alpha, rows, orig, ldo, beta, out,
                                                               ld_out);
    default:
        return launch_row_gather_mixed<gkoc_c64, gkoc_c128, I>(s, n_gather, cols, alpha, rows, orig, ldo, beta, out,
                                                               ld_out);
    }
}

}  // namespace
}  // namespace gkoc

#define GKOC_DEF_SPMV_MIXED(I, IN)                                                                     \
    extern "C" int gkoc_csr_spmv_mixed_##IN(gkoc_stream_t s, int mt, int it, int ot, int64_t n_rows,   \
                                            int64_t n_cols, const void* alpha, const I* row_ptrs,      \
                                            const I* col_idxs, const void* vals, const void* b,        \
                                            int64_t ldb, const void* beta, void* c, int64_t ldc,       \
                                            int64_t nrhs)                                              \
    {                                                                                                  \
        return gkoc::spmv_mixed<false, I>(                                                             \
            mt, it, ot, {s, n_rows, n_cols, 0, 0, alpha, row_ptrs, col_idxs, vals, b, ldb, beta, c, ldc, nrhs}); \
    }                                                                                                  \
    extern "C" int gkoc_ell_spmv_mixed_##IN(gkoc_stream_t s, int mt, int it, int ot, int64_t n_rows,   \
                                            int64_t n_cols, int64_t num_stored_per_row, int64_t stride, \
                                            const void* alpha, const I* col_idxs, const void* vals,    \
                                            const void* b, int64_t ldb, const void* beta, void* c,     \
                                            int64_t ldc, int64_t nrhs)                                 \
    {                                                                                                  \
        return gkoc::spmv_mixed<true, I>(mt, it, ot,                                                   \
                                         {s, n_rows, n_cols, num_stored_per_row, stride, alpha, nullptr, \
                                          col_idxs, vals, b, ldb, beta, c, ldc, nrhs});                \
    }
GKOC_DEF_SPMV_MIXED(int32_t, i32)
GKOC_DEF_SPMV_MIXED(int64_t, i64)

#define GKOC_DEF_ROW_GATHER_MIXED(I, IN)                                                               \
    extern "C" int gkoc_dense_row_gather_mixed_##IN(gkoc_stream_t s, int vt, int ot, int64_t n_gather, \
                                                    int64_t cols, const void* alpha, const I* rows,    \
                                                    const void* orig, int64_t ld_orig, const void* beta, \
                                                    void* out, int64_t ld_out)                         \
    {                                                                                                  \
        return gkoc::row_gather_mixed<I>(s, vt, ot, n_gather, cols, alpha, rows, orig, ld_orig, beta, out, \
                                         ld_out);                                                      \
    }
GKOC_DEF_ROW_GATHER_MIXED(int32_t, i32)
GKOC_DEF_ROW_GATHER_MIXED(int64_t, i64)

// coo::conj_array on complex values (core/matrix/coo.cpp: conj_transpose of a Coo) and
// dense::add_scaled_identity<complex, real> (reference/matrix/dense_kernels.cpp: m = beta m + alpha I
// with REAL scalars on a complex matrix)
namespace gkoc {
namespace {
template <typename T>
__global__ __launch_bounds__(256) void conj_array_kernel(int64_t n, T* __restrict__ x)
{
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        x[i] = conj_v(x[i]);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void add_scaled_identity_real_kernel(int64_t rows, int64_t cols,
                                                                       const real_t<T>* __restrict__ alpha,
                                                                       const real_t<T>* __restrict__ beta,
                                                                       T* __restrict__ m, int64_t ld)
{
    const real_t<T> a = alpha[0], b = beta[0];
    const int64_t total = rows * cols;
    for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += int64_t(gridDim.x) * blockDim.x) {
        const int64_t i = t / cols, j = t - i * cols;
        T v = m[i * ld + j] * b;
        if (i == j) v.re += a;
        m[i * ld + j] = v;
    }
}
}  // namespace
}  // namespace gkoc

#define GKOC_DEF_CONJ_ARRAY(T, TN, R)                                                                  \
    extern "C" int gkoc_conj_array_##TN(gkoc_stream_t s, int64_t n, T* x)                              \
    {                                                                                                  \
        if (n <= 0) return GKOC_OK;                                                                    \
        GKOC_REQUIRE(x, GKOC_E_INVALID, "null pointer");                                               \
        const int64_t want = gkoc::ceildiv(n, 256);                                                    \
        gkoc::conj_array_kernel<T><<<dim3(unsigned(want < gkoc::max_stream_blocks ? want : gkoc::max_stream_blocks)), \
                                     dim3(256), 0, gkoc::as_stream(s)>>>(n, x);                        \
        GKOC_LAUNCH_OK();                                                                              \
        return GKOC_OK;                                                                                \
    }                                                                                                  \
    extern "C" int gkoc_dense_add_scaled_identity_real_##TN(gkoc_stream_t s, int64_t rows, int64_t cols, \
                                                            const R* alpha, const R* beta, T* m, int64_t ld) \
    {                                                                                                  \
        if (rows <= 0 || cols <= 0) return GKOC_OK;                                                    \
        GKOC_REQUIRE(alpha && beta && m && ld >= cols, GKOC_E_INVALID, "null pointer or stride");      \
        const int64_t want = gkoc::ceildiv(rows * cols, 256);                                          \
        gkoc::add_scaled_identity_real_kernel<T>                                                       \
            <<<dim3(unsigned(want < gkoc::max_stream_blocks ? want : gkoc::max_stream_blocks)), dim3(256), 0, \
               gkoc::as_stream(s)>>>(rows, cols, alpha, beta, m, ld);                                  \
        GKOC_LAUNCH_OK();                                                                              \
        return GKOC_OK;                                                                                \
    }
GKOC_DEF_CONJ_ARRAY(gkoc_c128, c128, double)
GKOC_DEF_CONJ_ARRAY(gkoc_c64, c64, float)
