// Dense x dense product and precision conversion of gko::matrix::Dense.
//
// Replaces gko::kernels::hip::dense::{simple_apply, apply} (core/matrix/dense_kernels.hpp:23-32;
// semantics reference/matrix/dense_kernels.cpp:38-92) and the mixed-precision instances of
// dense::copy (:95-106).  Not on the SpMV / CG hot path (Ginkgo's solver tests and small dense
// operators use it): an LDS-tiled kernel, one thread per entry of C, the inner dimension walked
// in order with separate multiply and add => every entry is the reference's left-to-right sum,
// bit for bit (alpha form: (alpha * a) * b added term by term onto beta * c).
#include "common.hpp"

namespace gkoc {
namespace {

constexpr int gemm_tile = 16;

template <typename T, bool ADV>
__global__ __launch_bounds__(gemm_tile* gemm_tile) void dense_gemm_kernel(
    int64_t m, int64_t n, int64_t k, const T* __restrict__ alpha_p, const T* __restrict__ a,
    int64_t lda, const T* __restrict__ b, int64_t ldb, const T* __restrict__ beta_p,
    T* __restrict__ c, int64_t ldc)
{
    __shared__ T ta[gemm_tile][gemm_tile + 1];
    __shared__ T tb[gemm_tile][gemm_tile + 1];
    const int tx = threadIdx.x % gemm_tile, ty = threadIdx.x / gemm_tile;
    const int64_t row = int64_t(blockIdx.y) * gemm_tile + ty;
    const int64_t col = int64_t(blockIdx.x) * gemm_tile + tx;
    const bool mine = row < m && col < n;
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    T sum = T(0);
    if (ADV && mine && beta != T(0)) sum = c[row * ldc + col] * beta;
    for (int64_t k0 = 0; k0 < k; k0 += gemm_tile) {
        const int64_t ak = k0 + tx, bk = k0 + ty;
        ta[ty][tx] = (row < m && ak < k) ? a[row * lda + ak] : T(0);
        tb[ty][tx] = (bk < k && col < n) ? b[bk * ldb + col] : T(0);
        __syncthreads();
        const int lim = k - k0 < gemm_tile ? int(k - k0) : gemm_tile;
        for (int i = 0; i < lim; ++i) {
            const T t = ADV ? (alpha * ta[ty][i]) * tb[i][tx] : ta[ty][i] * tb[i][tx];
            sum += t;
        }
        __syncthreads();
    }
    if (mine) c[row * ldc + col] = sum;
}

template <typename T, bool ADV>
int launch_gemm(gkoc_stream_t s, int64_t m, int64_t n, int64_t k, const T* alpha, const T* a,
                int64_t lda, const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc)
{
    GKOC_REQUIRE(m >= 0 && n >= 0 && k >= 0, GKOC_E_INVALID, "negative dimension");
    if (m == 0 || n == 0) return GKOC_OK;
    GKOC_REQUIRE(c && (k == 0 || (a && b)), GKOC_E_INVALID, "null pointer");
    GKOC_REQUIRE(lda >= k && ldb >= n && ldc >= n, GKOC_E_INVALID, "stride smaller than row length");
    if (ADV) GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha/beta");
    const int64_t gx = ceildiv(n, gemm_tile), gy = ceildiv(m, gemm_tile);
    GKOC_REQUIRE(gy <= 65535, GKOC_E_NOT_SUPPORTED, "more than 1 M rows in a dense product");
    dense_gemm_kernel<T, ADV><<<dim3(unsigned(gx), unsigned(gy)), dim3(gemm_tile * gemm_tile), 0,
                                as_stream(s)>>>(m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dense_convert_kernel(int64_t rows, int64_t cols,
                                                            const TI* __restrict__ x, int64_t ldx,
                                                            TO* __restrict__ y, int64_t ldy)
{
    const int64_t total = rows * cols;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / cols, c = i - r * cols;
        y[r * ldy + c] = static_cast<TO>(x[r * ldx + c]);
    }
}

template <typename TI, typename TO>
int launch_convert(gkoc_stream_t s, int64_t rows, int64_t cols, const TI* x, int64_t ldx, TO* y,
                   int64_t ldy)
{
    GKOC_REQUIRE(rows >= 0 && cols >= 0, GKOC_E_INVALID, "negative dimension");
    if (rows == 0 || cols == 0) return GKOC_OK;
    GKOC_REQUIRE(x && y && ldx >= cols && ldy >= cols, GKOC_E_INVALID, "bad operand");
    int64_t nb = ceildiv(rows * cols, 256);
    if (nb > max_stream_blocks) nb = max_stream_blocks;
    dense_convert_kernel<TI, TO><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(rows, cols, x,
                                                                                    ldx, y, ldy);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_GEMM(T, TN)                                                                     \
    extern "C" int gkoc_dense_simple_apply_##TN(gkoc_stream_t s, int64_t m, int64_t n, int64_t k, \
                                                const T* a, int64_t lda, const T* b,             \
                                                int64_t ldb, T* c, int64_t ldc)                  \
    {                                                                                            \
        return launch_gemm<T, false>(s, m, n, k, nullptr, a, lda, b, ldb, nullptr, c, ldc);      \
    }                                                                                            \
    extern "C" int gkoc_dense_apply_##TN(gkoc_stream_t s, int64_t m, int64_t n, int64_t k,       \
                                         const T* alpha, const T* a, int64_t lda, const T* b,    \
                                         int64_t ldb, const T* beta, T* c, int64_t ldc)          \
    {                                                                                            \
        return launch_gemm<T, true>(s, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);            \
    }
GKOC_DEF_GEMM(double, f64)
GKOC_DEF_GEMM(float, f32)
GKOC_DEF_GEMM(gkoc_c128, c128)
GKOC_DEF_GEMM(gkoc_c64, c64)

extern "C" int gkoc_dense_convert_f64_f32(gkoc_stream_t s, int64_t rows, int64_t cols,
                                          const double* x, int64_t ldx, float* y, int64_t ldy)
{
    return launch_convert<double, float>(s, rows, cols, x, ldx, y, ldy);
}

extern "C" int gkoc_dense_convert_f32_f64(gkoc_stream_t s, int64_t rows, int64_t cols,
                                          const float* x, int64_t ldx, double* y, int64_t ldy)
{
    return launch_convert<float, double>(s, rows, cols, x, ldx, y, ldy);
}
