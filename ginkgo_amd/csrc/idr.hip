// IDR(s) step kernels for gfx950.
//
// Replaces gko::kernels::hip::idr::{initialize, step_1, step_2, step_3, compute_omega}
// (decl core/solver/idr_kernels.hpp:22-72; semantics reference/solver/idr_kernels.cpp:27-290;
// driver core/solver/idr.cpp:150-300).  SURVEY.md 8(f) rank 3.
//
// Shapes (row-major, s = subspace dimension, n = rows, nrhs right-hand sides):
//   p  s x n   (P^H: every shadow vector is one contiguous row)
//   m  s x (s nrhs), f / c  s x nrhs, g / u  n x (s nrhs), g_k / v / residual / x  n x nrhs.
// The vector updates are element-wise with an inner loop over at most s terms, evaluated in the
// reference's order (bit-identical given the same scalars); the dots <p_j, g_k> use a fixed
// two-level tree (partials of 256-row chunks folded in index order), so every run gives the same
// bits, the reference's sequential sums agree to rounding.  The s x s lower-triangular solve runs
// one thread per right-hand side.
#include <cmath>
#include <random>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace gkoc {
namespace {

constexpr int idr_block = 256;
constexpr int idr_max_partials = 1024;

inline unsigned idr_blocks(int64_t n)
{
    int64_t nb = ceildiv(n, idr_block);
    return unsigned(nb > max_stream_blocks ? max_stream_blocks : (nb < 1 ? 1 : nb));
}

template <typename T>
__global__ __launch_bounds__(idr_block) void idr_init_m_kernel(int64_t s, int64_t nrhs, T* m, int64_t ldm,
                                                               uint8_t* stop)
{
    const int64_t i = int64_t(blockIdx.x) * idr_block + threadIdx.x;
    if (i < nrhs) stop[i] = 0;
    if (i < s * s * nrhs) {
        const int64_t row = i / (s * nrhs), col = i % (s * nrhs);
        m[row * ldm + col] = row == col / nrhs ? T(1) : T(0);
    }
}

// out[col] = sum_row a[row * lda + acol] * b[row * ldb + col], col < ncols; two stages
template <typename T, bool CONJ = false>
__global__ __launch_bounds__(idr_block) void idr_dot_stage1(int64_t n, const T* __restrict__ a, int64_t lda,
                                                            const T* __restrict__ b, int64_t ldb,
                                                            int64_t ncols, T* __restrict__ partial)
{
    __shared__ T lds[idr_block / 64];
    const int64_t col = blockIdx.y;
    T acc = T(0);
    const int64_t stride = int64_t(gridDim.x) * idr_block;
    for (int64_t row = int64_t(blockIdx.x) * idr_block + threadIdx.x; row < n; row += stride) {
        acc += a[row * lda] * (CONJ ? conj_v(b[row * ldb + col]) : b[row * ldb + col]);
    }
    const T r = block_sum<idr_block>(acc, lds);
    if (threadIdx.x == 0) partial[col * gridDim.x + blockIdx.x] = r;
}

// folds the partials of column `col`; what happens with the sum is OP's business (one thread)
template <typename T, typename OP>
__global__ __launch_bounds__(idr_block) void idr_dot_stage2(int n_partials, const T* __restrict__ partial,
                                                            const uint8_t* __restrict__ stop, OP op)
{
    __shared__ T lds[idr_block / 64];
    const int64_t col = blockIdx.x;
    T acc = T(0);
    for (int i = threadIdx.x; i < n_partials; i += idr_block) acc += partial[col * n_partials + i];
    const T r = block_sum<idr_block>(acc, lds);
    if (threadIdx.x == 0 && !(stop && status_has_stopped(stop[col]))) op(col, r);
}

template <typename T>
struct store_scaled {
    T* out;          // out[col] = sum / divisor[col * div_stride] (divisor may be NULL)
    const T* divisor;
    int64_t div_stride;
    __device__ void operator()(int64_t col, T sum) const
    {
        out[col] = divisor ? sum / divisor[col * div_stride] : sum;
    }
};

template <typename T, bool CONJ = false>
int idr_dots(hipStream_t st, int64_t n, const T* a, int64_t lda, const T* b, int64_t ldb, int64_t ncols,
             T* partial, const uint8_t* stop, store_scaled<T> op)
{
    int64_t nb = ceildiv(n, idr_block * 4);
    if (nb > idr_max_partials) nb = idr_max_partials;
    if (nb < 1) nb = 1;
    idr_dot_stage1<T, CONJ><<<dim3(unsigned(nb), unsigned(ncols)), dim3(idr_block), 0, st>>>(n, a, lda, b, ldb,
                                                                                     ncols, partial);
    GKOC_LAUNCH_OK();
    idr_dot_stage2<T, store_scaled<T>><<<dim3(unsigned(ncols)), dim3(idr_block), 0, st>>>(int(nb), partial,
                                                                                          stop, op);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// detail::get_rand_value (include/ginkgo/core/base/matrix_data.hpp:36-50)
template <typename T, typename D, typename G>
T idr_rand_value(D& dist, G& gen)
{
    if constexpr (std::is_same<T, real_t<T>>::value) {
        return T(dist(gen));
    } else {
        const auto re = dist(gen);
        const auto im = dist(gen);
        return T{static_cast<real_t<T>>(re), static_cast<real_t<T>>(im)};
    }
}

// row r of p: p_r -= dot * p_i; and p_r /= norm (orthonormalisation of the shadow space)
template <typename T>
__global__ __launch_bounds__(idr_block) void idr_row_axpy_kernel(int64_t n, T* __restrict__ pr,
                                                                 const T* __restrict__ pi,
                                                                 const T* __restrict__ dot)
{
    const T d = dot[0];
    const int64_t stride = int64_t(gridDim.x) * idr_block;
    for (int64_t j = int64_t(blockIdx.x) * idr_block + threadIdx.x; j < n; j += stride) pr[j] -= d * pi[j];
}
template <typename T>
__global__ __launch_bounds__(idr_block) void idr_row_normalise_kernel(int64_t n, T* __restrict__ pr,
                                                                      const T* __restrict__ sq)
{
    const real_t<T> norm = sqrt(real_v(sq[0]));
    const int64_t stride = int64_t(gridDim.x) * idr_block;
    for (int64_t j = int64_t(blockIdx.x) * idr_block + threadIdx.x; j < n; j += stride) pr[j] = pr[j] / norm;
}

// c = M \ f, one thread per right-hand side (reference solve_lower_triangular)
template <typename T>
__global__ void idr_solve_lower_kernel(int64_t s, int64_t nrhs, const T* __restrict__ m, int64_t ldm,
                                       const T* __restrict__ f, int64_t ldf, T* __restrict__ c, int64_t ldc,
                                       const uint8_t* __restrict__ stop)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nrhs || status_has_stopped(stop[i])) return;
    for (int64_t row = 0; row < s; ++row) {
        T temp = f[row * ldf + i];
        for (int64_t col = 0; col < row; ++col) temp -= m[row * ldm + col * nrhs + i] * c[col * ldc + i];
        c[row * ldc + i] = temp / m[row * ldm + row * nrhs + i];
    }
}

// step_1: v = residual - sum_{j >= k} c_j g_j ; step_2: u_k = omega pv + sum_{j >= k} c_j u_j
template <typename T, bool STEP2>
__global__ __launch_bounds__(idr_block) void idr_combine_kernel(
    int64_t n, int64_t nrhs, int64_t s, int64_t k, const T* __restrict__ c, int64_t ldc,
    const T* __restrict__ omega, const T* __restrict__ first, int64_t ld_first, const T* basis,
    int64_t ld_basis, T* out, int64_t ld_out, int64_t out_col0, const uint8_t* __restrict__ stop)
{
    const int64_t total = n * nrhs;
    const int64_t stride = int64_t(gridDim.x) * idr_block;
    for (int64_t e = int64_t(blockIdx.x) * idr_block + threadIdx.x; e < total; e += stride) {
        const int64_t row = e / nrhs, i = e - row * nrhs;
        if (status_has_stopped(stop[i])) continue;
        T temp = STEP2 ? omega[i] * first[row * ld_first + i] : first[row * ld_first + i];
        for (int64_t j = k; j < s; ++j) {
            const T t = c[j * ldc + i] * basis[row * ld_basis + j * nrhs + i];
            temp = STEP2 ? temp + t : temp - t;
        }
        out[row * ld_out + out_col0 + i] = temp;
    }
}

// g_k -= alpha g_j ; u_k -= alpha u_j   (alpha per right-hand side, on the device)
template <typename T>
__global__ __launch_bounds__(idr_block) void idr_orth_update_kernel(
    int64_t n, int64_t nrhs, int64_t j, int64_t k, const T* __restrict__ alpha, const T* g, int64_t ldg,
    T* g_k, int64_t ldgk, T* u, int64_t ldu, const uint8_t* __restrict__ stop)
{
    const int64_t total = n * nrhs;
    const int64_t stride = int64_t(gridDim.x) * idr_block;
    for (int64_t e = int64_t(blockIdx.x) * idr_block + threadIdx.x; e < total; e += stride) {
        const int64_t row = e / nrhs, i = e - row * nrhs;
        if (status_has_stopped(stop[i])) continue;
        const T a = alpha[i];
        g_k[row * ldgk + i] -= a * g[row * ldg + j * nrhs + i];
        u[row * ldu + k * nrhs + i] -= a * u[row * ldu + j * nrhs + i];
    }
}

template <typename T>
__global__ __launch_bounds__(idr_block) void idr_store_gk_kernel(int64_t n, int64_t nrhs, int64_t k,
                                                                 const T* __restrict__ g_k, int64_t ldgk,
                                                                 T* g, int64_t ldg,
                                                                 const uint8_t* __restrict__ stop)
{
    const int64_t total = n * nrhs;
    const int64_t stride = int64_t(gridDim.x) * idr_block;
    for (int64_t e = int64_t(blockIdx.x) * idr_block + threadIdx.x; e < total; e += stride) {
        const int64_t row = e / nrhs, i = e - row * nrhs;
        if (status_has_stopped(stop[i])) continue;
        g[row * ldg + k * nrhs + i] = g_k[row * ldgk + i];
    }
}

// beta = f_k / m_kk ; residual -= beta g_k ; x += beta u_k   (beta recomputed per thread from
// read-only inputs: f and m change only in the kernel that follows)
template <typename T>
__global__ __launch_bounds__(idr_block) void idr_update_x_kernel(
    int64_t n, int64_t nrhs, int64_t k, const T* __restrict__ f, int64_t ldf, const T* __restrict__ m,
    int64_t ldm, const T* __restrict__ g, int64_t ldg, const T* __restrict__ u, int64_t ldu, T* residual,
    int64_t ldr, T* x, int64_t ldx, const uint8_t* __restrict__ stop)
{
    const int64_t total = n * nrhs;
    const int64_t stride = int64_t(gridDim.x) * idr_block;
    for (int64_t e = int64_t(blockIdx.x) * idr_block + threadIdx.x; e < total; e += stride) {
        const int64_t row = e / nrhs, i = e - row * nrhs;
        if (status_has_stopped(stop[i])) continue;
        const T beta = f[k * ldf + i] / m[k * ldm + k * nrhs + i];
        residual[row * ldr + i] -= beta * g[row * ldg + k * nrhs + i];
        x[row * ldx + i] += beta * u[row * ldu + k * nrhs + i];
    }
}

template <typename T>
__global__ void idr_update_f_kernel(int64_t s, int64_t nrhs, int64_t k, T* f, int64_t ldf,
                                    const T* __restrict__ m, int64_t ldm, const uint8_t* __restrict__ stop)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nrhs || status_has_stopped(stop[i])) return;
    if (k + 1 < s) {
        const T beta = f[k * ldf + i] / m[k * ldm + k * nrhs + i];
        f[k * ldf + i] = T(0);
        for (int64_t j = k + 1; j < s; ++j) f[j * ldf + i] -= beta * m[j * ldm + k * nrhs + i];
    }
}

template <typename T>
__global__ void idr_omega_kernel(int64_t nrhs, real_t<T> kappa, const T* __restrict__ tht,
                                 const real_t<T>* __restrict__ residual_norm, T* omega,
                                 const uint8_t* __restrict__ stop)
{
    using R = real_t<T>;
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nrhs || status_has_stopped(stop[i])) return;
    const T thr = omega[i];
    const R normt = sqrt(real_v(tht[i]));
    T om = omega[i] / tht[i];
    const R absrho = abs_v(thr / (normt * residual_norm[i]));
    if (absrho < kappa) om = om * (kappa / absrho);
    if (normt == R(0)) om = T(0);
    omega[i] = om;
}

template <typename T>
int idr_initialize(gkoc_stream_t s_, int64_t nrhs, int64_t s, T* m, int64_t ldm, int64_t n, T* p, int64_t ldp,
                   int deterministic, uint8_t* stop)
{
    hipStream_t st = as_stream(s_);
    GKOC_REQUIRE(nrhs >= 0 && s >= 0 && n >= 0, GKOC_E_INVALID, "negative dimension");
    if (s == 0) return GKOC_OK;
    GKOC_REQUIRE(m && p && stop, GKOC_E_INVALID, "null pointer");
    const int64_t cnt = std::max<int64_t>(s * s * nrhs, nrhs);
    idr_init_m_kernel<T><<<dim3(unsigned(ceildiv(cnt, idr_block))), dim3(idr_block), 0, st>>>(s, nrhs, m, ldm,
                                                                                           stop);
    GKOC_LAUNCH_OK();
    if (n == 0) return GKOC_OK;
    if (!deterministic) {
        // the reference draws the shadow vectors from a normal distribution with a random seed
        // (reference/solver/idr_kernels.cpp:118-128): same here, generated on the host
        std::vector<T> host(size_t(s) * n);
        std::normal_distribution<> dist(0.0, 1.0);
        std::default_random_engine gen(std::random_device{}());
        for (auto& v : host) v = idr_rand_value<T>(dist, gen);   // (complex: two draws, real part first)
        for (int64_t r = 0; r < s; ++r) {
            GKOC_HIP(hipMemcpyAsync(p + r * ldp, host.data() + r * n, sizeof(T) * n, hipMemcpyHostToDevice,
                                    st));
        }
        GKOC_HIP(hipStreamSynchronize(st));
    }
    T* scratch = nullptr;   // [0]: scalar, [1 ...]: partials
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&scratch), sizeof(T) * (1 + idr_max_partials)));
    for (int64_t r = 0; r < s; ++r) {
        T* pr = p + r * ldp;
        for (int64_t i = 0; i < r; ++i) {
            const T* pi = p + i * ldp;
            GKOC_TRY((idr_dots<T, true>(st, n, pr, 1, pi, 1, 1, scratch + 1, nullptr,
                                        store_scaled<T>{scratch, nullptr, 0})));
            idr_row_axpy_kernel<T><<<dim3(idr_blocks(n)), dim3(idr_block), 0, st>>>(n, pr, pi, scratch);
            GKOC_LAUNCH_OK();
        }
        GKOC_TRY((idr_dots<T, true>(st, n, pr, 1, pr, 1, 1, scratch + 1, nullptr,
                                    store_scaled<T>{scratch, nullptr, 0})));
        idr_row_normalise_kernel<T><<<dim3(idr_blocks(n)), dim3(idr_block), 0, st>>>(n, pr, scratch);
        GKOC_LAUNCH_OK();
    }
    GKOC_TRY(scratch_free(st, scratch));
    return GKOC_OK;
}

template <typename T>
int idr_step_3(gkoc_stream_t s_, int64_t n, int64_t nrhs, int64_t s, int64_t k, const T* p, int64_t ldp, T* g,
               int64_t ldg, T* g_k, int64_t ldgk, T* u, int64_t ldu, T* m, int64_t ldm, T* f, int64_t ldf,
               T* residual, int64_t ldr, T* x, int64_t ldx, const uint8_t* stop)
{
    hipStream_t st = as_stream(s_);
    GKOC_REQUIRE(n >= 0 && nrhs >= 0 && k >= 0 && k < s, GKOC_E_INVALID, "bad dimensions");
    if (nrhs == 0) return GKOC_OK;
    T* scratch = nullptr;   // nrhs scalars + partials
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&scratch),
                            sizeof(T) * size_t(nrhs) * (1 + idr_max_partials)));
    T* alpha = scratch;
    T* partial = scratch + nrhs;
    const unsigned nb = idr_blocks(n * nrhs);
    // orthogonalise g_k (and u_k) against the first k columns: alpha = <p_j, g_k> / m_jj
    for (int64_t j = 0; j < k; ++j) {
        GKOC_TRY(idr_dots<T>(st, n, p + j * ldp, 1, g_k, ldgk, nrhs, partial, stop,
                             store_scaled<T>{alpha, m + j * ldm + j * nrhs, 1}));
        idr_orth_update_kernel<T><<<dim3(nb), dim3(idr_block), 0, st>>>(n, nrhs, j, k, alpha, g, ldg, g_k, ldgk,
                                                                       u, ldu, stop);
        GKOC_LAUNCH_OK();
    }
    idr_store_gk_kernel<T><<<dim3(nb), dim3(idr_block), 0, st>>>(n, nrhs, k, g_k, ldgk, g, ldg, stop);
    GKOC_LAUNCH_OK();
    // m(j, k) = <p_j, g_k> for j >= k
    for (int64_t j = k; j < s; ++j) {
        GKOC_TRY(idr_dots<T>(st, n, p + j * ldp, 1, g + k * nrhs, ldg, nrhs, partial, stop,
                             store_scaled<T>{m + j * ldm + k * nrhs, nullptr, 0}));
    }
    idr_update_x_kernel<T><<<dim3(nb), dim3(idr_block), 0, st>>>(n, nrhs, k, f, ldf, m, ldm, g, ldg, u, ldu,
                                                                residual, ldr, x, ldx, stop);
    GKOC_LAUNCH_OK();
    idr_update_f_kernel<T><<<dim3(unsigned(ceildiv(nrhs, 64))), dim3(64), 0, st>>>(s, nrhs, k, f, ldf, m, ldm,
                                                                                  stop);
    GKOC_LAUNCH_OK();
    GKOC_TRY(scratch_free(st, scratch));
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_IDR(T, TN, R)                                                                              \
    extern "C" int gkoc_idr_initialize_##TN(gkoc_stream_t s, int64_t nrhs, int64_t subspace_dim, T* m,    \
                                            int64_t ldm, int64_t n, T* subspace_vectors, int64_t ldp,     \
                                            int deterministic, uint8_t* stop_status)                     \
    {                                                                                                    \
        return idr_initialize<T>(s, nrhs, subspace_dim, m, ldm, n, subspace_vectors, ldp, deterministic, \
                                 stop_status);                                                           \
    }                                                                                                    \
    extern "C" int gkoc_idr_step_1_##TN(gkoc_stream_t s, int64_t n, int64_t nrhs, int64_t subspace_dim,  \
                                        int64_t k, const T* m, int64_t ldm, const T* f, int64_t ldf,      \
                                        const T* residual, int64_t ldr, const T* g, int64_t ldg, T* c,    \
                                        int64_t ldc, T* v, int64_t ldv, const uint8_t* stop_status)      \
    {                                                                                                    \
        if (nrhs <= 0) return GKOC_OK;                                                                   \
        idr_solve_lower_kernel<T><<<dim3(unsigned(ceildiv(nrhs, 64))), dim3(64), 0, as_stream(s)>>>(      \
            subspace_dim, nrhs, m, ldm, f, ldf, c, ldc, stop_status);                                    \
        GKOC_LAUNCH_OK();                                                                                \
        if (n <= 0) return GKOC_OK;                                                                      \
        idr_combine_kernel<T, false><<<dim3(idr_blocks(n * nrhs)), dim3(idr_block), 0, as_stream(s)>>>(   \
            n, nrhs, subspace_dim, k, c, ldc, nullptr, residual, ldr, g, ldg, v, ldv, 0, stop_status);   \
        GKOC_LAUNCH_OK();                                                                                \
        return GKOC_OK;                                                                                  \
    }                                                                                                    \
    extern "C" int gkoc_idr_step_2_##TN(gkoc_stream_t s, int64_t n, int64_t nrhs, int64_t subspace_dim,  \
                                        int64_t k, const T* omega, const T* preconditioned_vector,       \
                                        int64_t ldpv, const T* c, int64_t ldc, T* u, int64_t ldu,         \
                                        const uint8_t* stop_status)                                      \
    {                                                                                                    \
        if (nrhs <= 0 || n <= 0) return GKOC_OK;                                                         \
        idr_combine_kernel<T, true><<<dim3(idr_blocks(n * nrhs)), dim3(idr_block), 0, as_stream(s)>>>(    \
            n, nrhs, subspace_dim, k, c, ldc, omega, preconditioned_vector, ldpv, u, ldu, u, ldu,        \
            k * nrhs, stop_status);                                                                      \
        GKOC_LAUNCH_OK();                                                                                \
        return GKOC_OK;                                                                                  \
    }                                                                                                    \
    extern "C" int gkoc_idr_step_3_##TN(gkoc_stream_t s, int64_t n, int64_t nrhs, int64_t subspace_dim,  \
                                        int64_t k, const T* p, int64_t ldp, T* g, int64_t ldg, T* g_k,    \
                                        int64_t ldgk, T* u, int64_t ldu, T* m, int64_t ldm, T* f,         \
                                        int64_t ldf, T* residual, int64_t ldr, T* x, int64_t ldx,         \
                                        const uint8_t* stop_status)                                      \
    {                                                                                                    \
        return idr_step_3<T>(s, n, nrhs, subspace_dim, k, p, ldp, g, ldg, g_k, ldgk, u, ldu, m, ldm, f,   \
                             ldf, residual, ldr, x, ldx, stop_status);                                   \
    }                                                                                                    \
    extern "C" int gkoc_idr_compute_omega_##TN(gkoc_stream_t s, int64_t nrhs, R kappa, const T* tht,      \
                                               const R* residual_norm, T* omega,                         \
                                               const uint8_t* stop_status)                               \
    {                                                                                                    \
        if (nrhs <= 0) return GKOC_OK;                                                                   \
        idr_omega_kernel<T><<<dim3(unsigned(ceildiv(nrhs, 64))), dim3(64), 0, as_stream(s)>>>(            \
            nrhs, kappa, tht, residual_norm, omega, stop_status);                                        \
        GKOC_LAUNCH_OK();                                                                                \
        return GKOC_OK;                                                                                  \
    }
GKOC_DEF_IDR(double, f64, double)
GKOC_DEF_IDR(float, f32, float)
GKOC_DEF_IDR(gkoc_c128, c128, double)
GKOC_DEF_IDR(gkoc_c64, c64, float)
