// GMRES inner-loop kernels for gfx950.
//
// Replaces gko::kernels::hip::gmres::{restart, multi_axpy, multi_dot} and
// common_gmres::{initialize, hessenberg_qr, solve_krylov}
// (decl core/solver/gmres_kernels.hpp:23-45, common_gmres_kernels.hpp:23-48;
// semantics reference/solver/gmres_kernels.cpp:26-100,
// reference/solver/common_gmres_kernels.cpp:28-193; stock GPU versions
// common/unified/solver/gmres_kernels.cpp:25-122, common_gmres_kernels.cpp:25-160).
//
// The Krylov basis is one tall Dense ((krylov_dim+1)*n x nrhs); basis vector i
// occupies rows [i*n, (i+1)*n).
//  * multi_dot (classical Gram-Schmidt): every workgroup keeps its 1024-row
//    chunk of next_krylov in registers and streams the matching chunk of each
//    basis vector past it => next_krylov is read ONCE for all k+1 dots
//    ((k+2)*n values of HBM traffic instead of 2(k+1)*n for k+1 separate
//    dots); fixed two-level reduction tree, deterministic.
//  * multi_axpy: one pass, each thread accumulates its row over the basis
//    vectors in j order (bit-identical to the reference).
//  * restart / initialize: element-wise.  hessenberg_qr / solve_krylov: one
//    thread per right-hand side, same operation order as the reference
//    (bit-identical: IEEE divide and sqrt, no FMA contraction).
#include <cmath>

#include "common.hpp"
#include "elementwise.hpp"
#include "fused.hpp"

namespace gkoc {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void gmres_restart_kernel(
    int64_t rows, int64_t cols, const T* __restrict__ residual, int64_t ldr,
    const real_t<T>* __restrict__ residual_norm, T* __restrict__ rnc,
    T* __restrict__ krylov, int64_t ldk, uint64_t* __restrict__ final_iter_nums)
{
    const int64_t total = rows * cols;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total;
         idx += stride) {
        const int64_t i = cols == 1 ? idx : idx / cols;
        const int64_t j = cols == 1 ? 0 : idx - i * cols;
        krylov[i * ldk + j] = residual[i * ldr + j] / residual_norm[j];
    }
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (t < cols) {
        rnc[t] = T(residual_norm[t]);  // row 0 of residual_norm_collection
        final_iter_nums[t] = 0;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gmres_init_kernel(
    int64_t rows, int64_t cols, const T* __restrict__ b, int64_t ldb,
    T* __restrict__ residual, int64_t ldr, T* __restrict__ gsin, int64_t lds,
    T* __restrict__ gcos, int64_t ldc, int64_t krylov_dim,
    uint8_t* __restrict__ stop)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    const int64_t tid = int64_t(blockIdx.x) * 256 + threadIdx.x;
    for (int64_t idx = tid; idx < rows * cols; idx += stride) {
        const int64_t i = cols == 1 ? idx : idx / cols;
        const int64_t j = cols == 1 ? 0 : idx - i * cols;
        residual[i * ldr + j] = b[i * ldb + j];
    }
    for (int64_t idx = tid; idx < krylov_dim * cols; idx += stride) {
        const int64_t i = idx / cols;
        const int64_t j = idx - i * cols;
        gsin[i * lds + j] = T(0);
        gcos[i * ldc + j] = T(0);
    }
    if (tid < cols) stop[tid] = 0;
}

// before_preconditioner(i,k) = sum_{j < final_iter_nums[k]} krylov(i + j*rows, k) * y(j,k)
template <typename T>
__global__ __launch_bounds__(256) void gmres_multi_axpy_kernel(
    int64_t rows, int64_t cols, const T* __restrict__ krylov, int64_t ldk,
    const T* __restrict__ y, int64_t ldy, T* __restrict__ out, int64_t ldo,
    const uint64_t* __restrict__ final_iter_nums,
    const uint8_t* __restrict__ stop)
{
    const int64_t total = rows * cols;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total;
         idx += stride) {
        const int64_t i = cols == 1 ? idx : idx / cols;
        const int64_t k = cols == 1 ? 0 : idx - i * cols;
        if (stop[k] & 0x40) continue;  // is_finalized
        const int64_t nj = int64_t(final_iter_nums[k]);
        T acc = T(0);
        int64_t j = 0;
        for (; j + 4 <= nj; j += 4) {
            const T a0 = krylov[(i + j * rows) * ldk + k];
            const T a1 = krylov[(i + (j + 1) * rows) * ldk + k];
            const T a2 = krylov[(i + (j + 2) * rows) * ldk + k];
            const T a3 = krylov[(i + (j + 3) * rows) * ldk + k];
            acc += a0 * y[j * ldy + k];
            acc += a1 * y[(j + 1) * ldy + k];
            acc += a2 * y[(j + 2) * ldy + k];
            acc += a3 * y[(j + 3) * ldy + k];
        }
        for (; j < nj; ++j) acc += krylov[(i + j * rows) * ldk + k] * y[j * ldy + k];
        out[i * ldo + k] = acc;
    }
}

// next_krylov(i,k) -= sum_{d < num} h(d,k) * basis_d(i,k), one term after the
// other in d order with the roundings of dense::sub_scaled (t = h*v; w = w - t;
// a single zero h skips its term) => bit-identical to the num sub_scaled calls
// of the classical Gram-Schmidt step (gmres.cpp:222-236), but next_krylov is
// read and written once instead of num times.
template <typename T>
__global__ __launch_bounds__(256) void gmres_multi_sub_scaled_kernel(
    int64_t rows, int64_t cols, int num, const T* __restrict__ krylov,
    int64_t ldk, const T* __restrict__ h, int64_t ldh, T* __restrict__ w,
    int64_t ldw)
{
    const int64_t total = rows * cols;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total;
         idx += stride) {
        const int64_t i = cols == 1 ? idx : idx / cols;
        const int64_t k = cols == 1 ? 0 : idx - i * cols;
        T acc = w[i * ldw + k];
        int d = 0;
        for (; d + 4 <= num; d += 4) {
            const T v0 = krylov[(i + int64_t(d) * rows) * ldk + k];
            const T v1 = krylov[(i + int64_t(d + 1) * rows) * ldk + k];
            const T v2 = krylov[(i + int64_t(d + 2) * rows) * ldk + k];
            const T v3 = krylov[(i + int64_t(d + 3) * rows) * ldk + k];
            const T a0 = h[d * ldh + k], a1 = h[(d + 1) * ldh + k];
            const T a2 = h[(d + 2) * ldh + k], a3 = h[(d + 3) * ldh + k];
            const bool one = cols == 1;
            const T t0 = a0 * v0, t1 = a1 * v1, t2 = a2 * v2, t3 = a3 * v3;
            acc = (one && a0 == T(0)) ? acc : acc - t0;
            acc = (one && a1 == T(0)) ? acc : acc - t1;
            acc = (one && a2 == T(0)) ? acc : acc - t2;
            acc = (one && a3 == T(0)) ? acc : acc - t3;
        }
        for (; d < num; ++d) {
            const T a = h[d * ldh + k];
            const T t = a * krylov[(i + int64_t(d) * rows) * ldk + k];
            acc = (cols == 1 && a == T(0)) ? acc : acc - t;
        }
        w[i * ldw + k] = acc;
    }
}

// One modified Gram-Schmidt step fused with the next dot (one column, unit
// strides): w -= h_cur * v_cur with the roundings of dense::sub_scaled (a zero
// h_cur skips the update), and from the registers that hold the new w this
// block's part of <v_next, w>.  w is bit-identical to the unfused step.
template <typename T>
__global__ __launch_bounds__(256) void gmres_mgs_step_kernel(
    int64_t n, T* __restrict__ w, const T* __restrict__ v_cur,
    const T* __restrict__ h_cur, const T* __restrict__ v_next,
    T* __restrict__ partial, bool vec_ok)
{
    __shared__ T lds[4];
    using V = vec16<T>;
    constexpr int W = V::width;
    const T a = h_cur[0];
    const bool noop = a == T(0);
    T acc = T(0);
    const int64_t tid = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t stride = int64_t(gridDim.x) * 256;
    int64_t done = 0;
    if (vec_ok) {
        const int64_t n_vec = n / W;
        for (int64_t i = tid; i < n_vec; i += stride) {
            V wv = reinterpret_cast<const V*>(w)[i];
            const V nv = reinterpret_cast<const V*>(v_next)[i];
            if (!noop) {
                const V cv = reinterpret_cast<const V*>(v_cur)[i];
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    const T t = a * cv.v[e];
                    wv.v[e] = wv.v[e] - t;
                }
                reinterpret_cast<V*>(w)[i] = wv;
            }
#pragma unroll
            for (int e = 0; e < W; ++e) acc += conj_v(nv.v[e]) * wv.v[e];
        }
        done = n_vec * W;
    }
    for (int64_t i = done + tid; i < n; i += stride) {
        T wv = w[i];
        if (!noop) {
            const T t = a * v_cur[i];
            wv = wv - t;
            w[i] = wv;
        }
        acc += conj_v(v_next[i]) * wv;
    }
    const T r = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// stop_status[k].finalize() for stopped, not yet finalized columns (runs after
// the axpy kernel, which must still see the old flags)
__global__ void gmres_finalize_kernel(int64_t cols, uint8_t* stop)
{
    const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k < cols) {
        const uint8_t s = stop[k];
        if (!(s & 0x40) && (s & 0x3f)) stop[k] = s | uint8_t(0x40);
    }
}

// stage 1 of multi_dot for one rhs column: block = 1024-row chunk, loops the
// basis vectors; partial[(d*cols + col) * nblocks + block]
constexpr int md_items = 4;
template <typename T>
__global__ __launch_bounds__(256) void gmres_multi_dot_stage1(
    int64_t rows, int64_t cols, int num_dots, const T* __restrict__ krylov,
    int64_t ldk, const T* __restrict__ next, int64_t ldn, T* __restrict__ partial)
{
    __shared__ T lds[4];
    const int64_t col = blockIdx.y;
    const int64_t base = int64_t(blockIdx.x) * (256 * md_items);
    T nv[md_items];
#pragma unroll
    for (int u = 0; u < md_items; ++u) {
        const int64_t r = base + u * 256 + threadIdx.x;
        nv[u] = r < rows ? next[r * ldn + col] : T(0);
    }
    for (int d = 0; d < num_dots; ++d) {
        T acc = T(0);
#pragma unroll
        for (int u = 0; u < md_items; ++u) {
            const int64_t r = base + u * 256 + threadIdx.x;
            const T kv = r < rows ? krylov[(int64_t(d) * rows + r) * ldk + col] : T(0);
            acc += conj_v(kv) * nv[u];       // conj(krylov) * next (gmres_kernels.cpp multi_dot)
        }
        const T s = block_sum<256>(acc, lds);
        if (threadIdx.x == 0) {
            partial[(int64_t(d) * cols + col) * gridDim.x + blockIdx.x] = s;
        }
        __syncthreads();
    }
}

// stage 2: one block per (dot, col): hessenberg_col(d, col) = sum of partials
template <typename T>
__global__ __launch_bounds__(256) void gmres_multi_dot_stage2(
    int64_t nblocks, int64_t cols, const T* __restrict__ partial,
    T* __restrict__ hcol, int64_t ldh)
{
    __shared__ T lds[4];
    const int64_t d = blockIdx.x / cols, col = blockIdx.x % cols;
    T acc = T(0);
    for (int64_t i = threadIdx.x; i < nblocks; i += 256) {
        acc += partial[int64_t(blockIdx.x) * nblocks + i];
    }
    const T s = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) hcol[d * ldh + col] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void gmres_hessenberg_qr_kernel(
    int64_t cols, T* __restrict__ gsin, int64_t lds_, T* __restrict__ gcos,
    int64_t ldc, real_t<T>* __restrict__ residual_norm, T* __restrict__ rnc,
    int64_t ldr, T* __restrict__ h, int64_t ldh, int64_t iter,
    uint64_t* __restrict__ final_iter_nums, const uint8_t* __restrict__ stop)
{
    using R = real_t<T>;
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= cols) return;
    if (stop[i] & 0x3f) return;
    final_iter_nums[i]++;
    // givens_rotation (common_gmres_kernels.cpp:52-90; conj_v is the identity for real T)
    for (int64_t j = 0; j < iter; ++j) {
        const T c = gcos[j * ldc + i], s = gsin[j * lds_ + i];
        const T hj = h[j * ldh + i], hj1 = h[(j + 1) * ldh + i];
        const T temp = c * hj + s * hj1;
        h[(j + 1) * ldh + i] = -conj_v(s) * hj + conj_v(c) * hj1;
        h[j * ldh + i] = temp;
    }
    // calculate_sin_and_cos (:29-48)
    const T this_h = h[iter * ldh + i];
    const T next_h = h[(iter + 1) * ldh + i];
    T c, s;
    if (this_h == T(0)) {
        c = T(0);
        s = T(1);
    } else {
        const R scale = abs_v(this_h) + abs_v(next_h);
        const R hyp = scale * sqrt(abs_v(this_h / scale) * abs_v(this_h / scale) +
                                   abs_v(next_h / scale) * abs_v(next_h / scale));
        c = conj_v(this_h) / hyp;
        s = conj_v(next_h) / hyp;
    }
    gcos[iter * ldc + i] = c;
    gsin[iter * lds_ + i] = s;
    h[iter * ldh + i] = c * this_h + s * next_h;
    h[(iter + 1) * ldh + i] = T(0);
    // calculate_next_residual_norm (:93-113)
    const T r = rnc[iter * ldr + i];
    const T rn = -conj_v(s) * r;
    rnc[(iter + 1) * ldr + i] = rn;
    rnc[iter * ldr + i] = c * r;
    residual_norm[i] = abs_v(rn);
}

template <typename T>
__global__ __launch_bounds__(256) void gmres_solve_krylov_kernel(
    int64_t cols, const T* __restrict__ rnc, int64_t ldr,
    const T* __restrict__ h, int64_t ldh, T* __restrict__ y, int64_t ldy,
    const uint64_t* __restrict__ final_iter_nums,
    const uint8_t* __restrict__ stop)
{
    const int64_t k = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (k >= cols) return;
    if (stop[k] & 0x40) return;  // is_finalized
    const int64_t m = int64_t(final_iter_nums[k]);
    for (int64_t i = m - 1; i >= 0; --i) {
        T temp = rnc[i * ldr + k];
        for (int64_t j = i + 1; j < m; ++j) {
            temp -= h[j * ldh + i * cols + k] * y[j * ldy + k];
        }
        y[i * ldy + k] = temp / h[i * ldh + i * cols + k];
    }
}

inline unsigned stream_blocks(int64_t n)
{
    int64_t b = ceildiv(n > 0 ? n : 1, 256);
    if (b > 4 * max_stream_blocks) b = 4 * max_stream_blocks;
    return unsigned(b);
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

extern "C" size_t gkoc_gmres_multi_dot_workspace_bytes(int64_t rows, int64_t nrhs,
                                                       int64_t num_dots,
                                                       size_t value_size)
{
    const int64_t nb = ceildiv(rows > 0 ? rows : 1, 256 * md_items);
    return size_t(nb) * size_t(nrhs > 0 ? nrhs : 1) *
           size_t(num_dots > 0 ? num_dots : 1) * value_size;
}

#define GKOC_DEF_GMRES(T, TN)                                                  \
    extern "C" int gkoc_common_gmres_initialize_##TN(                          \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, const T* b, int64_t ldb,  \
        T* residual, int64_t ldr, T* givens_sin, int64_t ld_sin,               \
        T* givens_cos, int64_t ld_cos, int64_t krylov_dim,                     \
        uint8_t* stop_status)                                                  \
    {                                                                          \
        if (nrhs <= 0) return GKOC_OK;                                         \
        const int64_t work = rows * nrhs > krylov_dim * nrhs ? rows * nrhs     \
                                                             : krylov_dim * nrhs; \
        gmres_init_kernel<T><<<dim3(stream_blocks(work > nrhs ? work : nrhs)), \
                               dim3(256), 0, as_stream(s)>>>(                  \
            rows, nrhs, b, ldb, residual, ldr, givens_sin, ld_sin, givens_cos, \
            ld_cos, krylov_dim, stop_status);                                  \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_gmres_restart_##TN(                                    \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, const T* residual,        \
        int64_t ldr, const gkoc::real_t<T>* residual_norm,                     \
        T* residual_norm_collection,                                           \
        T* krylov_bases, int64_t ldk, uint64_t* final_iter_nums)               \
    {                                                                          \
        if (nrhs <= 0) return GKOC_OK;                                         \
        gmres_restart_kernel<T>                                                \
            <<<dim3(stream_blocks(rows * nrhs > nrhs ? rows * nrhs : nrhs)),   \
               dim3(256), 0, as_stream(s)>>>(rows, nrhs, residual, ldr,        \
                                             residual_norm,                    \
                                             residual_norm_collection,         \
                                             krylov_bases, ldk,                \
                                             final_iter_nums);                 \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_gmres_multi_axpy_##TN(                                 \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, const T* krylov_bases,    \
        int64_t ldk, const T* y, int64_t ldy, T* before_preconditioner,        \
        int64_t ldo, const uint64_t* final_iter_nums, uint8_t* stop_status)    \
    {                                                                          \
        if (nrhs <= 0) return GKOC_OK;                                         \
        if (rows > 0) {                                                        \
            gmres_multi_axpy_kernel<T>                                         \
                <<<dim3(stream_blocks(rows * nrhs)), dim3(256), 0,             \
                   as_stream(s)>>>(rows, nrhs, krylov_bases, ldk, y, ldy,      \
                                   before_preconditioner, ldo,                 \
                                   final_iter_nums, stop_status);              \
            GKOC_LAUNCH_OK();                                                  \
        }                                                                      \
        gmres_finalize_kernel<<<dim3(unsigned(ceildiv(nrhs, 256))), dim3(256), \
                                0, as_stream(s)>>>(nrhs, stop_status);         \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_x_gmres_mgs_step_##TN(                                 \
        gkoc_stream_t s, int64_t rows, T* next_krylov, const T* basis_cur,     \
        const T* h_cur, const T* basis_next, T* h_next, void* work,            \
        size_t work_bytes)                                                     \
    {                                                                          \
        GKOC_REQUIRE(rows >= 0 && h_next, GKOC_E_INVALID, "bad argument");     \
        if (rows == 0) {                                                       \
            GKOC_HIP(hipMemsetAsync(h_next, 0, sizeof(T), as_stream(s)));      \
            return GKOC_OK;                                                    \
        }                                                                      \
        GKOC_REQUIRE(next_krylov && basis_cur && h_cur && basis_next && work,  \
                     GKOC_E_INVALID, "null pointer");                          \
        GKOC_REQUIRE(work_bytes >= fused_workspace_bytes(rows, sizeof(T)),     \
                     GKOC_E_WORKSPACE,                                         \
                     "workspace too small (gkoc_x_workspace_bytes)");          \
        int64_t nb = ceildiv(rows, int64_t(256) * 8);                          \
        if (nb > 2048) nb = 2048;                                              \
        T* partial = static_cast<T*>(work);                                    \
        T* scratch = partial + (fused_workspace_bytes(rows, sizeof(T)) /       \
                                    sizeof(T) - fold_chunks);                  \
        gmres_mgs_step_kernel<T>                                               \
            <<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(              \
                rows, next_krylov, basis_cur, h_cur, basis_next, partial,      \
                (reinterpret_cast<uintptr_t>(next_krylov) |                    \
                 reinterpret_cast<uintptr_t>(basis_cur) |                      \
                 reinterpret_cast<uintptr_t>(basis_next)) % 16 == 0);          \
        GKOC_LAUNCH_OK();                                                      \
        return fold_partials<T>(s, nb, partial, scratch, h_next, false);       \
    }                                                                          \
    extern "C" int gkoc_x_gmres_multi_sub_scaled_##TN(                         \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t num,              \
        const T* krylov_bases, int64_t ldk, const T* h, int64_t ldh,           \
        T* next_krylov, int64_t ldn)                                           \
    {                                                                          \
        if (rows <= 0 || nrhs <= 0 || num <= 0) return GKOC_OK;                \
        GKOC_REQUIRE(krylov_bases && h && next_krylov, GKOC_E_INVALID,         \
                     "null pointer");                                          \
        gmres_multi_sub_scaled_kernel<T>                                       \
            <<<dim3(stream_blocks(rows * nrhs)), dim3(256), 0, as_stream(s)>>>( \
                rows, nrhs, int(num), krylov_bases, ldk, h, ldh, next_krylov,  \
                ldn);                                                          \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_gmres_multi_dot_##TN(                                  \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t num_dots,         \
        const T* krylov_bases, int64_t ldk, const T* next_krylov, int64_t ldn, \
        T* hessenberg_col, int64_t ldh, void* work, size_t work_bytes)         \
    {                                                                          \
        if (nrhs <= 0 || num_dots <= 0) return GKOC_OK;                        \
        if (rows == 0) {                                                       \
            for (int64_t d = 0; d < num_dots; ++d) {                           \
                GKOC_HIP(hipMemsetAsync(hessenberg_col + d * ldh, 0,           \
                                        sizeof(T) * nrhs, as_stream(s)));      \
            }                                                                  \
            return GKOC_OK;                                                    \
        }                                                                      \
        GKOC_REQUIRE(work_bytes >= gkoc_gmres_multi_dot_workspace_bytes(       \
                                       rows, nrhs, num_dots, sizeof(T)),       \
                     GKOC_E_WORKSPACE, "multi_dot workspace too small");       \
        const int64_t nb = ceildiv(rows, 256 * md_items);                      \
        gmres_multi_dot_stage1<T>                                              \
            <<<dim3(unsigned(nb), unsigned(nrhs)), dim3(256), 0,               \
               as_stream(s)>>>(rows, nrhs, int(num_dots), krylov_bases, ldk,   \
                               next_krylov, ldn, static_cast<T*>(work));       \
        GKOC_LAUNCH_OK();                                                      \
        gmres_multi_dot_stage2<T>                                              \
            <<<dim3(unsigned(num_dots * nrhs)), dim3(256), 0, as_stream(s)>>>( \
                nb, nrhs, static_cast<const T*>(work), hessenberg_col, ldh);   \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_common_gmres_hessenberg_qr_##TN(                       \
        gkoc_stream_t s, int64_t nrhs, T* givens_sin, int64_t ld_sin,          \
        T* givens_cos, int64_t ld_cos, gkoc::real_t<T>* residual_norm,         \
        T* residual_norm_collection, int64_t ld_rnc, T* hessenberg_iter,       \
        int64_t ld_h, int64_t iter, uint64_t* final_iter_nums,                 \
        const uint8_t* stop_status)                                            \
    {                                                                          \
        if (nrhs <= 0) return GKOC_OK;                                         \
        gmres_hessenberg_qr_kernel<T>                                          \
            <<<dim3(unsigned(ceildiv(nrhs, 256))), dim3(256), 0,               \
               as_stream(s)>>>(nrhs, givens_sin, ld_sin, givens_cos, ld_cos,   \
                               residual_norm, residual_norm_collection,        \
                               ld_rnc, hessenberg_iter, ld_h, iter,            \
                               final_iter_nums, stop_status);                  \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_common_gmres_solve_krylov_##TN(                        \
        gkoc_stream_t s, int64_t nrhs, const T* residual_norm_collection,      \
        int64_t ld_rnc, const T* hessenberg, int64_t ld_h, T* y, int64_t ldy,  \
        const uint64_t* final_iter_nums, const uint8_t* stop_status)           \
    {                                                                          \
        if (nrhs <= 0) return GKOC_OK;                                         \
        gmres_solve_krylov_kernel<T>                                           \
            <<<dim3(unsigned(ceildiv(nrhs, 256))), dim3(256), 0,               \
               as_stream(s)>>>(nrhs, residual_norm_collection, ld_rnc,         \
                               hessenberg, ld_h, y, ldy, final_iter_nums,      \
                               stop_status);                                   \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }

GKOC_DEF_GMRES(double, f64)
GKOC_DEF_GMRES(float, f32)
// complex value types: the same templates (conj_v / abs_v / real_t are the identity for real T)
GKOC_DEF_GMRES(gkoc_c128, c128)
GKOC_DEF_GMRES(gkoc_c64, c64)
