// CB-GMRES for COMPLEX value types (round 5; the real types are cb_gmres.hip).
//
// Replaces gko::kernels::hip::cb_gmres::{restart, arnoldi, solve_krylov} for std::complex<double> /
// std::complex<float> with the Krylov basis stored as complex<double> or complex<float>
// (accessor reduced_row_major: GKO_INSTANTIATE_FOR_EACH_CB_GMRES_TYPE, core/solver/cb_gmres_kernels.hpp:37-94;
// the stock device instantiations common/cuda_hip/solver/cb_gmres_kernels.cpp:723, 979, 1060).
// Semantics: reference/solver/cb_gmres_kernels.cpp:31-420 with conj where the reference has it
// (hessenberg(k) = sum_j next(j) conj(basis_k(j)), the Givens coefficients, calculate_next_residual_norm),
// and - as in the stock device kernels (update_next_krylov_kernel) - WITHOUT the conjugate the sequential
// reference applies to the basis in its re-orthogonalisation update (reference/...:109), which is a
// different operation from its first pass and not what classical Gram-Schmidt does.
//
// Same shape as the real kernels: no host round trip inside an Arnoldi step.  One pass reads next_krylov
// once for its norm and all iter + 1 dots; one pass applies all updates and leaves the partials of the new
// norm; the decision to re-orthogonalise lives in a device flag per column, the (at most two) further rounds
// are always enqueued and leave at once where no column asks for them.  Reductions: per block a fixed
// tree, the blocks' partials folded in block order (deterministic).  Not tuned beyond that: one row per
// thread and step (the real kernels' four-row vector loads and fused second pass are not repeated here).
#include <cmath>

#include "common.hpp"
#include "complex_type.hpp"

namespace gkoc {
namespace {

constexpr int CXB = 256;            // threads per block
constexpr int CX_ROWS = 1024;       // rows per block

struct cx_ctrl {
    int any[3];
};

template <typename T, typename S>
__device__ __forceinline__ T cx_load(const S* p)
{
    const S v = *p;
    return T(real_t<T>(v.re), real_t<T>(v.im));
}
template <typename T, typename S>
__device__ __forceinline__ void cx_store(S* p, T v)
{
    *p = S(real_t<S>(v.re), real_t<S>(v.im));
}

// ---- restart ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(CXB) void cx_sqnorm_partial(int64_t rows, const T* __restrict__ v, int64_t ld,
                                                         real_t<T>* __restrict__ part)
{
    using R = real_t<T>;
    __shared__ R lds[CXB / 64];
    const int64_t col = blockIdx.y;
    R acc = R(0);
    const int64_t r0 = int64_t(blockIdx.x) * CX_ROWS;
    for (int64_t r = r0 + threadIdx.x; r < r0 + CX_ROWS && r < rows; r += CXB) acc += squared_norm_v(v[r * ld + col]);
    const R s = block_sum<CXB>(acc, lds);
    if (threadIdx.x == 0) part[col * gridDim.x + blockIdx.x] = s;
}

template <typename T>
__global__ __launch_bounds__(CXB) void cx_restart_fold(int64_t nblocks, int64_t krylov_dim,
                                                       const real_t<T>* __restrict__ part,
                                                       real_t<T>* __restrict__ residual_norm, T* __restrict__ rnc,
                                                       int64_t ld_rnc, uint64_t* __restrict__ final_iter_nums)
{
    using R = real_t<T>;
    __shared__ R lds[CXB / 64];
    const int64_t col = blockIdx.x;
    R acc = R(0);
    for (int64_t i = threadIdx.x; i < nblocks; i += CXB) acc += part[col * nblocks + i];
    const R s = block_sum<CXB>(acc, lds);
    if (threadIdx.x == 0) {
        const R nrm = ::sqrt(s);
        residual_norm[col] = nrm;
        rnc[col] = T(nrm, R(0));
        for (int64_t i = 1; i <= krylov_dim; ++i) rnc[i * ld_rnc + col] = T(R(0), R(0));
        final_iter_nums[col] = 0;
    }
}

template <typename T, typename S>
__global__ __launch_bounds__(CXB) void cx_restart_fill(int64_t rows, int64_t cols, int64_t krylov_dim,
                                                       const T* __restrict__ residual, int64_t ldr,
                                                       const real_t<T>* __restrict__ residual_norm,
                                                       S* __restrict__ bases, int64_t st0, int64_t st1,
                                                       T* __restrict__ next, int64_t ldn)
{
    const int64_t idx = int64_t(blockIdx.x) * CXB + threadIdx.x;
    if (idx >= rows * cols) return;
    const int64_t r = idx / cols, c = idx - r * cols;
    const T v = residual[r * ldr + c] / residual_norm[c];
    next[r * ldn + c] = v;
    cx_store<T, S>(bases + r * st1 + c, v);
    for (int64_t k = 1; k <= krylov_dim; ++k) cx_store<T, S>(bases + k * st0 + r * st1 + c, T(real_t<T>(0), real_t<T>(0)));
}

// ---- arnoldi: pass A - |next|^2 (round 0) and the dots next . conj(basis_k), k = 0 .. num_k - 1 -----
template <typename T, typename S>
__global__ __launch_bounds__(CXB) void cx_dots_partial(int64_t rows, int64_t cols, int num_k,
                                                       const T* __restrict__ next, int64_t ldn,
                                                       const S* __restrict__ bases, int64_t st0, int64_t st1,
                                                       T* __restrict__ pdot, real_t<T>* __restrict__ pnrm,
                                                       const uint8_t* __restrict__ stop,
                                                       const uint8_t* __restrict__ active, const cx_ctrl* ctrl, int round)
{
    using R = real_t<T>;
    __shared__ R ldr[CXB / 64];
    __shared__ R ldi[CXB / 64];
    if (round > 0 && ctrl->any[round - 1] == 0) return;
    const int64_t col = blockIdx.y;
    const bool skip = round > 0 ? !active[col] : status_has_stopped(stop[col]);
    if (round > 0 && skip) return;
    const int64_t r0 = int64_t(blockIdx.x) * CX_ROWS;
    T nv[CX_ROWS / CXB];
#pragma unroll
    for (int e = 0; e < CX_ROWS / CXB; ++e) {
        const int64_t r = r0 + threadIdx.x + int64_t(e) * CXB;
        nv[e] = r < rows ? next[r * ldn + col] : T(R(0), R(0));
    }
    if (round == 0) {
        R acc = R(0);
#pragma unroll
        for (int e = 0; e < CX_ROWS / CXB; ++e) acc += squared_norm_v(nv[e]);
        const R s = block_sum<CXB>(acc, ldr);
        if (threadIdx.x == 0) pnrm[col * gridDim.x + blockIdx.x] = s;
        __syncthreads();
        if (skip) return;
    }
    for (int k = 0; k < num_k; ++k) {
        T acc = T(R(0), R(0));
#pragma unroll
        for (int e = 0; e < CX_ROWS / CXB; ++e) {
            const int64_t r = r0 + threadIdx.x + int64_t(e) * CXB;
            if (r < rows) acc += nv[e] * conj_v(cx_load<T, S>(bases + int64_t(k) * st0 + r * st1 + col));
        }
        const R sr = block_sum<CXB>(acc.re, ldr);
        const R si = block_sum<CXB>(acc.im, ldi);
        if (threadIdx.x == 0) pdot[(int64_t(k) * cols + col) * gridDim.x + blockIdx.x] = T(sr, si);
        __syncthreads();
    }
}

// fold of pass A: grid (num_k + 1, cols); x < num_k: coefficient k; x == num_k: the old norm (round 0)
template <typename T>
__global__ __launch_bounds__(CXB) void cx_dots_fold(int64_t nblocks, int64_t cols, int num_k,
                                                    const T* __restrict__ pdot, const real_t<T>* __restrict__ pnrm,
                                                    T* __restrict__ h, int64_t ldh, T* __restrict__ coef,
                                                    T* __restrict__ buffer, int64_t ldb, real_t<T>* __restrict__ an,
                                                    uint64_t* __restrict__ final_iter_nums,
                                                    const uint8_t* __restrict__ stop, const uint8_t* __restrict__ active,
                                                    const cx_ctrl* ctrl, int round)
{
    using R = real_t<T>;
    __shared__ R ldr[CXB / 64];
    __shared__ R ldi[CXB / 64];
    if (round > 0 && ctrl->any[round - 1] == 0) return;
    const int64_t col = blockIdx.y;
    const int k = blockIdx.x;
    const bool stopped = status_has_stopped(stop[col]);
    if (k == num_k) {
        if (round > 0) return;
        R acc = R(0);
        for (int64_t i = threadIdx.x; i < nblocks; i += CXB) acc += pnrm[col * nblocks + i];
        const R s = block_sum<CXB>(acc, ldr);
        if (threadIdx.x == 0) {
            an[col] = R(1.0 / ::sqrt(2.0)) * ::sqrt(s);        // row 0 of arnoldi_norm: eta * old norm
            final_iter_nums[col] += stopped ? 0 : 1;
        }
        return;
    }
    if (round > 0 ? !active[col] : stopped) return;
    T acc = T(R(0), R(0));
    const T* p = pdot + (int64_t(k) * cols + col) * nblocks;
    for (int64_t i = threadIdx.x; i < nblocks; i += CXB) acc += p[i];
    const R sr = block_sum<CXB>(acc.re, ldr);
    const R si = block_sum<CXB>(acc.im, ldi);
    if (threadIdx.x == 0) {
        const T s = T(sr, si);
        coef[int64_t(k) * cols + col] = s;          // what pass B subtracts in this round
        if (round == 0) {
            h[int64_t(k) * ldh + col] = s;
        } else {
            if (buffer) buffer[int64_t(k) * ldb + col] = s;
            h[int64_t(k) * ldh + col] += s;
        }
    }
}

// ---- pass B: next -= sum_k coef(k) basis_k, term by term; partials of the new squared norm
template <typename T, typename S>
__global__ __launch_bounds__(CXB) void cx_update_partial(int64_t rows, int64_t cols, int num_k, T* __restrict__ next,
                                                         int64_t ldn, const S* __restrict__ bases, int64_t st0,
                                                         int64_t st1, const T* __restrict__ coef,
                                                         real_t<T>* __restrict__ pnrm, const uint8_t* __restrict__ stop,
                                                         const uint8_t* __restrict__ active, const cx_ctrl* ctrl,
                                                         int round)
{
    using R = real_t<T>;
    __shared__ R lds[CXB / 64];
    if (round > 0 && ctrl->any[round - 1] == 0) return;
    const int64_t col = blockIdx.y;
    if (round > 0 ? !active[col] : status_has_stopped(stop[col])) return;
    const int64_t r0 = int64_t(blockIdx.x) * CX_ROWS;
    R acc = R(0);
    for (int e = 0; e < CX_ROWS / CXB; ++e) {
        const int64_t r = r0 + threadIdx.x + int64_t(e) * CXB;
        if (r >= rows) break;
        T v = next[r * ldn + col];
        for (int k = 0; k < num_k; ++k) {
            v = v - coef[int64_t(k) * cols + col] * cx_load<T, S>(bases + int64_t(k) * st0 + r * st1 + col);
        }
        next[r * ldn + col] = v;
        acc += squared_norm_v(v);
    }
    const R s = block_sum<CXB>(acc, lds);
    if (threadIdx.x == 0) pnrm[col * gridDim.x + blockIdx.x] = s;
}

// fold of pass B and the decision: one block per column
template <typename T>
__global__ __launch_bounds__(CXB) void cx_norm_decide(int64_t nblocks, const real_t<T>* __restrict__ pnrm,
                                                      real_t<T>* __restrict__ an, int64_t ld_an,
                                                      const uint8_t* __restrict__ stop, uint8_t* __restrict__ active,
                                                      cx_ctrl* ctrl, int round)
{
    using R = real_t<T>;
    __shared__ R lds[CXB / 64];
    if (round > 0 && ctrl->any[round - 1] == 0) return;
    const int64_t col = blockIdx.x;
    if (round > 0 ? !active[col] : status_has_stopped(stop[col])) {
        if (round == 0 && threadIdx.x == 0) active[col] = 0;
        return;
    }
    R acc = R(0);
    for (int64_t i = threadIdx.x; i < nblocks; i += CXB) acc += pnrm[col * nblocks + i];
    const R s = block_sum<CXB>(acc, lds);
    if (threadIdx.x == 0) {
        const R nrm = ::sqrt(s);
        an[ld_an + col] = nrm;                                   // row 1: the norm
        // another round while the norm dropped below eta * the norm before (at most two more)
        const bool again = round < 2 && nrm < an[col];
        active[col] = again ? 1 : 0;
        if (again) {
            an[col] = R(1.0 / ::sqrt(2.0)) * nrm;
            atomicAdd(&ctrl->any[round], 1);
        }
    }
}

// hessenberg(iter + 1) = norm; next /= norm; basis(iter + 1) = next
template <typename T, typename S>
__global__ __launch_bounds__(CXB) void cx_finish(int64_t rows, int64_t cols, int64_t iter, T* __restrict__ next,
                                                 int64_t ldn, S* __restrict__ bases, int64_t st0, int64_t st1,
                                                 const real_t<T>* __restrict__ an, int64_t ld_an, T* __restrict__ h,
                                                 int64_t ldh, const uint8_t* __restrict__ stop)
{
    using R = real_t<T>;
    const int64_t idx = int64_t(blockIdx.x) * CXB + threadIdx.x;
    if (idx >= rows * cols) return;
    const int64_t r = idx / cols, c = idx - r * cols;
    if (status_has_stopped(stop[c])) return;
    const R nrm = an[ld_an + c];
    if (r == 0) h[(iter + 1) * ldh + c] = T(nrm, R(0));
    const T v = next[r * ldn + c] / nrm;
    next[r * ldn + c] = v;
    cx_store<T, S>(bases + (iter + 1) * st0 + r * st1 + c, v);
}

// Givens rotations of the new Hessenberg column and the next residual norm: one thread per column
template <typename T>
__global__ void cx_givens(int64_t cols, int64_t iter, T* __restrict__ gsin, int64_t lds_, T* __restrict__ gcos,
                          int64_t ldc, T* __restrict__ h, int64_t ldh, real_t<T>* __restrict__ residual_norm,
                          T* __restrict__ rnc, int64_t ld_rnc, const uint8_t* __restrict__ stop)
{
    using R = real_t<T>;
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols || status_has_stopped(stop[c])) return;
    for (int64_t j = 0; j < iter; ++j) {
        const T hj = h[j * ldh + c], hj1 = h[(j + 1) * ldh + c];
        const T cs = gcos[j * ldc + c], sn = gsin[j * lds_ + c];
        h[j * ldh + c] = cs * hj + sn * hj1;
        h[(j + 1) * ldh + c] = -conj_v(sn) * hj + conj_v(cs) * hj1;
    }
    const T this_h = h[iter * ldh + c], next_h = h[(iter + 1) * ldh + c];
    T cs, sn;
    if (this_h.re == R(0) && this_h.im == R(0)) {
        cs = T(R(0), R(0));
        sn = T(R(1), R(0));
    } else {
        const R scale = abs_v(this_h) + abs_v(next_h);
        const R a = abs_v(this_h / scale), b = abs_v(next_h / scale);
        const R hyp = scale * ::sqrt(a * a + b * b);
        cs = conj_v(this_h) / hyp;
        sn = conj_v(next_h) / hyp;
    }
    gcos[iter * ldc + c] = cs;
    gsin[iter * lds_ + c] = sn;
    h[iter * ldh + c] = cs * this_h + sn * next_h;
    h[(iter + 1) * ldh + c] = T(R(0), R(0));
    const T rn = rnc[iter * ld_rnc + c];
    const T nxt = -conj_v(sn) * rn;
    rnc[(iter + 1) * ld_rnc + c] = nxt;
    rnc[iter * ld_rnc + c] = cs * rn;
    residual_norm[c] = abs_v(nxt);
}

// ---- solve_krylov ---------------------------------------------------------------------------------
template <typename T>
__global__ void cx_solve_upper(int64_t cols, const T* __restrict__ rnc, int64_t ld_rnc, const T* __restrict__ hess,
                               int64_t ldh, T* __restrict__ y, int64_t ldy, const uint64_t* __restrict__ fin)
{
    const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= cols) return;
    const int64_t n = int64_t(fin[k]);
    for (int64_t i = n - 1; i >= 0; --i) {
        T t = rnc[i * ld_rnc + k];
        for (int64_t j = i + 1; j < n; ++j) t = t - hess[i * ldh + j * cols + k] * y[j * ldy + k];
        y[i * ldy + k] = t / hess[i * ldh + i * cols + k];
    }
}

template <typename T, typename S>
__global__ __launch_bounds__(CXB) void cx_qy(int64_t rows, int64_t cols, const S* __restrict__ bases, int64_t st0,
                                             int64_t st1, const T* __restrict__ y, int64_t ldy, T* __restrict__ out,
                                             int64_t ldo, const uint64_t* __restrict__ fin)
{
    using R = real_t<T>;
    const int64_t idx = int64_t(blockIdx.x) * CXB + threadIdx.x;
    if (idx >= rows * cols) return;
    const int64_t r = idx / cols, c = idx - r * cols;
    T acc = T(R(0), R(0));
    const int64_t n = int64_t(fin[c]);
    for (int64_t j = 0; j < n; ++j) acc += cx_load<T, S>(bases + j * st0 + r * st1 + c) * y[j * ldy + c];
    out[r * ldo + c] = acc;
}

// per-stream scratch of the arnoldi passes (partials, coefficients, flags, control block)
struct cx_scratch {
    void* p = nullptr;
    size_t bytes = 0;
};
thread_local cx_scratch t_scratch;

int scratch(size_t bytes, void** out)
{
    if (t_scratch.bytes < bytes) {
        if (t_scratch.p) (void)gkoc_free(t_scratch.p);
        t_scratch.p = nullptr;
        t_scratch.bytes = 0;
        void* q = nullptr;
        GKOC_TRY(gkoc_malloc(&q, bytes));
        t_scratch.p = q;
        t_scratch.bytes = bytes;
    }
    *out = t_scratch.p;
    return GKOC_OK;
}

template <typename T, typename S>
int restart_impl(gkoc_stream_t s, int64_t rows, int64_t cols, int64_t krylov_dim, const T* residual, int64_t ldr,
                 real_t<T>* residual_norm, T* rnc, int64_t ld_rnc, S* bases, int64_t st0, int64_t st1, T* next,
                 int64_t ldn, uint64_t* fin)
{
    using R = real_t<T>;
    const int64_t nb = ceildiv(rows, int64_t(CX_ROWS));
    void* w = nullptr;
    GKOC_TRY(scratch(size_t(nb) * size_t(cols) * sizeof(R) + 256, &w));
    R* part = static_cast<R*>(w);
    hipStream_t st = as_stream(s);
    cx_sqnorm_partial<T><<<dim3(unsigned(nb), unsigned(cols)), dim3(CXB), 0, st>>>(rows, residual, ldr, part);
    GKOC_LAUNCH_OK();
    cx_restart_fold<T><<<dim3(unsigned(cols)), dim3(CXB), 0, st>>>(nb, krylov_dim, part, residual_norm, rnc, ld_rnc, fin);
    GKOC_LAUNCH_OK();
    cx_restart_fill<T, S><<<dim3(unsigned(ceildiv(rows * cols, int64_t(CXB)))), dim3(CXB), 0, st>>>(
        rows, cols, krylov_dim, residual, ldr, residual_norm, bases, st0, st1, next, ldn);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename T, typename S>
int arnoldi_impl(gkoc_stream_t s, int64_t rows, int64_t cols, int64_t iter, T* next, int64_t ldn, T* gsin,
                 int64_t ld_sin, T* gcos, int64_t ld_cos, real_t<T>* residual_norm, T* rnc, int64_t ld_rnc, S* bases,
                 int64_t st0, int64_t st1, T* h, int64_t ldh, T* buffer, int64_t ldb, real_t<T>* an, int64_t ld_an,
                 uint64_t* fin, const uint8_t* stop)
{
    using R = real_t<T>;
    const int64_t nb = ceildiv(rows, int64_t(CX_ROWS));
    const int num_k = int(iter) + 1;
    const size_t b_pdot = size_t(num_k) * size_t(cols) * size_t(nb) * sizeof(T);
    const size_t b_pnrm = size_t(cols) * size_t(nb) * sizeof(R);
    const size_t b_coef = size_t(num_k) * size_t(cols) * sizeof(T);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    void* w = nullptr;
    GKOC_TRY(scratch(up(b_pdot) + up(b_pnrm) + up(b_coef) + up(size_t(cols)) + 256, &w));
    char* base = static_cast<char*>(w);
    T* pdot = reinterpret_cast<T*>(base);
    R* pnrm = reinterpret_cast<R*>(base + up(b_pdot));
    T* coef = reinterpret_cast<T*>(base + up(b_pdot) + up(b_pnrm));
    uint8_t* active = reinterpret_cast<uint8_t*>(base + up(b_pdot) + up(b_pnrm) + up(b_coef));
    cx_ctrl* ctrl = reinterpret_cast<cx_ctrl*>(base + up(b_pdot) + up(b_pnrm) + up(b_coef) + up(size_t(cols)));
    hipStream_t st = as_stream(s);
    GKOC_HIP(hipMemsetAsync(ctrl, 0, sizeof(cx_ctrl), st));
    const dim3 grid_rc{static_cast<unsigned>(nb), static_cast<unsigned>(cols), 1u};
    const dim3 blk{unsigned(CXB), 1u, 1u};
    for (int round = 0; round < 3; ++round) {
        cx_dots_partial<T, S><<<grid_rc, blk, 0, st>>>(rows, cols, num_k, next, ldn, bases, st0, st1, pdot, pnrm, stop,
                                                       active, ctrl, round);
        GKOC_LAUNCH_OK();
        cx_dots_fold<T><<<dim3(unsigned(num_k + 1), unsigned(cols)), blk, 0, st>>>(
            nb, cols, num_k, pdot, pnrm, h, ldh, coef, buffer, ldb, an, fin, stop, active, ctrl, round);
        GKOC_LAUNCH_OK();
        cx_update_partial<T, S><<<grid_rc, blk, 0, st>>>(rows, cols, num_k, next, ldn, bases, st0, st1, coef, pnrm, stop,
                                                         active, ctrl, round);
        GKOC_LAUNCH_OK();
        cx_norm_decide<T><<<dim3(unsigned(cols)), blk, 0, st>>>(nb, pnrm, an, ld_an, stop, active, ctrl, round);
        GKOC_LAUNCH_OK();
    }
    cx_finish<T, S><<<dim3(unsigned(ceildiv(rows * cols, int64_t(CXB)))), blk, 0, st>>>(rows, cols, iter, next, ldn, bases,
                                                                                       st0, st1, an, ld_an, h, ldh, stop);
    GKOC_LAUNCH_OK();
    cx_givens<T><<<dim3(unsigned(ceildiv(cols, int64_t(64)))), dim3(64), 0, st>>>(cols, iter, gsin, ld_sin, gcos, ld_cos, h,
                                                                                 ldh, residual_norm, rnc, ld_rnc, stop);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

template <typename T, typename S>
int solve_impl(gkoc_stream_t s, int64_t rows, int64_t cols, const T* rnc, int64_t ld_rnc, const S* bases, int64_t st0,
               int64_t st1, const T* hess, int64_t ldh, T* y, int64_t ldy, T* out, int64_t ldo, const uint64_t* fin)
{
    hipStream_t st = as_stream(s);
    cx_solve_upper<T><<<dim3(unsigned(ceildiv(cols, int64_t(64)))), dim3(64), 0, st>>>(cols, rnc, ld_rnc, hess, ldh, y, ldy,
                                                                                      fin);
    GKOC_LAUNCH_OK();
    cx_qy<T, S><<<dim3(unsigned(ceildiv(rows * cols, int64_t(CXB)))), dim3(CXB), 0, st>>>(rows, cols, bases, st0, st1, y,
                                                                                          ldy, out, ldo, fin);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_CX_DISPATCH(CALL64, CALL32)                                                            \
    if (storage_kind == GKOC_CB_KEEP) return CALL64;                                                \
    if (storage_kind == GKOC_CB_F32) return CALL32;                                                 \
    set_last_error("cb_gmres (complex): storage kind %d is not a complex storage type", storage_kind); \
    return GKOC_E_NOT_SUPPORTED;

extern "C" {

int gkoc_cb_gmres_restart_c128(gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t krylov_dim, const gkoc_c128* residual,
                               int64_t ldr, double* residual_norm, gkoc_c128* rnc, int64_t ld_rnc, int storage_kind,
                               void* bases, int64_t st0, int64_t st1, gkoc_c128* next, int64_t ldn,
                               uint64_t* final_iter_nums)
{
    GKOC_REQUIRE(rows >= 0 && nrhs >= 0 && krylov_dim >= 0, GKOC_E_INVALID, "negative dimension");
    if (rows == 0 || nrhs == 0) return GKOC_OK;
    GKOC_CX_DISPATCH((restart_impl<gkoc_c128, gkoc_c128>(s, rows, nrhs, krylov_dim, residual, ldr, residual_norm, rnc, ld_rnc,
                                                         static_cast<gkoc_c128*>(bases), st0, st1, next, ldn,
                                                         final_iter_nums)),
                     (restart_impl<gkoc_c128, gkoc_c64>(s, rows, nrhs, krylov_dim, residual, ldr, residual_norm, rnc, ld_rnc,
                                                        static_cast<gkoc_c64*>(bases), st0, st1, next, ldn,
                                                        final_iter_nums)))
}

int gkoc_cb_gmres_restart_c64(gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t krylov_dim, const gkoc_c64* residual,
                              int64_t ldr, float* residual_norm, gkoc_c64* rnc, int64_t ld_rnc, int storage_kind,
                              void* bases, int64_t st0, int64_t st1, gkoc_c64* next, int64_t ldn,
                              uint64_t* final_iter_nums)
{
    GKOC_REQUIRE(rows >= 0 && nrhs >= 0 && krylov_dim >= 0, GKOC_E_INVALID, "negative dimension");
    GKOC_REQUIRE(storage_kind == GKOC_CB_KEEP, GKOC_E_NOT_SUPPORTED, "complex<float>: the basis is stored as complex<float>");
    if (rows == 0 || nrhs == 0) return GKOC_OK;
    return restart_impl<gkoc_c64, gkoc_c64>(s, rows, nrhs, krylov_dim, residual, ldr, residual_norm, rnc, ld_rnc,
                                            static_cast<gkoc_c64*>(bases), st0, st1, next, ldn, final_iter_nums);
}

int gkoc_cb_gmres_arnoldi_c128(gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t iter, gkoc_c128* next, int64_t ldn,
                               gkoc_c128* gsin, int64_t ld_sin, gkoc_c128* gcos, int64_t ld_cos, double* residual_norm,
                               gkoc_c128* rnc, int64_t ld_rnc, int storage_kind, void* bases, int64_t st0, int64_t st1,
                               gkoc_c128* h, int64_t ld_h, gkoc_c128* buffer, int64_t ld_buf, double* arnoldi_norm,
                               int64_t ld_an, uint64_t* final_iter_nums, const uint8_t* stop)
{
    GKOC_REQUIRE(rows >= 0 && nrhs >= 0 && iter >= 0, GKOC_E_INVALID, "negative dimension");
    if (rows == 0 || nrhs == 0) return GKOC_OK;
    GKOC_CX_DISPATCH((arnoldi_impl<gkoc_c128, gkoc_c128>(s, rows, nrhs, iter, next, ldn, gsin, ld_sin, gcos, ld_cos,
                                                         residual_norm, rnc, ld_rnc, static_cast<gkoc_c128*>(bases), st0,
                                                         st1, h, ld_h, buffer, ld_buf, arnoldi_norm, ld_an,
                                                         final_iter_nums, stop)),
                     (arnoldi_impl<gkoc_c128, gkoc_c64>(s, rows, nrhs, iter, next, ldn, gsin, ld_sin, gcos, ld_cos,
                                                        residual_norm, rnc, ld_rnc, static_cast<gkoc_c64*>(bases), st0,
                                                        st1, h, ld_h, buffer, ld_buf, arnoldi_norm, ld_an, final_iter_nums,
                                                        stop)))
}

int gkoc_cb_gmres_arnoldi_c64(gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t iter, gkoc_c64* next, int64_t ldn,
                              gkoc_c64* gsin, int64_t ld_sin, gkoc_c64* gcos, int64_t ld_cos, float* residual_norm,
                              gkoc_c64* rnc, int64_t ld_rnc, int storage_kind, void* bases, int64_t st0, int64_t st1,
                              gkoc_c64* h, int64_t ld_h, gkoc_c64* buffer, int64_t ld_buf, float* arnoldi_norm,
                              int64_t ld_an, uint64_t* final_iter_nums, const uint8_t* stop)
{
    GKOC_REQUIRE(rows >= 0 && nrhs >= 0 && iter >= 0, GKOC_E_INVALID, "negative dimension");
    GKOC_REQUIRE(storage_kind == GKOC_CB_KEEP, GKOC_E_NOT_SUPPORTED, "complex<float>: the basis is stored as complex<float>");
    if (rows == 0 || nrhs == 0) return GKOC_OK;
    return arnoldi_impl<gkoc_c64, gkoc_c64>(s, rows, nrhs, iter, next, ldn, gsin, ld_sin, gcos, ld_cos, residual_norm, rnc,
                                            ld_rnc, static_cast<gkoc_c64*>(bases), st0, st1, h, ld_h, buffer, ld_buf,
                                            arnoldi_norm, ld_an, final_iter_nums, stop);
}

int gkoc_cb_gmres_solve_krylov_c128(gkoc_stream_t s, int64_t rows, int64_t nrhs, const gkoc_c128* rnc, int64_t ld_rnc,
                                    int storage_kind, const void* bases, int64_t st0, int64_t st1, const gkoc_c128* hess,
                                    int64_t ld_h, gkoc_c128* y, int64_t ldy, gkoc_c128* out, int64_t ldo,
                                    const uint64_t* final_iter_nums)
{
    GKOC_REQUIRE(rows >= 0 && nrhs >= 0, GKOC_E_INVALID, "negative dimension");
    if (nrhs == 0) return GKOC_OK;
    GKOC_CX_DISPATCH((solve_impl<gkoc_c128, gkoc_c128>(s, rows, nrhs, rnc, ld_rnc, static_cast<const gkoc_c128*>(bases), st0,
                                                       st1, hess, ld_h, y, ldy, out, ldo, final_iter_nums)),
                     (solve_impl<gkoc_c128, gkoc_c64>(s, rows, nrhs, rnc, ld_rnc, static_cast<const gkoc_c64*>(bases), st0,
                                                      st1, hess, ld_h, y, ldy, out, ldo, final_iter_nums)))
}

int gkoc_cb_gmres_solve_krylov_c64(gkoc_stream_t s, int64_t rows, int64_t nrhs, const gkoc_c64* rnc, int64_t ld_rnc,
                                   int storage_kind, const void* bases, int64_t st0, int64_t st1, const gkoc_c64* hess,
                                   int64_t ld_h, gkoc_c64* y, int64_t ldy, gkoc_c64* out, int64_t ldo,
                                   const uint64_t* final_iter_nums)
{
    GKOC_REQUIRE(rows >= 0 && nrhs >= 0, GKOC_E_INVALID, "negative dimension");
    GKOC_REQUIRE(storage_kind == GKOC_CB_KEEP, GKOC_E_NOT_SUPPORTED, "complex<float>: the basis is stored as complex<float>");
    if (nrhs == 0) return GKOC_OK;
    return solve_impl<gkoc_c64, gkoc_c64>(s, rows, nrhs, rnc, ld_rnc, static_cast<const gkoc_c64*>(bases), st0, st1, hess,
                                          ld_h, y, ldy, out, ldo, final_iter_nums);
}

}  // extern "C"
