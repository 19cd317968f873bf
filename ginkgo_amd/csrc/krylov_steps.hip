// Fused vector updates of the other Krylov solvers:
//   bicgstab::{initialize, step_1, step_2, step_3, finalize}
//     (decl core/solver/bicgstab_kernels.hpp; reference/solver/bicgstab_kernels.cpp:24-180;
//      stock GPU version common/unified/solver/bicgstab_kernels.cpp)
//   cgs::{initialize, step_1, step_2, step_3}       (reference/solver/cgs_kernels.cpp:24-146)
//   fcg::{initialize, step_1, step_2}               (reference/solver/fcg_kernels.cpp:24-106)
//   pipe_cg::{initialize_1, initialize_2, step_1, step_2}
//                                                   (reference/solver/pipe_cg_kernels.cpp:24-164)
//   bicg::{initialize, step_1, step_2}              (reference/solver/bicg_kernels.cpp:24-110)
//   gcr::{initialize, restart, step_1}              (reference/solver/gcr_kernels.cpp:24-88)
//   minres::{initialize, step_1, step_2}            (reference/solver/minres_kernels.cpp:24-150)
//   chebyshev::{init_update, update}                (reference/solver/chebyshev_kernels.cpp:20-66)
//   ir::initialize                                  (reference/solver/ir_kernels.cpp:20-27)
// All of them are "per column: a few scalars; per element: a short update that
// is skipped when the column has stopped" - the shape of elementwise.hpp: 16-byte
// loads, every operand of an element group in flight before the first store.
// Each result element is computed with the reference's expression (separate
// multiplies, adds and divides, -ffp-contract=off) => bit-identical.
// The scalars some kernels update (alpha, beta, omega) are written by one thread
// per column (OP::store); every thread derives the value it needs from the
// kernel's read-only inputs, so no thread depends on that write.
// Algorithmic HBM traffic per element (values): bicgstab step_1 4, step_2 3,
// step_3 8; cgs step_1 5, step_2 4, step_3 6; fcg step_1 3, step_2 7;
// pipe_cg step_1 14, step_2 12.
#include <cmath>

#include "common.hpp"
#include "elementwise.hpp"
#include "fused.hpp"

namespace gkoc {
namespace {

template <typename T, int NIN, int NOUT>
struct operand_list {
    ew_operands<T, NIN, NOUT> a{};
    int ni = 0, no = 0;
    operand_list& in(const T* p, int64_t ld)
    {
        a.in[ni] = p;
        a.ld_in[ni++] = ld;
        return *this;
    }
    operand_list& out(T* p, int64_t ld)
    {
        a.out[no] = p;
        a.ld_out[no++] = ld;
        return *this;
    }
};

// scalars of the initialize kernels: value list per column, stop reset
template <typename T, int N>
struct scalar_init {
    T* ptr[N];
    T value[N];
};

template <typename T, int N>
__global__ void init_scalars_kernel(int64_t cols, scalar_init<T, N> s, uint8_t* stop)
{
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j < cols) {
#pragma unroll
        for (int k = 0; k < N; ++k) s.ptr[k][j] = s.value[k];
        if (stop) stop[j] = 0;
    }
}

template <typename T, int N>
int launch_init_scalars(gkoc_stream_t s, int64_t cols, const scalar_init<T, N>& si, uint8_t* stop)
{
    if (cols <= 0) return GKOC_OK;
    init_scalars_kernel<T, N><<<dim3(unsigned(ceildiv(cols, 256))), dim3(256), 0, as_stream(s)>>>(
        cols, si, stop);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// out[0] = in[0], out[1..NOUT) = 0   (and NCOPY leading outputs all copy in[0])
template <typename T, int NCOPY, int NOUT>
struct op_copy_and_zero {
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
#pragma unroll
        for (int k = 0; k < NOUT; ++k) out[k] = k < NCOPY ? in[0] : T(0);
    }
};

// ---------------------------------------------------------------- bicgstab
// p = r + tmp (p - omega v), tmp = rho / prev_rho * alpha / omega ; p = r if
// prev_rho * omega == 0.   in = {r, p, v}, out = {p}
template <typename T>
struct op_bicgstab_step1 {
    const T *rho, *prev_rho, *alpha, *omega;
    const uint8_t* stop;
    struct scalars {
        T tmp, omega;
        bool plain, stopped;
    };
    __device__ scalars load(int64_t c) const
    {
        const T pr = prev_rho[c], om = omega[c];
        const bool nz = pr * om != T(0);
        return {nz ? rho[c] / pr * alpha[c] / om : T(0), om, !nz, status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = s.plain ? in[0] : in[0] + s.tmp * (in[1] - s.omega * in[2]);
    }
};

// alpha = rho / beta ; s = r - alpha v   (alpha = 0, s = r if beta == 0)
// in = {r, v}, out = {s}
template <typename T>
struct op_bicgstab_step2 {
    const T *rho, *beta;
    T* alpha;
    const uint8_t* stop;
    struct scalars {
        T alpha;
        bool plain, stopped;
    };
    __device__ scalars load(int64_t c) const
    {
        const T bt = beta[c];
        const bool nz = bt != T(0);
        return {nz ? rho[c] / bt : T(0), !nz, status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void store(int64_t c, const scalars& s) const { alpha[c] = s.alpha; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = s.plain ? in[0] : in[0] - s.alpha * in[1];
    }
};

// omega = gamma / beta (0 if beta == 0) ; x += alpha y + omega z ; r = s - omega t
// in = {x, s, t, y, z}, out = {x, r}
template <typename T>
struct op_bicgstab_step3 {
    const T *alpha, *beta, *gamma;
    T* omega;
    const uint8_t* stop;
    struct scalars {
        T alpha, omega;
        bool stopped;
    };
    __device__ scalars load(int64_t c) const
    {
        const T bt = beta[c];
        return {alpha[c], bt != T(0) ? gamma[c] / bt : T(0), status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void store(int64_t c, const scalars& s) const { omega[c] = s.omega; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = in[0] + (s.alpha * in[3] + s.omega * in[4]);
        out[1] = in[1] - s.omega * in[2];
    }
};

// x += alpha y for columns that stopped but are not finalized.  in = {x, y}, out = {x}
template <typename T>
struct op_bicgstab_finalize {
    const T* alpha;
    const uint8_t* stop;
    struct scalars {
        T alpha;
        bool skip;
    };
    __device__ scalars load(int64_t c) const
    {
        const uint8_t st = stop[c];
        return {alpha[c], !(status_has_stopped(st) && !(st & 0x40))};
    }
    __device__ bool skip(const scalars& s) const { return s.skip; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = in[0] + s.alpha * in[1];
    }
};

// after the update: stopped columns become finalized (stopping_status::finalize).
// A separate launch: the update kernel reads the flag this one sets.
__global__ void finalize_status_kernel(int64_t cols, uint8_t* stop)
{
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j < cols && status_has_stopped(stop[j])) stop[j] |= uint8_t(0x40);
}

// --------------------------------------------------------------------- cgs
// beta = rho / rho_prev (kept if rho_prev == 0) ; u = r + beta q ;
// p = u + beta (q + beta p).   in = {r, q, p}, out = {u, p}
template <typename T>
struct op_cgs_step1 {
    const T *rho, *rho_prev;
    T* beta;
    const uint8_t* stop;
    struct scalars {
        T beta;
        bool stopped;
    };
    __device__ scalars load(int64_t c) const
    {
        const T pr = rho_prev[c];
        // rho_prev == 0: the old beta is used and written back unchanged
        return {pr != T(0) ? rho[c] / pr : beta[c], status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void store(int64_t c, const scalars& s) const { beta[c] = s.beta; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        const T u = in[0] + s.beta * in[1];
        out[0] = u;
        out[1] = u + s.beta * (in[1] + s.beta * in[2]);
    }
};

// alpha = rho / gamma (kept if gamma == 0) ; q = u - alpha v_hat ; t = u + q
// in = {u, v_hat}, out = {q, t}
template <typename T>
struct op_cgs_step2 {
    const T *rho, *gamma;
    T* alpha;
    const uint8_t* stop;
    struct scalars {
        T alpha;
        bool stopped;
    };
    __device__ scalars load(int64_t c) const
    {
        const T g = gamma[c];
        return {g != T(0) ? rho[c] / g : alpha[c], status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void store(int64_t c, const scalars& s) const { alpha[c] = s.alpha; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        const T q = in[0] - s.alpha * in[1];
        out[0] = q;
        out[1] = in[0] + q;
    }
};

// x += alpha u_hat ; r -= alpha t.   in = {x, u_hat, r, t}, out = {x, r}
template <typename T>
struct op_cgs_step3 {
    const T* alpha;
    const uint8_t* stop;
    struct scalars {
        T alpha;
        bool stopped;
    };
    __device__ scalars load(int64_t c) const { return {alpha[c], status_has_stopped(stop[c])}; }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = in[0] + s.alpha * in[1];
        out[1] = in[2] - s.alpha * in[3];
    }
};

// --------------------------------------------------------------------- fcg
// p = z + (rho_t / prev_rho) p ; p = z if prev_rho == 0.   in = {z, p}, out = {p}
template <typename T>
struct op_fcg_step1 {
    const T *rho_t, *prev_rho;
    const uint8_t* stop;
    struct scalars {
        T tmp;
        bool plain, stopped;
    };
    __device__ scalars load(int64_t c) const
    {
        const T pr = prev_rho[c];
        const bool z = pr == T(0);
        return {z ? T(0) : rho_t[c] / pr, z, status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = s.plain ? in[0] : in[0] + s.tmp * in[1];
    }
};

// tmp = rho / beta ; x += tmp p ; r_new = r - tmp q ; t = r_new - r  (beta != 0)
// in = {x, r, p, q}, out = {x, r, t}
template <typename T>
struct op_fcg_step2 {
    const T *beta, *rho;
    const uint8_t* stop;
    struct scalars {
        T tmp;
        bool noop;
    };
    __device__ scalars load(int64_t c) const
    {
        const T bt = beta[c];
        const bool nz = bt != T(0);
        return {nz ? rho[c] / bt : T(0), !nz || status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.noop; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        const T rn = in[1] - s.tmp * in[3];
        out[0] = in[0] + s.tmp * in[2];
        out[1] = rn;
        out[2] = rn - in[1];
    }
};

// ----------------------------------------------------------------- pipe_cg
// p = z, q = w, f = m, g = n.   in = {z, w, m, n}, out = {p, q, f, g}
template <typename T>
struct op_copy4 {
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = in[k];
    }
};

template <typename T>
__global__ void copy_scalars_kernel(int64_t cols, T* dst, const T* src)
{
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j < cols) dst[j] = src[j];
}

// tmp = rho / beta ; x += tmp p ; r -= tmp q ; z1 -= tmp f ; z2 = z1 ; w -= tmp g
// in = {x, r, z1, w, p, q, f, g}, out = {x, r, z1, z2, w}
template <typename T>
struct op_pipe_cg_step1 {
    const T *rho, *beta;
    const uint8_t* stop;
    struct scalars {
        T tmp;
        bool noop;
    };
    __device__ scalars load(int64_t c) const
    {
        const T bt = beta[c];
        const bool nz = bt != T(0);
        return {nz ? rho[c] / bt : T(0), !nz || status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.noop; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        const T z = in[2] - s.tmp * in[6];
        out[0] = in[0] + s.tmp * in[4];
        out[1] = in[1] - s.tmp * in[5];
        out[2] = z;
        out[3] = z;
        out[4] = in[3] - s.tmp * in[7];
    }
};

// tmp = rho / prev_rho ; beta = delta - |tmp|^2 beta (delta if that is 0) ;
// p = z + tmp p ; q = w + tmp q ; f = m + tmp f ; g = n + tmp g
// prev_rho == 0: beta = delta and plain copies.
// in = {z, w, m, n, p, q, f, g}, out = {p, q, f, g}
template <typename T>
struct op_pipe_cg_step2 {
    const T *prev_rho, *rho, *delta;
    T* beta;
    const uint8_t* stop;
    struct scalars {
        T tmp;
        bool plain, stopped;
    };
    __device__ scalars load(int64_t c) const
    {
        const T pr = prev_rho[c];
        const bool nz = pr != T(0);
        return {nz ? rho[c] / pr : T(0), !nz, status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    // the only reader and writer of beta is this one thread
    __device__ void store(int64_t c, const scalars& s) const
    {
        if (s.plain) {
            beta[c] = delta[c];
        } else {
            const real_t<T> a = abs_v(s.tmp);          // |rho / prev_rho| (real for complex T as well)
            T b = delta[c] - a * a * beta[c];
            if (b == T(0)) b = delta[c];
            beta[c] = b;
        }
    }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = s.plain ? in[k] : in[k] + s.tmp * in[4 + k];
    }
};

// -------------------------------------------------------------------- bicg
// reference/solver/bicg_kernels.cpp:24-110.
// step_1: p = z + t p ; p2 = z2 + t p2 (t = rho / prev_rho; plain copies if prev_rho == 0)
// in = {z, p, z2, p2}, out = {p, p2}
template <typename T>
struct op_bicg_step1 {
    const T *rho, *prev_rho;
    const uint8_t* stop;
    struct scalars {
        T tmp;
        bool plain, stopped;
    };
    __device__ scalars load(int64_t c) const
    {
        const T pr = prev_rho[c];
        const bool z = pr == T(0);
        return {z ? T(0) : rho[c] / pr, z, status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = s.plain ? in[0] : in[0] + s.tmp * in[1];
        out[1] = s.plain ? in[2] : in[2] + s.tmp * in[3];
    }
};

// step_2: t = rho / beta ; x += t p ; r -= t q ; r2 -= t q2   (beta != 0)
// in = {x, r, r2, p, q, q2}, out = {x, r, r2}
template <typename T>
struct op_bicg_step2 {
    const T *beta, *rho;
    const uint8_t* stop;
    struct scalars {
        T tmp;
        bool noop;
    };
    __device__ scalars load(int64_t c) const
    {
        const T bt = beta[c];
        const bool nz = bt != T(0);
        return {nz ? rho[c] / bt : T(0), !nz || status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.noop; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = in[0] + s.tmp * in[3];
        out[1] = in[1] - s.tmp * in[4];
        out[2] = in[2] - s.tmp * in[5];
    }
};

// ------------------------------------------------------------------ minres
// reference/solver/minres_kernels.cpp:24-150.  safe_divide(a, b) = b == 0 ? 0 : a / b.
template <typename T>
__device__ __forceinline__ T safe_div(T a, T b)
{
    return b == T(0) ? T(0) : a / b;
}

// scalars of initialize: beta = sqrt(<r, z>) etc.  One thread per column; runs before
// the vector kernel, which reads the new beta.
template <typename T>
__global__ void minres_init_scalars_kernel(int64_t cols, T* beta, T* gamma, T* delta,
                                           T* cos_prev, T* cosv, T* sin_prev, T* sinv,
                                           T* eta_next, T* eta, uint8_t* stop)
{
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= cols) return;
    delta[j] = gamma[j] = cos_prev[j] = sin_prev[j] = sinv[j] = T(0);
    cosv[j] = T(1);
    const T b = sqrt(beta[j]);
    eta_next[j] = eta[j] = beta[j] = b;
    stop[j] = 0;
}

// q = r / beta ; z = z / beta ; p = p_prev = q_prev = q_tilde = 0
// in = {r, z}, out = {q, z, p, p_prev, q_prev, q_tilde}
template <typename T>
struct op_minres_init {
    const T* beta;
    struct scalars {
        T beta;
    };
    __device__ scalars load(int64_t c) const { return {beta[c]}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = safe_div(in[0], s.beta);
        out[1] = safe_div(in[1], s.beta);
        out[2] = out[3] = out[4] = out[5] = T(0);
    }
};

// the Givens update of step_1: all scalars, one thread per column
template <typename T>
__global__ void minres_step1_kernel(int64_t cols, T* alpha, T* beta, T* gamma, T* delta,
                                    T* cos_prev, T* cosv, T* sin_prev, T* sinv, T* eta,
                                    T* eta_next, T* tau, const uint8_t* stop)
{
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= cols || status_has_stopped(stop[j])) return;
    const T bt = sqrt(beta[j]);
    beta[j] = bt;
    delta[j] = sin_prev[j] * gamma[j];
    const T tmp_d = gamma[j], tmp_a = alpha[j];
    const T c_old = cosv[j], s_old = sinv[j], cp_old = cos_prev[j];
    gamma[j] = cp_old * c_old * tmp_d + s_old * tmp_a;
    T a = -conj_v(s_old) * cp_old * tmp_d + c_old * tmp_a;
    cos_prev[j] = c_old;                      // swap(cos, cos_prev), swap(sin, sin_prev)
    sin_prev[j] = s_old;
    T c, sn;
    if (a == T(0)) {
        c = T(0);
        sn = T(1);
    } else {
        const real_t<T> scale = abs_v(a) + abs_v(bt);
        const real_t<T> as = abs_v(a / scale), bs = abs_v(bt / scale);
        const real_t<T> hyp = scale * sqrt(as * as + bs * bs);
        c = conj_v(a) / hyp;
        sn = conj_v(bt) / hyp;
    }
    a = c * a + sn * bt;
    alpha[j] = a;
    cosv[j] = c;
    sinv[j] = sn;
    tau[j] = sn * sn * tau[j];
    const T e = eta_next[j];
    eta[j] = e;
    eta_next[j] = -conj_v(sn) * e;
}

// p = (z - gamma p_prev - delta p) / alpha ; x += cos eta p ; q_prev = v ;
// q_new = v / beta ; v = q_old beta ; z = z_tilde / beta
// in = {x, p, p_prev, z, z_tilde, q, v}, out = {x, p, z, q, q_prev, v}
template <typename T>
struct op_minres_step2 {
    const T *alpha, *beta, *gamma, *delta, *cosv, *eta;
    const uint8_t* stop;
    struct scalars {
        T alpha, beta, gamma, delta, cos, eta;
        bool stopped;
    };
    __device__ scalars load(int64_t c) const
    {
        return {alpha[c], beta[c], gamma[c], delta[c], cosv[c], eta[c], status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        const T p = safe_div(in[3] - s.gamma * in[2] - s.delta * in[1], s.alpha);
        out[0] = in[0] + s.cos * s.eta * p;
        out[1] = p;
        out[2] = safe_div(in[4], s.beta);
        out[3] = safe_div(in[6], s.beta);
        out[4] = in[6];
        out[5] = in[5] * s.beta;
    }
};

// --------------------------------------------------------------------- gcr
// reference/solver/gcr_kernels.cpp:24-88.
// step_1: t = rAp / Ap_norm ; x += t p ; residual -= t Ap   (Ap_norm != 0)
// in = {x, residual, p, Ap}, out = {x, residual}
template <typename T>
struct op_gcr_step1 {
    const real_t<T>* ap_norm;     // Dense<remove_complex<ValueType>> (core/solver/gcr_kernels.hpp:42)
    const T* rap;
    const uint8_t* stop;
    struct scalars {
        T tmp;
        bool noop;
    };
    __device__ scalars load(int64_t c) const
    {
        const real_t<T> nrm = ap_norm[c];
        const bool nz = nrm != real_t<T>(0);
        return {nz ? rap[c] / nrm : T(0), !nz || status_has_stopped(stop[c])};
    }
    __device__ bool skip(const scalars& s) const { return s.noop; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = in[0] + s.tmp * in[2];
        out[1] = in[1] - s.tmp * in[3];
    }
};

// p_bases(0:rows) = residual ; Ap_bases(0:rows) = A_residual.  in = {res, Ares}, out = {p, Ap}
template <typename T>
struct op_copy2 {
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
        out[0] = in[0];
        out[1] = in[1];
    }
};

__global__ void zero_u64_kernel(int64_t n, uint64_t* data)
{
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j < n) data[j] = 0;
}

// --------------------------------------------------------------- chebyshev
// coefficients are host scalars of the highest precision (solver::detail::coeff_type
// = double); every element is widened, updated and narrowed back like the reference
// (reference/solver/chebyshev_kernels.cpp:20-66).
// init_update: update = inner ; output += alpha inner.   in = {inner, output}, out = {update, output}
template <typename T>
struct op_cheb_init {
    double alpha;
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
        const double v = static_cast<double>(in[0]);
        out[0] = static_cast<T>(v);
        out[1] = static_cast<T>(static_cast<double>(in[1]) + alpha * v);
    }
};

// val = inner + beta update ; inner = update = val ; output += alpha val
// in = {inner, update, output}, out = {inner, update, output}
template <typename T>
struct op_cheb_update {
    double alpha, beta;
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
        const double v = static_cast<double>(in[0]) + beta * static_cast<double>(in[1]);
        out[0] = static_cast<T>(v);
        out[1] = static_cast<T>(v);
        out[2] = static_cast<T>(static_cast<double>(in[2]) + alpha * v);
    }
};

// ... with complex coefficients (coeff_type<complex<float | double>> = complex<double>)
template <typename R>
__device__ __forceinline__ gkoc_c128 cheb_widen(gkoc_cplx<R> v)
{
    return gkoc_c128{static_cast<double>(v.re), static_cast<double>(v.im)};
}
template <typename T>
__device__ __forceinline__ T cheb_narrow(gkoc_c128 v)
{
    using R = real_t<T>;
    return T{static_cast<R>(v.re), static_cast<R>(v.im)};
}
template <typename T>
struct op_ccheb_init {
    gkoc_c128 alpha;
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
        const gkoc_c128 v = cheb_widen(in[0]);
        out[0] = cheb_narrow<T>(v);
        out[1] = cheb_narrow<T>(cheb_widen(in[1]) + alpha * v);
    }
};
template <typename T>
struct op_ccheb_update {
    gkoc_c128 alpha, beta;
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
        const gkoc_c128 v = cheb_widen(in[0]) + beta * cheb_widen(in[1]);
        out[0] = cheb_narrow<T>(v);
        out[1] = cheb_narrow<T>(v);
        out[2] = cheb_narrow<T>(cheb_widen(in[2]) + alpha * v);
    }
};

__global__ void reset_status_kernel(int64_t cols, uint8_t* stop)
{
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j < cols) stop[j] = 0;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

// ir::initialize (reference/solver/ir_kernels.cpp:20-27): stop_status.reset()
extern "C" int gkoc_ir_initialize(gkoc_stream_t s, int64_t cols, uint8_t* stop_status)
{
    GKOC_REQUIRE(cols >= 0 && (cols == 0 || stop_status), GKOC_E_INVALID, "bad argument");
    if (cols == 0) return GKOC_OK;
    reset_status_kernel<<<dim3(unsigned(ceildiv(cols, 256))), dim3(256), 0, as_stream(s)>>>(
        cols, stop_status);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// pipe_cg::step_1 fused with the three reductions that follow it in a distributed PipeCg
// iteration (one column, unit strides): x += t p ; r -= t q ; z -= t f ; w -= t g with
// t = rho / beta (vectors bit-identical to op_pipe_cg_step1), and from the registers that hold
// the new r, z, w this block's parts of <r, z>, <w, z> and <r, r>.  The eight vectors are read
// once (step_1 + two dots + a norm: 136 B per row, fused: 96 B) and the whole iteration needs
// ONE all-reduce, of the three values this kernel's fold leaves next to each other.
template <typename T>
__global__ __launch_bounds__(256) void pipe_cg_step1_dots_kernel(
    int64_t n, T* __restrict__ x, T* __restrict__ r, T* __restrict__ z, T* __restrict__ w,
    const T* __restrict__ p, const T* __restrict__ q, const T* __restrict__ f,
    const T* __restrict__ g, const T* __restrict__ rho, const T* __restrict__ beta,
    const uint8_t* __restrict__ stop, T* __restrict__ partial, int64_t pstride, bool vec_ok)
{
    __shared__ T lds[4];
    using V = vec16<T>;
    constexpr int W = V::width;
    const T bt = beta[0];
    const bool noop = bt == T(0) || status_has_stopped(stop[0]);
    const T tmp = noop ? T(0) : rho[0] / bt;
    T a_rz = T(0), a_wz = T(0), a_rr = T(0);
    const int64_t tid = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t nthreads = int64_t(gridDim.x) * 256;
    int64_t done = 0;
    if (vec_ok) {
        const int64_t n_vec = n / W;
        for (int64_t i = tid; i < n_vec; i += nthreads) {
            V rv = reinterpret_cast<const V*>(r)[i];
            V zv = reinterpret_cast<const V*>(z)[i];
            V wv = reinterpret_cast<const V*>(w)[i];
            if (!noop) {
                V xv = reinterpret_cast<const V*>(x)[i];
                const V pv = reinterpret_cast<const V*>(p)[i];
                const V qv = reinterpret_cast<const V*>(q)[i];
                const V fv = reinterpret_cast<const V*>(f)[i];
                const V gv = reinterpret_cast<const V*>(g)[i];
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    xv.v[e] = xv.v[e] + tmp * pv.v[e];
                    rv.v[e] = rv.v[e] - tmp * qv.v[e];
                    zv.v[e] = zv.v[e] - tmp * fv.v[e];
                    wv.v[e] = wv.v[e] - tmp * gv.v[e];
                }
                reinterpret_cast<V*>(x)[i] = xv;
                reinterpret_cast<V*>(r)[i] = rv;
                reinterpret_cast<V*>(z)[i] = zv;
                reinterpret_cast<V*>(w)[i] = wv;
            }
#pragma unroll
            for (int e = 0; e < W; ++e) {
                a_rz += rv.v[e] * zv.v[e];
                a_wz += wv.v[e] * zv.v[e];
                a_rr += rv.v[e] * rv.v[e];
            }
        }
        done = n_vec * W;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        T rv = r[i], zv = z[i], wv = w[i];
        if (!noop) {
            x[i] = x[i] + tmp * p[i];
            rv = rv - tmp * q[i];
            zv = zv - tmp * f[i];
            wv = wv - tmp * g[i];
            r[i] = rv;
            z[i] = zv;
            w[i] = wv;
        }
        a_rz += rv * zv;
        a_wz += wv * zv;
        a_rr += rv * rv;
    }
    const T s0 = block_sum<256>(a_rz, lds);
    __syncthreads();
    const T s1 = block_sum<256>(a_wz, lds);
    __syncthreads();
    const T s2 = block_sum<256>(a_rr, lds);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s0;
        partial[pstride + blockIdx.x] = s1;
        partial[2 * pstride + blockIdx.x] = s2;
    }
}

template <typename T>
int launch_pipe_cg_step1_dots(gkoc_stream_t s, int64_t n, T* x, T* r, T* z, T* w, const T* p,
                              const T* q, const T* f, const T* g, const T* rho, const T* beta,
                              const uint8_t* stop, T* out3, void* work, size_t work_bytes)
{
    GKOC_REQUIRE(n >= 0 && out3, GKOC_E_INVALID, "bad argument");
    if (n == 0) {
        GKOC_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(T), as_stream(s)));
        return GKOC_OK;
    }
    GKOC_REQUIRE(x && r && z && w && p && q && f && g && rho && beta && stop && work,
                 GKOC_E_INVALID, "null pointer");
    constexpr int64_t max_blocks = 1024;
    GKOC_REQUIRE(work_bytes >= size_t(3 * max_blocks) * sizeof(T), GKOC_E_WORKSPACE,
                 "workspace too small (gkoc_x_workspace_bytes)");
    const uintptr_t bits = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(r) |
                           reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(w) |
                           reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(q) |
                           reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(g);
    int64_t nb = ceildiv(n, int64_t(256) * vec16<T>::width * 2);
    if (nb > max_blocks) nb = max_blocks;
    T* partial = static_cast<T*>(work);
    pipe_cg_step1_dots_kernel<T><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(
        n, x, r, z, w, p, q, f, g, rho, beta, stop, partial, max_blocks, bits % 16 == 0);
    GKOC_LAUNCH_OK();
    fold_rows_kernel<T><<<dim3(3), dim3(1024), 0, as_stream(s)>>>(nb, max_blocks, partial, out3);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// pipe_cg::step_2 of iteration k AND step_1 of iteration k + 1 (pipe_cg.cpp:247-262, then :196-203)
// with the three reductions that follow: nothing sits between the two element-wise steps but the
// host's (lagged) look at the stopping criterion, and both are masked by stop_status on the device.
//   t2 = rho / prev_rho ; beta' = delta - |t2|^2 beta (delta if that is 0; delta and plain copies if
//   prev_rho == 0) ; p = z + t2 p ; q = w + t2 q ; f = m + t2 f ; g = n + t2 g          (step_2)
//   t1 = rho / beta' ; x += t1 p ; r -= t1 q ; z -= t1 f ; w -= t1 g                      (step_1)
//   partials of <r, z>, <w, z>, <r, r>
// Ten vectors in, eight out (144 B per row; the two kernels it replaces: 96 + 96).  Every element
// goes through the expressions of op_pipe_cg_step2 / op_pipe_cg_step1 in the same order, so the
// vectors are bit-identical to the two kernels.  beta is read from beta_in by every thread and
// written to beta_out (another address: the caller alternates two) by one.
template <typename T>
__global__ __launch_bounds__(256) void pipe_cg_step2_step1_dots_kernel(
    int64_t n, T* __restrict__ x, T* __restrict__ r, T* __restrict__ z, T* __restrict__ w,
    T* __restrict__ p, T* __restrict__ q, T* __restrict__ f, T* __restrict__ g,
    const T* __restrict__ m, const T* __restrict__ nv, const T* __restrict__ prev_rho,
    const T* __restrict__ rho, const T* __restrict__ delta, const T* __restrict__ beta_in,
    T* __restrict__ beta_out, uint8_t* stop, T* __restrict__ partial,
    int64_t pstride, bool vec_ok, step_gate_dev<T> gate)
{
    __shared__ T lds[4];
    using V = vec16<T>;
    constexpr int W = V::width;
    const bool stopped = step_gate_enter(gate, stop, blockIdx.x == 0 && threadIdx.x == 0);
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
    // step_1 of the next iteration: a no-op if beta' is zero or the column has stopped
    const bool noop1 = bnew == T(0) || stopped;
    const T t1 = noop1 ? T(0) : rho[0] / bnew;
    const int64_t tid = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (tid == 0) beta_out[0] = bnew;
    const int64_t nthreads = int64_t(gridDim.x) * 256;
    T a_rz = T(0), a_wz = T(0), a_rr = T(0);
    int64_t done = 0;
    if (vec_ok) {
        const int64_t n_vec = n / W;
        for (int64_t i = tid; i < n_vec; i += nthreads) {
            V rv = reinterpret_cast<const V*>(r)[i];
            V zv = reinterpret_cast<const V*>(z)[i];
            V wv = reinterpret_cast<const V*>(w)[i];
            if (!stopped) {
                V pv = reinterpret_cast<const V*>(p)[i];
                V qv = reinterpret_cast<const V*>(q)[i];
                V fv = reinterpret_cast<const V*>(f)[i];
                V gv = reinterpret_cast<const V*>(g)[i];
                const V mv = reinterpret_cast<const V*>(m)[i];
                const V nn = reinterpret_cast<const V*>(nv)[i];
                V xv = reinterpret_cast<const V*>(x)[i];
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    pv.v[e] = plain ? zv.v[e] : zv.v[e] + t2 * pv.v[e];
                    qv.v[e] = plain ? wv.v[e] : wv.v[e] + t2 * qv.v[e];
                    fv.v[e] = plain ? mv.v[e] : mv.v[e] + t2 * fv.v[e];
                    gv.v[e] = plain ? nn.v[e] : nn.v[e] + t2 * gv.v[e];
                }
                reinterpret_cast<V*>(p)[i] = pv;
                reinterpret_cast<V*>(q)[i] = qv;
                reinterpret_cast<V*>(f)[i] = fv;
                reinterpret_cast<V*>(g)[i] = gv;
                if (!noop1) {
#pragma unroll
                    for (int e = 0; e < W; ++e) {
                        xv.v[e] = xv.v[e] + t1 * pv.v[e];
                        rv.v[e] = rv.v[e] - t1 * qv.v[e];
                        zv.v[e] = zv.v[e] - t1 * fv.v[e];
                        wv.v[e] = wv.v[e] - t1 * gv.v[e];
                    }
                    reinterpret_cast<V*>(x)[i] = xv;
                    reinterpret_cast<V*>(r)[i] = rv;
                    reinterpret_cast<V*>(z)[i] = zv;
                    reinterpret_cast<V*>(w)[i] = wv;
                }
            }
#pragma unroll
            for (int e = 0; e < W; ++e) {
                a_rz += rv.v[e] * zv.v[e];
                a_wz += wv.v[e] * zv.v[e];
                a_rr += rv.v[e] * rv.v[e];
            }
        }
        done = n_vec * W;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        T rv = r[i], zv = z[i], wv = w[i];
        if (!stopped) {
            const T pv = plain ? zv : zv + t2 * p[i];
            const T qv = plain ? wv : wv + t2 * q[i];
            const T fv = plain ? m[i] : m[i] + t2 * f[i];
            const T gv = plain ? nv[i] : nv[i] + t2 * g[i];
            p[i] = pv;
            q[i] = qv;
            f[i] = fv;
            g[i] = gv;
            if (!noop1) {
                x[i] = x[i] + t1 * pv;
                rv = rv - t1 * qv;
                zv = zv - t1 * fv;
                wv = wv - t1 * gv;
                r[i] = rv;
                z[i] = zv;
                w[i] = wv;
            }
        }
        a_rz += rv * zv;
        a_wz += wv * zv;
        a_rr += rv * rv;
    }
    const T s0 = block_sum<256>(a_rz, lds);
    __syncthreads();
    const T s1 = block_sum<256>(a_wz, lds);
    __syncthreads();
    const T s2 = block_sum<256>(a_rr, lds);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s0;
        partial[pstride + blockIdx.x] = s1;
        partial[2 * pstride + blockIdx.x] = s2;
    }
}

template <typename T>
int launch_pipe_cg_step2_step1_dots(gkoc_stream_t s, int64_t n, T* x, T* r, T* z, T* w, T* p, T* q, T* f,
                                    T* g, const T* m, const T* nv, const T* prev_rho, const T* rho,
                                    const T* delta, const T* beta_in, T* beta_out, const uint8_t* stop,
                                    T* out3, void* work, size_t work_bytes, const gkoc_step_gate* gate_in)
{
    GKOC_REQUIRE(n >= 0 && out3, GKOC_E_INVALID, "bad argument");
    GKOC_REQUIRE(!gate_in || !gate_in->tau || (gate_in->orig_tau && gate_in->flags), GKOC_E_INVALID,
                 "gkoc_step_gate: a criterion needs orig_tau and flags");
    GKOC_REQUIRE(!gate_in || n > 0, GKOC_E_NOT_SUPPORTED,
                 "gkoc_step_gate on an empty local part: run the criterion's own kernel");
    const step_gate_dev<T> gate = step_gate_of<T>(gate_in);
    GKOC_REQUIRE(beta_in && beta_out && beta_in != beta_out, GKOC_E_INVALID,
                 "beta_in and beta_out must be two different scalars");
    if (n == 0) {
        GKOC_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(T), as_stream(s)));
        return GKOC_OK;
    }
    GKOC_REQUIRE(x && r && z && w && p && q && f && g && m && nv && prev_rho && rho && delta && stop && work,
                 GKOC_E_INVALID, "null pointer");
    constexpr int64_t max_blocks = 1024;
    GKOC_REQUIRE(work_bytes >= size_t(3 * max_blocks) * sizeof(T), GKOC_E_WORKSPACE,
                 "workspace too small (gkoc_x_workspace_bytes)");
    const uintptr_t bits = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(r) |
                           reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(w) |
                           reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(q) |
                           reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(g) |
                           reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(nv);
    int64_t nb = ceildiv(n, int64_t(256) * vec16<T>::width * 2);
    if (nb > max_blocks) nb = max_blocks;
    T* partial = static_cast<T*>(work);
    pipe_cg_step2_step1_dots_kernel<T><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(
        n, x, r, z, w, p, q, f, g, m, nv, prev_rho, rho, delta, beta_in, beta_out, const_cast<uint8_t*>(stop),
        partial, max_blocks, bits % 16 == 0, gate);
    GKOC_LAUNCH_OK();
    fold_rows_kernel<T><<<dim3(3), dim3(1024), 0, as_stream(s)>>>(nb, max_blocks, partial, out3);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

#define GKOC_DEF_KRYLOV(T, TN)                                                            \
    /* ------------------------------------------------------------ bicgstab */           \
    extern "C" int gkoc_bicgstab_initialize_##TN(                                         \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* b, int64_t ldb, T* r,       \
        int64_t ldr, T* rr, int64_t ldrr, T* y, int64_t ldy, T* sv, int64_t lds, T* t,    \
        int64_t ldt, T* z, int64_t ldz, T* v, int64_t ldv, T* p, int64_t ldp,             \
        T* prev_rho, T* rho, T* alpha, T* beta, T* gamma, T* omega, uint8_t* stop_status) \
    {                                                                                     \
        scalar_init<T, 6> si{{prev_rho, rho, alpha, beta, gamma, omega},                  \
                             {T(1), T(1), T(1), T(1), T(1), T(1)}};                       \
        int rc = launch_init_scalars<T, 6>(s, cols, si, stop_status);                     \
        if (rc != GKOC_OK) return rc;                                                     \
        operand_list<T, 1, 8> o;                                                          \
        o.in(b, ldb).out(r, ldr).out(rr, ldrr).out(y, ldy).out(sv, lds).out(t, ldt)       \
            .out(z, ldz).out(v, ldv).out(p, ldp);                                         \
        return launch_elementwise<T, op_copy_and_zero<T, 1, 8>, 1, 8>(                    \
            s, rows, cols, o.a, op_copy_and_zero<T, 1, 8>{}, true);                       \
    }                                                                                     \
    extern "C" int gkoc_bicgstab_step_1_##TN(                                             \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* r, int64_t ldr, T* p,       \
        int64_t ldp, const T* v, int64_t ldv, const T* rho, const T* prev_rho,            \
        const T* alpha, const T* omega, const uint8_t* stop_status)                       \
    {                                                                                     \
        operand_list<T, 3, 1> o;                                                          \
        o.in(r, ldr).in(p, ldp).in(v, ldv).out(p, ldp);                                   \
        return launch_elementwise<T, op_bicgstab_step1<T>, 3, 1>(                         \
            s, rows, cols, o.a,                                                           \
            op_bicgstab_step1<T>{rho, prev_rho, alpha, omega, stop_status}, false);       \
    }                                                                                     \
    extern "C" int gkoc_bicgstab_step_2_##TN(                                             \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* r, int64_t ldr, T* sv,      \
        int64_t lds, const T* v, int64_t ldv, const T* rho, T* alpha, const T* beta,      \
        const uint8_t* stop_status)                                                       \
    {                                                                                     \
        operand_list<T, 2, 1> o;                                                          \
        o.in(r, ldr).in(v, ldv).out(sv, lds);                                             \
        return launch_elementwise<T, op_bicgstab_step2<T>, 2, 1>(                         \
            s, rows, cols, o.a, op_bicgstab_step2<T>{rho, beta, alpha, stop_status},      \
            false);                                                                       \
    }                                                                                     \
    extern "C" int gkoc_bicgstab_step_3_##TN(                                             \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx, T* r,             \
        int64_t ldr, const T* sv, int64_t lds, const T* t, int64_t ldt, const T* y,       \
        int64_t ldy, const T* z, int64_t ldz, const T* alpha, const T* beta,              \
        const T* gamma, T* omega, const uint8_t* stop_status)                             \
    {                                                                                     \
        operand_list<T, 5, 2> o;                                                          \
        o.in(x, ldx).in(sv, lds).in(t, ldt).in(y, ldy).in(z, ldz).out(x, ldx)             \
            .out(r, ldr);                                                                 \
        return launch_elementwise<T, op_bicgstab_step3<T>, 5, 2>(                         \
            s, rows, cols, o.a,                                                           \
            op_bicgstab_step3<T>{alpha, beta, gamma, omega, stop_status}, false);         \
    }                                                                                     \
    extern "C" int gkoc_bicgstab_finalize_##TN(                                           \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx, const T* y,       \
        int64_t ldy, const T* alpha, uint8_t* stop_status)                                \
    {                                                                                     \
        operand_list<T, 2, 1> o;                                                          \
        o.in(x, ldx).in(y, ldy).out(x, ldx);                                              \
        int rc = launch_elementwise<T, op_bicgstab_finalize<T>, 2, 1>(                    \
            s, rows, cols, o.a, op_bicgstab_finalize<T>{alpha, stop_status}, false);      \
        if (rc != GKOC_OK || cols <= 0 || rows <= 0) return rc;                           \
        finalize_status_kernel<<<dim3(unsigned(ceildiv(cols, 256))), dim3(256), 0,        \
                                 as_stream(s)>>>(cols, stop_status);                      \
        GKOC_LAUNCH_OK();                                                                 \
        return GKOC_OK;                                                                   \
    }                                                                                     \
    /* ----------------------------------------------------------------- cgs */           \
    extern "C" int gkoc_cgs_initialize_##TN(                                              \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* b, int64_t ldb, T* r,       \
        int64_t ldr, T* r_tld, int64_t ldrt, T* p, int64_t ldp, T* q, int64_t ldq, T* u,  \
        int64_t ldu, T* u_hat, int64_t lduh, T* v_hat, int64_t ldvh, T* t, int64_t ldt,   \
        T* alpha, T* beta, T* gamma, T* rho_prev, T* rho, uint8_t* stop_status)           \
    {                                                                                     \
        scalar_init<T, 5> si{{rho, rho_prev, alpha, beta, gamma},                         \
                             {T(0), T(1), T(1), T(1), T(1)}};                             \
        int rc = launch_init_scalars<T, 5>(s, cols, si, stop_status);                     \
        if (rc != GKOC_OK) return rc;                                                     \
        operand_list<T, 1, 8> o;                                                          \
        o.in(b, ldb).out(r, ldr).out(r_tld, ldrt).out(p, ldp).out(q, ldq).out(u, ldu)     \
            .out(u_hat, lduh).out(v_hat, ldvh).out(t, ldt);                               \
        return launch_elementwise<T, op_copy_and_zero<T, 2, 8>, 1, 8>(                    \
            s, rows, cols, o.a, op_copy_and_zero<T, 2, 8>{}, true);                       \
    }                                                                                     \
    extern "C" int gkoc_cgs_step_1_##TN(                                                  \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* r, int64_t ldr, T* u,       \
        int64_t ldu, T* p, int64_t ldp, const T* q, int64_t ldq, T* beta, const T* rho,   \
        const T* rho_prev, const uint8_t* stop_status)                                    \
    {                                                                                     \
        operand_list<T, 3, 2> o;                                                          \
        o.in(r, ldr).in(q, ldq).in(p, ldp).out(u, ldu).out(p, ldp);                       \
        return launch_elementwise<T, op_cgs_step1<T>, 3, 2>(                              \
            s, rows, cols, o.a, op_cgs_step1<T>{rho, rho_prev, beta, stop_status},        \
            false);                                                                       \
    }                                                                                     \
    extern "C" int gkoc_cgs_step_2_##TN(                                                  \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* u, int64_t ldu,             \
        const T* v_hat, int64_t ldvh, T* q, int64_t ldq, T* t, int64_t ldt, T* alpha,     \
        const T* rho, const T* gamma, const uint8_t* stop_status)                         \
    {                                                                                     \
        operand_list<T, 2, 2> o;                                                          \
        o.in(u, ldu).in(v_hat, ldvh).out(q, ldq).out(t, ldt);                             \
        return launch_elementwise<T, op_cgs_step2<T>, 2, 2>(                              \
            s, rows, cols, o.a, op_cgs_step2<T>{rho, gamma, alpha, stop_status}, false);  \
    }                                                                                     \
    extern "C" int gkoc_cgs_step_3_##TN(                                                  \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* t, int64_t ldt,             \
        const T* u_hat, int64_t lduh, T* r, int64_t ldr, T* x, int64_t ldx,               \
        const T* alpha, const uint8_t* stop_status)                                       \
    {                                                                                     \
        operand_list<T, 4, 2> o;                                                          \
        o.in(x, ldx).in(u_hat, lduh).in(r, ldr).in(t, ldt).out(x, ldx).out(r, ldr);       \
        return launch_elementwise<T, op_cgs_step3<T>, 4, 2>(                              \
            s, rows, cols, o.a, op_cgs_step3<T>{alpha, stop_status}, false);              \
    }                                                                                     \
    /* ----------------------------------------------------------------- fcg */           \
    extern "C" int gkoc_fcg_initialize_##TN(                                              \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* b, int64_t ldb, T* r,       \
        int64_t ldr, T* z, int64_t ldz, T* p, int64_t ldp, T* q, int64_t ldq, T* t,       \
        int64_t ldt, T* prev_rho, T* rho, T* rho_t, uint8_t* stop_status)                 \
    {                                                                                     \
        scalar_init<T, 3> si{{rho, prev_rho, rho_t}, {T(0), T(1), T(1)}};                 \
        int rc = launch_init_scalars<T, 3>(s, cols, si, stop_status);                     \
        if (rc != GKOC_OK) return rc;                                                     \
        operand_list<T, 1, 5> o;                                                          \
        o.in(b, ldb).out(r, ldr).out(t, ldt).out(z, ldz).out(p, ldp).out(q, ldq);         \
        return launch_elementwise<T, op_copy_and_zero<T, 2, 5>, 1, 5>(                    \
            s, rows, cols, o.a, op_copy_and_zero<T, 2, 5>{}, true);                       \
    }                                                                                     \
    extern "C" int gkoc_fcg_step_1_##TN(                                                  \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* p, int64_t ldp, const T* z,       \
        int64_t ldz, const T* rho_t, const T* prev_rho, const uint8_t* stop_status)       \
    {                                                                                     \
        operand_list<T, 2, 1> o;                                                          \
        o.in(z, ldz).in(p, ldp).out(p, ldp);                                              \
        return launch_elementwise<T, op_fcg_step1<T>, 2, 1>(                              \
            s, rows, cols, o.a, op_fcg_step1<T>{rho_t, prev_rho, stop_status}, false);    \
    }                                                                                     \
    extern "C" int gkoc_fcg_step_2_##TN(                                                  \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx, T* r,             \
        int64_t ldr, T* t, int64_t ldt, const T* p, int64_t ldp, const T* q,              \
        int64_t ldq, const T* beta, const T* rho, const uint8_t* stop_status)             \
    {                                                                                     \
        operand_list<T, 4, 3> o;                                                          \
        o.in(x, ldx).in(r, ldr).in(p, ldp).in(q, ldq).out(x, ldx).out(r, ldr)             \
            .out(t, ldt);                                                                 \
        return launch_elementwise<T, op_fcg_step2<T>, 4, 3>(                              \
            s, rows, cols, o.a, op_fcg_step2<T>{beta, rho, stop_status}, false);          \
    }                                                                                     \
    /* ------------------------------------------------------------- pipe_cg */           \
    extern "C" int gkoc_pipe_cg_initialize_1_##TN(                                        \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* b, int64_t ldb, T* r,       \
        int64_t ldr, T* prev_rho, uint8_t* stop_status)                                   \
    {                                                                                     \
        scalar_init<T, 1> si{{prev_rho}, {T(1)}};                                         \
        int rc = launch_init_scalars<T, 1>(s, cols, si, stop_status);                     \
        if (rc != GKOC_OK) return rc;                                                     \
        operand_list<T, 1, 1> o;                                                          \
        o.in(b, ldb).out(r, ldr);                                                         \
        return launch_elementwise<T, op_copy_and_zero<T, 1, 1>, 1, 1>(                    \
            s, rows, cols, o.a, op_copy_and_zero<T, 1, 1>{}, true);                       \
    }                                                                                     \
    extern "C" int gkoc_pipe_cg_initialize_2_##TN(                                        \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* p, int64_t ldp, T* q,             \
        int64_t ldq, T* f, int64_t ldf, T* g, int64_t ldg, T* beta, const T* z,           \
        int64_t ldz, const T* w, int64_t ldw, const T* m, int64_t ldm, const T* n,        \
        int64_t ldn, const T* delta)                                                      \
    {                                                                                     \
        if (cols > 0) {                                                                   \
            copy_scalars_kernel<T><<<dim3(unsigned(ceildiv(cols, 256))), dim3(256), 0,    \
                                     as_stream(s)>>>(cols, beta, delta);                  \
            GKOC_LAUNCH_OK();                                                             \
        }                                                                                 \
        operand_list<T, 4, 4> o;                                                          \
        o.in(z, ldz).in(w, ldw).in(m, ldm).in(n, ldn).out(p, ldp).out(q, ldq)             \
            .out(f, ldf).out(g, ldg);                                                     \
        return launch_elementwise<T, op_copy4<T>, 4, 4>(s, rows, cols, o.a,               \
                                                        op_copy4<T>{}, true);             \
    }                                                                                     \
    extern "C" int gkoc_pipe_cg_step_1_##TN(                                              \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx, T* r,             \
        int64_t ldr, T* z1, int64_t ldz1, T* z2, int64_t ldz2, T* w, int64_t ldw,         \
        const T* p, int64_t ldp, const T* q, int64_t ldq, const T* f, int64_t ldf,        \
        const T* g, int64_t ldg, const T* rho, const T* beta,                             \
        const uint8_t* stop_status)                                                       \
    {                                                                                     \
        operand_list<T, 8, 5> o;                                                          \
        o.in(x, ldx).in(r, ldr).in(z1, ldz1).in(w, ldw).in(p, ldp).in(q, ldq)             \
            .in(f, ldf).in(g, ldg).out(x, ldx).out(r, ldr).out(z1, ldz1).out(z2, ldz2)    \
            .out(w, ldw);                                                                 \
        return launch_elementwise<T, op_pipe_cg_step1<T>, 8, 5>(                          \
            s, rows, cols, o.a, op_pipe_cg_step1<T>{rho, beta, stop_status}, false);      \
    }                                                                                     \
    extern "C" int gkoc_pipe_cg_step_2_##TN(                                              \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* beta, T* p, int64_t ldp, T* q,    \
        int64_t ldq, T* f, int64_t ldf, T* g, int64_t ldg, const T* z, int64_t ldz,       \
        const T* w, int64_t ldw, const T* m, int64_t ldm, const T* n, int64_t ldn,        \
        const T* prev_rho, const T* rho, const T* delta, const uint8_t* stop_status)      \
    {                                                                                     \
        operand_list<T, 8, 4> o;                                                          \
        o.in(z, ldz).in(w, ldw).in(m, ldm).in(n, ldn).in(p, ldp).in(q, ldq).in(f, ldf)    \
            .in(g, ldg).out(p, ldp).out(q, ldq).out(f, ldf).out(g, ldg);                  \
        return launch_elementwise<T, op_pipe_cg_step2<T>, 8, 4>(                          \
            s, rows, cols, o.a,                                                           \
            op_pipe_cg_step2<T>{prev_rho, rho, delta, beta, stop_status}, false);         \
    }

#define GKOC_DEF_KRYLOV_X(T, TN)                                                          \
    extern "C" int gkoc_x_pipe_cg_step_2_step_1_dots_##TN(                                \
        gkoc_stream_t s, int64_t rows, T* x, T* r, T* z, T* w, T* p, T* q, T* f, T* g,    \
        const T* m, const T* n, const T* prev_rho, const T* rho, const T* delta,          \
        const T* beta_in, T* beta_out, const uint8_t* stop_status, T* out3, void* work,   \
        size_t work_bytes, const gkoc_step_gate* gate)                                    \
    {                                                                                     \
        return launch_pipe_cg_step2_step1_dots<T>(s, rows, x, r, z, w, p, q, f, g, m, n,  \
                                                  prev_rho, rho, delta, beta_in, beta_out, \
                                                  stop_status, out3, work, work_bytes,    \
                                                  gate);                                  \
    }                                                                                     \
    extern "C" int gkoc_x_pipe_cg_step_1_dots_##TN(                                       \
        gkoc_stream_t s, int64_t rows, T* x, T* r, T* z, T* w, const T* p, const T* q,    \
        const T* f, const T* g, const T* rho, const T* beta, const uint8_t* stop_status,  \
        T* out3, void* work, size_t work_bytes)                                           \
    {                                                                                     \
        return launch_pipe_cg_step1_dots<T>(s, rows, x, r, z, w, p, q, f, g, rho, beta,   \
                                            stop_status, out3, work, work_bytes);         \
    }

GKOC_DEF_KRYLOV(double, f64)
GKOC_DEF_KRYLOV(float, f32)
// (complex value types: the same templates on gkoc_cplx; agree with the reference to rounding)
GKOC_DEF_KRYLOV(gkoc_c128, c128)
GKOC_DEF_KRYLOV(gkoc_c64, c64)
GKOC_DEF_KRYLOV_X(double, f64)
GKOC_DEF_KRYLOV_X(float, f32)

#define GKOC_DEF_BICG(T, TN)                                                              \
    extern "C" int gkoc_bicg_initialize_##TN(                                             \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* b, int64_t ldb, T* r,       \
        int64_t ldr, T* z, int64_t ldz, T* p, int64_t ldp, T* q, int64_t ldq,             \
        T* prev_rho, T* rho, T* r2, int64_t ldr2, T* z2, int64_t ldz2, T* p2,             \
        int64_t ldp2, T* q2, int64_t ldq2, uint8_t* stop_status)                          \
    {                                                                                     \
        scalar_init<T, 2> si{{rho, prev_rho}, {T(0), T(1)}};                              \
        int rc = launch_init_scalars<T, 2>(s, cols, si, stop_status);                     \
        if (rc != GKOC_OK) return rc;                                                     \
        operand_list<T, 1, 8> o;                                                          \
        o.in(b, ldb).out(r, ldr).out(r2, ldr2).out(z, ldz).out(p, ldp).out(q, ldq)        \
            .out(z2, ldz2).out(p2, ldp2).out(q2, ldq2);                                   \
        return launch_elementwise<T, op_copy_and_zero<T, 2, 8>, 1, 8>(                    \
            s, rows, cols, o.a, op_copy_and_zero<T, 2, 8>{}, true);                       \
    }                                                                                     \
    extern "C" int gkoc_bicg_step_1_##TN(                                                 \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* p, int64_t ldp, const T* z,       \
        int64_t ldz, T* p2, int64_t ldp2, const T* z2, int64_t ldz2, const T* rho,        \
        const T* prev_rho, const uint8_t* stop_status)                                    \
    {                                                                                     \
        operand_list<T, 4, 2> o;                                                          \
        o.in(z, ldz).in(p, ldp).in(z2, ldz2).in(p2, ldp2).out(p, ldp).out(p2, ldp2);      \
        return launch_elementwise<T, op_bicg_step1<T>, 4, 2>(                             \
            s, rows, cols, o.a, op_bicg_step1<T>{rho, prev_rho, stop_status}, false);     \
    }                                                                                     \
    extern "C" int gkoc_bicg_step_2_##TN(                                                 \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx, T* r,             \
        int64_t ldr, T* r2, int64_t ldr2, const T* p, int64_t ldp, const T* q,            \
        int64_t ldq, const T* q2, int64_t ldq2, const T* beta, const T* rho,              \
        const uint8_t* stop_status)                                                       \
    {                                                                                     \
        operand_list<T, 6, 3> o;                                                          \
        o.in(x, ldx).in(r, ldr).in(r2, ldr2).in(p, ldp).in(q, ldq).in(q2, ldq2)           \
            .out(x, ldx).out(r, ldr).out(r2, ldr2);                                       \
        return launch_elementwise<T, op_bicg_step2<T>, 6, 3>(                             \
            s, rows, cols, o.a, op_bicg_step2<T>{beta, rho, stop_status}, false);         \
    }
GKOC_DEF_BICG(double, f64)
GKOC_DEF_BICG(float, f32)
GKOC_DEF_BICG(gkoc_c128, c128)
GKOC_DEF_BICG(gkoc_c64, c64)

#define GKOC_DEF_MINRES(T, TN)                                                            \
    extern "C" int gkoc_minres_initialize_##TN(                                           \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* r, int64_t ldr, T* z,       \
        int64_t ldz, T* p, int64_t ldp, T* p_prev, int64_t ldpp, T* q, int64_t ldq,       \
        T* q_prev, int64_t ldqp, T* q_tilde, int64_t ldqt, T* beta, T* gamma, T* delta,   \
        T* cos_prev, T* cosv, T* sin_prev, T* sinv, T* eta_next, T* eta,                  \
        uint8_t* stop_status)                                                             \
    {                                                                                     \
        if (cols > 0) {                                                                   \
            minres_init_scalars_kernel<T>                                                 \
                <<<dim3(unsigned(ceildiv(cols, 256))), dim3(256), 0, as_stream(s)>>>(     \
                    cols, beta, gamma, delta, cos_prev, cosv, sin_prev, sinv, eta_next,   \
                    eta, stop_status);                                                    \
            GKOC_LAUNCH_OK();                                                             \
        }                                                                                 \
        operand_list<T, 2, 6> o;                                                          \
        o.in(r, ldr).in(z, ldz).out(q, ldq).out(z, ldz).out(p, ldp).out(p_prev, ldpp)     \
            .out(q_prev, ldqp).out(q_tilde, ldqt);                                        \
        return launch_elementwise<T, op_minres_init<T>, 2, 6>(s, rows, cols, o.a,         \
                                                              op_minres_init<T>{beta},    \
                                                              false);                     \
    }                                                                                     \
    extern "C" int gkoc_minres_step_1_##TN(                                               \
        gkoc_stream_t s, int64_t cols, T* alpha, T* beta, T* gamma, T* delta,             \
        T* cos_prev, T* cosv, T* sin_prev, T* sinv, T* eta, T* eta_next, T* tau,          \
        const uint8_t* stop_status)                                                       \
    {                                                                                     \
        if (cols <= 0) return GKOC_OK;                                                    \
        minres_step1_kernel<T><<<dim3(unsigned(ceildiv(cols, 256))), dim3(256), 0,        \
                                 as_stream(s)>>>(cols, alpha, beta, gamma, delta,         \
                                                 cos_prev, cosv, sin_prev, sinv, eta,     \
                                                 eta_next, tau, stop_status);             \
        GKOC_LAUNCH_OK();                                                                 \
        return GKOC_OK;                                                                   \
    }                                                                                     \
    extern "C" int gkoc_minres_step_2_##TN(                                               \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx, T* p,             \
        int64_t ldp, const T* p_prev, int64_t ldpp, T* z, int64_t ldz, const T* z_tilde,  \
        int64_t ldzt, T* q, int64_t ldq, T* q_prev, int64_t ldqp, T* v, int64_t ldv,      \
        const T* alpha, const T* beta, const T* gamma, const T* delta, const T* cosv,     \
        const T* eta, const uint8_t* stop_status)                                         \
    {                                                                                     \
        operand_list<T, 7, 6> o;                                                          \
        o.in(x, ldx).in(p, ldp).in(p_prev, ldpp).in(z, ldz).in(z_tilde, ldzt).in(q, ldq)  \
            .in(v, ldv).out(x, ldx).out(p, ldp).out(z, ldz).out(q, ldq)                   \
            .out(q_prev, ldqp).out(v, ldv);                                               \
        return launch_elementwise<T, op_minres_step2<T>, 7, 6>(                           \
            s, rows, cols, o.a,                                                           \
            op_minres_step2<T>{alpha, beta, gamma, delta, cosv, eta, stop_status},        \
            false);                                                                       \
    }
GKOC_DEF_MINRES(double, f64)
GKOC_DEF_MINRES(float, f32)
GKOC_DEF_MINRES(gkoc_c128, c128)
GKOC_DEF_MINRES(gkoc_c64, c64)

#define GKOC_DEF_GCR(T, TN)                                                               \
    extern "C" int gkoc_gcr_initialize_##TN(gkoc_stream_t s, int64_t rows, int64_t cols,   \
                                            const T* b, int64_t ldb, T* residual,         \
                                            int64_t ldr, uint8_t* stop_status)            \
    {                                                                                     \
        int rc = gkoc_ir_initialize(s, cols, stop_status);                                \
        if (rc != GKOC_OK) return rc;                                                     \
        operand_list<T, 1, 1> o;                                                          \
        o.in(b, ldb).out(residual, ldr);                                                  \
        return launch_elementwise<T, op_copy_and_zero<T, 1, 1>, 1, 1>(                    \
            s, rows, cols, o.a, op_copy_and_zero<T, 1, 1>{}, true);                       \
    }                                                                                     \
    extern "C" int gkoc_gcr_restart_##TN(                                                 \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* residual, int64_t ldr,      \
        const T* a_residual, int64_t ldar, T* p_bases, int64_t ldp, T* ap_bases,          \
        int64_t ldap, uint64_t* final_iter_nums)                                          \
    {                                                                                     \
        if (cols > 0) {                                                                   \
            zero_u64_kernel<<<dim3(unsigned(ceildiv(cols, 256))), dim3(256), 0,           \
                              as_stream(s)>>>(cols, final_iter_nums);                     \
            GKOC_LAUNCH_OK();                                                             \
        }                                                                                 \
        operand_list<T, 2, 2> o;                                                          \
        o.in(residual, ldr).in(a_residual, ldar).out(p_bases, ldp).out(ap_bases, ldap);   \
        return launch_elementwise<T, op_copy2<T>, 2, 2>(s, rows, cols, o.a, op_copy2<T>{}, \
                                                        true);                            \
    }                                                                                     \
    extern "C" int gkoc_gcr_step_1_##TN(                                                  \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx, T* residual,      \
        int64_t ldr, const T* p, int64_t ldp, const T* ap, int64_t ldap,                  \
        const gkoc::real_t<T>* ap_norm, const T* rap, const uint8_t* stop_status)         \
    {                                                                                     \
        operand_list<T, 4, 2> o;                                                          \
        o.in(x, ldx).in(residual, ldr).in(p, ldp).in(ap, ldap).out(x, ldx)                \
            .out(residual, ldr);                                                          \
        return launch_elementwise<T, op_gcr_step1<T>, 4, 2>(                              \
            s, rows, cols, o.a, op_gcr_step1<T>{ap_norm, rap, stop_status}, false);       \
    }
GKOC_DEF_GCR(double, f64)
GKOC_DEF_GCR(float, f32)
GKOC_DEF_GCR(gkoc_c128, c128)
GKOC_DEF_GCR(gkoc_c64, c64)

#define GKOC_DEF_CHEB(T, TN)                                                              \
    extern "C" int gkoc_chebyshev_init_update_##TN(                                       \
        gkoc_stream_t s, int64_t rows, int64_t cols, double alpha, const T* inner_sol,    \
        int64_t ldi, T* update_sol, int64_t ldu, T* output, int64_t ldo)                  \
    {                                                                                     \
        operand_list<T, 2, 2> o;                                                          \
        o.in(inner_sol, ldi).in(output, ldo).out(update_sol, ldu).out(output, ldo);       \
        return launch_elementwise<T, op_cheb_init<T>, 2, 2>(s, rows, cols, o.a,           \
                                                            op_cheb_init<T>{alpha}, true); \
    }                                                                                     \
    extern "C" int gkoc_chebyshev_update_##TN(                                            \
        gkoc_stream_t s, int64_t rows, int64_t cols, double alpha, double beta,           \
        T* inner_sol, int64_t ldi, T* update_sol, int64_t ldu, T* output, int64_t ldo)    \
    {                                                                                     \
        operand_list<T, 3, 3> o;                                                          \
        o.in(inner_sol, ldi).in(update_sol, ldu).in(output, ldo).out(inner_sol, ldi)      \
            .out(update_sol, ldu).out(output, ldo);                                       \
        return launch_elementwise<T, op_cheb_update<T>, 3, 3>(                            \
            s, rows, cols, o.a, op_cheb_update<T>{alpha, beta}, true);                    \
    }
GKOC_DEF_CHEB(double, f64)
GKOC_DEF_CHEB(float, f32)
// complex values: the coefficients are HOST complex<double> values, passed by address
#define GKOC_DEF_CCHEB(T, TN)                                                             \
    extern "C" int gkoc_chebyshev_init_update_##TN(                                       \
        gkoc_stream_t s, int64_t rows, int64_t cols, const gkoc_c128* alpha_host,         \
        const T* inner_sol, int64_t ldi, T* update_sol, int64_t ldu, T* output, int64_t ldo) \
    {                                                                                     \
        GKOC_REQUIRE(alpha_host, GKOC_E_INVALID, "null coefficient");                     \
        operand_list<T, 2, 2> o;                                                          \
        o.in(inner_sol, ldi).in(output, ldo).out(update_sol, ldu).out(output, ldo);       \
        return launch_elementwise<T, op_ccheb_init<T>, 2, 2>(s, rows, cols, o.a,          \
                                                             op_ccheb_init<T>{*alpha_host}, true); \
    }                                                                                     \
    extern "C" int gkoc_chebyshev_update_##TN(                                            \
        gkoc_stream_t s, int64_t rows, int64_t cols, const gkoc_c128* alpha_host,         \
        const gkoc_c128* beta_host, T* inner_sol, int64_t ldi, T* update_sol, int64_t ldu, \
        T* output, int64_t ldo)                                                           \
    {                                                                                     \
        GKOC_REQUIRE(alpha_host && beta_host, GKOC_E_INVALID, "null coefficient");        \
        operand_list<T, 3, 3> o;                                                          \
        o.in(inner_sol, ldi).in(update_sol, ldu).in(output, ldo).out(inner_sol, ldi)      \
            .out(update_sol, ldu).out(output, ldo);                                       \
        return launch_elementwise<T, op_ccheb_update<T>, 3, 3>(                           \
            s, rows, cols, o.a, op_ccheb_update<T>{*alpha_host, *beta_host}, true);       \
    }
GKOC_DEF_CCHEB(gkoc_c128, c128)
GKOC_DEF_CCHEB(gkoc_c64, c64)
