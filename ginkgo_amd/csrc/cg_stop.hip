// Fused CG vector updates, stopping-criterion kernels and scalar Jacobi.
//
// Replaces gko::kernels::hip::cg::{initialize, step_1, step_2}
//   (decl core/solver/cg_kernels.hpp:25-48; semantics
//    reference/solver/cg_kernels.cpp:25-100; stock GPU version
//    common/unified/solver/cg_kernels.cpp:25-130),
// residual_norm::residual_norm, implicit_residual_norm::implicit_residual_norm,
// set_all_statuses
//   (reference/stop/residual_norm_kernels.cpp:27-90,
//    reference/stop/criterion_kernels.cpp; stock GPU version
//    common/cuda_hip/stop/residual_norm_kernels.cpp:33-171),
// jacobi::{invert_diagonal, simple_scalar_apply, scalar_apply}
//   (reference/preconditioner/jacobi_kernels.cpp:533-590).
//
// Vector updates: 16-byte loads, all operands of an element group loaded
// before the first store; results bit-identical to the reference (divide,
// multiply, add kept separate).  Algorithmic HBM bytes: initialize 5n,
// step_1 3n, step_2 6n values.
// Stop check: ONE kernel + ONE 2-byte D2H copy per call (the stock backend
// uses two kernels and two blocking 1-byte copies).
#include <cmath>

#include <chrono>

#include "common.hpp"
#include "elementwise.hpp"
#include "fused.hpp"

namespace gkoc {
namespace {

// r = b ; z = p = q = 0.  in[0] = b ; out = r, z, p, q
template <typename T>
struct op_cg_init {
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
        out[0] = in[0];
        out[1] = T(0);
        out[2] = T(0);
        out[3] = T(0);
    }
};

template <typename T>
__global__ void cg_init_scalars_kernel(int64_t cols, T* prev_rho, T* rho,
                                       uint8_t* stop)
{
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j < cols) {
        rho[j] = T(0);
        prev_rho[j] = T(1);
        stop[j] = 0;
    }
}

// p = z + (rho / prev_rho) * p, p = z if prev_rho == 0 ; in = {z, p}, out = {p}
template <typename T>
struct op_cg_step1 {
    const T* rho;
    const T* prev_rho;
    const uint8_t* stop;
    struct scalars {
        T tmp;
        bool zero_prev;
        bool stopped;
    };
    __device__ scalars load(int64_t col) const
    {
        const T pr = prev_rho[col];
        const bool zp = pr == T(0);
        return {zp ? T(0) : rho[col] / pr, zp, status_has_stopped(stop[col])};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = s.zero_prev ? in[0] : in[0] + s.tmp * in[1];
    }
};

// The same with the stopping criterion in front of it (one column): ImplicitResidualNorm /
// ResidualNorm on tau (residual_norm_kernel below) is evaluated by EVERY thread from the device
// scalars, one thread records the verdict (stop status, the two flag bytes), and a column that has
// converged is left alone exactly as if the criterion's own kernel had run first.  Saves that
// kernel - 4.7 us of a 220 us iteration per rank of an 8-rank 256^3 run.  Threads
// that read stop[0] while the recording thread writes it still decide alike: the status can only
// change to "stopped" for the reason they evaluate themselves.
template <typename T, bool IMPLICIT>
struct op_cg_step1_check {
    const T* rho;
    const T* prev_rho;
    const T* tau;
    const T* orig_tau;
    T goal;
    uint8_t stopping_id;
    bool set_finalized;
    uint8_t* stop;
    uint8_t* flags;
    struct scalars {
        T tmp;
        bool zero_prev;
        bool stopped;
        bool converged;
        uint8_t st;
    };
    __device__ scalars load(int64_t col) const
    {
        const T pr = prev_rho[col];
        const bool zp = pr == T(0);
        const uint8_t st = stop[col];
        const T t = IMPLICIT ? sqrt(fabs(tau[col])) : tau[col];
        const bool conv = t <= goal * orig_tau[col];
        return {zp ? T(0) : rho[col] / pr, zp, status_has_stopped(st) || conv, conv, st};
    }
    __device__ bool skip(const scalars& s) const { return s.stopped; }
    __device__ void note(int64_t col, const scalars& s) const
    {
        uint8_t st = s.st;
        if (s.converged && (st & 0x3f) == 0) {
            // stopping_status::converge (stopping_status.hpp:98-107)
            st |= uint8_t(0x80) | (stopping_id & 0x3f);
            if (set_finalized) st |= uint8_t(0x40);
            stop[col] = st;
        }
        flags[0] = uint8_t((st & 0x3f) != 0);      // all_converged (one column)
        flags[1] = uint8_t(s.converged);           // one_changed, as residual_norm_kernel counts it
        __threadfence_system();                    // flags may be pinned host memory that is polled
    }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = s.zero_prev ? in[0] : in[0] + s.tmp * in[1];
    }
};

// t = rho / beta ; x += t p ; r -= t q  (only if beta != 0)
// in = {x, r, p, q}, out = {x, r}
template <typename T>
struct op_cg_step2 {
    const T* beta;
    const T* rho;
    const uint8_t* stop;
    struct scalars {
        T tmp;
        bool noop;
    };
    __device__ scalars load(int64_t col) const
    {
        const T bt = beta[col];
        const bool nz = bt != T(0);
        return {nz ? rho[col] / bt : T(0),
                !nz || status_has_stopped(stop[col])};
    }
    __device__ bool skip(const scalars& s) const { return s.noop; }
    __device__ void apply(const scalars& s, const T* in, T* out) const
    {
        out[0] = in[0] + s.tmp * in[2];
        out[1] = in[1] - s.tmp * in[3];
    }
};

// ---------------------------------------------------- step_2 + ||r||_2 fused
// one column, unit strides: x += t p, r -= t q (t = rho / beta, the masks of
// op_cg_step2) and, from the registers that hold the new r, this block's part
// of sum r^2.  x and r are bit-identical to cg::step_2.
template <typename T>
__global__ __launch_bounds__(256) void cg_step2_norm_kernel(
    int64_t n, T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p,
    const T* __restrict__ q, const T* __restrict__ beta,
    const T* __restrict__ rho, const uint8_t* __restrict__ stop,
    T* __restrict__ partial, bool vec_ok)
{
    __shared__ T lds[4];
    using V = vec16<T>;
    constexpr int W = V::width;
    const T bt = beta[0];
    const bool noop = bt == T(0) || status_has_stopped(stop[0]);
    const T tmp = noop ? T(0) : rho[0] / bt;
    T acc = T(0);
    const int64_t tid = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t nthreads = int64_t(gridDim.x) * 256;
    int64_t done = 0;
    if (vec_ok) {
        const int64_t n_vec = n / W;
        for (int64_t i = tid; i < n_vec; i += nthreads) {
            V rv = reinterpret_cast<const V*>(r)[i];
            if (!noop) {
                V xv = reinterpret_cast<const V*>(x)[i];
                const V pv = reinterpret_cast<const V*>(p)[i];
                const V qv = reinterpret_cast<const V*>(q)[i];
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    xv.v[e] = xv.v[e] + tmp * pv.v[e];
                    rv.v[e] = rv.v[e] - tmp * qv.v[e];
                }
                reinterpret_cast<V*>(x)[i] = xv;
                reinterpret_cast<V*>(r)[i] = rv;
            }
#pragma unroll
            for (int e = 0; e < W; ++e) acc += rv.v[e] * rv.v[e];
        }
        done = n_vec * W;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        T rv = r[i];
        if (!noop) {
            x[i] = x[i] + tmp * p[i];
            rv = rv - tmp * q[i];
            r[i] = rv;
        }
        acc += rv * rv;
    }
    const T s = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

template <typename T>
int launch_step2_norm(gkoc_stream_t s, int64_t n, T* x, T* r, const T* p,
                      const T* q, const T* beta, const T* rho,
                      const uint8_t* stop, T* norm_out, int take_sqrt,
                      void* work, size_t work_bytes)
{
    GKOC_REQUIRE(n >= 0, GKOC_E_INVALID, "negative dimension");
    GKOC_REQUIRE(norm_out, GKOC_E_INVALID, "null result");
    if (n == 0) {
        GKOC_HIP(hipMemsetAsync(norm_out, 0, sizeof(T), as_stream(s)));
        return GKOC_OK;
    }
    GKOC_REQUIRE(x && r && p && q && beta && rho && stop && work, GKOC_E_INVALID,
                 "null pointer");
    GKOC_REQUIRE(work_bytes >= fused_workspace_bytes(n, sizeof(T)), GKOC_E_WORKSPACE,
                 "workspace too small (gkoc_x_workspace_bytes)");
    T* partial = static_cast<T*>(work);
    T* scratch = partial + (fused_workspace_bytes(n, sizeof(T)) / sizeof(T) - fold_chunks);
    const bool vec_ok = (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(r) |
                         reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(q)) % 16 == 0;
    int64_t nb = ceildiv(n, int64_t(256) * vec16<T>::width * 2);
    if (nb > 2048) nb = 2048;
    cg_step2_norm_kernel<T><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(
        n, x, r, p, q, beta, rho, stop, partial, vec_ok);
    GKOC_LAUNCH_OK();
    return fold_partials<T>(s, nb, partial, scratch, norm_out, take_sqrt != 0);
}

// ------------------------------------------------------------------ stop
// flags[0] = all_converged, flags[1] = one_changed
template <typename T, bool IMPLICIT>
__global__ __launch_bounds__(256) void residual_norm_kernel(
    int64_t cols, const T* __restrict__ tau, const T* __restrict__ orig_tau,
    T goal, uint8_t stopping_id, bool set_finalized,
    uint8_t* __restrict__ stop, uint8_t* __restrict__ flags)
{
    int changed = 0;
    int all_stopped = 1;
    for (int64_t j = threadIdx.x; j < cols; j += 256) {
        const T t = IMPLICIT ? sqrt(fabs(tau[j])) : tau[j];
        uint8_t st = stop[j];
        if (t <= goal * orig_tau[j]) {
            // stopping_status::converge (stopping_status.hpp:98-107)
            if ((st & 0x3f) == 0) {
                st |= uint8_t(0x80) | (stopping_id & 0x3f);
                if (set_finalized) st |= uint8_t(0x40);
                stop[j] = st;
            }
            changed = 1;
        }
        if ((st & 0x3f) == 0) all_stopped = 0;
    }
    changed = __syncthreads_or(changed);
    all_stopped = __syncthreads_and(all_stopped);
    if (threadIdx.x == 0) {
        flags[0] = uint8_t(all_stopped != 0);
        flags[1] = uint8_t(changed != 0);
        // flags may live in pinned host memory that the host polls without waiting for an event
        __threadfence_system();
    }
}

__global__ void set_all_statuses_kernel(int64_t cols, uint8_t id,
                                        bool set_finalized, uint8_t* stop)
{
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j < cols) {
        // stopping_status::stop (stopping_status.hpp:85-93)
        uint8_t st = stop[j];
        if ((st & 0x3f) == 0) {
            st |= (id & 0x3f);
            if (set_finalized) st |= uint8_t(0x40);
            stop[j] = st;
        }
    }
}

struct nothing_behind {
    int operator()() const { return GKOC_OK; }
};

// `behind`: what the caller wants ENQUEUED right behind the criterion's kernel - before the host starts to
// wait for the answer (gkoc_x_residual_norm_then_cg_step_1_*: the device goes on while the host decides)
template <typename T, bool IMPLICIT, typename Behind = nothing_behind>
int launch_residual_norm(gkoc_stream_t s, int64_t cols, const T* tau,
                         const T* orig_tau, T goal, uint8_t id,
                         int set_finalized, uint8_t* stop, uint8_t* flags,
                         int* all_converged, int* one_changed, Behind behind = Behind{})
{
    GKOC_REQUIRE((all_converged == nullptr) == (one_changed == nullptr),
                 GKOC_E_INVALID, "pass both host results or neither");
    GKOC_REQUIRE(cols >= 0, GKOC_E_INVALID, "negative dimension");
    GKOC_REQUIRE(flags, GKOC_E_INVALID, "null flag storage");
    if (all_converged) {
        // The synchronous form (Ginkgo's core reads the two answers at once, residual_norm.cpp): the
        // kernel writes them into a pinned word of this thread and the host polls it - the answer is
        // there a few microseconds after the kernel has run, where a 2-byte copy + stream
        // synchronisation costs an interrupt round trip per iteration.  Falls back to the copy if
        // pinned memory is not to be had or the word does not arrive.
        thread_local volatile uint8_t* pinned = nullptr;
        thread_local bool tried = false;
        if (!tried) {
            tried = true;
            void* hp = nullptr;
            if (hipHostMalloc(&hp, 64, hipHostMallocDefault) == hipSuccess) {
                pinned = static_cast<volatile uint8_t*>(hp);
            } else {
                (void)hipGetLastError();
            }
        }
        if (pinned) {
            pinned[0] = pinned[1] = 0xFF;
            residual_norm_kernel<T, IMPLICIT><<<dim3(1), dim3(256), 0, as_stream(s)>>>(
                cols, tau, orig_tau, goal, id, set_finalized != 0, stop, const_cast<uint8_t*>(pinned));
            GKOC_LAUNCH_OK();
            GKOC_TRY(behind());
            // The host is usually a whole iteration AHEAD of the device here (the launches of an iteration take
            // 35 us, its kernels 1.4 ms on 16.7 M rows), so the answer is an iteration away: poll by the CLOCK.
            // (Round 5 gave up after 2^22 looks = 1 ms and let the stream drain - which, with cg::step_1
            // enqueued behind the criterion, waits for that step too and leaves the device idle until the host
            // is back with the product: 40 us per iteration, profiles/r06/r06_ginkgo_api_timeline.txt.)
            const auto poll_start = std::chrono::steady_clock::now();
            bool drained = false;
            for (long spins = 0; pinned[0] == 0xFF || pinned[1] == 0xFF; ++spins) {
                if ((spins & 0xfff) != 0xfff) continue;
                const double waited =
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - poll_start).count();
                if (!drained && waited > 2.0) {
                    // not there after two seconds of polling: let the stream drain (the stores are visible then)
                    GKOC_HIP(hipStreamSynchronize(as_stream(s)));
                    drained = true;
                } else if (drained && waited > 2.5) {
                    set_last_error("residual_norm: the criterion's flags did not arrive in pinned memory");
                    return GKOC_E_INVALID;
                }
            }
            *all_converged = pinned[0];
            *one_changed = pinned[1];
            return GKOC_OK;
        }
    }
    residual_norm_kernel<T, IMPLICIT><<<dim3(1), dim3(256), 0, as_stream(s)>>>(
        cols, tau, orig_tau, goal, id, set_finalized != 0, stop, flags);
    GKOC_LAUNCH_OK();
    GKOC_TRY(behind());
    // asynchronous form: results stay in flags[0..1] on the device and the
    // caller fetches them when it wants to (no host sync here)
    if (!all_converged) return GKOC_OK;
    uint8_t host[2];
    GKOC_HIP(hipMemcpyAsync(host, flags, 2, hipMemcpyDeviceToHost, as_stream(s)));
    GKOC_HIP(hipStreamSynchronize(as_stream(s)));
    *all_converged = host[0];
    *one_changed = host[1];
    return GKOC_OK;
}

// ------------------------------------------------------------ scalar Jacobi
template <typename T>
struct op_invert {
    struct scalars {};
    __device__ scalars load(int64_t) const { return {}; }
    __device__ bool skip(const scalars&) const { return false; }
    __device__ void apply(const scalars&, const T* in, T* out) const
    {
        // reference invert_diagonal: a zero entry is replaced by one
        const T dv = in[0] == T(0) ? T(1) : in[0];
        out[0] = T(1) / dv;
    }
};

// x(i,j) = b(i,j) * d[i]  /  x = beta*x + alpha*b*d ; the diagonal is indexed
// by ROW, so these get their own kernels instead of the column-scalar ops.
template <typename T, bool ADV>
__global__ __launch_bounds__(256) void scalar_jacobi_kernel(
    int64_t rows, int64_t cols, const T* __restrict__ d,
    const T* __restrict__ alpha_p, const T* __restrict__ b, int64_t ldb,
    const T* __restrict__ beta_p, T* __restrict__ x, int64_t ldx)
{
    T alpha = T(1), beta = T(0);
    if (ADV) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    const int64_t total = rows * cols;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total;
         idx += stride) {
        const int64_t row = cols == 1 ? idx : idx / cols;
        const int64_t col = cols == 1 ? 0 : idx - row * cols;
        const T bv = b[row * ldb + col];
        if (ADV) {
            x[row * ldx + col] = beta * x[row * ldx + col] + alpha * bv * d[row];
        } else {
            x[row * ldx + col] = bv * d[row];
        }
    }
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

// cg::{initialize, step_1, step_2} for all four value types (the complex instantiations run the same
// templates on gkoc_cplx, complex_type.hpp)
#define GKOC_DEF_CG_STEPS(T, TN)                                                     \
    extern "C" int gkoc_cg_initialize_##TN(                                    \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* b, int64_t ldb,  \
        T* r, int64_t ldr, T* z, int64_t ldz, T* p, int64_t ldp, T* q,         \
        int64_t ldq, T* prev_rho, T* rho, uint8_t* stop_status)                \
    {                                                                          \
        if (cols > 0) {                                                        \
            cg_init_scalars_kernel<T>                                          \
                <<<dim3(unsigned(ceildiv(cols, 256))), dim3(256), 0,           \
                   as_stream(s)>>>(cols, prev_rho, rho, stop_status);          \
            GKOC_LAUNCH_OK();                                                  \
        }                                                                      \
        ew_operands<T, 1, 4> a{};                                              \
        a.in[0] = b;                                                           \
        a.ld_in[0] = ldb;                                                      \
        a.out[0] = r;                                                          \
        a.ld_out[0] = ldr;                                                     \
        a.out[1] = z;                                                          \
        a.ld_out[1] = ldz;                                                     \
        a.out[2] = p;                                                          \
        a.ld_out[2] = ldp;                                                     \
        a.out[3] = q;                                                          \
        a.ld_out[3] = ldq;                                                     \
        return launch_elementwise<T, op_cg_init<T>, 1, 4>(                     \
            s, rows, cols, a, op_cg_init<T>{}, true);                          \
    }                                                                          \
    extern "C" int gkoc_cg_step_1_##TN(                                        \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* p, int64_t ldp,        \
        const T* z, int64_t ldz, const T* rho, const T* prev_rho,              \
        const uint8_t* stop_status)                                            \
    {                                                                          \
        ew_operands<T, 2, 1> a{};                                              \
        a.in[0] = z;                                                           \
        a.ld_in[0] = ldz;                                                      \
        a.in[1] = p;                                                           \
        a.ld_in[1] = ldp;                                                      \
        a.out[0] = p;                                                          \
        a.ld_out[0] = ldp;                                                     \
        return launch_elementwise<T, op_cg_step1<T>, 2, 1>(                    \
            s, rows, cols, a, op_cg_step1<T>{rho, prev_rho, stop_status},      \
            false);                                                            \
    }                                                                          \
    extern "C" int gkoc_cg_step_2_##TN(                                        \
        gkoc_stream_t s, int64_t rows, int64_t cols, T* x, int64_t ldx, T* r,  \
        int64_t ldr, const T* p, int64_t ldp, const T* q, int64_t ldq,         \
        const T* beta, const T* rho, const uint8_t* stop_status)               \
    {                                                                          \
        ew_operands<T, 4, 2> a{};                                              \
        a.in[0] = x;                                                           \
        a.ld_in[0] = ldx;                                                      \
        a.in[1] = r;                                                           \
        a.ld_in[1] = ldr;                                                      \
        a.in[2] = p;                                                           \
        a.ld_in[2] = ldp;                                                      \
        a.in[3] = q;                                                           \
        a.ld_in[3] = ldq;                                                      \
        a.out[0] = x;                                                          \
        a.ld_out[0] = ldx;                                                     \
        a.out[1] = r;                                                          \
        a.ld_out[1] = ldr;                                                     \
        return launch_elementwise<T, op_cg_step2<T>, 4, 2>(                    \
            s, rows, cols, a, op_cg_step2<T>{beta, rho, stop_status}, false);  \
    }
GKOC_DEF_CG_STEPS(double, f64)
GKOC_DEF_CG_STEPS(float, f32)
GKOC_DEF_CG_STEPS(gkoc_c128, c128)
GKOC_DEF_CG_STEPS(gkoc_c64, c64)

#define GKOC_DEF_CG(T, TN)                                                     \
    extern "C" int gkoc_x_cg_step_1_check_##TN(                                \
        gkoc_stream_t s, int64_t rows, T* p, const T* z, const T* rho,         \
        const T* prev_rho, const T* tau, const T* orig_tau, T goal,            \
        int implicit, uint8_t id, int set_finalized, uint8_t* stop_status,     \
        uint8_t* flags)                                                        \
    {                                                                          \
        GKOC_REQUIRE(rows >= 0 && rho && prev_rho && tau && orig_tau &&        \
                         stop_status && flags,                                 \
                     GKOC_E_INVALID, "null pointer");                          \
        if (rows == 0) {                                                       \
            /* a rank without rows still owes the verdict */                   \
            return implicit ? launch_residual_norm<T, true>(                   \
                                  s, 1, tau, orig_tau, goal, id,               \
                                  set_finalized, stop_status, flags, nullptr,  \
                                  nullptr)                                     \
                            : launch_residual_norm<T, false>(                  \
                                  s, 1, tau, orig_tau, goal, id,               \
                                  set_finalized, stop_status, flags, nullptr,  \
                                  nullptr);                                    \
        }                                                                      \
        ew_operands<T, 2, 1> a{};                                              \
        a.in[0] = z;                                                           \
        a.ld_in[0] = 1;                                                        \
        a.in[1] = p;                                                           \
        a.ld_in[1] = 1;                                                        \
        a.out[0] = p;                                                          \
        a.ld_out[0] = 1;                                                       \
        if (implicit) {                                                        \
            return launch_elementwise<T, op_cg_step1_check<T, true>, 2, 1>(    \
                s, rows, 1, a,                                                 \
                op_cg_step1_check<T, true>{rho, prev_rho, tau, orig_tau, goal, \
                                           id, set_finalized != 0,             \
                                           stop_status, flags},                \
                false);                                                        \
        }                                                                      \
        return launch_elementwise<T, op_cg_step1_check<T, false>, 2, 1>(       \
            s, rows, 1, a,                                                     \
            op_cg_step1_check<T, false>{rho, prev_rho, tau, orig_tau, goal,    \
                                        id, set_finalized != 0, stop_status,   \
                                        flags},                                \
            false);                                                            \
    }                                                                          \
    extern "C" int gkoc_x_cg_step_2_norm_##TN(                                 \
        gkoc_stream_t s, int64_t rows, T* x, T* r, const T* p, const T* q,     \
        const T* beta, const T* rho, const uint8_t* stop_status, T* norm_out,  \
        int take_sqrt, void* work, size_t work_bytes)                          \
    {                                                                          \
        return launch_step2_norm<T>(s, rows, x, r, p, q, beta, rho,            \
                                    stop_status, norm_out, take_sqrt, work,    \
                                    work_bytes);                               \
    }                                                                          \
    extern "C" int gkoc_residual_norm_##TN(                                    \
        gkoc_stream_t s, int64_t cols, const T* tau, const T* orig_tau,        \
        T goal, uint8_t id, int set_finalized, uint8_t* stop_status,           \
        uint8_t* flags_dev, int* all_converged, int* one_changed)              \
    {                                                                          \
        return launch_residual_norm<T, false>(s, cols, tau, orig_tau, goal,    \
                                              id, set_finalized, stop_status,  \
                                              flags_dev, all_converged,        \
                                              one_changed);                    \
    }                                                                          \
    extern "C" int gkoc_implicit_residual_norm_##TN(                           \
        gkoc_stream_t s, int64_t cols, const T* tau, const T* orig_tau,        \
        T goal, uint8_t id, int set_finalized, uint8_t* stop_status,           \
        uint8_t* flags_dev, int* all_converged, int* one_changed)              \
    {                                                                          \
        return launch_residual_norm<T, true>(s, cols, tau, orig_tau, goal,     \
                                             id, set_finalized, stop_status,   \
                                             flags_dev, all_converged,         \
                                             one_changed);                     \
    }                                                                          \
    extern "C" int gkoc_x_residual_norm_then_cg_step_1_##TN(                   \
        gkoc_stream_t s, const T* tau, const T* orig_tau, T goal, uint8_t id,  \
        int set_finalized, int implicit, uint8_t* stop_status,                 \
        uint8_t* flags_dev, int* all_converged, int* one_changed,              \
        int64_t rows, T* p, const T* z, const T* rho, const T* prev_rho)       \
    {                                                                          \
        GKOC_REQUIRE(all_converged && one_changed && p && z && rho && prev_rho, \
                     GKOC_E_INVALID, "null pointer");                          \
        auto step_1 = [&] {                                                    \
            return gkoc_cg_step_1_##TN(s, rows, 1, p, 1, z, 1, rho, prev_rho,  \
                                       stop_status);                           \
        };                                                                     \
        if (implicit) {                                                        \
            return launch_residual_norm<T, true>(                              \
                s, 1, tau, orig_tau, goal, id, set_finalized, stop_status,     \
                flags_dev, all_converged, one_changed, step_1);                \
        }                                                                      \
        return launch_residual_norm<T, false>(                                 \
            s, 1, tau, orig_tau, goal, id, set_finalized, stop_status,         \
            flags_dev, all_converged, one_changed, step_1);                    \
    }                                                                          \
    extern "C" int gkoc_jacobi_invert_diagonal_##TN(gkoc_stream_t s,           \
                                                    int64_t n, const T* diag,  \
                                                    T* inv_diag)               \
    {                                                                          \
        ew_operands<T, 1, 1> a{};                                              \
        a.in[0] = diag;                                                        \
        a.ld_in[0] = 1;                                                        \
        a.out[0] = inv_diag;                                                   \
        a.ld_out[0] = 1;                                                       \
        return launch_elementwise<T, op_invert<T>, 1, 1>(s, n, 1, a,           \
                                                         op_invert<T>{},       \
                                                         true);                \
    }                                                                          \
    extern "C" int gkoc_jacobi_simple_scalar_apply_##TN(                       \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* inv_diag,        \
        const T* b, int64_t ldb, T* x, int64_t ldx)                            \
    {                                                                          \
        if (rows <= 0 || cols <= 0) return GKOC_OK;                            \
        int64_t blocks = ceildiv(rows * cols, 256);                            \
        if (blocks > 4 * max_stream_blocks) blocks = 4 * max_stream_blocks;    \
        scalar_jacobi_kernel<T, false>                                         \
            <<<dim3(unsigned(blocks)), dim3(256), 0, as_stream(s)>>>(          \
                rows, cols, inv_diag, nullptr, b, ldb, nullptr, x, ldx);       \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }                                                                          \
    extern "C" int gkoc_jacobi_scalar_apply_##TN(                              \
        gkoc_stream_t s, int64_t rows, int64_t cols, const T* inv_diag,        \
        const T* alpha, const T* b, int64_t ldb, const T* beta, T* x,          \
        int64_t ldx)                                                           \
    {                                                                          \
        if (rows <= 0 || cols <= 0) return GKOC_OK;                            \
        GKOC_REQUIRE(alpha && beta, GKOC_E_INVALID, "null alpha/beta");        \
        int64_t blocks = ceildiv(rows * cols, 256);                            \
        if (blocks > 4 * max_stream_blocks) blocks = 4 * max_stream_blocks;    \
        scalar_jacobi_kernel<T, true>                                          \
            <<<dim3(unsigned(blocks)), dim3(256), 0, as_stream(s)>>>(          \
                rows, cols, inv_diag, alpha, b, ldb, beta, x, ldx);            \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }

GKOC_DEF_CG(double, f64)
GKOC_DEF_CG(float, f32)

extern "C" int gkoc_set_all_statuses(gkoc_stream_t s, int64_t cols,
                                     uint8_t stopping_id, int set_finalized,
                                     uint8_t* stop_status)
{
    if (cols <= 0) return GKOC_OK;
    set_all_statuses_kernel<<<dim3(unsigned(ceildiv(cols, 256))), dim3(256), 0,
                              as_stream(s)>>>(cols, stopping_id,
                                              set_finalized != 0, stop_status);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}
