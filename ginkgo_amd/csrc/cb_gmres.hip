// CB-GMRES (compressed-basis GMRES) kernels for gfx950.
//
// Replaces gko::kernels::hip::cb_gmres::{restart, arnoldi, solve_krylov}
// (decl core/solver/cb_gmres_kernels.hpp:101-142; semantics
// reference/solver/cb_gmres_kernels.cpp:31-420; driver core/solver/cb_gmres.cpp:205-480);
// cb_gmres::initialize is common_gmres::initialize (gmres.hip).
//
// The Krylov basis is a 3-d array (krylov_dim + 1) x rows x nrhs, row-major with storage strides
// (st0, st1), kept in a STORAGE type narrower than the arithmetic type T (accessor
// reduced_row_major: double / float / half) or as integers times one scalar per (vector, column)
// (accessor scaled_reduced_row_major<..., 0b101>: int64 / int32 / int16, value = storage * scalar,
// storage = trunc(value / scalar), accessor/scaled_reduced_row_major_reference.hpp:68-82).  Reading
// the basis is what an Arnoldi step costs, so narrower storage is fewer HBM bytes - the kernels
// below widen on load and do all arithmetic in T.
//
// Design (not the stock backend's): the whole Arnoldi step is enqueued without a host round trip.
// The stock kernels copy the number of columns that need re-orthogonalisation to the host after
// every norm (common/cuda_hip/solver/cb_gmres_kernels.cpp:826,893); here the decision lives in a
// device flag per column plus one counter per round, the two possible re-orthogonalisation rounds
// are always enqueued and their blocks leave at once when the counter is zero.  One pass reads
// next_krylov ONCE for its norm and all iter+1 dots (the block keeps its 1024 rows in registers),
// one pass applies all iter+1 updates and produces the partials of the new 2-norm and inf-norm
// from the registers that hold the result.  Reductions: fixed two-level tree (deterministic).
// With one right-hand side every lane works on four consecutive rows (one 32/16/8-byte load per
// basis vector); several right-hand sides take the strided path.
#include <cmath>
#include <limits>
#include <type_traits>

#include "common.hpp"

namespace gkoc {
namespace {

enum : int { CB_KEEP = 0, CB_F32 = 1, CB_F16 = 2, CB_I64 = 3, CB_I32 = 4, CB_I16 = 5 };

// gko::half as stored by the reference executor: via float, round to nearest even; values below
// the normal half range become signed zero and subnormal halves read as zero
// (include/ginkgo/core/base/half.hpp:405-446)
__device__ __forceinline__ uint16_t float_to_half_bits(float v)
{
    const uint32_t f = __float_as_uint(v);
    const uint16_t sign = uint16_t((f >> 16) & 0x8000u);
    const uint32_t e = (f >> 23) & 0xffu, m = f & 0x007fffffu;
    if (e == 0xffu) return uint16_t(sign | 0x7c00u | (m ? 0x03ffu : 0u));
    if (e <= 112u) return sign;
    if (e - 112u >= 31u) return uint16_t(sign | 0x7c00u);
    const uint16_t res = uint16_t(sign | ((e - 112u) << 10) | (m >> 13));
    const uint32_t tail = m & 0x1fffu;
    return uint16_t(res + ((tail > 0x1000u || (tail == 0x1000u && (res & 1u))) ? 1u : 0u));
}

__device__ __forceinline__ float half_bits_to_float(uint16_t h)
{
    _Float16 hv;
    __builtin_memcpy(&hv, &h, 2);
    return (h & 0x7c00u) == 0 ? __uint_as_float(uint32_t(h & 0x8000u) << 16) : float(hv);
}

template <typename T, int KIND>
struct cb_store;
template <typename T>
struct cb_store<T, CB_KEEP> {
    using type = T;
    static constexpr bool scaled = false;
    __device__ static T load(type v, T) { return v; }
    __device__ static type store(T v, T) { return v; }
};
template <>
struct cb_store<double, CB_F32> {
    using type = float;
    static constexpr bool scaled = false;
    __device__ static double load(type v, double) { return double(v); }
    __device__ static type store(double v, double) { return float(v); }
};
template <typename T>
struct cb_store<T, CB_F16> {
    using type = uint16_t;
    static constexpr bool scaled = false;
    __device__ static T load(type v, T) { return T(half_bits_to_float(v)); }
    __device__ static type store(T v, T) { return float_to_half_bits(float(v)); }
};
template <typename T, typename I>
struct cb_store_int {
    using type = I;
    static constexpr bool scaled = true;
    __device__ static T load(type v, T scal) { return T(v) * scal; }
    __device__ static type store(T v, T scal) { return I(v / scal); }
};
template <>
struct cb_store<double, CB_I64> : cb_store_int<double, int64_t> {};
template <typename T>
struct cb_store<T, CB_I32> : cb_store_int<T, int32_t> {};
template <typename T>
struct cb_store<T, CB_I16> : cb_store_int<T, int16_t> {};

// write_scalar's correction factor (core/solver/cb_gmres_accessor.hpp:134-143)
template <typename T, typename S>
__host__ __device__ constexpr T cb_correction()
{
    return std::numeric_limits<S>::is_integer ? T(2) / T(std::numeric_limits<S>::max()) : T(1);
}

template <typename T>
__device__ __forceinline__ T tabs(T v)
{
    return v < T(0) ? -v : v;
}

constexpr int cb_rows_per_thread = 4;
constexpr int cb_chunk = 256 * cb_rows_per_thread;

template <typename S>
struct alignas(sizeof(S) * 4) quad {
    S v[4];
};

// four consecutive rows of one vector (unit stride) or four rows `step` apart
template <bool UNIT, typename S>
__device__ __forceinline__ void load4(const S* __restrict__ p, int64_t step, int64_t avail, S (&o)[4])
{
    if (UNIT) {
        if (avail >= 4 && reinterpret_cast<uintptr_t>(p) % (sizeof(S) * 4) == 0) {
            const quad<S> q = *reinterpret_cast<const quad<S>*>(p);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = q.v[e];
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = e < avail ? p[e] : S(0);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = e < avail ? p[e * step] : S(0);
    }
}

template <bool UNIT, typename S>
__device__ __forceinline__ void store4(S* __restrict__ p, int64_t step, int64_t avail, const S (&o)[4])
{
    if (UNIT) {
        if (avail >= 4 && reinterpret_cast<uintptr_t>(p) % (sizeof(S) * 4) == 0) {
            quad<S> q;
#pragma unroll
            for (int e = 0; e < 4; ++e) q.v[e] = o[e];
            *reinterpret_cast<quad<S>*>(p) = q;
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e < avail) p[e] = o[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e < avail) p[e * step] = o[e];
        }
    }
}

// control block of one arnoldi call (device memory, zeroed per call)
struct cb_ctrl {
    int any[3];  // any[l]: columns that take re-orthogonalisation round l+1
};

// block -> (first row, rows available) of this thread; UNIT: 4 consecutive rows, else rows
// tid, tid+256, ... inside the block's 1024-row chunk
template <bool UNIT>
__device__ __forceinline__ void thread_rows(int64_t rows, int64_t& r0, int64_t& rstep, int64_t& avail)
{
    const int64_t base = int64_t(blockIdx.x) * cb_chunk;
    if (UNIT) {
        r0 = base + int64_t(threadIdx.x) * 4;
        rstep = 1;
        avail = rows - r0;
    } else {
        r0 = base + threadIdx.x;
        rstep = 256;
        avail = r0 < rows ? (rows - r0 + 255) / 256 : 0;
    }
    if (avail > 4) avail = 4;
    if (avail < 0) avail = 0;
}

// ---- pass A: norm of next_krylov and its dots with basis vectors 0..num_k-1 -------------------
// mode 0 (first pass): every column gets its norm partial, stopped columns no dots.
// mode 1 (re-orthogonalisation round): only columns with active[col], no norm; the whole grid
// leaves when ctrl->any[round-1] == 0.
template <typename T, int KIND, bool UNIT>
__global__ __launch_bounds__(256) void cb_dots_stage1(
    int64_t rows, int64_t cols, int num_k, const T* __restrict__ next, int64_t ldn,
    const typename cb_store<T, KIND>::type* __restrict__ bases, int64_t st0, int64_t st1,
    const T* __restrict__ scal, int64_t sst, T* __restrict__ pdot, T* __restrict__ pnrm,
    const uint8_t* __restrict__ stop, const uint8_t* __restrict__ active, const cb_ctrl* ctrl,
    int round)
{
    using St = cb_store<T, KIND>;
    using S = typename St::type;
    __shared__ T lds[4];
    if (round > 0 && ctrl->any[round - 1] == 0) return;
    const int64_t col = blockIdx.y;
    const bool skip = round > 0 ? !active[col] : status_has_stopped(stop[col]);
    if (round > 0 && skip) return;
    int64_t r0, rstep, avail;
    thread_rows<UNIT>(rows, r0, rstep, avail);
    T nv[4];
    load4<UNIT, T>(next + r0 * ldn + col, rstep * ldn, avail, nv);
    if (round == 0) {
        T acc = T(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += nv[e] * nv[e];
        const T s = block_sum<256>(acc, lds);
        if (threadIdx.x == 0) pnrm[col * gridDim.x + blockIdx.x] = s;
        __syncthreads();
        if (skip) return;
    }
    for (int k = 0; k < num_k; ++k) {
        S sv[4];
        load4<UNIT, S>(bases + int64_t(k) * st0 + r0 * st1 + col, rstep * st1, avail, sv);
        const T sc = St::scaled ? scal[int64_t(k) * sst + col] : T(1);
        T acc = T(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += nv[e] * St::load(sv[e], sc);
        const T s = block_sum<256>(acc, lds);
        if (threadIdx.x == 0) pdot[(int64_t(k) * cols + col) * gridDim.x + blockIdx.x] = s;
        __syncthreads();
    }
}

// fold of pass A: grid (num_k + 1, cols); x < num_k: coefficient k, x == num_k: the norm (round 0)
template <typename T>
__global__ __launch_bounds__(256) void cb_dots_stage2(
    int64_t nblocks, int64_t cols, int num_k, const T* __restrict__ pdot,
    const T* __restrict__ pnrm, T* __restrict__ h, int64_t ldh, T* __restrict__ buffer,
    int64_t ldb, T* __restrict__ an, uint64_t* __restrict__ final_iter_nums,
    const uint8_t* __restrict__ stop, const uint8_t* __restrict__ active, const cb_ctrl* ctrl,
    int round)
{
    __shared__ T lds[4];
    if (round > 0 && ctrl->any[round - 1] == 0) return;
    const int64_t col = blockIdx.y;
    const int k = blockIdx.x;
    const bool stopped = status_has_stopped(stop[col]);
    if (k == num_k) {
        if (round > 0) return;
        T acc = T(0);
        for (int64_t i = threadIdx.x; i < nblocks; i += 256) acc += pnrm[col * nblocks + i];
        const T s = block_sum<256>(acc, lds);
        if (threadIdx.x == 0) {
            const T eta = T(1.0 / sqrt(2.0));
            an[col] = eta * sqrt(s);  // row 0 of arnoldi_norm
            final_iter_nums[col] += stopped ? 0 : 1;
        }
        return;
    }
    if (round > 0 ? !active[col] : stopped) return;
    T acc = T(0);
    const T* p = pdot + (int64_t(k) * cols + col) * nblocks;
    for (int64_t i = threadIdx.x; i < nblocks; i += 256) acc += p[i];
    const T s = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) {
        if (round == 0) {
            h[int64_t(k) * ldh + col] = s;
        } else {
            buffer[int64_t(k) * ldb + col] = s;
            h[int64_t(k) * ldh + col] += s;
        }
    }
}

// ---- pass B: next_krylov -= sum_k coef(k) * basis_k, term by term in k order; partials of the
// new squared 2-norm and of the inf-norm
template <typename T, int KIND, bool UNIT>
__global__ __launch_bounds__(256) void cb_update_stage1(
    int64_t rows, int64_t cols, int num_k, T* __restrict__ next, int64_t ldn,
    const typename cb_store<T, KIND>::type* __restrict__ bases, int64_t st0, int64_t st1,
    const T* __restrict__ scal, int64_t sst, const T* __restrict__ coef, int64_t ldc,
    T* __restrict__ pnrm, T* __restrict__ pmax, const uint8_t* __restrict__ stop,
    const uint8_t* __restrict__ active, const cb_ctrl* ctrl, int round)
{
    using St = cb_store<T, KIND>;
    using S = typename St::type;
    __shared__ T lds[4];
    if (round > 0 && ctrl->any[round - 1] == 0) return;
    const int64_t col = blockIdx.y;
    if (round > 0 ? !active[col] : status_has_stopped(stop[col])) return;
    int64_t r0, rstep, avail;
    thread_rows<UNIT>(rows, r0, rstep, avail);
    T nv[4];
    load4<UNIT, T>(next + r0 * ldn + col, rstep * ldn, avail, nv);
    int k = 0;
    for (; k + 2 <= num_k; k += 2) {
        S s0[4], s1[4];
        load4<UNIT, S>(bases + int64_t(k) * st0 + r0 * st1 + col, rstep * st1, avail, s0);
        load4<UNIT, S>(bases + int64_t(k + 1) * st0 + r0 * st1 + col, rstep * st1, avail, s1);
        const T c0 = coef[int64_t(k) * ldc + col], c1 = coef[int64_t(k + 1) * ldc + col];
        const T sc0 = St::scaled ? scal[int64_t(k) * sst + col] : T(1);
        const T sc1 = St::scaled ? scal[int64_t(k + 1) * sst + col] : T(1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const T t0 = c0 * St::load(s0[e], sc0);
            nv[e] = nv[e] - t0;
            const T t1 = c1 * St::load(s1[e], sc1);
            nv[e] = nv[e] - t1;
        }
    }
    for (; k < num_k; ++k) {
        S s0[4];
        load4<UNIT, S>(bases + int64_t(k) * st0 + r0 * st1 + col, rstep * st1, avail, s0);
        const T c0 = coef[int64_t(k) * ldc + col];
        const T sc0 = St::scaled ? scal[int64_t(k) * sst + col] : T(1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const T t0 = c0 * St::load(s0[e], sc0);
            nv[e] = nv[e] - t0;
        }
    }
    store4<UNIT, T>(next + r0 * ldn + col, rstep * ldn, avail, nv);
    T acc = T(0), mx = T(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (e < avail) {
            acc += nv[e] * nv[e];
            const T a = tabs(nv[e]);
            mx = mx >= a ? mx : a;
        }
    }
    const T s = block_sum<256>(acc, lds);
    __syncthreads();
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        T m = lds[0];
        for (int w = 1; w < 4; ++w) m = m >= lds[w] ? m : lds[w];
        pnrm[col * gridDim.x + blockIdx.x] = s;
        pmax[col * gridDim.x + blockIdx.x] = m;
    }
}

// fold of pass B, one block per column: arnoldi_norm rows 1 (2-norm) and 2 (inf-norm, scaled
// storage only), then the re-orthogonalisation decision of cb_gmres_kernels.cpp:96-99:
// while (norm_new < norm_old_scaled && round < 2) { norm_old_scaled = eta * norm_new; ... }
template <typename T, bool SCALED>
__global__ __launch_bounds__(256) void cb_update_stage2(
    int64_t nblocks, const T* __restrict__ pnrm, const T* __restrict__ pmax, T* __restrict__ an,
    int64_t ld_an, const uint8_t* __restrict__ stop, uint8_t* __restrict__ active, cb_ctrl* ctrl,
    int round)
{
    __shared__ T lds[4];
    if (round > 0 && ctrl->any[round - 1] == 0) return;
    const int64_t col = blockIdx.x;
    if (round > 0 ? !active[col] : status_has_stopped(stop[col])) {
        if (round == 0 && threadIdx.x == 0) active[col] = 0;
        return;
    }
    T acc = T(0), mx = T(0);
    for (int64_t i = threadIdx.x; i < nblocks; i += 256) {
        acc += pnrm[col * nblocks + i];
        const T a = pmax[col * nblocks + i];
        mx = mx >= a ? mx : a;
    }
    const T s = block_sum<256>(acc, lds);
    __syncthreads();
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        T m = lds[0];
        for (int w = 1; w < 4; ++w) m = m >= lds[w] ? m : lds[w];
        const T n1 = sqrt(s);
        an[ld_an + col] = n1;
        if (SCALED || round > 0) an[2 * ld_an + col] = m;
        const bool again = round < 2 && n1 < an[col];
        active[col] = again ? 1 : 0;
        if (again) {
            const T eta = T(1.0 / sqrt(2.0));
            an[col] = eta * n1;
            atomicAdd(&ctrl->any[round], 1);
        }
    }
}

// ---- pass C: hessenberg(iter+1) = norm, next_krylov /= norm, basis(iter+1) = next_krylov
template <typename T, int KIND, bool UNIT>
__global__ __launch_bounds__(256) void cb_finish_kernel(
    int64_t rows, int64_t cols, int64_t iter, T* __restrict__ next, int64_t ldn,
    typename cb_store<T, KIND>::type* __restrict__ bases, int64_t st0, int64_t st1,
    T* __restrict__ scal, int64_t sst, T* __restrict__ h, int64_t ldh,
    const T* __restrict__ an, int64_t ld_an, const uint8_t* __restrict__ stop)
{
    using St = cb_store<T, KIND>;
    using S = typename St::type;
    const int64_t col = blockIdx.y;
    if (status_has_stopped(stop[col])) return;
    const T n1 = an[ld_an + col];
    T sc = T(1);
    if (St::scaled) sc = (an[2 * ld_an + col] / n1) * cb_correction<T, S>();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (St::scaled) scal[(iter + 1) * sst + col] = sc;
        h[(iter + 1) * ldh + col] = n1;
    }
    int64_t r0, rstep, avail;
    thread_rows<UNIT>(rows, r0, rstep, avail);
    if (avail <= 0) return;
    T nv[4];
    S sv[4];
    load4<UNIT, T>(next + r0 * ldn + col, rstep * ldn, avail, nv);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        nv[e] = nv[e] / n1;
        sv[e] = St::store(nv[e], sc);
    }
    store4<UNIT, T>(next + r0 * ldn + col, rstep * ldn, avail, nv);
    store4<UNIT, S>(bases + (iter + 1) * st0 + r0 * st1 + col, rstep * st1, avail, sv);
}

// givens_rotation + calculate_sin_and_cos + calculate_next_residual_norm
// (reference/solver/cb_gmres_kernels.cpp:151-231), one thread per column, the reference's
// operation order
template <typename T>
__global__ __launch_bounds__(256) void cb_givens_kernel(
    int64_t cols, int64_t iter, T* __restrict__ gsin, int64_t lds_, T* __restrict__ gcos,
    int64_t ldc, T* __restrict__ residual_norm, T* __restrict__ rnc, int64_t ldr,
    T* __restrict__ h, int64_t ldh, const uint8_t* __restrict__ stop)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= cols || status_has_stopped(stop[i])) return;
    for (int64_t j = 0; j < iter; ++j) {
        const T c = gcos[j * ldc + i], s = gsin[j * lds_ + i];
        const T hj = h[j * ldh + i], hj1 = h[(j + 1) * ldh + i];
        const T temp = c * hj + s * hj1;
        h[(j + 1) * ldh + i] = -s * hj + c * hj1;
        h[j * ldh + i] = temp;
    }
    const T this_h = h[iter * ldh + i];
    const T next_h = h[(iter + 1) * ldh + i];
    T c, s;
    if (this_h == T(0)) {
        c = T(0);
        s = T(1);
    } else {
        const T scale = tabs(this_h) + tabs(next_h);
        const T hyp = scale * sqrt(tabs(this_h / scale) * tabs(this_h / scale) +
                                   tabs(next_h / scale) * tabs(next_h / scale));
        c = this_h / hyp;
        s = next_h / hyp;
    }
    gcos[iter * ldc + i] = c;
    gsin[iter * lds_ + i] = s;
    h[iter * ldh + i] = c * this_h + s * next_h;
    h[(iter + 1) * ldh + i] = T(0);
    const T r = rnc[iter * ldr + i];
    const T rn = -s * r;
    rnc[(iter + 1) * ldr + i] = rn;
    rnc[iter * ldr + i] = c * r;
    residual_norm[i] = tabs(rn);
}

// ---- restart ------------------------------------------------------------------------------
template <typename T, bool UNIT>
__global__ __launch_bounds__(256) void cb_restart_stage1(int64_t rows, const T* __restrict__ res,
                                                         int64_t ldr, T* __restrict__ pnrm,
                                                         T* __restrict__ pmax)
{
    __shared__ T lds[4];
    const int64_t col = blockIdx.y;
    int64_t r0, rstep, avail;
    thread_rows<UNIT>(rows, r0, rstep, avail);
    T rv[4];
    load4<UNIT, T>(res + r0 * ldr + col, rstep * ldr, avail, rv);
    T acc = T(0), mx = T(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        acc += rv[e] * rv[e];
        const T a = tabs(rv[e]);
        mx = mx >= a ? mx : a;
    }
    const T s = block_sum<256>(acc, lds);
    __syncthreads();
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        T m = lds[0];
        for (int w = 1; w < 4; ++w) m = m >= lds[w] ? m : lds[w];
        pnrm[col * gridDim.x + blockIdx.x] = s;
        pmax[col * gridDim.x + blockIdx.x] = m;
    }
}

// one block per column: residual_norm, arnoldi_norm row 2, the scalars of every basis vector,
// residual_norm_collection column, final_iter_nums
template <typename T, typename S, bool SCALED>
__global__ __launch_bounds__(256) void cb_restart_stage2(
    int64_t nblocks, int64_t krylov_dim, const T* __restrict__ pnrm, const T* __restrict__ pmax,
    T* __restrict__ residual_norm, T* __restrict__ rnc, int64_t ld_rnc, T* __restrict__ an,
    int64_t ld_an, T* __restrict__ scal, int64_t sst, uint64_t* __restrict__ final_iter_nums)
{
    __shared__ T lds[4];
    const int64_t col = blockIdx.x;
    T acc = T(0), mx = T(0);
    for (int64_t i = threadIdx.x; i < nblocks; i += 256) {
        acc += pnrm[col * nblocks + i];
        const T a = pmax[col * nblocks + i];
        mx = mx >= a ? mx : a;
    }
    const T s = block_sum<256>(acc, lds);
    __syncthreads();
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        T m = lds[0];
        for (int w = 1; w < 4; ++w) m = m >= lds[w] ? m : lds[w];
        const T rn = sqrt(s);
        residual_norm[col] = rn;
        rnc[col] = rn;
        final_iter_nums[col] = 0;
        if (SCALED) {
            an[2 * ld_an + col] = m;
            scal[col] = (m / rn) * cb_correction<T, S>();
        }
    }
    for (int64_t k = 1 + threadIdx.x; k < krylov_dim + 1; k += 256) {
        rnc[k * ld_rnc + col] = T(0);
        if (SCALED) scal[k * sst + col] = T(1) * cb_correction<T, S>();
    }
}

template <typename T, int KIND, bool UNIT>
__global__ __launch_bounds__(256) void cb_restart_stage3(
    int64_t rows, const T* __restrict__ res, int64_t ldr, const T* __restrict__ residual_norm,
    typename cb_store<T, KIND>::type* __restrict__ bases, int64_t st1, const T* __restrict__ scal,
    T* __restrict__ next, int64_t ldn)
{
    using St = cb_store<T, KIND>;
    using S = typename St::type;
    const int64_t col = blockIdx.y;
    int64_t r0, rstep, avail;
    thread_rows<UNIT>(rows, r0, rstep, avail);
    if (avail <= 0) return;
    const T rn = residual_norm[col];
    const T sc = St::scaled ? scal[col] : T(1);
    T rv[4];
    S sv[4];
    load4<UNIT, T>(res + r0 * ldr + col, rstep * ldr, avail, rv);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        rv[e] = rv[e] / rn;
        sv[e] = St::store(rv[e], sc);
    }
    store4<UNIT, S>(bases + r0 * st1 + col, rstep * st1, avail, sv);
    store4<UNIT, T>(next + r0 * ldn + col, rstep * ldn, avail, rv);
}

// zero of basis vectors 1..krylov_dim when the storage is not one compact block
template <typename S>
__global__ __launch_bounds__(256) void cb_zero_bases_kernel(int64_t rows, int64_t cols,
                                                            int64_t krylov_dim, S* __restrict__ bases,
                                                            int64_t st0, int64_t st1)
{
    const int64_t total = krylov_dim * rows * cols;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += stride) {
        const int64_t c = idx % cols;
        const int64_t r = (idx / cols) % rows;
        const int64_t k = idx / (cols * rows) + 1;
        bases[k * st0 + r * st1 + c] = S(0);
    }
}

// ---- solve_krylov ---------------------------------------------------------------------------
// reference/solver/cb_gmres_kernels.cpp:234-253: H(i, j) of column k at hessenberg(i, j*cols + k)
template <typename T>
__global__ __launch_bounds__(256) void cb_solve_upper_kernel(int64_t cols, const T* __restrict__ rnc,
                                                             int64_t ldr, const T* __restrict__ h,
                                                             int64_t ldh, T* __restrict__ y,
                                                             int64_t ldy,
                                                             const uint64_t* __restrict__ fin)
{
    const int64_t k = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (k >= cols) return;
    const int64_t m = int64_t(fin[k]);
    for (int64_t i = m - 1; i >= 0; --i) {
        T temp = rnc[i * ldr + k];
        for (int64_t j = i + 1; j < m; ++j) temp -= h[i * ldh + j * cols + k] * y[j * ldy + k];
        y[i * ldy + k] = temp / h[i * ldh + i * cols + k];
    }
}

// before_preconditioner(r, c) = sum_{j < final_iter_nums[c]} basis_j(r, c) * y(j, c), in j order
template <typename T, int KIND, bool UNIT>
__global__ __launch_bounds__(256) void cb_qy_kernel(
    int64_t rows, const typename cb_store<T, KIND>::type* __restrict__ bases, int64_t st0,
    int64_t st1, const T* __restrict__ scal, int64_t sst, const T* __restrict__ y, int64_t ldy,
    T* __restrict__ out, int64_t ldo, const uint64_t* __restrict__ fin)
{
    using St = cb_store<T, KIND>;
    using S = typename St::type;
    const int64_t col = blockIdx.y;
    int64_t r0, rstep, avail;
    thread_rows<UNIT>(rows, r0, rstep, avail);
    if (avail <= 0) return;
    const int64_t m = int64_t(fin[col]);
    T acc[4] = {T(0), T(0), T(0), T(0)};
    int64_t j = 0;
    for (; j + 2 <= m; j += 2) {
        S s0[4], s1[4];
        load4<UNIT, S>(bases + j * st0 + r0 * st1 + col, rstep * st1, avail, s0);
        load4<UNIT, S>(bases + (j + 1) * st0 + r0 * st1 + col, rstep * st1, avail, s1);
        const T y0 = y[j * ldy + col], y1 = y[(j + 1) * ldy + col];
        const T sc0 = St::scaled ? scal[j * sst + col] : T(1);
        const T sc1 = St::scaled ? scal[(j + 1) * sst + col] : T(1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] += St::load(s0[e], sc0) * y0;
            acc[e] += St::load(s1[e], sc1) * y1;
        }
    }
    for (; j < m; ++j) {
        S s0[4];
        load4<UNIT, S>(bases + j * st0 + r0 * st1 + col, rstep * st1, avail, s0);
        const T y0 = y[j * ldy + col];
        const T sc0 = St::scaled ? scal[j * sst + col] : T(1);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += St::load(s0[e], sc0) * y0;
    }
    store4<UNIT, T>(out + r0 * ldo + col, rstep * ldo, avail, acc);
}

// ---------------------------------------------------------------------------------- launchers
template <typename T, int KIND>
struct kind_ok {
    static constexpr bool value = true;
};
template <>
struct kind_ok<float, CB_F32> {
    static constexpr bool value = false;  // float / float is CB_KEEP
};
template <>
struct kind_ok<float, CB_I64> {
    static constexpr bool value = false;
};

struct cb_args {
    int64_t rows, nrhs, krylov_dim;
    void* bases;
    int64_t st0, st1;
    void* scal;
    int64_t sst;
};

inline bool unit_case(const cb_args& a, int64_t ldn) { return a.nrhs == 1 && a.st1 == 1 && ldn == 1; }

template <typename T, int KIND>
int cb_restart_impl(hipStream_t st, const cb_args& a, const T* residual, int64_t ldr,
                    T* residual_norm, T* rnc, int64_t ld_rnc, T* an, int64_t ld_an, T* next,
                    int64_t ldn, uint64_t* fin)
{
    using St = cb_store<T, KIND>;
    using S = typename St::type;
    const int64_t nb = ceildiv(a.rows > 0 ? a.rows : 1, cb_chunk);
    T* work = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&work), size_t(2 * nb * a.nrhs) * sizeof(T)));
    T* pnrm = work;
    T* pmax = work + nb * a.nrhs;
    const dim3 grid(unsigned(nb), unsigned(a.nrhs));
    const bool unit = unit_case(a, ldn) && ldr == 1;
    if (unit) {
        cb_restart_stage1<T, true><<<grid, 256, 0, st>>>(a.rows, residual, ldr, pnrm, pmax);
    } else {
        cb_restart_stage1<T, false><<<grid, 256, 0, st>>>(a.rows, residual, ldr, pnrm, pmax);
    }
    cb_restart_stage2<T, S, St::scaled><<<unsigned(a.nrhs), 256, 0, st>>>(
        nb, a.krylov_dim, pnrm, pmax, residual_norm, rnc, ld_rnc, an, ld_an,
        static_cast<T*>(a.scal), a.sst, fin);
    if (a.rows > 0) {
        if (unit) {
            cb_restart_stage3<T, KIND, true><<<grid, 256, 0, st>>>(
                a.rows, residual, ldr, residual_norm, static_cast<S*>(a.bases), a.st1,
                static_cast<const T*>(a.scal), next, ldn);
        } else {
            cb_restart_stage3<T, KIND, false><<<grid, 256, 0, st>>>(
                a.rows, residual, ldr, residual_norm, static_cast<S*>(a.bases), a.st1,
                static_cast<const T*>(a.scal), next, ldn);
        }
        if (a.krylov_dim > 0) {
            if (a.st1 == a.nrhs && a.st0 == a.rows * a.nrhs) {
                GKOC_HIP(hipMemsetAsync(static_cast<S*>(a.bases) + a.st0, 0,
                                        size_t(a.krylov_dim) * size_t(a.st0) * sizeof(S), st));
            } else {
                int64_t zb = ceildiv(a.krylov_dim * a.rows * a.nrhs, 256);
                if (zb > 4 * max_stream_blocks) zb = 4 * max_stream_blocks;
                cb_zero_bases_kernel<S><<<unsigned(zb), 256, 0, st>>>(
                    a.rows, a.nrhs, a.krylov_dim, static_cast<S*>(a.bases), a.st0, a.st1);
            }
        }
    }
    GKOC_LAUNCH_OK();
    return scratch_free(st, work);
}

template <typename T, int KIND>
int cb_arnoldi_impl(hipStream_t st, const cb_args& a, T* next, int64_t ldn, T* gsin, int64_t ld_sin,
                    T* gcos, int64_t ld_cos, T* residual_norm, T* rnc, int64_t ld_rnc, T* h,
                    int64_t ldh, T* buffer, int64_t ldb, T* an, int64_t ld_an, int64_t iter,
                    uint64_t* fin, const uint8_t* stop)
{
    using St = cb_store<T, KIND>;
    using S = typename St::type;
    const int num_k = int(iter + 1);
    const int64_t nb = ceildiv(a.rows > 0 ? a.rows : 1, cb_chunk);
    // scratch: dot partials, norm / max partials, a buffer if the caller's is absent, the
    // per-column flags and the control block
    const size_t n_pdot = size_t(num_k) * a.nrhs * nb, n_p = size_t(a.nrhs) * nb;
    const bool own_buffer = buffer == nullptr;
    const size_t n_buf = own_buffer ? size_t(num_k) * a.nrhs : 0;
    const size_t val_bytes = (n_pdot + 2 * n_p + n_buf) * sizeof(T);
    const size_t flag_off = (val_bytes + 15) / 16 * 16;
    const size_t ctrl_off = (flag_off + size_t(a.nrhs) + 15) / 16 * 16;
    char* work = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&work), ctrl_off + sizeof(cb_ctrl)));
    T* pdot = reinterpret_cast<T*>(work);
    T* pnrm = pdot + n_pdot;
    T* pmax = pnrm + n_p;
    if (own_buffer) {
        buffer = pmax + n_p;
        ldb = a.nrhs;
    }
    uint8_t* active = reinterpret_cast<uint8_t*>(work + flag_off);
    cb_ctrl* ctrl = reinterpret_cast<cb_ctrl*>(work + ctrl_off);
    GKOC_HIP(hipMemsetAsync(ctrl, 0, sizeof(cb_ctrl), st));
    const dim3 grid(unsigned(nb), unsigned(a.nrhs));
    const dim3 grid_fold(unsigned(num_k + 1), unsigned(a.nrhs));
    const bool unit = unit_case(a, ldn);
    const S* bases = static_cast<const S*>(a.bases);
    const T* scal = static_cast<const T*>(a.scal);
    for (int round = 0; round < 3; ++round) {
        T* coef = round == 0 ? h : buffer;
        const int64_t ldc = round == 0 ? ldh : ldb;
        if (unit) {
            cb_dots_stage1<T, KIND, true><<<grid, 256, 0, st>>>(a.rows, a.nrhs, num_k, next, ldn, bases,
                                                             a.st0, a.st1, scal, a.sst, pdot, pnrm,
                                                             stop, active, ctrl, round);
        } else {
            cb_dots_stage1<T, KIND, false><<<grid, 256, 0, st>>>(a.rows, a.nrhs, num_k, next, ldn,
                                                              bases, a.st0, a.st1, scal, a.sst, pdot,
                                                              pnrm, stop, active, ctrl, round);
        }
        cb_dots_stage2<T><<<grid_fold, 256, 0, st>>>(nb, a.nrhs, num_k, pdot, pnrm, h, ldh, buffer, ldb,
                                                     an, fin, stop, active, ctrl, round);
        if (unit) {
            cb_update_stage1<T, KIND, true><<<grid, 256, 0, st>>>(a.rows, a.nrhs, num_k, next, ldn, bases,
                                                               a.st0, a.st1, scal, a.sst, coef, ldc,
                                                               pnrm, pmax, stop, active, ctrl, round);
        } else {
            cb_update_stage1<T, KIND, false><<<grid, 256, 0, st>>>(a.rows, a.nrhs, num_k, next, ldn,
                                                                bases, a.st0, a.st1, scal, a.sst, coef,
                                                                ldc, pnrm, pmax, stop, active, ctrl,
                                                                round);
        }
        cb_update_stage2<T, St::scaled><<<unsigned(a.nrhs), 256, 0, st>>>(nb, pnrm, pmax, an, ld_an, stop,
                                                                        active, ctrl, round);
    }
    if (unit) {
        cb_finish_kernel<T, KIND, true><<<grid, 256, 0, st>>>(a.rows, a.nrhs, iter, next, ldn,
                                                           static_cast<S*>(a.bases), a.st0, a.st1,
                                                           static_cast<T*>(a.scal), a.sst, h, ldh, an,
                                                           ld_an, stop);
    } else {
        cb_finish_kernel<T, KIND, false><<<grid, 256, 0, st>>>(a.rows, a.nrhs, iter, next, ldn,
                                                            static_cast<S*>(a.bases), a.st0, a.st1,
                                                            static_cast<T*>(a.scal), a.sst, h, ldh, an,
                                                            ld_an, stop);
    }
    cb_givens_kernel<T><<<unsigned(ceildiv(a.nrhs, 256)), 256, 0, st>>>(
        a.nrhs, iter, gsin, ld_sin, gcos, ld_cos, residual_norm, rnc, ld_rnc, h, ldh, stop);
    GKOC_LAUNCH_OK();
    return scratch_free(st, work);
}

template <typename T, int KIND>
int cb_solve_impl(hipStream_t st, const cb_args& a, const T* rnc, int64_t ld_rnc, const T* h,
                  int64_t ldh, T* y, int64_t ldy, T* out, int64_t ldo, const uint64_t* fin)
{
    using S = typename cb_store<T, KIND>::type;
    cb_solve_upper_kernel<T><<<unsigned(ceildiv(a.nrhs, 256)), 256, 0, st>>>(a.nrhs, rnc, ld_rnc, h, ldh,
                                                                            y, ldy, fin);
    if (a.rows > 0) {
        const dim3 grid(unsigned(ceildiv(a.rows, cb_chunk)), unsigned(a.nrhs));
        if (unit_case(a, ldo)) {
            cb_qy_kernel<T, KIND, true><<<grid, 256, 0, st>>>(a.rows, static_cast<const S*>(a.bases),
                                                           a.st0, a.st1, static_cast<const T*>(a.scal),
                                                           a.sst, y, ldy, out, ldo, fin);
        } else {
            cb_qy_kernel<T, KIND, false><<<grid, 256, 0, st>>>(a.rows, static_cast<const S*>(a.bases),
                                                            a.st0, a.st1, static_cast<const T*>(a.scal),
                                                            a.sst, y, ldy, out, ldo, fin);
        }
    }
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// storage kind -> template instance; f is called with std::integral_constant<int, KIND>
template <typename T, typename F>
int cb_dispatch(int kind, F f)
{
    switch (kind) {
    case CB_KEEP: return f(std::integral_constant<int, CB_KEEP>{});
    case CB_F32:
        if constexpr (kind_ok<T, CB_F32>::value) return f(std::integral_constant<int, CB_F32>{});
        break;
    case CB_F16: return f(std::integral_constant<int, CB_F16>{});
    case CB_I64:
        if constexpr (kind_ok<T, CB_I64>::value) return f(std::integral_constant<int, CB_I64>{});
        break;
    case CB_I32: return f(std::integral_constant<int, CB_I32>{});
    case CB_I16: return f(std::integral_constant<int, CB_I16>{});
    default: break;
    }
    set_last_error("cb_gmres: storage kind %d not available for this value type", kind);
    return GKOC_E_NOT_SUPPORTED;
}

inline bool kind_scaled(int kind) { return kind >= CB_I64; }

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_CB_GMRES(T, TN)                                                                    \
    extern "C" int gkoc_cb_gmres_restart_##TN(                                                      \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t krylov_dim, const T* residual,         \
        int64_t ldr, T* residual_norm, T* residual_norm_collection, int64_t ld_rnc,                 \
        T* arnoldi_norm, int64_t ld_an, int storage_kind, void* bases, int64_t st0, int64_t st1,    \
        T* scalars, int64_t sst, T* next_krylov, int64_t ldn, uint64_t* final_iter_nums)            \
    {                                                                                               \
        if (nrhs <= 0) return GKOC_OK;                                                              \
        GKOC_REQUIRE(rows >= 0 && krylov_dim >= 0 && residual_norm && residual_norm_collection &&   \
                         final_iter_nums && (rows == 0 || (residual && bases && next_krylov)),      \
                     GKOC_E_INVALID, "bad argument");                                               \
        GKOC_REQUIRE(!kind_scaled(storage_kind) || (scalars && arnoldi_norm), GKOC_E_INVALID,       \
                     "scaled storage needs scalars and arnoldi_norm");                              \
        const cb_args a{rows, nrhs, krylov_dim, bases, st0, st1, scalars, sst};                     \
        return cb_dispatch<T>(storage_kind, [&](auto kind) {                                        \
            return cb_restart_impl<T, decltype(kind)::value>(                                       \
                as_stream(s), a, residual, ldr, residual_norm, residual_norm_collection, ld_rnc,    \
                arnoldi_norm, ld_an, next_krylov, ldn, final_iter_nums);                            \
        });                                                                                         \
    }                                                                                               \
    extern "C" int gkoc_cb_gmres_arnoldi_##TN(                                                      \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, int64_t iter, T* next_krylov, int64_t ldn,     \
        T* givens_sin, int64_t ld_sin, T* givens_cos, int64_t ld_cos, T* residual_norm,             \
        T* residual_norm_collection, int64_t ld_rnc, int storage_kind, void* bases, int64_t st0,    \
        int64_t st1, T* scalars, int64_t sst, T* hessenberg_iter, int64_t ld_h, T* buffer_iter,     \
        int64_t ld_buf, T* arnoldi_norm, int64_t ld_an, uint64_t* final_iter_nums,                  \
        const uint8_t* stop_status)                                                                 \
    {                                                                                               \
        if (nrhs <= 0) return GKOC_OK;                                                              \
        GKOC_REQUIRE(rows >= 0 && iter >= 0 && givens_sin && givens_cos && residual_norm &&         \
                         residual_norm_collection && hessenberg_iter && arnoldi_norm &&             \
                         final_iter_nums && stop_status && (rows == 0 || (next_krylov && bases)),   \
                     GKOC_E_INVALID, "bad argument");                                               \
        GKOC_REQUIRE(!kind_scaled(storage_kind) || scalars, GKOC_E_INVALID,                         \
                     "scaled storage needs scalars");                                               \
        const cb_args a{rows, nrhs, 0, bases, st0, st1, scalars, sst};                              \
        return cb_dispatch<T>(storage_kind, [&](auto kind) {                                        \
            return cb_arnoldi_impl<T, decltype(kind)::value>(                                       \
                as_stream(s), a, next_krylov, ldn, givens_sin, ld_sin, givens_cos, ld_cos,          \
                residual_norm, residual_norm_collection, ld_rnc, hessenberg_iter, ld_h,             \
                buffer_iter, ld_buf, arnoldi_norm, ld_an, iter, final_iter_nums, stop_status);      \
        });                                                                                         \
    }                                                                                               \
    extern "C" int gkoc_cb_gmres_solve_krylov_##TN(                                                 \
        gkoc_stream_t s, int64_t rows, int64_t nrhs, const T* residual_norm_collection,             \
        int64_t ld_rnc, int storage_kind, const void* bases, int64_t st0, int64_t st1,              \
        const T* scalars, int64_t sst, const T* hessenberg, int64_t ld_h, T* y, int64_t ldy,        \
        T* before_preconditioner, int64_t ldo, const uint64_t* final_iter_nums)                     \
    {                                                                                               \
        if (nrhs <= 0) return GKOC_OK;                                                              \
        GKOC_REQUIRE(rows >= 0 && residual_norm_collection && y && final_iter_nums &&               \
                         (rows == 0 || (bases && before_preconditioner)),                           \
                     GKOC_E_INVALID, "bad argument");                                               \
        GKOC_REQUIRE(!kind_scaled(storage_kind) || scalars, GKOC_E_INVALID,                         \
                     "scaled storage needs scalars");                                               \
        const cb_args a{rows, nrhs, 0, const_cast<void*>(bases), st0, st1,                          \
                        const_cast<T*>(scalars), sst};                                              \
        return cb_dispatch<T>(storage_kind, [&](auto kind) {                                        \
            return cb_solve_impl<T, decltype(kind)::value>(                                         \
                as_stream(s), a, residual_norm_collection, ld_rnc, hessenberg, ld_h, y, ldy,        \
                before_preconditioner, ldo, final_iter_nums);                                       \
        });                                                                                         \
    }

GKOC_DEF_CB_GMRES(double, f64)
GKOC_DEF_CB_GMRES(float, f32)
