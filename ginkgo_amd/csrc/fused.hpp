// SPDX-License-Identifier: BSD-3-Clause
// Deterministic fold of per-block partial sums written by the fused producer
// kernels (gkoc_x_*): fixed chunking, fixed tree, optional sqrt.
#pragma once
#include "common.hpp"

namespace gkoc {

#ifdef __HIPCC__

constexpr int fold_block = 1024;
constexpr int64_t fold_single_max = 16384;   // one block folds up to this many
constexpr int fold_chunks = 1024;

template <typename T, bool SQRT>
__global__ __launch_bounds__(fold_block) void fold_partials_kernel(
    int64_t count, const T* __restrict__ partial, T* __restrict__ result)
{
    __shared__ T lds[fold_block / 64];
    T acc = T(0);
    for (int64_t i = threadIdx.x; i < count; i += fold_block) acc += partial[i];
    const T r = block_sum<fold_block>(acc, lds);
    if constexpr (SQRT) {
        if (threadIdx.x == 0) result[0] = sqrt(r);
    } else {
        if (threadIdx.x == 0) result[0] = r;
    }
}

// One launch for tens of thousands of partial sums (a rank's share of a strong-scaling run: one per
// wave of its SpMV): eight loads in flight per thread, eight running sums per thread added in a
// fixed order, then the block's tree - fixed chunking, the value does not depend on timing.
// (fold_partials_kernel's one-load-at-a-time loop needs 13 us for 32 768 values, this one 4.)
template <typename T>
__global__ __launch_bounds__(fold_block) void fold_partials_wide_kernel(
    int64_t count, const T* __restrict__ partial, T* __restrict__ result)
{
    __shared__ T lds[fold_block / 64];
    T a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = T(0);
    int64_t i = threadIdx.x;
    for (; i + 7 * fold_block < count; i += 8 * fold_block) {
        T v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = partial[i + k * fold_block];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (i + k * fold_block < count) a[k] += partial[i + k * fold_block];
    }
    const T acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    const T r = block_sum<fold_block>(acc, lds);
    if (threadIdx.x == 0) result[0] = r;
}

// level 1 of a two-level fold: block k folds the k-th contiguous chunk
template <typename T>
__global__ __launch_bounds__(256) void fold_chunks_kernel(
    int64_t count, int64_t chunk, const T* __restrict__ partial,
    T* __restrict__ out)
{
    __shared__ T lds[256 / 64];
    const int64_t lo = int64_t(blockIdx.x) * chunk;
    const int64_t hi = lo + chunk < count ? lo + chunk : count;
    T acc = T(0);
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) acc += partial[i];
    const T r = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// partial[0..count) -> result[0]; scratch holds fold_chunks values
template <typename T>
int fold_partials(gkoc_stream_t s, int64_t count, const T* partial, T* scratch,
                  T* result, bool take_sqrt)
{
    const T* src = partial;
    int64_t cnt = count;
    if (count > fold_single_max) {
        const int64_t chunk = ceildiv(count, int64_t(fold_chunks));
        const int64_t nb = ceildiv(count, chunk);
        fold_chunks_kernel<T><<<dim3(unsigned(nb)), dim3(256), 0, as_stream(s)>>>(
            count, chunk, partial, scratch);
        GKOC_LAUNCH_OK();
        src = scratch;
        cnt = nb;
    }
    if (take_sqrt) {
        fold_partials_kernel<T, true>
            <<<dim3(1), dim3(fold_block), 0, as_stream(s)>>>(cnt, src, result);
    } else {
        fold_partials_kernel<T, false>
            <<<dim3(1), dim3(fold_block), 0, as_stream(s)>>>(cnt, src, result);
    }
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// out[k] = sum of partial[k * pstride .. + count), k = blockIdx.x (fixed tree); several sums of one
// producer kernel folded by one launch (count <= a few thousand)
template <typename T>
__global__ __launch_bounds__(1024) void fold_rows_kernel(int64_t count, int64_t pstride,
                                                          const T* __restrict__ partial,
                                                          T* __restrict__ out)
{
    __shared__ T lds[1024 / 64];
    T acc = T(0);
    const T* p = partial + int64_t(blockIdx.x) * pstride;
    for (int64_t i = threadIdx.x; i < count; i += 1024) acc += p[i];
    const T r = block_sum<1024>(acc, lds);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// fold_partials for TWO sums of one producer kernel (rows of `partial`, pstride apart) in the
// same two launches: the same chunking and the same trees as fold_partials, so each result has
// the bits the one-sum fold would give it.  scratch holds 2 * fold_chunks values.
template <typename T>
__global__ __launch_bounds__(256) void fold_chunks2_kernel(int64_t count, int64_t chunk,
                                                            int64_t pstride,
                                                            const T* __restrict__ partial,
                                                            T* __restrict__ out)
{
    __shared__ T lds[256 / 64];
    const T* p = partial + int64_t(blockIdx.y) * pstride;
    const int64_t lo = int64_t(blockIdx.x) * chunk;
    const int64_t hi = lo + chunk < count ? lo + chunk : count;
    T acc = T(0);
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) acc += p[i];
    const T r = block_sum<256>(acc, lds);
    if (threadIdx.x == 0) out[int64_t(blockIdx.y) * fold_chunks + blockIdx.x] = r;
}

template <typename T>
__global__ __launch_bounds__(fold_block) void fold_partials2_kernel(int64_t count, int64_t pstride,
                                                                    const T* __restrict__ partial,
                                                                    T* __restrict__ out0,
                                                                    T* __restrict__ out1, int sqrt_row)
{
    __shared__ T lds[fold_block / 64];
    const T* p = partial + int64_t(blockIdx.x) * pstride;
    T acc = T(0);
    for (int64_t i = threadIdx.x; i < count; i += fold_block) acc += p[i];
    T r = block_sum<fold_block>(acc, lds);
    if (threadIdx.x == 0) {
        if (int(blockIdx.x) == sqrt_row) r = sqrt(r);
        (blockIdx.x == 0 ? out0 : out1)[0] = r;
    }
}

template <typename T>
int fold_partials2(gkoc_stream_t s, int64_t count, int64_t pstride, const T* partial, T* scratch,
                   T* result0, T* result1, int sqrt_row)
{
    const T* src = partial;
    int64_t cnt = count, stride = pstride;
    if (count > fold_single_max) {
        const int64_t chunk = ceildiv(count, int64_t(fold_chunks));
        const int64_t nb = ceildiv(count, chunk);
        fold_chunks2_kernel<T><<<dim3(unsigned(nb), 2), dim3(256), 0, as_stream(s)>>>(
            count, chunk, pstride, partial, scratch);
        GKOC_LAUNCH_OK();
        src = scratch;
        cnt = nb;
        stride = fold_chunks;
    }
    fold_partials2_kernel<T><<<dim3(2), dim3(fold_block), 0, as_stream(s)>>>(cnt, stride, src, result0,
                                                                            result1, sqrt_row);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// ---- a step kernel that waits for its scalars and judges the criterion itself -------------------
// PipeCg's step kernel needs the all-reduced {rho, delta, ||r||^2}.  They travel on the exchange's
// stream while n = A m runs; a join (event) in front of the step kernel and the criterion's own
// kernel cost the main queue 6-7 + 4.6 us per iteration (profiles/r04_dist_sim_timelines.txt).
// Instead: every block's first thread polls the word that gkoc_gate_open sets on the exchange's
// stream BEHIND the reduction (normally long set - it was needed by the product's boundary rows),
// then every thread evaluates the criterion from the device scalars and ONE thread records it
// (stop status + the two flag bytes, exactly as residual_norm_kernel does).
template <typename T>
struct step_gate_dev {
    const uint32_t* word;     // nullptr: no wait
    uint32_t number;
    const T* tau;             // nullptr: no criterion
    const T* orig_tau;
    T goal;
    int implicit;
    uint8_t stopping_id;
    uint8_t set_finalized;
    uint8_t* flags;
    int fence;                // gate_fence_policy() at launch (common.hpp)
};

template <typename T>
step_gate_dev<T> step_gate_of(const gkoc_step_gate* g)
{
    step_gate_dev<T> d{};
    d.fence = gate_fence_policy();
    if (g) {
        d.word = g->wait_word;
        d.number = g->wait_number;
        d.tau = static_cast<const T*>(g->tau);
        d.orig_tau = static_cast<const T*>(g->orig_tau);
        d.goal = T(g->goal);
        d.implicit = g->implicit;
        d.stopping_id = g->stopping_id;
        d.set_finalized = g->set_finalized;
        d.flags = g->flags;
    }
    return d;
}

// true if the column has stopped (before, or by the criterion now).  All threads of the block call
// it; `recorder`: the one thread of the grid that writes the verdict.
template <typename T>
__device__ __forceinline__ bool step_gate_enter(const step_gate_dev<T>& g, uint8_t* stop, bool recorder)
{
    if (g.word != nullptr) {
        if (threadIdx.x == 0) {
            bool waited = false;
            long spins = 0;
            while (int32_t(__hip_atomic_load(g.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - g.number) <
                   0) {
                __builtin_amdgcn_s_sleep(16);
                waited = true;
                if (++spins > (long(1) << 23)) {     // ~10 s: give up and say so (the caller checks word[1])
                    __hip_atomic_store(const_cast<uint32_t*>(g.word) + 1, 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
            // (nothing of this launch has read the scalars before the word was seen; a block that
            // waited drops what its CU may hold)
            if (g.fence >= 2) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // (a peer on another device: system scope, always)
            } else if (waited || g.fence == 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
    }
    const uint8_t st = stop[0];
    bool stopped = (st & 0x3f) != 0;
    if (g.tau != nullptr) {
        const T t = g.implicit ? sqrt(fabs(g.tau[0])) : g.tau[0];
        const bool conv = t <= g.goal * g.orig_tau[0];
        if (recorder) {
            uint8_t sn = st;
            if (conv && (sn & 0x3f) == 0) {
                sn |= uint8_t(0x80) | (g.stopping_id & 0x3f);      // stopping_status::converge
                if (g.set_finalized) sn |= uint8_t(0x40);
                stop[0] = sn;
            }
            g.flags[0] = uint8_t((sn & 0x3f) != 0);
            g.flags[1] = uint8_t(conv);
            __threadfence_system();
        }
        stopped = stopped || conv;
    }
    return stopped;
}

// workspace layout: [partials (max_partials) | scratch (fold_chunks)]
inline size_t fused_workspace_bytes(int64_t n, size_t value_size)
{
    const int64_t partials = (n + 63) / 64 + 4096;
    return size_t(partials + fold_chunks) * value_size;
}

#endif

}  // namespace gkoc
