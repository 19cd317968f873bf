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
    if (threadIdx.x == 0) result[0] = SQRT ? sqrt(r) : r;
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

// workspace layout: [partials (max_partials) | scratch (fold_chunks)]
inline size_t fused_workspace_bytes(int64_t n, size_t value_size)
{
    const int64_t partials = (n + 63) / 64 + 4096;
    return size_t(partials + fold_chunks) * value_size;
}

#endif

}  // namespace gkoc
