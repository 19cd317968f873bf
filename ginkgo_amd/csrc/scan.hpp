// Device-wide exclusive prefix sum (reduce-then-scan, 2048-element tiles).
// Used by components::prefix_sum_nonnegative, sellp::compute_slice_sets,
// jacobi::find_blocks and the format conversions.  Integer-exact.
#pragma once
#include "common.hpp"

namespace gkoc {

#ifdef __HIPCC__

constexpr int scan_block = 256;
constexpr int scan_items = 8;
constexpr int scan_tile = scan_block * scan_items;

template <typename T>
__global__ __launch_bounds__(scan_block) void scan_tile_sums(
    int64_t n, const T* __restrict__ data, T* __restrict__ sums)
{
    __shared__ T lds[scan_block / 64];
    const int64_t base = int64_t(blockIdx.x) * scan_tile;
    T acc = T(0);
#pragma unroll
    for (int u = 0; u < scan_items; ++u) {
        const int64_t i = base + u * scan_block + threadIdx.x;
        if (i < n) acc += data[i];
    }
    const T r = block_sum<scan_block>(acc, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = r;
}

// exclusive scan of one tile, thread t owns items [t*8, t*8+8) of the tile
template <typename T>
__global__ __launch_bounds__(scan_block) void scan_tiles(
    int64_t n, T* __restrict__ data, const T* __restrict__ offsets)
{
    __shared__ T wave_tot[scan_block / 64];
    const int64_t base = int64_t(blockIdx.x) * scan_tile + threadIdx.x * scan_items;
    T v[scan_items];
    T local = T(0);
#pragma unroll
    for (int u = 0; u < scan_items; ++u) {
        const int64_t i = base + u;
        v[u] = i < n ? data[i] : T(0);
        local += v[u];
    }
    // inclusive scan of `local` across the wave
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    T incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    T wave_off = T(0);
    for (int w = 0; w < wid; ++w) wave_off += wave_tot[w];
    T run = (offsets ? offsets[blockIdx.x] : T(0)) + wave_off + (incl - local);
#pragma unroll
    for (int u = 0; u < scan_items; ++u) {
        const int64_t i = base + u;
        if (i < n) data[i] = run;
        run += v[u];
    }
}

// in-place exclusive scan of data[0..n); data[n-1] ends up with the sum of the
// first n-1 inputs (Ginkgo's prefix_sum_nonnegative contract,
// core/components/prefix_sum_kernels.hpp)
template <typename T>
int device_exclusive_scan(hipStream_t st, T* data, int64_t n)
{
    if (n <= 0) return GKOC_OK;
    const int64_t tiles = ceildiv(n, scan_tile);
    if (tiles == 1) {
        scan_tiles<T><<<dim3(1), dim3(scan_block), 0, st>>>(n, data, nullptr);
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    T* sums = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&sums), sizeof(T) * tiles));
    scan_tile_sums<T><<<dim3(unsigned(tiles)), dim3(scan_block), 0, st>>>(n, data, sums);
    GKOC_LAUNCH_OK();
    int rc = device_exclusive_scan<T>(st, sums, tiles);
    if (rc != GKOC_OK) return rc;
    scan_tiles<T><<<dim3(unsigned(tiles)), dim3(scan_block), 0, st>>>(n, data, sums);
    GKOC_LAUNCH_OK();
    GKOC_TRY(scratch_free(st, sums));
    return GKOC_OK;
}

// ---- overflow check of Ginkgo's prefix_sum_nonnegative contract -----------
// (reference/components/prefix_sum_kernels.cpp:15-33: OverflowError as soon as a
// partial sum of the first n-1 non-negative entries exceeds the type's maximum).
// The 128-bit total is computed BEFORE the in-place scan; a total above `max` means
// some partial sum overflowed (the entries are non-negative).
static __global__ __launch_bounds__(256) void scan_total128_kernel(
    int64_t n, const void* __restrict__ data, int elem_bytes, unsigned long long* __restrict__ hi_lo)
{
    unsigned long long lo = 0, hi = 0;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
        const unsigned long long v =
            elem_bytes == 4 ? (unsigned long long)(static_cast<const unsigned int*>(data)[i])
                            : static_cast<const unsigned long long*>(data)[i];
        lo += v;
        hi += lo < v ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long olo = __shfl_xor(lo, off, 64);
        const unsigned long long ohi = __shfl_xor(hi, off, 64);
        lo += olo;
        hi += ohi + (lo < olo ? 1 : 0);
    }
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long old = atomicAdd(&hi_lo[1], lo);
        if (old + lo < old) atomicAdd(&hi_lo[0], 1ull);
        if (hi) atomicAdd(&hi_lo[0], hi);
    }
}

// 0 = fits, 1 = overflow, <0 = error; synchronises the stream
inline int scan_overflows(hipStream_t st, const void* data, int64_t n_summed, int elem_bytes,
                          unsigned long long max_value, int* overflow)
{
    *overflow = 0;
    if (n_summed <= 0) return GKOC_OK;
    unsigned long long* acc = nullptr;
    GKOC_TRY(scratch_malloc(st, reinterpret_cast<void**>(&acc), 16));
    GKOC_HIP(hipMemsetAsync(acc, 0, 16, st));
    int64_t nb = ceildiv(n_summed, 256);
    if (nb > max_stream_blocks) nb = max_stream_blocks;
    scan_total128_kernel<<<dim3(unsigned(nb)), dim3(256), 0, st>>>(n_summed, data, elem_bytes, acc);
    GKOC_LAUNCH_OK();
    unsigned long long h[2] = {0, 0};
    GKOC_HIP(hipMemcpyAsync(h, acc, 16, hipMemcpyDeviceToHost, st));
    GKOC_HIP(hipStreamSynchronize(st));
    GKOC_TRY(scratch_free(st, acc));
    *overflow = (h[0] != 0 || h[1] > max_value) ? 1 : 0;
    return GKOC_OK;
}

// the same with caller-provided scratch (no allocation on the stream):
// scratch holds scan_scratch_count(n) values
inline int64_t scan_scratch_count(int64_t n)
{
    int64_t total = 0;
    for (int64_t t = ceildiv(n > 0 ? n : 1, int64_t(scan_tile)); t > 1;
         t = ceildiv(t, int64_t(scan_tile))) {
        total += t;
    }
    return total + 1;
}

template <typename T>
int device_exclusive_scan(hipStream_t st, T* data, int64_t n, T* scratch)
{
    if (n <= 0) return GKOC_OK;
    const int64_t tiles = ceildiv(n, scan_tile);
    if (tiles == 1) {
        scan_tiles<T><<<dim3(1), dim3(scan_block), 0, st>>>(n, data, nullptr);
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    scan_tile_sums<T><<<dim3(unsigned(tiles)), dim3(scan_block), 0, st>>>(n, data, scratch);
    GKOC_LAUNCH_OK();
    int rc = device_exclusive_scan<T>(st, scratch, tiles, scratch + tiles);
    if (rc != GKOC_OK) return rc;
    scan_tiles<T><<<dim3(unsigned(tiles)), dim3(scan_block), 0, st>>>(n, data, scratch);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

#endif

}  // namespace gkoc
