// Shared host/device helpers for libgko_cdna4 (gfx950 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "complex_type.hpp"
#include "gko_cdna4.h"

namespace gkoc {

constexpr int wave_size = 64;
// grid cap for HBM-bound grid-stride kernels: 256 CUs x 8 resident blocks
// of 256 threads (cdna_hip_programming.md, Guideline 11)
constexpr int max_stream_blocks = 2048;

void set_last_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define GKOC_HIP(call)                                                  \
    do {                                                                \
        hipError_t gkoc_e_ = (call);                                    \
        if (gkoc_e_ != hipSuccess) {                                    \
            return ::gkoc::hip_fail(gkoc_e_, #call, __FILE__, __LINE__); \
        }                                                               \
    } while (0)

#define GKOC_LAUNCH_OK() GKOC_HIP(hipGetLastError())

#define GKOC_REQUIRE(cond, code, msg)                                   \
    do {                                                                \
        if (!(cond)) {                                                  \
            ::gkoc::set_last_error("%s:%d: %s", __FILE__, __LINE__, msg); \
            return (code);                                              \
        }                                                               \
    } while (0)

inline hipStream_t as_stream(gkoc_stream_t s)
{
    return reinterpret_cast<hipStream_t>(s);
}

inline int64_t ceildiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct device_props {
    int num_cu;
    int num_xcd;
};
// cached per-device query (hipGetDeviceProperties is slow)
const device_props& current_device_props();

// gkoc_malloc / gkoc_free go through the arena (arena.hip)
int arena_malloc(void** ptr, size_t bytes, int role);
int arena_free(void* ptr);
bool arena_owns(const void* ptr);
// Stream-ordered scratch for the library's own temporaries (flags, scan partials,
// find_blocks work arrays): taken from the arena, handed back once `st` has passed the
// point of scratch_free.  NOT hipMallocAsync / hipFreeAsync: their pool unmaps and
// re-maps virtual addresses, and on this system a kernel can still reach the OLD
// physical memory through a re-mapped address (tools/vmm_tlb.hip) - observed as
// sporadically wrong flags and scan offsets in Ginkgo's own test-suite.
int scratch_malloc(hipStream_t st, void** ptr, size_t bytes);
int scratch_free(hipStream_t st, void* ptr);

#define GKOC_TRY(call)                              \
    do {                                            \
        int gkoc_rc_ = (call);                      \
        if (gkoc_rc_ != GKOC_OK) return gkoc_rc_;   \
    } while (0)

// A zero-initialised device word per (device, stream) for kernels whose last block finishes a
// reduction (the word is 0 again when such a kernel ends; launches of one stream do not overlap).
int stream_ticket(hipStream_t st, unsigned** word);

// csr::spmv's memory of which segments of a matrix hold very long rows (csr_spmv.hip): forgotten when the
// row-pointer array is freed
void csr_long_rows_forget(const void* ptr);
// csr::spmv / advanced_spmv on complex values through the row-segment kernel of the real types
// (csr_spmv.hip; complex_blas.hip's gkoc_ccsr_spmv_* calls it).  GKOC_E_NOT_SUPPORTED: the arrays are not
// aligned for its loads - the caller keeps its thread-per-row kernel for that.
template <typename T, typename I>
int csr_spmv_complex(gkoc_stream_t s, int64_t n_rows, int64_t nrhs, const I* row_ptrs, const I* col_idxs,
                     const T* vals, const T* alpha, const T* b, int64_t ldb, const T* beta, T* c, int64_t ldc);
// What a wave (block) that reads data behind a gate word pays before its first read (csr_spmv_pipe.hpp GATE,
// fused.hpp step_gate_enter): 0 = the cheap gate (an agent-scope acquire only if it had to wait), 1 = every
// such wave an agent-scope acquire, 2 = every one a SYSTEM-scope acquire.  A communicator that has a peer on
// ANOTHER device sets 2 when it comes up (comm.hip) - the cheap gate's argument has only ever been soaked with
// writers on the same device - and the caller may lower it again once its own self-check on that communicator
// has passed (gkoc_gate_fence_policy).
int gate_fence_policy();
void gate_fence_policy_set(int policy);
// tuning switches (runtime.hip; keys = GKOC_TUNE_* of gko_cdna4.h)
constexpr int tune_num_keys = 18;
int64_t tune_value(int key);

#ifdef __HIPCC__

// ---- complex pairs (gkoc_c128 / gkoc_c64): what the data-movement kernels need ----------
template <typename T>
__host__ __device__ __forceinline__ T zero_of()
{
    return T(0);
}
template <>
__host__ __device__ __forceinline__ gkoc_c128 zero_of<gkoc_c128>()
{
    return gkoc_c128{0.0, 0.0};
}
template <>
__host__ __device__ __forceinline__ gkoc_c64 zero_of<gkoc_c64>()
{
    return gkoc_c64{0.0f, 0.0f};
}
// (==, !=, +=, ... of the two complex types: complex_type.hpp)

// ---- wave / block reductions (64-lane) ---------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off, 64);
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        T o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// deterministic block sum for blockDim.x == BLOCK (multiple of 64);
// result valid in thread 0 (and wave 0)
template <int BLOCK, typename T>
__device__ __forceinline__ T block_sum(T v, T* lds /* BLOCK/64 entries */)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    T r = T(0);
    if (wid == 0) {
        r = lane < BLOCK / 64 ? lds[lane] : T(0);
        r = wave_sum(r);
    }
    return r;
}

// Workgroup b runs on XCD b % 8 (observed dispatch rule; a wrong guess costs speed, never
// correctness - the map is a bijection on [0, nblocks)).  Hand every XCD chunks of `chunk`
// consecutive logical blocks in turn instead of every 8th block: with a chunk of 1/8 of a matrix'
// far band offset (the stencil's plane), the b rows an XCD touches through the far bands are the
// rows of its OWN chunks of the neighbouring planes - its L2 then has to hold 3 chunks of b instead
// of two whole planes.  chunk <= 0: identity.
__device__ __forceinline__ int64_t xcd_chunked_block(int64_t b, int64_t nblocks, int64_t chunk)
{
    if (chunk <= 0) return b;
    const int64_t period = 8 * chunk;
    if (b >= (nblocks / period) * period) return b;
    const int64_t xcd = b & 7, slot = b >> 3;
    const int64_t round = slot / chunk;
    return (round * 8 + xcd) * chunk + (slot - round * chunk);
}

// make this wave's LDS writes visible to its own other lanes (single-wave
// producer/consumer through LDS; no cross-wave barrier needed)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stopping_status helpers (include/ginkgo/core/stop/stopping_status.hpp)
__device__ __forceinline__ bool status_has_stopped(uint8_t s)
{
    return (s & 0x3f) != 0;
}

#endif  // __HIPCC__

}  // namespace gkoc
