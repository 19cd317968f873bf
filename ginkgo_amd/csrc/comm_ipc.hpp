// The communicator's second transport: mailboxes in peer-mapped device memory (hipIpc), written and
// polled by kernels of this library - no RCCL, no host in the loop.
//
// Why: a CG iteration at 8 GPUs is ~120 us of device work with two 8-byte all-reduces and one
// 1-2 plane halo exchange in it.  A ring / tree collective made for megabytes costs
// tens of microseconds for 8 bytes; over point-to-point xGMI every rank can simply STORE its few
// values into every peer's memory and sum what it finds in its own: one hop, one kernel.
// It stands where the reference puts a collective_communicator behind the distributed classes
// (include/ginkgo/core/distributed/collective_communicator.hpp:31-71; the reductions of
// distributed::Vector, core/distributed/vector.cpp:473-592; the exchange of
// distributed::Matrix::apply, core/distributed/matrix.cpp:450-509).
//
// One WINDOW per rank, one plain device allocation exported with hipIpcGetMemHandle and mapped by
// every peer (several processes on ONE GPU map each other's windows the same way - that is how this
// transport is tested without an 8-GPU node):
//
//   LL slots   [2 parities][MAX_RANKS sources][LL_WORDS] x 8 bytes
//       all-reduce: every rank stores its values into slot [parity][its rank] of EVERY window as
//       8-byte words {4 bytes of data, 4 bytes of epoch} (one store = data and flag, atomic), polls
//       its own window until all words of all ranks carry the epoch, and sums them in RANK ORDER:
//       every rank gets the same bits, run after run (RCCL's order depends on its algorithm choice).
//       Two parities: a rank can be at most one all-reduce ahead of another (finishing epoch e needs
//       everybody's contribution to e, which a rank sends only after it has finished e - 1).
//   flags      [MAX_RANKS] x 128 bytes   flag[src] = number of the last complete message from src
//   acks       [MAX_RANKS] x 128 bytes   ack[dst]  = number of my last message dst has copied out
//   data       [2 parities][n_ranks sources][slot_bytes]
//       exchange: the sender copies its segment into ITS slot of the receiver's window (message s
//       into parity s % 2, after the receiver has acknowledged s - 2), the last of its workgroups
//       stores the flag; the receiver's workgroups poll the flag, copy the slot into the user's
//       receive buffer and the last one acknowledges.  Messages are numbered per PAIR, so a rank
//       talks to its neighbours only.
//
// Everything that waits has a patience (GKOC_IPC_PATIENCE_MS, two minutes): a wait that runs out sets a
// bit in the communicator's status word (gkoc_comm_status) and lets the kernel end - a wrong number
// that is reported, not a hung device; every later wait of that communicator gives up within a
// millisecond (out_of_patience), so a solve that has lost its transport ends in seconds.
#ifndef GKOC_COMM_IPC_HPP_
#define GKOC_COMM_IPC_HPP_

#include <cstdint>

#include "common.hpp"

namespace gkoc {
namespace ipc {

constexpr int MAX_RANKS = 16;
constexpr int LL_WORDS = 64;                 // 4-byte pieces per all-reduce launch: 32 doubles
constexpr size_t LL_BYTES = size_t(2) * MAX_RANKS * LL_WORDS * 8;
constexpr size_t FLAG_STRIDE = 128;          // one cache line per flag
constexpr size_t FLAGS_OFF = LL_BYTES;
constexpr size_t ACKS_OFF = FLAGS_OFF + MAX_RANKS * FLAG_STRIDE;
constexpr size_t DATA_OFF = ACKS_OFF + MAX_RANKS * FLAG_STRIDE;
constexpr int64_t MIN_CHUNK = 32 * 1024;     // bytes one workgroup copies at least
constexpr int MAX_WG_PER_MSG = 16;           // a link is saturated by a few CUs; the rest belong to the SpMV
constexpr int COPY_THREADS = 256;

constexpr uint32_t ST_ALLREDUCE_TIMEOUT = 1u;
constexpr uint32_t ST_FLAG_TIMEOUT = 2u;
constexpr uint32_t ST_ACK_TIMEOUT = 4u;

struct peers_t {
    char* win[MAX_RANKS];
};

struct ar_args {
    peers_t peers;
    void* buf;
    uint32_t* status;       // host-pinned, device-visible
    long long patience;     // ticks of the 100 MHz wall clock
    int me, n_ranks, n_words, value_size;
    uint32_t epoch, parity;
};

// A transport on which one wait has run out is DEAD (its numbering is off by one from then on): every
// later wait would otherwise sit out the whole patience again - minutes per operation of a solve that is
// already lost.  A waiter therefore looks at the status word once per millisecond of waiting (a load from
// pinned host memory: never on the fast path) and gives up at once when a bit is set.
constexpr long long DEAD_CHECK_TICKS = 100000;      // 1 ms of the 100 MHz clock

__device__ __forceinline__ bool out_of_patience(long long waited, long long patience, long long& next_look,
                                                const uint32_t* status)
{
    if (waited > patience) return true;
    if (waited > next_look) {
        next_look = waited + DEAD_CHECK_TICKS;
        if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return true;
    }
    return false;
}

__device__ __forceinline__ uint64_t* ll_slot(char* win, uint32_t parity, int src)
{
    return reinterpret_cast<uint64_t*>(win) + (size_t(parity) * MAX_RANKS + src) * LL_WORDS;
}

// ONE workgroup.  buf[0 .. n_words * 4 / value_size) <- sum over ranks, in rank order.
__global__ __launch_bounds__(256) void all_reduce_kernel(ar_args a)
{
    __shared__ uint32_t piece[MAX_RANKS][LL_WORDS];
    const int tid = threadIdx.x;
    const int total = a.n_ranks * a.n_words;
    const uint32_t* in32 = static_cast<const uint32_t*>(a.buf);
    for (int idx = tid; idx < total; idx += blockDim.x) {
        const int p = idx / a.n_words, w = idx - p * a.n_words;
        const uint64_t v = (uint64_t(a.epoch) << 32) | in32[w];
        __hip_atomic_store(ll_slot(a.peers.win[p], a.parity, a.me) + w, v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const long long t0 = wall_clock64();
    long long next_look = DEAD_CHECK_TICKS;
    bool gave_up = false;
    for (int idx = tid; idx < total; idx += blockDim.x) {
        const int p = idx / a.n_words, w = idx - p * a.n_words;
        const uint64_t* src = ll_slot(a.peers.win[a.me], a.parity, p) + w;
        uint64_t v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while (uint32_t(v >> 32) != a.epoch) {
            if (out_of_patience(wall_clock64() - t0, a.patience, next_look, a.status)) {
                gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        piece[p][w] = uint32_t(v);
    }
    if (gave_up) __hip_atomic_fetch_or(a.status, ST_ALLREDUCE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();      // also: every thread has read its share of buf before anybody overwrites it
    if (a.value_size == 8) {
        const int n = a.n_words / 2;
        if (tid < n) {
            double s = 0.0;
            for (int p = 0; p < a.n_ranks; ++p) {
                const uint64_t bits = uint64_t(piece[p][2 * tid]) | (uint64_t(piece[p][2 * tid + 1]) << 32);
                s = (p == 0) ? __longlong_as_double(static_cast<long long>(bits))
                             : s + __longlong_as_double(static_cast<long long>(bits));
            }
            static_cast<double*>(a.buf)[tid] = s;
        }
    } else {
        if (tid < a.n_words) {
            float s = 0.f;
            for (int p = 0; p < a.n_ranks; ++p) {
                s = (p == 0) ? __uint_as_float(piece[p][tid]) : s + __uint_as_float(piece[p][tid]);
            }
            static_cast<float*>(a.buf)[tid] = s;
        }
    }
}

// ---- exchange ---------------------------------------------------------------------------------
struct msg_t {
    int64_t off;        // bytes into the user's send / receive buffer
    int64_t len;        // bytes
    int64_t chunk;      // bytes per workgroup (multiple of 16)
    int first_wg;       // workgroups [first_wg, first_wg + n_wg) carry this message
    int n_wg;
    int peer;
    uint32_t seq;       // number of this message between the two ranks (1, 2, ...)
};

struct xchg_args {
    peers_t peers;
    msg_t send[MAX_RANKS];
    msg_t recv[MAX_RANKS];
    const char* send_base;
    char* recv_base;
    uint32_t* done;     // [2 * MAX_RANKS] device words, zero between launches: finished workgroups per message
    uint32_t* status;
    long long patience;
    int64_t slot_bytes;
    int me, n_ranks, n_send, n_recv, send_wgs;
};

__device__ __forceinline__ uint32_t* flag_of(char* win, int src)
{
    return reinterpret_cast<uint32_t*>(win + FLAGS_OFF + size_t(src) * FLAG_STRIDE);
}
__device__ __forceinline__ uint32_t* ack_of(char* win, int dst)
{
    return reinterpret_cast<uint32_t*>(win + ACKS_OFF + size_t(dst) * FLAG_STRIDE);
}
__device__ __forceinline__ char* data_of(char* win, uint32_t parity, int src, int n_ranks, int64_t slot_bytes)
{
    return win + DATA_OFF + (size_t(parity) * n_ranks + src) * size_t(slot_bytes);
}

template <typename W>
__device__ __forceinline__ void copy_as(char* dst, const char* src, int64_t len, int tid, int nthreads)
{
    const int64_t n = len / int64_t(sizeof(W));
    const W* s = reinterpret_cast<const W*>(src);
    W* d = reinterpret_cast<W*>(dst);
    for (int64_t i = tid; i < n; i += nthreads) d[i] = s[i];
}

__device__ __forceinline__ void copy_bytes(char* dst, const char* src, int64_t len, int tid, int nthreads)
{
    const uintptr_t bits = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | uintptr_t(len);
    if ((bits & 15) == 0) {
        copy_as<uint4>(dst, src, len, tid, nthreads);
    } else if ((bits & 7) == 0) {
        copy_as<uint64_t>(dst, src, len, tid, nthreads);
    } else if ((bits & 3) == 0) {
        copy_as<uint32_t>(dst, src, len, tid, nthreads);
    } else {
        copy_as<unsigned char>(dst, src, len, tid, nthreads);
    }
}

// waits (thread 0 of the workgroup polls, the others wait at the barrier) until *word has reached
// `want`; false if the patience ran out
__device__ __forceinline__ bool wait_reached(const uint32_t* word, uint32_t want, long long patience,
                                             const uint32_t* status)
{
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        long long next_look = DEAD_CHECK_TICKS;
        int good = 1;
        while (int32_t(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
            if (out_of_patience(wall_clock64() - t0, patience, next_look, status)) {
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        ok = good;
    }
    __syncthreads();
    const bool r = ok != 0;
    __threadfence_system();      // acquire: what the writer stored before the word is read from memory
    __syncthreads();             // (ok may be rewritten by the next wait)
    return r;
}

// Workgroups [0, send_wgs): the sending side of every message (dispatched first); the rest: the
// receiving side.  The grid is a few dozen workgroups: all of them are resident at once, a waiting
// receiver never keeps a sender from starting.
__global__ __launch_bounds__(COPY_THREADS) void exchange_kernel(xchg_args a)
{
    const int wg = blockIdx.x, tid = threadIdx.x;
    __shared__ int last;
    if (wg < a.send_wgs) {
        int m = 0;
        while (m + 1 < a.n_send && wg >= a.send[m + 1].first_wg) ++m;
        const msg_t& g = a.send[m];
        const int part = wg - g.first_wg;
        const int64_t begin = int64_t(part) * g.chunk;
        const int64_t len = (begin + g.chunk <= g.len) ? g.chunk : (g.len - begin);
        // the slot of parity seq % 2 still holds message seq - 2 until the receiver has copied it out
        if (!wait_reached(ack_of(a.peers.win[a.me], g.peer), g.seq - 2u, a.patience, a.status)) {
            if (tid == 0) __hip_atomic_fetch_or(a.status, ST_ACK_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        char* dst = data_of(a.peers.win[g.peer], g.seq & 1u, a.me, a.n_ranks, a.slot_bytes) + begin;
        if (len > 0) copy_bytes(dst, a.send_base + g.off + begin, len, tid, blockDim.x);
        __threadfence_system();          // this workgroup's stores have arrived ...
        __syncthreads();
        if (tid == 0) {
            const uint32_t before = __hip_atomic_fetch_add(a.done + m, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            last = (before + 1u == uint32_t(g.n_wg));
            if (last) {
                __hip_atomic_store(a.done + m, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();  // ... and so have everybody else's: the message is complete
                __hip_atomic_store(flag_of(a.peers.win[g.peer], a.me), g.seq, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    const int rwg = wg - a.send_wgs;
    int m = 0;
    while (m + 1 < a.n_recv && rwg >= a.recv[m + 1].first_wg) ++m;
    const msg_t& g = a.recv[m];
    const int part = rwg - g.first_wg;
    const int64_t begin = int64_t(part) * g.chunk;
    const int64_t len = (begin + g.chunk <= g.len) ? g.chunk : (g.len - begin);
    if (!wait_reached(flag_of(a.peers.win[a.me], g.peer), g.seq, a.patience, a.status)) {
        if (tid == 0) __hip_atomic_fetch_or(a.status, ST_FLAG_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const char* src = data_of(a.peers.win[a.me], g.seq & 1u, g.peer, a.n_ranks, a.slot_bytes) + begin;
    if (len > 0) copy_bytes(a.recv_base + g.off + begin, src, len, tid, blockDim.x);
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const uint32_t before =
            __hip_atomic_fetch_add(a.done + MAX_RANKS + m, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1u == uint32_t(g.n_wg)) {
            __hip_atomic_store(a.done + MAX_RANKS + m, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the slot is free again: tell the sender
            __hip_atomic_store(ack_of(a.peers.win[g.peer], a.me), g.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace ipc
}  // namespace gkoc

#endif  // GKOC_COMM_IPC_HPP_
