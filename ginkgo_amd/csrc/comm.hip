// Device-buffer communicator of the distributed path on RCCL (xGMI inside a node).
// It stands where experimental::mpi::communicator stands in the reference
// (include/ginkgo/core/base/mpi.hpp:419; all_reduce :838, i_all_to_all_v :1441),
// restricted to what distributed::Vector reductions (vector.cpp:473-592) and the
// RowGatherer exchange (row_gatherer.cpp:67-190) need.
//
// Why not go through a host framework's process group: every collective of a CG
// iteration is a 16-byte all-reduce or a 1-2 plane neighbour exchange; at 8 GPUs
// the device side of an iteration is ~250 us, and a framework call costs 20-45 us
// of host time each plus cross-stream event hops on the device.  Here a collective
// is ONE enqueue on the caller's stream.
//
// RCCL is bound with dlopen/dlsym at run time: libgko_cdna4.so keeps no link
// dependency on it and loads on machines without RCCL.
#include <dlfcn.h>
#include <time.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "comm_ipc.hpp"
#include "common.hpp"

namespace {

// the part of the RCCL/NCCL ABI used here (rccl.h): opaque communicator, 128-byte
// id passed by value, enums as ints
struct nccl_unique_id {
    char internal[GKOC_COMM_ID_BYTES];
};
using nccl_comm = void*;
constexpr int nccl_success = 0;
constexpr int nccl_float32 = 7, nccl_float64 = 8, nccl_uint8 = 1;
constexpr int nccl_sum = 0;

struct rccl_api {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    // optional (diagnostics of gkoc_comm_topology_get): absent symbols leave the fields at 0
    int (*CommCount)(nccl_comm, int*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
    bool ok = false;
};

rccl_api g_rccl;
std::mutex g_rccl_mtx;

int load_rccl(const char* path)
{
    std::lock_guard<std::mutex> g(g_rccl_mtx);
    if (g_rccl.ok) return GKOC_OK;
    const char* candidates[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* c : candidates) {
        if (c == nullptr || *c == 0) continue;
        h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        gkoc::set_last_error("gkoc_comm_load_rccl: librccl not found (%s)", dlerror());
        return GKOC_E_COMM;
    }
    rccl_api a;
    a.handle = h;
    bool all = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(h, name);
        if (!p) {
            gkoc::set_last_error("gkoc_comm_load_rccl: symbol %s missing", name);
            all = false;
        }
        return p;
    };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    if (!all) return GKOC_E_COMM;
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
    a.GetVersion = reinterpret_cast<decltype(a.GetVersion)>(dlsym(h, "ncclGetVersion"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.ok = true;
    g_rccl = a;
    return GKOC_OK;
}

int rccl_fail(int e, const char* what, int line)
{
    gkoc::set_last_error("comm.hip:%d: %s failed: %s (%d)", line, what,
                         g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?", e);
    return GKOC_E_COMM;
}

#define GKOC_RCCL(call)                                        \
    do {                                                       \
        int gkoc_r_ = (call);                                  \
        if (gkoc_r_ != nccl_success) {                         \
            return rccl_fail(gkoc_r_, #call, __LINE__);        \
        }                                                      \
    } while (0)

}  // namespace

// ---- a fork without an event ---------------------------------------------------------------
// "side waits for what main has enqueued so far" as an event record + hipStreamWaitEvent is a
// barrier packet on the MAIN queue: the device idles 6-7 us before main's next kernel starts
// (profiles/r03_dist_sim_timelines.txt).  Two one-thread kernels do the same without touching
// main's queue state: main stores the fork's number into a word, side polls the word.  The kernel
// that follows on the side stream starts after the poller has ended, i.e. after main's work in
// front of the store has been released to the device (end-of-kernel release of the storing
// kernel's predecessors, start-of-kernel acquire of the follower).  The poller is one wave; it is
// enqueued AFTER the store, so a queue that serialises the two streams still makes progress.
__global__ void fork_set_kernel(uint32_t* word, uint32_t number)
{
    __hip_atomic_store(word, number, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// word[0]: the number of the last fork whose store has happened; word[16] (another cache line): set
// by a poller that gave up - about a minute of polling, so that a store that never comes (its kernel
// was refused after the poller had been enqueued) does not hang the queue for good; what runs behind
// such a poller is NOT ordered behind the main stream, gkoc_comm_fork_timed_out tells (the solvers
// of the Python mirror ask it when they return)
__global__ void fork_wait_kernel(const uint32_t* word, uint32_t number)
{
    long spins = 0;
    while (int32_t(__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - number) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (long(1) << 28)) {
            __hip_atomic_store(const_cast<uint32_t*>(word) + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}

int fork_by_kernels(hipStream_t ms, hipStream_t xs, uint32_t* word, uint32_t number)
{
    fork_set_kernel<<<dim3(1), dim3(1), 0, ms>>>(word, number);
    GKOC_LAUNCH_OK();
    fork_wait_kernel<<<dim3(1), dim3(1), 0, xs>>>(word, number);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

struct gkoc_comm_s {
    uint32_t* fork_word = nullptr;   // forks are kernels (unless GKOC_COMM_FORK=event)
    uint32_t fork_number = 0;
    bool fork_deferred = false;      // the next fork's store is the caller's kernel (gkoc_comm_fork_deferred)
    nccl_comm comm = nullptr;
    int n_ranks = 0, rank = 0;
    hipEvent_t packed = nullptr;   // main -> side: the send buffer is ready
    hipEvent_t arrived = nullptr;  // side -> main: the halo is in recv_buf
    hipEvent_t reduce_in = nullptr;   // main -> side: the values to reduce are written
    hipEvent_t reduce_out = nullptr;  // side -> main: the reduced values are there
    bool pending_side = false;
    bool pending_reduce = false;
    hipStream_t side_in_use = nullptr;  // the stream that carries the pending operations
    // ---- transport 1: mailboxes in peer-mapped memory (comm_ipc.hpp); comm == nullptr then
    int transport = 0;
    char* window = nullptr;             // this rank's window (exported)
    size_t window_bytes = 0;
    int64_t slot_bytes = 0;
    bool window_uncached = false;
    bool connected = false;
    gkoc::ipc::peers_t peers{};         // every rank's window as mapped here
    bool opened[gkoc::ipc::MAX_RANKS] = {};
    uint32_t epoch = 0, reduces = 0;    // all-reduce: flag value and parity counter
    uint32_t send_seq[gkoc::ipc::MAX_RANKS] = {}, recv_seq[gkoc::ipc::MAX_RANKS] = {};
    uint32_t* done = nullptr;           // device: finished workgroups per message of the running exchange
    uint32_t* status = nullptr;         // pinned host word the kernels OR their give-ups into
    long long patience = 0;             // ticks of the 100 MHz clock
    // ---- who is where (gkoc_comm_topology_get): PCI bus id of every rank's device
    char bus_ids[gkoc::ipc::MAX_RANKS][GKOC_COMM_BUS_ID_BYTES] = {};
    int ranks_seen = 0;                 // ncclCommCount / window cards read
    int rccl_version = 0;
    bool cross_device = false;          // at least one peer's device is not mine
    bool all_uncached = false;          // transport 1: every rank's window is uncached memory
    long mute_after = -1, all_reduces_seen = 0;     // GKOC_IPC_INJECT_MUTE (tests)
};

// what travels with a window handle from gkoc_comm_ipc_create to gkoc_comm_ipc_connect
struct ipc_card {
    hipIpcMemHandle_t handle;
    char bus_id[GKOC_COMM_BUS_ID_BYTES];    // device that holds the window
    uint8_t window_uncached;
    uint8_t version;                        // 1
    uint8_t pad[GKOC_COMM_IPC_HANDLE_BYTES - sizeof(hipIpcMemHandle_t) - GKOC_COMM_BUS_ID_BYTES - 2];
};
static_assert(sizeof(ipc_card) == GKOC_COMM_IPC_HANDLE_BYTES, "card size");

// PCI bus id of the current device; GKOC_COMM_FAKE_BUS_ID="<rank>=<id>[,<rank>=<id>...]" (tests only) makes
// `rank` report another one, so that the rules for peers on OTHER devices can be exercised on a one-GPU box
static void own_bus_id(int rank, char* out)
{
    std::memset(out, 0, GKOC_COMM_BUS_ID_BYTES);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(out, GKOC_COMM_BUS_ID_BYTES - 1, dev) != hipSuccess) {
        (void)hipGetLastError();
        snprintf(out, GKOC_COMM_BUS_ID_BYTES, "dev%d", dev);
    }
    const char* fake = std::getenv("GKOC_COMM_FAKE_BUS_ID");
    while (fake && *fake) {
        char* end = nullptr;
        const long r = std::strtol(fake, &end, 10);
        if (end == fake || *end != '=') break;
        const char* v = end + 1;
        const char* stop = std::strchr(v, ',');
        const size_t len = stop ? size_t(stop - v) : std::strlen(v);
        if (r == rank && len > 0 && len < GKOC_COMM_BUS_ID_BYTES) {
            std::memset(out, 0, GKOC_COMM_BUS_ID_BYTES);
            std::memcpy(out, v, len);
        }
        fake = stop ? stop + 1 : nullptr;
    }
}

// a communicator with a peer on another device is up: the gated kernels pay the full fence until the
// caller's self-check has passed on it (common.hpp gate_fence_policy)
static void note_topology(gkoc_comm_s* c)
{
    c->cross_device = false;
    for (int p = 0; p < c->n_ranks && p < gkoc::ipc::MAX_RANKS; ++p) {
        if (std::strncmp(c->bus_ids[p], c->bus_ids[c->rank], GKOC_COMM_BUS_ID_BYTES) != 0) c->cross_device = true;
    }
    if (c->cross_device) gkoc::gate_fence_policy_set(2);
}

using namespace gkoc;

namespace {
// main -> side: everything enqueued on ms so far happens before what is enqueued on xs from now on
int comm_fork(gkoc_comm_s* comm, hipEvent_t ev, hipStream_t ms, hipStream_t xs)
{
    if (comm->fork_word && comm->fork_deferred) {
        // the number was handed to the caller, whose next kernel on ms stores it
        comm->fork_deferred = false;
        fork_wait_kernel<<<dim3(1), dim3(1), 0, xs>>>(comm->fork_word, comm->fork_number);
        GKOC_LAUNCH_OK();
        return GKOC_OK;
    }
    if (comm->fork_word) return fork_by_kernels(ms, xs, comm->fork_word, ++comm->fork_number);
    GKOC_HIP(hipEventRecord(ev, ms));
    GKOC_HIP(hipStreamWaitEvent(xs, ev, 0));
    return GKOC_OK;
}
}  // namespace

// ---- transport 1 on the host side: numbering, chunking, launches --------------------------------
namespace {
namespace ipcx = gkoc::ipc;

int ipc_all_reduce(gkoc_comm_s* c, hipStream_t st, void* buf, int64_t n, size_t value_size)
{
    GKOC_REQUIRE(c->connected, GKOC_E_INVALID, "communicator not connected (gkoc_comm_ipc_connect)");
    const int64_t per_launch = int64_t(ipcx::LL_WORDS) * 4 / int64_t(value_size);
    // tests: GKOC_IPC_INJECT_MUTE="<rank>:<k>" - from its k-th all-reduce on, `rank` contributes nothing (its
    // peers' waits run out of patience): the fault "a peer went away in the middle of a solve"
    if (c->mute_after < 0) {
        c->mute_after = 0;
        const char* mu = std::getenv("GKOC_IPC_INJECT_MUTE");
        int r = -1;
        long k = 0;
        if (mu && std::sscanf(mu, "%d:%ld", &r, &k) == 2 && r == c->rank && k > 0) c->mute_after = k;
    }
    if (c->mute_after > 0 && ++c->all_reduces_seen >= c->mute_after) return GKOC_OK;
    for (int64_t off = 0; off < n; off += per_launch) {
        const int64_t cnt = (n - off < per_launch) ? (n - off) : per_launch;
        ipcx::ar_args a;
        a.peers = c->peers;
        a.buf = static_cast<char*>(buf) + size_t(off) * value_size;
        a.status = c->status;
        a.patience = c->patience;
        a.me = c->rank;
        a.n_ranks = c->n_ranks;
        a.n_words = int(cnt * int64_t(value_size) / 4);
        a.value_size = int(value_size);
        if (++c->epoch == 0) ++c->epoch;        // 0 is what an untouched slot holds
        a.epoch = c->epoch;
        a.parity = (c->reduces++) & 1u;
        ipcx::all_reduce_kernel<<<dim3(1), dim3(256), 0, st>>>(a);
        GKOC_LAUNCH_OK();
    }
    return GKOC_OK;
}

// one message -> its workgroups; returns the number of workgroups
int ipc_plan(ipcx::msg_t& m, int peer, int64_t off, int64_t len, int first_wg, uint32_t seq)
{
    int n_wg = int((len + ipcx::MIN_CHUNK - 1) / ipcx::MIN_CHUNK);
    if (n_wg < 1) n_wg = 1;
    if (n_wg > ipcx::MAX_WG_PER_MSG) n_wg = ipcx::MAX_WG_PER_MSG;
    int64_t chunk = (len + n_wg - 1) / n_wg;
    chunk = (chunk + 15) / 16 * 16;
    n_wg = int((len + chunk - 1) / chunk);
    m.off = off;
    m.len = len;
    m.chunk = chunk;
    m.first_wg = first_wg;
    m.n_wg = n_wg;
    m.peer = peer;
    m.seq = seq;
    return n_wg;
}

// bytes and byte offsets per peer on both sides; everything on `st`
int ipc_exchange(gkoc_comm_s* c, hipStream_t st, const void* send_buf, const int64_t* send_bytes,
                 const int64_t* send_off, void* recv_buf, const int64_t* recv_bytes, const int64_t* recv_off)
{
    GKOC_REQUIRE(c->connected, GKOC_E_INVALID, "communicator not connected (gkoc_comm_ipc_connect)");
    ipcx::xchg_args a;
    a.peers = c->peers;
    a.send_base = static_cast<const char*>(send_buf);
    a.recv_base = static_cast<char*>(recv_buf);
    a.done = c->done;
    a.status = c->status;
    a.patience = c->patience;
    a.slot_bytes = c->slot_bytes;
    a.me = c->rank;
    a.n_ranks = c->n_ranks;
    a.n_send = a.n_recv = 0;
    int swg = 0, rwg = 0;
    for (int p = 0; p < c->n_ranks; ++p) {
        GKOC_REQUIRE(send_bytes[p] <= c->slot_bytes && recv_bytes[p] <= c->slot_bytes, GKOC_E_NOT_SUPPORTED,
                     "a message is larger than the window's slot per peer (GKOC_IPC_SLOT_MIB / slot_bytes of "
                     "gkoc_comm_ipc_create)");
    }
    for (int p = 0; p < c->n_ranks; ++p) {
        if (send_bytes[p] > 0) {
            swg += ipc_plan(a.send[a.n_send++], p, send_off[p], send_bytes[p], swg, ++c->send_seq[p]);
        }
        if (recv_bytes[p] > 0) {
            rwg += ipc_plan(a.recv[a.n_recv++], p, recv_off[p], recv_bytes[p], rwg, ++c->recv_seq[p]);
        }
    }
    a.send_wgs = swg;
    if (swg + rwg == 0) return GKOC_OK;
    ipcx::exchange_kernel<<<dim3(unsigned(swg + rwg)), dim3(ipcx::COPY_THREADS), 0, st>>>(a);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

// the counts / displacements of gkoc_comm_exchange_begin (values, receive side packed in rank order)
int ipc_exchange_counts(gkoc_comm_s* c, hipStream_t st, const void* send_buf, const int64_t* send_counts,
                        const int64_t* send_displs, void* recv_buf, const int64_t* recv_counts, size_t value_size)
{
    int64_t sb[ipcx::MAX_RANKS], so[ipcx::MAX_RANKS], rb[ipcx::MAX_RANKS], ro[ipcx::MAX_RANKS];
    int64_t spos = 0, rpos = 0;
    for (int p = 0; p < c->n_ranks; ++p) {
        sb[p] = send_counts[p] * int64_t(value_size);
        rb[p] = recv_counts[p] * int64_t(value_size);
        so[p] = send_displs ? send_displs[p] * int64_t(value_size) : spos;
        ro[p] = rpos;
        spos += sb[p];
        rpos += rb[p];
    }
    return ipc_exchange(c, st, send_buf, sb, so, recv_buf, rb, ro);
}

void ipc_release(gkoc_comm_s* c)
{
    for (int p = 0; p < c->n_ranks && p < ipcx::MAX_RANKS; ++p) {
        if (c->opened[p] && c->peers.win[p]) (void)hipIpcCloseMemHandle(c->peers.win[p]);
        c->opened[p] = false;
    }
    if (c->window) (void)hipFree(c->window);
    if (c->done) (void)hipFree(c->done);
    if (c->status) (void)hipHostFree(c->status);
    c->window = nullptr;
    c->done = nullptr;
    c->status = nullptr;
    (void)hipGetLastError();
}
}  // namespace

extern "C" {

int gkoc_stream_fork(gkoc_stream_t main_stream, gkoc_stream_t side, uint32_t* word, uint32_t number)
{
    GKOC_REQUIRE(word && side != main_stream, GKOC_E_INVALID, "bad argument");
    return fork_by_kernels(as_stream(main_stream), as_stream(side), word, number);
}

int gkoc_stream_fork_wait(gkoc_stream_t side, const uint32_t* word, uint32_t number)
{
    GKOC_REQUIRE(word, GKOC_E_INVALID, "word == NULL");
    fork_wait_kernel<<<dim3(1), dim3(1), 0, as_stream(side)>>>(word, number);
    GKOC_LAUNCH_OK();
    return GKOC_OK;
}

int gkoc_comm_fork_deferred(gkoc_comm_t comm, gkoc_stream_t main_stream, gkoc_stream_t side,
                            uint32_t** word, uint32_t* number)
{
    GKOC_REQUIRE(comm && word && number, GKOC_E_INVALID, "bad argument");
    *word = nullptr;
    *number = 0;
    if (!comm->fork_word || side == nullptr || side == main_stream) return GKOC_OK;
    // a poller enqueued BEFORE the kernel that stores must not sit in front of it in one hardware
    // queue: streams of different priority never share one
    int pm = 0, ps = 0;
    if (hipStreamGetPriority(as_stream(main_stream), &pm) != hipSuccess ||
        hipStreamGetPriority(as_stream(side), &ps) != hipSuccess || pm == ps) {
        (void)hipGetLastError();
        return GKOC_OK;
    }
    comm->fork_deferred = true;
    *word = comm->fork_word;
    *number = ++comm->fork_number;
    return GKOC_OK;
}

int gkoc_comm_fork_timed_out(gkoc_comm_t comm, int* timed_out)
{
    GKOC_REQUIRE(comm && timed_out, GKOC_E_INVALID, "bad argument");
    *timed_out = 0;
    if (!comm->fork_word) return GKOC_OK;
    uint32_t v = 0;
    GKOC_HIP(hipMemcpy(&v, comm->fork_word + 16, sizeof(v), hipMemcpyDeviceToHost));
    *timed_out = v != 0;
    return GKOC_OK;
}

int gkoc_comm_load_rccl(const char* librccl_path) { return load_rccl(librccl_path); }

int gkoc_comm_unique_id(void* id_out)
{
    GKOC_REQUIRE(id_out, GKOC_E_INVALID, "id_out == NULL");
    int rc = load_rccl(nullptr);
    if (rc != GKOC_OK) return rc;
    nccl_unique_id id;
    GKOC_RCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, id.internal, GKOC_COMM_ID_BYTES);
    return GKOC_OK;
}

int gkoc_comm_create(gkoc_comm_t* comm, int n_ranks, int rank, const void* id)
{
    GKOC_REQUIRE(comm && id, GKOC_E_INVALID, "comm or id == NULL");
    GKOC_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, GKOC_E_INVALID, "bad rank / n_ranks");
    int rc = load_rccl(nullptr);
    if (rc != GKOC_OK) return rc;
    nccl_unique_id uid;
    std::memcpy(uid.internal, id, GKOC_COMM_ID_BYTES);
    auto* c = new gkoc_comm_s;
    c->n_ranks = n_ranks;
    c->rank = rank;
    int e = g_rccl.CommInitRank(&c->comm, n_ranks, uid, rank);
    if (e != nccl_success) {
        delete c;
        return rccl_fail(e, "ncclCommInitRank", __LINE__);
    }
    if (hipEventCreateWithFlags(&c->packed, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->arrived, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->reduce_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->reduce_out, hipEventDisableTiming) != hipSuccess) {
        g_rccl.CommDestroy(c->comm);
        delete c;
        set_last_error("gkoc_comm_create: hipEventCreate failed");
        return GKOC_E_COMM;
    }
    // forks as kernels (see fork_set_kernel): GKOC_COMM_FORK=event selects the event pair
    const char* fk = std::getenv("GKOC_COMM_FORK");
    if (!(fk && std::strcmp(fk, "event") == 0)) {
        void* w = nullptr;
        if (gkoc_malloc(&w, 256) == GKOC_OK && hipMemset(w, 0, 256) == hipSuccess) {
            c->fork_word = static_cast<uint32_t*>(w);
        } else {
            (void)hipGetLastError();
            if (w) (void)gkoc_free(w);
        }
    }
    // who is where: what RCCL itself counts, its version, and every rank's device (one small all-gather -
    // collective like the init it follows; a failure here is recorded, not fatal).  RCCL admits one rank per
    // device, so with more than one rank every peer is on another device whatever the gather says.
    own_bus_id(rank, c->bus_ids[rank < ipc::MAX_RANKS ? rank : 0]);
    if (g_rccl.CommCount && g_rccl.CommCount(c->comm, &c->ranks_seen) != nccl_success) c->ranks_seen = -1;
    if (g_rccl.GetVersion && g_rccl.GetVersion(&c->rccl_version) != nccl_success) c->rccl_version = -1;
    if (n_ranks > 1 && n_ranks <= ipc::MAX_RANKS && g_rccl.AllGather) {
        void* d = nullptr;
        const size_t each = GKOC_COMM_BUS_ID_BYTES;
        if (hipMalloc(&d, each * size_t(n_ranks + 1)) == hipSuccess) {
            char* all = static_cast<char*>(d);
            char* mine = all + each * size_t(n_ranks);
            bool ok = hipMemcpy(mine, c->bus_ids[rank], each, hipMemcpyHostToDevice) == hipSuccess &&
                      g_rccl.AllGather(mine, all, each, nccl_uint8, c->comm, nullptr) == nccl_success &&
                      hipStreamSynchronize(nullptr) == hipSuccess &&
                      hipMemcpy(c->bus_ids, all, each * size_t(n_ranks), hipMemcpyDeviceToHost) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                std::memset(c->bus_ids, 0, sizeof(c->bus_ids));
                own_bus_id(rank, c->bus_ids[rank]);
            }
            (void)hipFree(d);
        }
        (void)hipGetLastError();
    }
    note_topology(c);
    if (n_ranks > 1) {
        c->cross_device = true;
        gate_fence_policy_set(2);
    }
    *comm = c;
    return GKOC_OK;
}

// ---- transport 1: creation in two steps (the handles travel through whatever the host program has)
int gkoc_comm_ipc_create(gkoc_comm_t* comm, int n_ranks, int rank, int64_t slot_bytes, void* handle_out)
{
    GKOC_REQUIRE(comm && handle_out, GKOC_E_INVALID, "comm or handle_out == NULL");
    GKOC_REQUIRE(n_ranks >= 1 && n_ranks <= ipcx::MAX_RANKS && rank >= 0 && rank < n_ranks, GKOC_E_INVALID,
                 "bad rank / n_ranks (at most 16 ranks)");
    static_assert(sizeof(hipIpcMemHandle_t) + GKOC_COMM_BUS_ID_BYTES + 2 <= GKOC_COMM_IPC_HANDLE_BYTES, "handle size");
    if (slot_bytes <= 0) {
        const char* e = std::getenv("GKOC_IPC_SLOT_MIB");
        const long mib = e ? std::atol(e) : 0;
        slot_bytes = int64_t(mib > 0 ? mib : 8) << 20;
    }
    slot_bytes = (slot_bytes + 255) / 256 * 256;
    auto* c = new gkoc_comm_s;
    c->transport = 1;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->slot_bytes = slot_bytes;
    c->window_bytes = ipcx::DATA_OFF + size_t(2) * size_t(n_ranks) * size_t(slot_bytes);
    const char* pe = std::getenv("GKOC_IPC_PATIENCE_MS");
    const long pms = pe ? std::atol(pe) : 0;
    // (two minutes: ranks of a real job reach their first collectives seconds apart - allocator surveys,
    // matrix set-up - and a peer that is merely late must not be taken for one that is gone)
    c->patience = (long long)(pms > 0 ? pms : 120000) * 100000ll;     // 100 MHz clock
    // Uncached device memory (what RCCL takes for its own flags and buffers): a peer's stores are seen by
    // this device's polling loads without relying on L2 behaviour; plain hipMalloc if that cannot be had
    // or exported (GKOC_IPC_WINDOW=plain asks for it).
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof(h));
    const char* wk = std::getenv("GKOC_IPC_WINDOW");
    bool have = false;
    if (!(wk && std::strcmp(wk, "plain") == 0)) {
        void* w = nullptr;
        if (hipExtMallocWithFlags(&w, c->window_bytes, hipDeviceMallocUncached) == hipSuccess) {
            // (only the words that are polled start at zero; the data slots are written before they are read)
            if (hipMemset(w, 0, ipcx::DATA_OFF) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
                hipIpcGetMemHandle(&h, w) == hipSuccess) {
                c->window = static_cast<char*>(w);
                c->window_uncached = true;
                have = true;
            } else {
                (void)hipFree(w);
            }
        }
        (void)hipGetLastError();
    }
    if (!have) {
        void* w = nullptr;
        if (hipMalloc(&w, c->window_bytes) != hipSuccess || hipMemset(w, 0, ipcx::DATA_OFF) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess || hipIpcGetMemHandle(&h, w) != hipSuccess) {
            set_last_error("gkoc_comm_ipc_create: cannot allocate / export a window of %zu bytes: %s", c->window_bytes,
                           hipGetErrorString(hipGetLastError()));
            if (w) (void)hipFree(w);
            delete c;
            return GKOC_E_COMM;
        }
        c->window = static_cast<char*>(w);
    }
    void* d = nullptr;
    void* st = nullptr;
    if (hipMalloc(&d, 2 * ipcx::MAX_RANKS * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(d, 0, 2 * ipcx::MAX_RANKS * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc(&st, 64, hipHostMallocMapped) != hipSuccess ||
        hipEventCreateWithFlags(&c->packed, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->arrived, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->reduce_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->reduce_out, hipEventDisableTiming) != hipSuccess) {
        set_last_error("gkoc_comm_ipc_create: %s", hipGetErrorString(hipGetLastError()));
        c->done = static_cast<uint32_t*>(d);
        c->status = static_cast<uint32_t*>(st);
        ipc_release(c);
        delete c;
        return GKOC_E_COMM;
    }
    c->done = static_cast<uint32_t*>(d);
    c->status = static_cast<uint32_t*>(st);
    std::memset(c->status, 0, 64);
    (void)hipDeviceSynchronize();
    const char* fk = std::getenv("GKOC_COMM_FORK");
    if (!(fk && std::strcmp(fk, "event") == 0)) {
        void* w = nullptr;
        if (gkoc_malloc(&w, 256) == GKOC_OK && hipMemset(w, 0, 256) == hipSuccess) {
            c->fork_word = static_cast<uint32_t*>(w);
        } else {
            (void)hipGetLastError();
            if (w) (void)gkoc_free(w);
        }
    }
    ipc_card card;
    std::memset(&card, 0, sizeof(card));
    card.handle = h;
    own_bus_id(rank, card.bus_id);
    card.window_uncached = c->window_uncached ? 1 : 0;
    card.version = 1;
    std::memcpy(handle_out, &card, sizeof(card));
    *comm = c;
    return GKOC_OK;
}

int gkoc_comm_ipc_connect(gkoc_comm_t comm, const void* handles)
{
    GKOC_REQUIRE(comm && handles && comm->transport == 1, GKOC_E_INVALID, "not a mailbox communicator");
    GKOC_REQUIRE(!comm->connected, GKOC_E_INVALID, "already connected");
    const char* hb = static_cast<const char*>(handles);
    // Who is where.  EVERY rank reads the same cards, so every rank reaches the same verdict without another
    // round of talking: a window that is plain (coarse-grained) device memory must not be polled from ANOTHER
    // device - this device's caches may serve a polling load from a line they already hold for ever, and the
    // failure would be a patience timeout on the first real multi-GPU run.  Between
    // processes that share one device it works and is tested.  The caller takes RCCL.
    bool all_uncached = true;
    for (int p = 0; p < comm->n_ranks; ++p) {
        ipc_card card;
        std::memcpy(&card, hb + size_t(p) * GKOC_COMM_IPC_HANDLE_BYTES, sizeof(card));
        GKOC_REQUIRE(card.version == 1, GKOC_E_INVALID, "not a window card of this library version");
        std::memcpy(comm->bus_ids[p], card.bus_id, GKOC_COMM_BUS_ID_BYTES);
        comm->bus_ids[p][GKOC_COMM_BUS_ID_BYTES - 1] = 0;
        all_uncached = all_uncached && card.window_uncached != 0;
    }
    comm->ranks_seen = comm->n_ranks;
    comm->all_uncached = all_uncached;
    note_topology(comm);
    const char* allow = std::getenv("GKOC_IPC_ALLOW_PLAIN_ACROSS_DEVICES");
    if (comm->cross_device && !all_uncached && !(allow && std::atoi(allow) != 0)) {
        set_last_error("gkoc_comm_ipc_connect: the ranks sit on different devices and at least one rank's window is "
                       "plain device memory (uncached memory could not be had or exported there): mailboxes in "
                       "coarse-grained memory are not polled across devices - refused on every rank, use RCCL");
        return GKOC_E_NOT_SUPPORTED;
    }
    for (int p = 0; p < comm->n_ranks; ++p) {
        if (p == comm->rank) {
            comm->peers.win[p] = comm->window;
            continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, hb + size_t(p) * GKOC_COMM_IPC_HANDLE_BYTES, sizeof(h));
        const char* deny = std::getenv("GKOC_IPC_INJECT_OPEN_FAIL");      // tests: "<rank>" whose opens are denied
        if (deny && *deny && std::atoi(deny) == comm->rank) {
            set_last_error("gkoc_comm_ipc_connect: hipIpcOpenMemHandle of rank %d's window failed: injected "
                           "(GKOC_IPC_INJECT_OPEN_FAIL)", p);
            return GKOC_E_COMM;
        }
        void* w = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&w, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            set_last_error("gkoc_comm_ipc_connect: hipIpcOpenMemHandle of rank %d's window failed: %s "
                           "(HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment? peer access between the devices?)",
                           p, hipGetErrorString(e));
            (void)hipGetLastError();
            return GKOC_E_COMM;
        }
        comm->peers.win[p] = static_cast<char*>(w);
        comm->opened[p] = true;
    }
    comm->connected = true;
    return GKOC_OK;
}

int gkoc_comm_status(gkoc_comm_t comm, uint32_t* status)
{
    GKOC_REQUIRE(comm && status, GKOC_E_INVALID, "bad argument");
    *status = comm->status ? __atomic_load_n(comm->status, __ATOMIC_RELAXED) : 0u;
    return GKOC_OK;
}

int gkoc_comm_set_patience_ms(gkoc_comm_t comm, int64_t ms)
{
    GKOC_REQUIRE(comm, GKOC_E_INVALID, "comm == NULL");
    if (ms <= 0) {
        const char* pe = std::getenv("GKOC_IPC_PATIENCE_MS");
        const long pms = pe ? std::atol(pe) : 0;
        ms = pms > 0 ? pms : 120000;
    }
    comm->patience = (long long)ms * 100000ll;     // 100 MHz clock
    return GKOC_OK;
}

int gkoc_comm_transport(gkoc_comm_t comm, int* transport, int* window_uncached)
{
    GKOC_REQUIRE(comm && transport, GKOC_E_INVALID, "bad argument");
    *transport = comm->transport;
    if (window_uncached) *window_uncached = comm->window_uncached ? 1 : 0;
    return GKOC_OK;
}

int gkoc_comm_topology_get(gkoc_comm_t comm, gkoc_comm_topology* out)
{
    GKOC_REQUIRE(comm && out, GKOC_E_INVALID, "bad argument");
    std::memset(out, 0, sizeof(*out));
    out->transport = comm->transport;
    out->n_ranks = comm->n_ranks;
    out->rank = comm->rank;
    out->ranks_seen = comm->ranks_seen;
    out->rccl_version = comm->rccl_version;
    out->cross_device = comm->cross_device ? 1 : 0;
    out->window_uncached = comm->transport == 1 ? (comm->connected ? comm->all_uncached : comm->window_uncached) : 0;
    out->gate_fence = gate_fence_policy();
    const int n = comm->n_ranks < ipc::MAX_RANKS ? comm->n_ranks : ipc::MAX_RANKS;
    for (int p = 0; p < n; ++p) std::memcpy(out->bus_id[p], comm->bus_ids[p], GKOC_COMM_BUS_ID_BYTES);
    return GKOC_OK;
}

int gkoc_comm_destroy(gkoc_comm_t comm)
{
    if (!comm) return GKOC_OK;
    if (comm->transport == 1) {
        (void)hipDeviceSynchronize();
        // a peer acknowledges my last message AFTER my kernel has ended: its store must find the window
        // alive - wait (a moment) until every message I sent has been acknowledged
        if (comm->connected && comm->window) {
            for (int p = 0; p < comm->n_ranks; ++p) {
                if (comm->send_seq[p] == 0) continue;
                for (int tries = 0; tries < 2000; ++tries) {
                    uint32_t a = 0;
                    if (hipMemcpy(&a, comm->window + ipcx::ACKS_OFF + size_t(p) * ipcx::FLAG_STRIDE, sizeof(a),
                                  hipMemcpyDeviceToHost) != hipSuccess) {
                        break;
                    }
                    if (int32_t(a - comm->send_seq[p]) >= 0) break;
                    struct timespec ts = {0, 1000000};
                    nanosleep(&ts, nullptr);
                }
            }
        }
        if (comm->fork_word) (void)gkoc_free(comm->fork_word);
        if (comm->packed) (void)hipEventDestroy(comm->packed);
        if (comm->arrived) (void)hipEventDestroy(comm->arrived);
        if (comm->reduce_in) (void)hipEventDestroy(comm->reduce_in);
        if (comm->reduce_out) (void)hipEventDestroy(comm->reduce_out);
        ipc_release(comm);
        delete comm;
        (void)hipGetLastError();
        return GKOC_OK;
    }
    if (comm->fork_word) {
        (void)hipDeviceSynchronize();
        (void)gkoc_free(comm->fork_word);
    }
    if (comm->packed) (void)hipEventDestroy(comm->packed);
    if (comm->arrived) (void)hipEventDestroy(comm->arrived);
    if (comm->reduce_in) (void)hipEventDestroy(comm->reduce_in);
    if (comm->reduce_out) (void)hipEventDestroy(comm->reduce_out);
    int e = comm->comm ? g_rccl.CommDestroy(comm->comm) : nccl_success;
    delete comm;
    if (e != nccl_success) return rccl_fail(e, "ncclCommDestroy", __LINE__);
    return GKOC_OK;
}

int gkoc_comm_size(gkoc_comm_t comm, int* n_ranks, int* rank)
{
    GKOC_REQUIRE(comm, GKOC_E_INVALID, "comm == NULL");
    if (n_ranks) *n_ranks = comm->n_ranks;
    if (rank) *rank = comm->rank;
    return GKOC_OK;
}

int gkoc_comm_all_reduce_sum(gkoc_comm_t comm, gkoc_stream_t s, void* buf, int64_t n,
                             size_t value_size)
{
    GKOC_REQUIRE(comm && buf && n >= 0, GKOC_E_INVALID, "bad argument");
    GKOC_REQUIRE(value_size == 8 || value_size == 4, GKOC_E_NOT_SUPPORTED, "value_size must be 4 or 8");
    if (n == 0) return GKOC_OK;
    if (comm->transport == 1) {
        GKOC_REQUIRE(!(comm->pending_side || comm->pending_reduce) || as_stream(s) == comm->side_in_use,
                     GKOC_E_INVALID, "gkoc_comm_all_reduce_sum on another stream while an overlapped "
                                     "exchange / all-reduce is pending (end it first)");
        return ipc_all_reduce(comm, as_stream(s), buf, n, value_size);
    }
    // One communicator, one order of operations on every rank: while an overlapped operation is
    // pending on the side stream, a collective on ANOTHER stream would reach RCCL in an order
    // that depends on timing (a hang, not a wrong number) - refuse it.
    GKOC_REQUIRE(!(comm->pending_side || comm->pending_reduce) || as_stream(s) == comm->side_in_use,
                 GKOC_E_INVALID,
                 "gkoc_comm_all_reduce_sum on another stream while an overlapped exchange / "
                 "all-reduce is pending (end it first)");
    GKOC_RCCL(g_rccl.AllReduce(buf, buf, static_cast<size_t>(n),
                               value_size == 8 ? nccl_float64 : nccl_float32, nccl_sum, comm->comm,
                               as_stream(s)));
    return GKOC_OK;
}

int gkoc_comm_all_reduce_begin(gkoc_comm_t comm, gkoc_stream_t main_stream, gkoc_stream_t side,
                               void* buf, int64_t n, size_t value_size)
{
    GKOC_REQUIRE(comm && buf && n >= 0, GKOC_E_INVALID, "bad argument");
    GKOC_REQUIRE(value_size == 8 || value_size == 4, GKOC_E_NOT_SUPPORTED, "value_size must be 4 or 8");
    GKOC_REQUIRE(!comm->pending_reduce, GKOC_E_INVALID,
                 "gkoc_comm_all_reduce_begin: previous all-reduce not ended");
    if (n == 0) return GKOC_OK;
    hipStream_t ms = as_stream(main_stream);
    const bool overlapped = side != nullptr && side != main_stream;
    hipStream_t xs = overlapped ? as_stream(side) : ms;
    GKOC_REQUIRE(!comm->pending_side || xs == comm->side_in_use, GKOC_E_INVALID,
                 "gkoc_comm_all_reduce_begin: an exchange is pending on another stream");
    if (overlapped) {
        GKOC_TRY(comm_fork(comm, comm->reduce_in, ms, xs));
    } else {
        comm->fork_deferred = false;     // a fork handed out for an overlapped begin is void now
    }
    if (comm->transport == 1) {
        GKOC_TRY(ipc_all_reduce(comm, xs, buf, n, value_size));
    } else {
        GKOC_RCCL(g_rccl.AllReduce(buf, buf, static_cast<size_t>(n),
                                   value_size == 8 ? nccl_float64 : nccl_float32, nccl_sum, comm->comm,
                                   xs));
    }
    if (overlapped) {
        GKOC_HIP(hipEventRecord(comm->reduce_out, xs));
        comm->pending_reduce = true;
        comm->side_in_use = xs;
    }
    return GKOC_OK;
}

int gkoc_comm_all_reduce_end(gkoc_comm_t comm, gkoc_stream_t main_stream)
{
    GKOC_REQUIRE(comm, GKOC_E_INVALID, "comm == NULL");
    if (comm->pending_reduce) {
        GKOC_HIP(hipStreamWaitEvent(as_stream(main_stream), comm->reduce_out, 0));
        comm->pending_reduce = false;
    }
    return GKOC_OK;
}

int gkoc_comm_exchange_begin(gkoc_comm_t comm, gkoc_stream_t main_stream, gkoc_stream_t side,
                             const void* send_buf, const int64_t* send_counts,
                             const int64_t* send_displs, void* recv_buf,
                             const int64_t* recv_counts, size_t value_size)
{
    GKOC_REQUIRE(comm && send_counts && recv_counts, GKOC_E_INVALID, "bad argument");
    GKOC_REQUIRE(value_size > 0, GKOC_E_INVALID, "value_size == 0");
    GKOC_REQUIRE(!comm->pending_side, GKOC_E_INVALID,
                 "gkoc_comm_exchange_begin: previous exchange not ended");
    hipStream_t ms = as_stream(main_stream);
    const bool overlapped = side != nullptr && side != main_stream;
    hipStream_t xs = overlapped ? as_stream(side) : ms;
    int64_t n_msgs = 0;
    for (int p = 0; p < comm->n_ranks; ++p) {
        GKOC_REQUIRE(send_counts[p] >= 0 && recv_counts[p] >= 0, GKOC_E_INVALID, "negative count");
        GKOC_REQUIRE(!send_displs || send_displs[p] >= 0, GKOC_E_INVALID, "negative displacement");
        n_msgs += (send_counts[p] > 0) + (recv_counts[p] > 0);
    }
    if (n_msgs == 0) {
        comm->fork_deferred = false;     // nothing to order: a fork handed to the caller is void
        return GKOC_OK;
    }
    GKOC_REQUIRE(send_buf && recv_buf, GKOC_E_INVALID, "NULL buffer with non-zero counts");
    GKOC_REQUIRE(!comm->pending_reduce || xs == comm->side_in_use, GKOC_E_INVALID,
                 "gkoc_comm_exchange_begin: an all-reduce is pending on another stream");
    if (overlapped) {
        GKOC_TRY(comm_fork(comm, comm->packed, ms, xs));
    } else {
        comm->fork_deferred = false;     // a fork handed out for an overlapped begin is void now
    }
    if (comm->transport == 1) {
        GKOC_TRY(ipc_exchange_counts(comm, xs, send_buf, send_counts, send_displs, recv_buf, recv_counts, value_size));
        if (overlapped) {
            GKOC_HIP(hipEventRecord(comm->arrived, xs));
            comm->pending_side = true;
            comm->side_in_use = xs;
        }
        return GKOC_OK;
    }
    const char* sp = static_cast<const char*>(send_buf);
    char* rp = static_cast<char*>(recv_buf);
    GKOC_RCCL(g_rccl.GroupStart());
    int e = nccl_success;
    for (int p = 0; p < comm->n_ranks && e == nccl_success; ++p) {
        // bytes as uint8: any value type travels unchanged
        const size_t sb = static_cast<size_t>(send_counts[p]) * value_size;
        const size_t rb = static_cast<size_t>(recv_counts[p]) * value_size;
        if (send_displs) {
            sp = static_cast<const char*>(send_buf) + static_cast<size_t>(send_displs[p]) * value_size;
        }
        if (sb) e = g_rccl.Send(sp, sb, nccl_uint8, p, comm->comm, xs);
        if (rb && e == nccl_success) e = g_rccl.Recv(rp, rb, nccl_uint8, p, comm->comm, xs);
        sp += sb;
        rp += rb;
    }
    int e2 = g_rccl.GroupEnd();
    if (e != nccl_success) return rccl_fail(e, "ncclSend/ncclRecv", __LINE__);
    if (e2 != nccl_success) return rccl_fail(e2, "ncclGroupEnd", __LINE__);
    if (overlapped) {
        GKOC_HIP(hipEventRecord(comm->arrived, xs));
        comm->pending_side = true;
        comm->side_in_use = xs;
    }
    return GKOC_OK;
}

// The overlapped all-reduce of a pipelined solver AND the halo exchange of the SpMV that hides it,
// behind ONE fork and ONE join: main records one event, the side stream waits for it, reduces,
// exchanges and records one event that gkoc_comm_exchange_end / _join wait for.  Every event
// record / cross-stream wait on the main stream is a barrier packet that leaves the device idle
// for 6-7 us before the next kernel (profiles/r03_dist_sim_timelines.txt); all_reduce_begin +
// exchange_begin + exchange_end + all_reduce_end are four of them, this pair is two.  Both the
// values to reduce and the send buffer must be final on the main stream at the call.
int gkoc_comm_all_reduce_exchange_begin(gkoc_comm_t comm, gkoc_stream_t main_stream, gkoc_stream_t side,
                                        void* reduce_buf, int64_t reduce_n, size_t reduce_value_size,
                                        const void* send_buf, const int64_t* send_counts,
                                        const int64_t* send_displs, void* recv_buf,
                                        const int64_t* recv_counts, size_t value_size)
{
    GKOC_REQUIRE(comm && reduce_buf && reduce_n > 0 && send_counts && recv_counts, GKOC_E_INVALID,
                 "bad argument");
    GKOC_REQUIRE(reduce_value_size == 8 || reduce_value_size == 4, GKOC_E_NOT_SUPPORTED,
                 "value_size must be 4 or 8");
    GKOC_REQUIRE(side != nullptr && side != main_stream, GKOC_E_INVALID, "needs a side stream");
    GKOC_REQUIRE(!comm->pending_side && !comm->pending_reduce, GKOC_E_INVALID,
                 "gkoc_comm_all_reduce_exchange_begin: a previous operation has not been ended");
    hipStream_t ms = as_stream(main_stream), xs = as_stream(side);
    int64_t n_msgs = 0;
    for (int p = 0; p < comm->n_ranks; ++p) {
        GKOC_REQUIRE(send_counts[p] >= 0 && recv_counts[p] >= 0, GKOC_E_INVALID, "negative count");
        GKOC_REQUIRE(!send_displs || send_displs[p] >= 0, GKOC_E_INVALID, "negative displacement");
        n_msgs += (send_counts[p] > 0) + (recv_counts[p] > 0);
    }
    GKOC_REQUIRE(n_msgs == 0 || (send_buf && recv_buf), GKOC_E_INVALID, "NULL buffer with non-zero counts");
    GKOC_TRY(comm_fork(comm, comm->packed, ms, xs));
    if (comm->transport == 1) {
        GKOC_TRY(ipc_all_reduce(comm, xs, reduce_buf, reduce_n, reduce_value_size));
        if (n_msgs > 0) {
            GKOC_TRY(ipc_exchange_counts(comm, xs, send_buf, send_counts, send_displs, recv_buf, recv_counts,
                                         value_size));
        }
        GKOC_HIP(hipEventRecord(comm->arrived, xs));
        comm->pending_side = true;
        comm->side_in_use = xs;
        return GKOC_OK;
    }
    GKOC_RCCL(g_rccl.AllReduce(reduce_buf, reduce_buf, static_cast<size_t>(reduce_n),
                               reduce_value_size == 8 ? nccl_float64 : nccl_float32, nccl_sum, comm->comm, xs));
    if (n_msgs > 0) {
        const char* sp = static_cast<const char*>(send_buf);
        char* rp = static_cast<char*>(recv_buf);
        GKOC_RCCL(g_rccl.GroupStart());
        int e = nccl_success;
        for (int p = 0; p < comm->n_ranks && e == nccl_success; ++p) {
            const size_t sb = static_cast<size_t>(send_counts[p]) * value_size;
            const size_t rb = static_cast<size_t>(recv_counts[p]) * value_size;
            if (send_displs) {
                sp = static_cast<const char*>(send_buf) + static_cast<size_t>(send_displs[p]) * value_size;
            }
            if (sb) e = g_rccl.Send(sp, sb, nccl_uint8, p, comm->comm, xs);
            if (rb && e == nccl_success) e = g_rccl.Recv(rp, rb, nccl_uint8, p, comm->comm, xs);
            sp += sb;
            rp += rb;
        }
        int e2 = g_rccl.GroupEnd();
        if (e != nccl_success) return rccl_fail(e, "ncclSend/ncclRecv", __LINE__);
        if (e2 != nccl_success) return rccl_fail(e2, "ncclGroupEnd", __LINE__);
    }
    GKOC_HIP(hipEventRecord(comm->arrived, xs));
    comm->pending_side = true;     // ended by gkoc_comm_exchange_end / _join: the reduction is done by then too
    comm->side_in_use = xs;
    return GKOC_OK;
}

// like gkoc_comm_exchange_end, but the main stream waits for EVERYTHING that has been enqueued on
// the exchange's stream since gkoc_comm_exchange_begin - the halo and the kernels that consume it
// there (the boundary rows of a slab partition, csr_rowlist_full_kernel)
int gkoc_comm_exchange_join(gkoc_comm_t comm, gkoc_stream_t main_stream)
{
    GKOC_REQUIRE(comm, GKOC_E_INVALID, "comm == NULL");
    if (comm->pending_side) {
        GKOC_HIP(hipEventRecord(comm->arrived, comm->side_in_use));
        GKOC_HIP(hipStreamWaitEvent(as_stream(main_stream), comm->arrived, 0));
        comm->pending_side = false;
    }
    return GKOC_OK;
}

// MPI_Alltoallv over RCCL: byte counts and byte offsets per peer on both sides, one grouped
// send / recv on `s`; the part a rank sends to itself is a device copy
// The consumer of the halo waits for it by itself (gkoc_csr_spmv_gated_* behind a gkoc_gate_open on
// the side stream): the exchange is over for the communicator, and the main stream does not wait for
// the side stream.  The next exchange is ordered behind whatever the main stream runs until then by
// its own fork.
int gkoc_comm_exchange_forget(gkoc_comm_t comm)
{
    GKOC_REQUIRE(comm, GKOC_E_INVALID, "comm == NULL");
    // (an all-reduce begun with gkoc_comm_all_reduce_begin keeps its own end; one begun TOGETHER with
    // the exchange - gkoc_comm_all_reduce_exchange_begin - needs gkoc_comm_exchange_join)
    comm->pending_side = false;
    return GKOC_OK;
}

int gkoc_comm_all_to_all_v_bytes(gkoc_comm_t comm, gkoc_stream_t s, const void* send_buf,
                                 const int64_t* send_bytes, const int64_t* send_offsets, void* recv_buf,
                                 const int64_t* recv_bytes, const int64_t* recv_offsets)
{
    GKOC_REQUIRE(comm && send_bytes && send_offsets && recv_bytes && recv_offsets, GKOC_E_INVALID,
                 "bad argument");
    GKOC_REQUIRE(!comm->pending_side && !comm->pending_reduce, GKOC_E_INVALID,
                 "gkoc_comm_all_to_all_v_bytes while an overlapped operation is pending");
    hipStream_t st = as_stream(s);
    const char* sp = static_cast<const char*>(send_buf);
    char* rp = static_cast<char*>(recv_buf);
    for (int p = 0; p < comm->n_ranks; ++p) {
        GKOC_REQUIRE(send_bytes[p] >= 0 && recv_bytes[p] >= 0 && send_offsets[p] >= 0 && recv_offsets[p] >= 0,
                     GKOC_E_INVALID, "negative count or offset");
    }
    const int me = comm->rank;
    if (comm->transport == 1) {
        return ipc_exchange(comm, st, send_buf, send_bytes, send_offsets, recv_buf, recv_bytes, recv_offsets);
    }
    if (send_bytes[me] > 0) {
        GKOC_REQUIRE(send_bytes[me] == recv_bytes[me], GKOC_E_INVALID, "self message sizes differ");
        GKOC_HIP(hipMemcpyAsync(rp + recv_offsets[me], sp + send_offsets[me], size_t(send_bytes[me]),
                                hipMemcpyDeviceToDevice, st));
    }
    GKOC_RCCL(g_rccl.GroupStart());
    int e = nccl_success;
    for (int p = 0; p < comm->n_ranks && e == nccl_success; ++p) {
        if (p == me) continue;
        if (send_bytes[p]) e = g_rccl.Send(sp + send_offsets[p], size_t(send_bytes[p]), nccl_uint8, p, comm->comm, st);
        if (recv_bytes[p] && e == nccl_success) {
            e = g_rccl.Recv(rp + recv_offsets[p], size_t(recv_bytes[p]), nccl_uint8, p, comm->comm, st);
        }
    }
    int e2 = g_rccl.GroupEnd();
    if (e != nccl_success) return rccl_fail(e, "ncclSend/ncclRecv", __LINE__);
    if (e2 != nccl_success) return rccl_fail(e2, "ncclGroupEnd", __LINE__);
    return GKOC_OK;
}

int gkoc_comm_exchange_end(gkoc_comm_t comm, gkoc_stream_t main_stream)
{
    GKOC_REQUIRE(comm, GKOC_E_INVALID, "comm == NULL");
    if (comm->pending_side) {
        GKOC_HIP(hipStreamWaitEvent(as_stream(main_stream), comm->arrived, 0));
        comm->pending_side = false;
    }
    return GKOC_OK;
}

}  // extern "C"
