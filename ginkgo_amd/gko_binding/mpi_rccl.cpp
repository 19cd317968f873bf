// libgkoc_mpi_rccl.so - device buffers behind Ginkgo's own MPI calls.
//
// Ginkgo's distributed classes talk to MPI directly (include/ginkgo/core/base/mpi.hpp): with
// GINKGO_HAVE_GPU_AWARE_MPI (CMake: GINKGO_FORCE_GPU_AWARE_MPI, CMakeLists.txt:185, 410-417)
// mpi::requires_host_buffer is false (core/base/mpi.cpp:65-70) and the core hands DEVICE pointers
// to MPI_Allreduce (distributed/vector.cpp:473-497 and the other reductions, solver/gmres.cpp:215),
// MPI_Ialltoallv (RowGatherer::apply_finalize, row_gatherer.cpp:118-174, through DenseCommunicator),
// MPI_Ineighbor_alltoallv (NeighborhoodCommunicator), MPI_Alltoall(v) / MPI_Allgather (assembly,
// partition helpers).  An MPI that is not GPU-aware cannot take them.  This library is the
// GPU-aware layer for such an MPI: it defines those MPI_* entry points (link it in front of
// libmpi, or LD_PRELOAD it) and forwards to PMPI_*, except that buffers in device memory are
// routed
//   * over RCCL (csrc/comm.hip: one communicator per MPI communicator, created from an id that
//     rank 0 broadcasts over MPI; all-reduce = ncclAllReduce, all-to-all-v = grouped ncclSend /
//     ncclRecv over xGMI) when every rank of the communicator drives its own GPU - no host
//     staging, no D2H / H2D copies; a non-blocking call returns at once and the exchange
//     overlaps whatever the caller enqueues next (the local SpMV of distributed::Matrix::apply),
//     MPI_Wait waits for the side stream;
//   * through pinned host staging buffers and the plain MPI call when ranks share a GPU (RCCL
//     refuses two ranks on one device; the single-GPU test box), for operations / datatypes RCCL
//     has no counterpart for, or with GKOC_MPI_MODE=staged.
// MPI itself stays the launcher, the control plane and the transport of every host buffer.
// Counters (gkoc_mpi_stats) tell tests which route each call took.
//
// Contract, as for any GPU-aware MPI: the contents of a device send buffer are final when MPI is
// called (Ginkgo synchronises its executor or the pack event before every such call), results are
// in the receive buffer when the blocking call or MPI_Wait returns.
#include <mpi.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "gko_cdna4.h"

extern "C" void gko_cdna4_launch_deferred() __attribute__((weak));

namespace {

enum class mode { undecided, rccl, staged };

struct comm_state {
    mode m = mode::undecided;
    gkoc_comm_t rccl = nullptr;    // the device-resident communicator: RCCL, or the library's mailboxes (ipc)
    int size = 0, rank = 0;
    bool ipc = false;              // rccl is a mailbox communicator (csrc/comm_ipc.hpp): ranks may share a GPU
    int64_t slot_bytes = 0;        //   ... whose messages are bounded
};

// A NONBLOCKING all-to-all-v whose route has to be agreed by the ranks first (route_of): the two-int
// agreement travels as PMPI_Iallreduce inside the request, and the exchange itself is carried out when the
// request is completed (MPI_Wait / a successful MPI_Test) - a blocking all-reduce inside MPI_Ialltoallv
// would synchronise the ranks in a call MPI defines as local, and can deadlock a legal program in which a
// rank posts the collective and then serves a point-to-point message its peer sends BEFORE posting its own
// (ADVICE round 4).  The core's pattern - post, launch the local product, wait - loses nothing: the kernel
// is already running when MPI_Wait starts the exchange.  The buffers and count arrays stay valid until
// completion (MPI-3.1, 5.12).
struct deferred_call {
    int which = 0;                 // 0 MPI_Ialltoallv, 1 MPI_Ineighbor_alltoallv
    const void* sendbuf = nullptr;
    const int *scounts = nullptr, *sdispls = nullptr;
    MPI_Datatype stype = MPI_DATATYPE_NULL;
    void* recvbuf = nullptr;
    const int *rcounts = nullptr, *rdispls = nullptr;
    MPI_Datatype rtype = MPI_DATATYPE_NULL;
    MPI_Comm comm = MPI_COMM_NULL;
    int n = 0;
    int mine[2] = {0, 0}, any[2] = {0, 0};
    long seq = 0;                  // order of issue on its communicator: carried out in that order on every rank
    int rc = MPI_SUCCESS;
    bool done = false;
    // round 6: MPI_Test starts the exchange in its NONBLOCKING form and looks at it again later,
    // instead of carrying it out to the end (which waits for the peers' kernels - MPI_Test must not)
    bool launched = false;
    MPI_Request sub = MPI_REQUEST_NULL;      // the started exchange: one of ours (rccl / staged) or MPI's
};

struct pending {
    int kind = 0;                  // 1 rccl (stream work), 2 staged (real request + copy back), 3 deferred
    std::shared_ptr<deferred_call> call;
    MPI_Request inner = MPI_REQUEST_NULL;
    // staged: device destination, host staging source
    void* dev_dst = nullptr;
    void* host_src = nullptr;
    size_t bytes = 0;
    void* host_send = nullptr;     // kept alive until completion
    std::vector<int> keep_i;       // count / displacement arrays MPI may read until completion
};

std::mutex g_mtx;
std::map<MPI_Comm, comm_state> g_comms;
std::map<MPI_Request, pending> g_pending;
gkoc_stream_t g_stream = nullptr;
long g_stats[8] = {};              // 0 rccl all-reduce, 1 staged all-reduce, 2 rccl all-to-all-v,
                                   // 3 staged all-to-all-v, 4 other staged, 5 D2H+H2D bytes, 6 host pass-through

int env_mode()
{
    const char* e = std::getenv("GKOC_MPI_MODE");
    if (!e) return 0;
    if (!std::strcmp(e, "staged")) return 2;
    if (!std::strcmp(e, "rccl")) return 1;
    if (!std::strcmp(e, "off")) return 3;
    return 0;
}

bool is_device(const void* p)
{
    if (p == nullptr || p == MPI_IN_PLACE) return false;
    int d = 0;
    if (gkoc_pointer_is_device(p, &d) != GKOC_OK) return false;
    return d != 0;
}

gkoc_stream_t stream()
{
    if (!g_stream) gkoc_stream_create(&g_stream);
    return g_stream;
}

// Wait for the layer's stream WITHOUT starving MPI: a kernel of the mailbox transport (or of RCCL) may be
// waiting for a peer whose own exchange starts only when ITS agreement - a PMPI_Iallreduce this rank takes
// part in - has completed, and nonblocking collectives of MPICH advance only inside MPI calls.  A rank that
// sat in hipStreamSynchronize would keep its peers' agreements from arriving (seen with two requests
// completed in different orders on different ranks: tests/dropin/mpi_layer_test.cpp).
void sync_progressing()
{
    for (;;) {
        int done = 0;
        if (gkoc_stream_query(stream(), &done) != GKOC_OK || done) break;
        int flag = 0;
        PMPI_Iprobe(MPI_ANY_SOURCE, MPI_ANY_TAG, MPI_COMM_WORLD, &flag, MPI_STATUS_IGNORE);   // drives the progress engine
    }
    gkoc_stream_synchronize(stream());
}

// What a GPU-aware MPI does implicitly when it is handed a device buffer: the kernels that produced
// it are complete before the buffer is read.  Ginkgo's reductions and the RowGatherer synchronize
// themselves (vector.cpp:484-657, row_gatherer.cpp:165: the wait below finds an idle device), but
// assemble_rows_from_neighbors (core/distributed/assembly.cpp:69-96) fills its send buffers on the
// executor's stream and calls MPI_Ialltoallv at once - a library that copies with hipMemcpy gets the
// ordering from the legacy null stream, the side stream of this layer does not.
void flush_binding()
{
    if (gko_cdna4_launch_deferred) gko_cdna4_launch_deferred();
    gkoc_device_synchronize();
}

// pinned host staging buffers, recycled
struct host_pool {
    std::vector<std::pair<void*, size_t>> free_list;
    void* get(size_t bytes)
    {
        for (size_t i = 0; i < free_list.size(); ++i) {
            if (free_list[i].second >= bytes) {
                void* p = free_list[i].first;
                sizes[p] = free_list[i].second;
                free_list.erase(free_list.begin() + i);
                return p;
            }
        }
        void* p = nullptr;
        const size_t cap = bytes < 4096 ? 4096 : bytes;
        if (gkoc_malloc_host(&p, cap) != GKOC_OK) return nullptr;
        sizes[p] = cap;
        return p;
    }
    void put(void* p)
    {
        if (p) free_list.emplace_back(p, sizes[p]);
    }
    std::map<void*, size_t> sizes;
} g_host;

// decide (collectively, once per communicator) how device buffers travel on `comm`
comm_state& state_of(MPI_Comm comm)
{
    comm_state& st = g_comms[comm];
    if (st.m != mode::undecided) return st;
    PMPI_Comm_size(comm, &st.size);
    PMPI_Comm_rank(comm, &st.rank);
    const int forced = env_mode();
    int want_rccl = forced != 2;
    if (want_rccl && st.size > 1 && forced != 1) {
        // one GPU per rank?  compare "host/pci-bus-id" of all ranks
        char mine[64] = {0};
        if (gkoc_device_identity(mine, sizeof(mine)) != GKOC_OK) mine[0] = 0;
        std::vector<char> all(size_t(64) * st.size);
        PMPI_Allgather(mine, 64, MPI_CHAR, all.data(), 64, MPI_CHAR, comm);
        for (int a = 0; a < st.size && want_rccl; ++a) {
            for (int b = a + 1; b < st.size; ++b) {
                if (!std::strncmp(&all[64 * a], &all[64 * b], 64)) {
                    want_rccl = 0;
                    break;
                }
            }
        }
    }
    int ok = 0;
    if (want_rccl) {
        unsigned char id[GKOC_COMM_ID_BYTES] = {0};
        int have = 1;
        if (gkoc_comm_load_rccl(nullptr) != GKOC_OK) have = 0;
        if (have && st.rank == 0 && gkoc_comm_unique_id(id) != GKOC_OK) have = 0;
        int all_have = 0;
        PMPI_Allreduce(&have, &all_have, 1, MPI_INT, MPI_MIN, comm);
        if (all_have) {
            PMPI_Bcast(id, GKOC_COMM_ID_BYTES, MPI_BYTE, 0, comm);
            ok = gkoc_comm_create(&st.rccl, st.size, st.rank, id) == GKOC_OK ? 1 : 0;
            int all_ok = 0;
            PMPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, comm);
            if (!all_ok && st.rccl) {
                gkoc_comm_destroy(st.rccl);
                st.rccl = nullptr;
            }
            ok = all_ok;
        }
    }
    // No RCCL communicator (ranks that share a GPU - RCCL refuses that -, no librccl, a failed bring-up): the
    // library's own transport, mailboxes in peer-mapped device memory (gkoc_comm_ipc_*), serves the same
    // gkoc_comm_* calls; GKOC_MPI_TRANSPORT=rccl keeps to RCCL only, =ipc asks for the mailboxes first.
    // Every step is agreed by all ranks (they all arrive here: state_of is collective).
    // (MPI knows no time-outs: a peer may reach its wait minutes after this rank - the kernels of the mailbox
    // transport get ten minutes of patience here unless the environment says otherwise)
    if (!std::getenv("GKOC_IPC_PATIENCE_MS")) setenv("GKOC_IPC_PATIENCE_MS", "600000", 0);
    const char* tr = std::getenv("GKOC_MPI_TRANSPORT");
    const bool ipc_allowed = forced != 2 && st.size > 1 && st.size <= 16 && !(tr && !std::strcmp(tr, "rccl"));
    if (!ok && ipc_allowed) {
        unsigned char mine_h[GKOC_COMM_IPC_HANDLE_BYTES] = {0};
        // Ginkgo creates communicators per matrix (dist-graph, split): a window per communicator must be cheap -
        // 1 MiB per peer and direction unless GKOC_IPC_SLOT_MIB says otherwise (a halo of 131 072 doubles per
        // neighbour; larger messages take the staged route, agreed by all ranks)
        const char* se = std::getenv("GKOC_IPC_SLOT_MIB");
        const long smib = se ? std::atol(se) : 0;
        const int64_t slot = int64_t(smib > 0 ? smib : 1) << 20;
        int have = gkoc_comm_ipc_create(&st.rccl, st.size, st.rank, slot, mine_h) == GKOC_OK ? 1 : 0;
        int all_have = 0;
        PMPI_Allreduce(&have, &all_have, 1, MPI_INT, MPI_MIN, comm);
        if (all_have) {
            std::vector<unsigned char> all(size_t(GKOC_COMM_IPC_HANDLE_BYTES) * st.size);
            PMPI_Allgather(mine_h, GKOC_COMM_IPC_HANDLE_BYTES, MPI_BYTE, all.data(), GKOC_COMM_IPC_HANDLE_BYTES,
                           MPI_BYTE, comm);
            int conn = gkoc_comm_ipc_connect(st.rccl, all.data()) == GKOC_OK ? 1 : 0;
            int all_conn = 0;
            PMPI_Allreduce(&conn, &all_conn, 1, MPI_INT, MPI_MIN, comm);
            ok = all_conn;
        }
        if (!ok && st.rccl) {
            PMPI_Barrier(comm);            // nobody unmaps while a peer is still mapping
            gkoc_comm_destroy(st.rccl);
            st.rccl = nullptr;
        }
        if (ok) {
            st.ipc = true;
            st.slot_bytes = slot;
        }
    }
    st.m = ok ? mode::rccl : mode::staged;
    if (std::getenv("GKOC_MPI_VERBOSE") && st.rank == 0) {
        std::fprintf(stderr, "[gkoc_mpi] communicator of %d ranks: device buffers go %s\n", st.size,
                     ok ? (st.ipc ? "through the library's mailboxes in peer-mapped device memory" : "over RCCL")
                        : "through host staging");
    }
    return st;
}

size_t type_bytes(MPI_Datatype t)
{
    int sz = 0;
    PMPI_Type_size(t, &sz);
    return size_t(sz);
}

// element size RCCL can sum (8 = double / complex<double> as pairs, 4 = float), 0 otherwise
size_t rccl_sum_element(MPI_Datatype t, MPI_Op op, int count, int64_t* n_elems)
{
    if (op != MPI_SUM) return 0;
    if (t == MPI_DOUBLE) { *n_elems = count; return 8; }
    if (t == MPI_FLOAT) { *n_elems = count; return 4; }
    if (t == MPI_C_DOUBLE_COMPLEX || t == MPI_CXX_DOUBLE_COMPLEX || t == MPI_DOUBLE_COMPLEX) { *n_elems = 2 * int64_t(count); return 8; }
    if (t == MPI_C_FLOAT_COMPLEX || t == MPI_CXX_FLOAT_COMPLEX || t == MPI_COMPLEX) { *n_elems = 2 * int64_t(count); return 4; }
    return 0;
}

// a completed-later request the application can hold: a generalized request
int gq_query(void*, MPI_Status* s)
{
    if (s) {
        MPI_Status_set_elements(s, MPI_BYTE, 0);
        MPI_Status_set_cancelled(s, 0);
        s->MPI_SOURCE = MPI_UNDEFINED;
        s->MPI_TAG = MPI_UNDEFINED;
    }
    return MPI_SUCCESS;
}
int gq_free(void*) { return MPI_SUCCESS; }
int gq_cancel(void*, int) { return MPI_SUCCESS; }

MPI_Request new_handle()
{
    MPI_Request r = MPI_REQUEST_NULL;
    PMPI_Grequest_start(gq_query, gq_free, gq_cancel, nullptr, &r);
    return r;
}

// bring a pending operation to its end (stream work done / inner request done + copied back)
int finish(pending& p)
{
    int rc = MPI_SUCCESS;
    if (p.kind == 1) {
        sync_progressing();
    } else if (p.kind == 2) {
        if (p.inner != MPI_REQUEST_NULL) rc = PMPI_Wait(&p.inner, MPI_STATUS_IGNORE);
        if (p.dev_dst && p.bytes) {
            gkoc_memcpy_h2d(p.dev_dst, p.host_src, p.bytes, stream());
            gkoc_stream_synchronize(stream());
            g_stats[5] += long(p.bytes);
        }
        g_host.put(p.host_src);
        g_host.put(p.host_send);
    }
    return rc;
}

struct span_bytes {
    size_t lo = 0, hi = 0;     // byte range touched inside a buffer
};

span_bytes extent(const int* counts, const int* displs, int n, size_t tb)
{
    span_bytes s;
    bool first = true;
    for (int i = 0; i < n; ++i) {
        if (counts[i] <= 0) continue;
        const size_t lo = size_t(displs[i]) * tb, hi = lo + size_t(counts[i]) * tb;
        if (first || lo < s.lo) s.lo = lo;
        if (first || hi > s.hi) s.hi = hi;
        first = false;
    }
    return s;
}

// How an exchange travels is a property of the COMMUNICATOR AND THE CALL, never of one rank's
// pointers: Ginkgo passes nullptr for empty arrays, so an end rank of a
// one-sided pattern, a rank without halo or an empty rank sees no device pointer at all - if it
// took the plain MPI call while its peers issued ncclSend / ncclRecv, the job would hang.  Every
// rank of the communicator enters here on every all-to-all-v (the comm_state is created
// collectively on the first one); where device buffers can go over RCCL the ranks agree with one
// two-int all-reduce - does anyone hold a non-empty HOST buffer, does anyone hold a DEVICE buffer:
//   0 plain MPI (nobody holds a device buffer, or the layer is off)
//   1 RCCL      (device buffers everywhere they are not empty; a rank with nothing to send or
//                receive takes part with zero counts and issues no RCCL call)
//   2 staged    (ranks share a GPU, or host and device buffers are mixed: the MPI call with
//                host copies of the device ranges - compatible with peers that pass host buffers)
enum { route_mpi = 0, route_rccl = 1, route_staged = 2 };

bool all_zero(const int* counts, int n)
{
    for (int i = 0; i < n; ++i) {
        if (counts[i] != 0) return false;
    }
    return true;
}

int route_from(const int any[2])
{
    if (!any[1]) return route_mpi;
    return any[0] ? route_staged : route_rccl;
}

// the route where it is a local matter, else -1 with mine[] = what this rank contributes to the agreement
int route_local(MPI_Comm comm, const void* sendbuf, bool send_empty, const void* recvbuf, bool recv_empty, int mine[2],
                size_t max_message_bytes = 0)
{
    if (env_mode() == 3) return route_mpi;
    const bool sdev = is_device(sendbuf), rdev = is_device(recvbuf);
    mode m;
    int size;
    int64_t slot = 0;
    {
        std::lock_guard<std::mutex> g(g_mtx);
        const comm_state& st = state_of(comm);       // collective on first use: every rank is here
        m = st.m;
        size = st.size;
        slot = st.ipc ? st.slot_bytes : 0;
    }
    if (m != mode::rccl || size == 1) {
        if (m == mode::rccl && sdev && rdev) return route_rccl;
        return (sdev || rdev) ? route_staged : route_mpi;
    }
    // (a message that does not fit the mailbox transport's slot sends everybody through the staged route,
    // like a host buffer does)
    const bool oversize = slot > 0 && int64_t(max_message_bytes) > slot;
    mine[0] = ((!sdev && !send_empty) || (!rdev && !recv_empty) || oversize) ? 1 : 0;
    mine[1] = (sdev || rdev) ? 1 : 0;
    return -1;
}

// the exchange as grouped ncclSend / ncclRecv on the layer's stream (g_mtx held)
int rccl_alltoallv_locked(comm_state& st, const void* sendbuf, const int* scounts, const int* sdispls, size_t sb,
                          void* recvbuf, const int* rcounts, const int* rdispls, size_t rb, MPI_Request* request,
                          int n)
{
    std::vector<int64_t> a(4 * size_t(n));
    for (int p = 0; p < n; ++p) {
        a[p] = int64_t(scounts[p]) * int64_t(sb);
        a[n + p] = int64_t(sdispls[p]) * int64_t(sb);
        a[2 * n + p] = int64_t(rcounts[p]) * int64_t(rb);
        a[3 * n + p] = int64_t(rdispls[p]) * int64_t(rb);
    }
    if (gkoc_comm_all_to_all_v_bytes(st.rccl, stream(), sendbuf, &a[0], &a[n], recvbuf, &a[2 * n], &a[3 * n]) !=
        GKOC_OK) {
        std::fprintf(stderr, "[gkoc_mpi] device all-to-all-v failed: %s\n", gkoc_last_error());
        for (int p = 0; p < n; ++p) {
            std::fprintf(stderr, "[gkoc_mpi]   peer %d: send %lld bytes at %lld, recv %lld bytes at %lld\n", p,
                         (long long)a[p], (long long)a[n + p], (long long)a[2 * n + p], (long long)a[3 * n + p]);
        }
        return MPI_ERR_OTHER;
    }
    g_stats[2]++;
    if (request) {
        *request = new_handle();
        pending p;
        p.kind = 1;
        g_pending[*request] = p;
    } else {
        sync_progressing();
    }
    return MPI_SUCCESS;
}

int rccl_alltoallv(const void* sendbuf, const int* scounts, const int* sdispls, MPI_Datatype stype, void* recvbuf,
                   const int* rcounts, const int* rdispls, MPI_Datatype rtype, MPI_Comm comm, MPI_Request* request,
                   int n)
{
    flush_binding();
    std::lock_guard<std::mutex> g(g_mtx);
    return rccl_alltoallv_locked(state_of(comm), sendbuf, scounts, sdispls, type_bytes(stype), recvbuf, rcounts,
                                 rdispls, type_bytes(rtype), request, n);
}

// the all-to-all-v family, blocking or not (request != nullptr), over the peers of `comm`
std::map<MPI_Comm, long> g_seq;             // deferred calls issued per communicator (g_mtx)

int alltoallv_routed(int route, const void* sendbuf, const int* scounts, const int* sdispls, MPI_Datatype stype,
                     void* recvbuf, const int* rcounts, const int* rdispls, MPI_Datatype rtype, MPI_Comm comm,
                     MPI_Request* request, int n);

int alltoallv_common(const void* sendbuf, const int* scounts, const int* sdispls, MPI_Datatype stype,
                     void* recvbuf, const int* rcounts, const int* rdispls, MPI_Datatype rtype, MPI_Comm comm,
                     MPI_Request* request, int n_peers_dense)
{
    const int n = n_peers_dense;
    int mine[2] = {0, 0};
    size_t biggest = 0;
    for (int p = 0; p < n; ++p) {
        biggest = std::max(biggest, std::max(size_t(scounts[p]) * type_bytes(stype), size_t(rcounts[p]) * type_bytes(rtype)));
    }
    int route = route_local(comm, sendbuf, all_zero(scounts, n), recvbuf, all_zero(rcounts, n), mine, biggest);
    if (route < 0 && request != nullptr) {
        auto d = std::make_shared<deferred_call>();
        d->which = 0;
        d->sendbuf = sendbuf;
        d->scounts = scounts;
        d->sdispls = sdispls;
        PMPI_Type_dup(stype, &d->stype);      // the caller may free its datatype before the wait (MPI-3.1, 4.1.9)
        d->recvbuf = recvbuf;
        d->rcounts = rcounts;
        d->rdispls = rdispls;
        PMPI_Type_dup(rtype, &d->rtype);
        d->comm = comm;
        d->n = n;
        d->mine[0] = mine[0];
        d->mine[1] = mine[1];
        pending p;
        p.kind = 3;
        p.call = d;
        int rc = PMPI_Iallreduce(d->mine, d->any, 2, MPI_INT, MPI_MAX, comm, &p.inner);
        std::lock_guard<std::mutex> g(g_mtx);
        d->seq = ++g_seq[comm];
        *request = new_handle();
        g_pending[*request] = p;
        return rc;
    }
    if (route < 0) {
        int any[2] = {0, 0};
        PMPI_Allreduce(mine, any, 2, MPI_INT, MPI_MAX, comm);      // a blocking collective may synchronise
        route = route_from(any);
    }
    return alltoallv_routed(route, sendbuf, scounts, sdispls, stype, recvbuf, rcounts, rdispls, rtype, comm, request,
                            n);
}

int alltoallv_routed(int route, const void* sendbuf, const int* scounts, const int* sdispls, MPI_Datatype stype,
                     void* recvbuf, const int* rcounts, const int* rdispls, MPI_Datatype rtype, MPI_Comm comm,
                     MPI_Request* request, int n)
{
    const bool sdev = is_device(sendbuf), rdev = is_device(recvbuf);
    if (route == route_mpi) {
        g_stats[6]++;
        return request ? PMPI_Ialltoallv(sendbuf, scounts, sdispls, stype, recvbuf, rcounts, rdispls, rtype, comm,
                                         request)
                       : PMPI_Alltoallv(sendbuf, scounts, sdispls, stype, recvbuf, rcounts, rdispls, rtype, comm);
    }
    flush_binding();
    std::lock_guard<std::mutex> g(g_mtx);
    comm_state& st = state_of(comm);
    const size_t sb = type_bytes(stype), rb = type_bytes(rtype);
    if (route == route_rccl) {
        return rccl_alltoallv_locked(st, sendbuf, scounts, sdispls, sb, recvbuf, rcounts, rdispls, rb, request, n);
    }
    // staged: whole touched byte range of each device buffer through pinned memory
    g_stats[3]++;
    const span_bytes se = extent(scounts, sdispls, n, sb), re = extent(rcounts, rdispls, n, rb);
    const void* s_use = sendbuf;
    void* r_use = recvbuf;
    void *hs = nullptr, *hr = nullptr;
    if (sdev && se.hi > se.lo) {
        hs = g_host.get(se.hi - se.lo);
        gkoc_memcpy_d2h(hs, static_cast<const char*>(sendbuf) + se.lo, se.hi - se.lo, stream());
        g_stats[5] += long(se.hi - se.lo);
        s_use = static_cast<const char*>(hs) - se.lo;
    }
    if (rdev && re.hi > re.lo) {
        hr = g_host.get(re.hi - re.lo);
        r_use = static_cast<char*>(hr) - re.lo;
    }
    if (!request) {
        int rc = PMPI_Alltoallv(s_use, scounts, sdispls, stype, r_use, rcounts, rdispls, rtype, comm);
        if (hr) {
            gkoc_memcpy_h2d(static_cast<char*>(recvbuf) + re.lo, hr, re.hi - re.lo, stream());
            gkoc_stream_synchronize(stream());
            g_stats[5] += long(re.hi - re.lo);
        }
        g_host.put(hs);
        g_host.put(hr);
        return rc;
    }
    pending p;
    p.kind = 2;
    p.host_send = hs;
    p.host_src = hr;
    p.dev_dst = hr ? static_cast<char*>(recvbuf) + re.lo : nullptr;
    p.bytes = hr ? re.hi - re.lo : 0;
    int rc = PMPI_Ialltoallv(s_use, scounts, sdispls, stype, r_use, rcounts, rdispls, rtype, comm, &p.inner);
    *request = new_handle();
    g_pending[*request] = p;
    return rc;
}

// generic staging of one blocking collective: device buffers are replaced by host copies
struct staged_buf {
    const void* orig = nullptr;
    void* host = nullptr;
    size_t bytes = 0;
    bool dev = false;
    staged_buf(const void* p, size_t b, bool copy_in) : orig(p), bytes(b)
    {
        dev = is_device(p) && b > 0;
        if (dev) {
            host = g_host.get(b);
            if (copy_in) {
                gkoc_memcpy_d2h(host, p, b, stream());
                g_stats[5] += long(b);
            }
        }
    }
    void* use() const { return dev ? host : const_cast<void*>(orig); }
    void copy_out()
    {
        if (dev) {
            gkoc_memcpy_h2d(const_cast<void*>(orig), host, bytes, stream());
            gkoc_stream_synchronize(stream());
            g_stats[5] += long(bytes);
        }
    }
    ~staged_buf() { g_host.put(host); }
};

}  // namespace

extern "C" {

// route counters for tests: [rccl all-reduce, staged all-reduce, rccl all-to-all-v, staged
// all-to-all-v, other staged collectives, bytes copied D2H + H2D, host pass-through]
void gkoc_mpi_stats(long* out7)
{
    std::lock_guard<std::mutex> g(g_mtx);
    for (int i = 0; i < 7; ++i) out7[i] = g_stats[i];
}

int MPI_Finalize(void)
{
    {
        std::lock_guard<std::mutex> g(g_mtx);
        bool any_ipc = false;
        for (auto& kv : g_comms) any_ipc = any_ipc || kv.second.ipc;
        if (any_ipc) {
            // (MPI_Finalize is collective: nobody unmaps a window a peer's kernel may still write into)
            gkoc_device_synchronize();
            PMPI_Barrier(MPI_COMM_WORLD);
        }
        for (auto& kv : g_comms) {
            if (kv.second.rccl && kv.second.ipc) {
                // a wait of the mailbox transport that ran out of patience is an error the job must hear of
                uint32_t st = 0;
                if (gkoc_comm_status(kv.second.rccl, &st) == GKOC_OK && st != 0) {
                    std::fprintf(stderr, "[gkoc_mpi] rank %d: the mailbox transport stopped waiting for a peer (status "
                                         "%#x: 1 all-reduce, 2 message, 4 acknowledgement) - results since are suspect\n",
                                 kv.second.rank, st);
                }
            }
            if (kv.second.rccl) gkoc_comm_destroy(kv.second.rccl);
        }
        g_comms.clear();
        if (std::getenv("GKOC_MPI_VERBOSE")) {
            std::fprintf(stderr,
                         "[gkoc_mpi] all-reduce rccl %ld staged %ld | all-to-all-v rccl %ld staged %ld | other staged "
                         "%ld | %ld bytes through the host | %ld host calls passed through\n",
                         g_stats[0], g_stats[1], g_stats[2], g_stats[3], g_stats[4], g_stats[5], g_stats[6]);
        }
    }
    return PMPI_Finalize();
}

// Ginkgo creates and frees dist-graph and split communicators per matrix: the state attached to a
// handle must go with it, or a recycled handle would pick up an RCCL communicator of other ranks
static void forget_comm(MPI_Comm comm)
{
    std::lock_guard<std::mutex> g(g_mtx);
    auto it = g_comms.find(comm);
    if (it == g_comms.end()) return;
    if (it->second.rccl) {
        gkoc_stream_synchronize(stream());
        // (MPI_Comm_free is collective: every rank is here; no rank unmaps its window while a peer's kernel
        // may still acknowledge into it)
        if (it->second.ipc) PMPI_Barrier(comm);
        gkoc_comm_destroy(it->second.rccl);
    }
    g_comms.erase(it);
}

int MPI_Comm_free(MPI_Comm* comm)
{
    if (comm && *comm != MPI_COMM_NULL) forget_comm(*comm);
    return PMPI_Comm_free(comm);
}

int MPI_Comm_disconnect(MPI_Comm* comm)
{
    if (comm && *comm != MPI_COMM_NULL) forget_comm(*comm);
    return PMPI_Comm_disconnect(comm);
}

int MPI_Allreduce(const void* sendbuf, void* recvbuf, int count, MPI_Datatype datatype, MPI_Op op, MPI_Comm comm)
{
    const bool sdev = is_device(sendbuf), rdev = is_device(recvbuf);
    if ((!sdev && !rdev) || env_mode() == 3 || count == 0) {
        g_stats[6]++;
        return PMPI_Allreduce(sendbuf, recvbuf, count, datatype, op, comm);
    }
    flush_binding();
    std::lock_guard<std::mutex> g(g_mtx);
    comm_state& st = state_of(comm);
    int64_t n = 0;
    const size_t es = rccl_sum_element(datatype, op, count, &n);
    if (st.m == mode::rccl && es && rdev && (sdev || sendbuf == MPI_IN_PLACE)) {
        if (sendbuf != MPI_IN_PLACE && sendbuf != recvbuf) {
            gkoc_memcpy_d2d(recvbuf, sendbuf, size_t(n) * es, stream());
        }
        if (gkoc_comm_all_reduce_sum(st.rccl, stream(), recvbuf, n, es) != GKOC_OK) {
            std::fprintf(stderr, "[gkoc_mpi] RCCL all-reduce failed: %s\n", gkoc_last_error());
            return MPI_ERR_OTHER;
        }
        sync_progressing();
        g_stats[0]++;
        return MPI_SUCCESS;
    }
    g_stats[1]++;
    const size_t bytes = size_t(count) * type_bytes(datatype);
    const bool in_place = sendbuf == MPI_IN_PLACE;
    staged_buf r(recvbuf, bytes, in_place);
    if (in_place) {
        int rc = PMPI_Allreduce(MPI_IN_PLACE, r.use(), count, datatype, op, comm);
        r.copy_out();
        return rc;
    }
    staged_buf s(sendbuf, bytes, true);
    int rc = PMPI_Allreduce(s.use(), r.use(), count, datatype, op, comm);
    r.copy_out();
    return rc;
}

int MPI_Iallreduce(const void* sendbuf, void* recvbuf, int count, MPI_Datatype datatype, MPI_Op op, MPI_Comm comm,
                   MPI_Request* request)
{
    if ((!is_device(sendbuf) && !is_device(recvbuf)) || env_mode() == 3) {
        g_stats[6]++;
        return PMPI_Iallreduce(sendbuf, recvbuf, count, datatype, op, comm, request);
    }
    // device buffers: the blocking route, handed back as an already complete request
    int rc = MPI_Allreduce(sendbuf, recvbuf, count, datatype, op, comm);
    std::lock_guard<std::mutex> g(g_mtx);
    *request = new_handle();
    pending p;
    p.kind = 0;
    g_pending[*request] = p;
    return rc;
}

int MPI_Alltoallv(const void* sendbuf, const int* sendcounts, const int* sdispls, MPI_Datatype sendtype,
                  void* recvbuf, const int* recvcounts, const int* rdispls, MPI_Datatype recvtype, MPI_Comm comm)
{
    int n = 0;
    PMPI_Comm_size(comm, &n);
    return alltoallv_common(sendbuf, sendcounts, sdispls, sendtype, recvbuf, recvcounts, rdispls, recvtype, comm,
                            nullptr, n);
}

int MPI_Ialltoallv(const void* sendbuf, const int* sendcounts, const int* sdispls, MPI_Datatype sendtype,
                   void* recvbuf, const int* recvcounts, const int* rdispls, MPI_Datatype recvtype, MPI_Comm comm,
                   MPI_Request* request)
{
    int n = 0;
    PMPI_Comm_size(comm, &n);
    return alltoallv_common(sendbuf, sendcounts, sdispls, sendtype, recvbuf, recvcounts, rdispls, recvtype, comm,
                            request, n);
}

int MPI_Alltoall(const void* sendbuf, int sendcount, MPI_Datatype sendtype, void* recvbuf, int recvcount,
                 MPI_Datatype recvtype, MPI_Comm comm)
{
    if ((!is_device(sendbuf) && !is_device(recvbuf)) || env_mode() == 3) {
        g_stats[6]++;
        return PMPI_Alltoall(sendbuf, sendcount, sendtype, recvbuf, recvcount, recvtype, comm);
    }
    int n = 0;
    PMPI_Comm_size(comm, &n);
    if (sendbuf == MPI_IN_PLACE) {
        flush_binding();
        std::lock_guard<std::mutex> g(g_mtx);
        g_stats[4]++;
        staged_buf r(recvbuf, size_t(n) * size_t(recvcount) * type_bytes(recvtype), true);
        int rc = PMPI_Alltoall(MPI_IN_PLACE, sendcount, sendtype, r.use(), recvcount, recvtype, comm);
        r.copy_out();
        return rc;
    }
    std::vector<int> c(4 * size_t(n));
    for (int p = 0; p < n; ++p) {
        c[p] = sendcount;
        c[n + p] = p * sendcount;
        c[2 * n + p] = recvcount;
        c[3 * n + p] = p * recvcount;
    }
    return alltoallv_common(sendbuf, &c[0], &c[n], sendtype, recvbuf, &c[2 * n], &c[3 * n], recvtype, comm, nullptr,
                            n);
}

int MPI_Allgather(const void* sendbuf, int sendcount, MPI_Datatype sendtype, void* recvbuf, int recvcount,
                  MPI_Datatype recvtype, MPI_Comm comm)
{
    if ((!is_device(sendbuf) && !is_device(recvbuf)) || env_mode() == 3) {
        g_stats[6]++;
        return PMPI_Allgather(sendbuf, sendcount, sendtype, recvbuf, recvcount, recvtype, comm);
    }
    flush_binding();
    std::lock_guard<std::mutex> g(g_mtx);
    g_stats[4]++;
    int n = 0;
    PMPI_Comm_size(comm, &n);
    const bool in_place = sendbuf == MPI_IN_PLACE;
    staged_buf r(recvbuf, size_t(n) * size_t(recvcount) * type_bytes(recvtype), in_place);
    int rc;
    if (in_place) {
        rc = PMPI_Allgather(MPI_IN_PLACE, sendcount, sendtype, r.use(), recvcount, recvtype, comm);
    } else {
        staged_buf s(sendbuf, size_t(sendcount) * type_bytes(sendtype), true);
        rc = PMPI_Allgather(s.use(), sendcount, sendtype, r.use(), recvcount, recvtype, comm);
    }
    r.copy_out();
    return rc;
}

int MPI_Bcast(void* buffer, int count, MPI_Datatype datatype, int root, MPI_Comm comm)
{
    if (!is_device(buffer) || env_mode() == 3) {
        g_stats[6]++;
        return PMPI_Bcast(buffer, count, datatype, root, comm);
    }
    flush_binding();
    std::lock_guard<std::mutex> g(g_mtx);
    g_stats[4]++;
    int rank = 0;
    PMPI_Comm_rank(comm, &rank);
    staged_buf b(buffer, size_t(count) * type_bytes(datatype), rank == root);
    int rc = PMPI_Bcast(b.use(), count, datatype, root, comm);
    if (rank != root) b.copy_out();
    return rc;
}

int MPI_Send(const void* buf, int count, MPI_Datatype datatype, int dest, int tag, MPI_Comm comm)
{
    if (!is_device(buf) || env_mode() == 3) return PMPI_Send(buf, count, datatype, dest, tag, comm);
    flush_binding();
    std::lock_guard<std::mutex> g(g_mtx);
    g_stats[4]++;
    staged_buf b(buf, size_t(count) * type_bytes(datatype), true);
    return PMPI_Send(b.use(), count, datatype, dest, tag, comm);
}

int MPI_Recv(void* buf, int count, MPI_Datatype datatype, int source, int tag, MPI_Comm comm, MPI_Status* status)
{
    if (!is_device(buf) || env_mode() == 3) return PMPI_Recv(buf, count, datatype, source, tag, comm, status);
    flush_binding();
    std::lock_guard<std::mutex> g(g_mtx);
    g_stats[4]++;
    staged_buf b(buf, size_t(count) * type_bytes(datatype), false);
    int rc = PMPI_Recv(b.use(), count, datatype, source, tag, comm, status);
    b.copy_out();
    return rc;
}

int neighbor_routed(int route, const void* sendbuf, const int* sendcounts, const int* sdispls, MPI_Datatype sendtype,
                    void* recvbuf, const int* recvcounts, const int* rdispls, MPI_Datatype recvtype, MPI_Comm comm,
                    MPI_Request* request);

// neighbourhood collective of NeighborhoodCommunicator: counts per neighbour of the graph topology
int MPI_Ineighbor_alltoallv(const void* sendbuf, const int* sendcounts, const int* sdispls, MPI_Datatype sendtype,
                            void* recvbuf, const int* recvcounts, const int* rdispls, MPI_Datatype recvtype,
                            MPI_Comm comm, MPI_Request* request)
{
    if (env_mode() == 3) {
        g_stats[6]++;
        return PMPI_Ineighbor_alltoallv(sendbuf, sendcounts, sdispls, sendtype, recvbuf, recvcounts, rdispls, recvtype,
                                        comm, request);
    }
    int indeg = 0, outdeg = 0, weighted = 0, n = 0;
    PMPI_Dist_graph_neighbors_count(comm, &indeg, &outdeg, &weighted);
    PMPI_Comm_size(comm, &n);
    // the route is agreed by all ranks of the communicator, not read off local pointers - without blocking:
    // the agreement travels in the request, the exchange is carried out at completion (deferred_call)
    int mine[2] = {0, 0};
    size_t biggest = 0;
    for (int i = 0; i < outdeg; ++i) biggest = std::max(biggest, size_t(sendcounts[i]) * type_bytes(sendtype));
    for (int i = 0; i < indeg; ++i) biggest = std::max(biggest, size_t(recvcounts[i]) * type_bytes(recvtype));
    const int route = route_local(comm, sendbuf, all_zero(sendcounts, outdeg), recvbuf, all_zero(recvcounts, indeg),
                                  mine, biggest);
    if (route < 0) {
        auto d = std::make_shared<deferred_call>();
        d->which = 1;
        d->sendbuf = sendbuf;
        d->scounts = sendcounts;
        d->sdispls = sdispls;
        PMPI_Type_dup(sendtype, &d->stype);   // the caller may free its datatype before the wait (MPI-3.1, 4.1.9)
        d->recvbuf = recvbuf;
        d->rcounts = recvcounts;
        d->rdispls = rdispls;
        PMPI_Type_dup(recvtype, &d->rtype);
        d->comm = comm;
        d->n = n;
        d->mine[0] = mine[0];
        d->mine[1] = mine[1];
        pending p;
        p.kind = 3;
        p.call = d;
        int rc = PMPI_Iallreduce(d->mine, d->any, 2, MPI_INT, MPI_MAX, comm, &p.inner);
        std::lock_guard<std::mutex> g(g_mtx);
        d->seq = ++g_seq[comm];
        *request = new_handle();
        g_pending[*request] = p;
        return rc;
    }
    return neighbor_routed(route, sendbuf, sendcounts, sdispls, sendtype, recvbuf, recvcounts, rdispls, recvtype, comm,
                           request);
}

// request == nullptr: the blocking form (a deferred call at its completion)
int neighbor_routed(int route, const void* sendbuf, const int* sendcounts, const int* sdispls, MPI_Datatype sendtype,
                    void* recvbuf, const int* recvcounts, const int* rdispls, MPI_Datatype recvtype, MPI_Comm comm,
                    MPI_Request* request)
{
    int indeg = 0, outdeg = 0, weighted = 0, n = 0;
    PMPI_Dist_graph_neighbors_count(comm, &indeg, &outdeg, &weighted);
    PMPI_Comm_size(comm, &n);
    if (route == route_mpi) {
        g_stats[6]++;
        return request ? PMPI_Ineighbor_alltoallv(sendbuf, sendcounts, sdispls, sendtype, recvbuf, recvcounts, rdispls,
                                                  recvtype, comm, request)
                       : PMPI_Neighbor_alltoallv(sendbuf, sendcounts, sdispls, sendtype, recvbuf, recvcounts, rdispls,
                                                 recvtype, comm);
    }
    if (route == route_rccl) {
        std::vector<int> src(indeg ? indeg : 1), dst(outdeg ? outdeg : 1), w(indeg + outdeg + 1);
        PMPI_Dist_graph_neighbors(comm, indeg, src.data(), w.data(), outdeg, dst.data(), w.data());
        // dense counts over the ranks of the communicator (a neighbour appears once in Ginkgo's graphs)
        std::vector<int> c(4 * size_t(n), 0);
        for (int i = 0; i < outdeg; ++i) {
            c[dst[i]] = sendcounts[i];
            c[n + dst[i]] = sdispls[i];
        }
        for (int i = 0; i < indeg; ++i) {
            c[2 * n + src[i]] = recvcounts[i];
            c[3 * n + src[i]] = rdispls[i];
        }
        return rccl_alltoallv(sendbuf, &c[0], &c[n], sendtype, recvbuf, &c[2 * n], &c[3 * n], recvtype, comm, request,
                              n);
    }
    // staged: keep the neighbourhood call, with host copies of the touched ranges
    flush_binding();
    std::lock_guard<std::mutex> g(g_mtx);
    g_stats[3]++;
    const size_t sb = type_bytes(sendtype), rb = type_bytes(recvtype);
    const span_bytes se = extent(sendcounts, sdispls, outdeg, sb), re = extent(recvcounts, rdispls, indeg, rb);
    const void* s_use = sendbuf;
    void* r_use = recvbuf;
    pending p;
    p.kind = 2;
    if (is_device(sendbuf) && se.hi > se.lo) {
        p.host_send = g_host.get(se.hi - se.lo);
        gkoc_memcpy_d2h(p.host_send, static_cast<const char*>(sendbuf) + se.lo, se.hi - se.lo, stream());
        g_stats[5] += long(se.hi - se.lo);
        s_use = static_cast<const char*>(p.host_send) - se.lo;
    }
    if (is_device(recvbuf) && re.hi > re.lo) {
        p.host_src = g_host.get(re.hi - re.lo);
        p.dev_dst = static_cast<char*>(recvbuf) + re.lo;
        p.bytes = re.hi - re.lo;
        r_use = static_cast<char*>(p.host_src) - re.lo;
    }
    if (!request) {
        int rc = PMPI_Neighbor_alltoallv(s_use, sendcounts, sdispls, sendtype, r_use, recvcounts, rdispls, recvtype,
                                         comm);
        if (p.dev_dst && p.bytes) {
            gkoc_memcpy_h2d(p.dev_dst, p.host_src, p.bytes, stream());
            gkoc_stream_synchronize(stream());
            g_stats[5] += long(p.bytes);
        }
        g_host.put(p.host_src);
        g_host.put(p.host_send);
        return rc;
    }
    int rc = PMPI_Ineighbor_alltoallv(s_use, sendcounts, sdispls, sendtype, r_use, recvcounts, rdispls, recvtype, comm,
                                      &p.inner);
    *request = new_handle();
    g_pending[*request] = p;
    return rc;
}

static void deferred_ended(deferred_call& d)
{
    if (d.done) return;
    PMPI_Type_free(&d.stype);
    PMPI_Type_free(&d.rtype);
    d.done = true;
}

// carry out one deferred call (its agreement first); g_mtx NOT held
static void run_deferred(pending& p)
{
    deferred_call& d = *p.call;
    if (d.done) return;
    if (d.launched) {
        // MPI_Test has started it: bring the started exchange to its end (MPI_Wait of this layer knows both kinds)
        if (d.sub != MPI_REQUEST_NULL) {
            const int rc = MPI_Wait(&d.sub, MPI_STATUS_IGNORE);
            if (rc != MPI_SUCCESS) d.rc = rc;
        }
        deferred_ended(d);
        return;
    }
    if (p.inner != MPI_REQUEST_NULL) PMPI_Wait(&p.inner, MPI_STATUS_IGNORE);
    const int route = route_from(d.any);
    d.rc = d.which == 0
               ? alltoallv_routed(route, d.sendbuf, d.scounts, d.sdispls, d.stype, d.recvbuf, d.rcounts, d.rdispls,
                                  d.rtype, d.comm, nullptr, d.n)
               : neighbor_routed(route, d.sendbuf, d.scounts, d.sdispls, d.stype, d.recvbuf, d.rcounts, d.rdispls,
                                 d.rtype, d.comm, nullptr);
    deferred_ended(d);
}

// MPI_Test's half of the same: START the exchange (the agreement has arrived), never wait for it.  g_mtx NOT held
static void start_deferred(deferred_call& d)
{
    if (d.done || d.launched) return;
    const int route = route_from(d.any);
    d.rc = d.which == 0
               ? alltoallv_routed(route, d.sendbuf, d.scounts, d.sdispls, d.stype, d.recvbuf, d.rcounts, d.rdispls,
                                  d.rtype, d.comm, &d.sub, d.n)
               : neighbor_routed(route, d.sendbuf, d.scounts, d.sdispls, d.stype, d.recvbuf, d.rcounts, d.rdispls,
                                 d.rtype, d.comm, &d.sub);
    d.launched = true;
}

// has a started exchange reached its end?  Looks, never waits (a look at the stream / PMPI_Test, which also
// drives MPI's progress engine); finishes the local part (copy back, release) when it has.  g_mtx NOT held
static bool started_has_ended(deferred_call& d)
{
    if (d.sub == MPI_REQUEST_NULL) return true;
    pending sp;
    bool ours = false;
    {
        std::lock_guard<std::mutex> g(g_mtx);
        auto it = g_pending.find(d.sub);
        ours = it != g_pending.end();
        if (ours) sp = it->second;
    }
    if (!ours) {
        int f = 0;
        const int rc = PMPI_Test(&d.sub, &f, MPI_STATUS_IGNORE);
        if (rc != MPI_SUCCESS) d.rc = rc;
        return f != 0;
    }
    if (sp.kind == 1) {
        int done = 0;
        if (gkoc_stream_query(stream(), &done) != GKOC_OK) done = 1;
        if (!done) {
            int flag = 0;
            PMPI_Iprobe(MPI_ANY_SOURCE, MPI_ANY_TAG, MPI_COMM_WORLD, &flag, MPI_STATUS_IGNORE);
            return false;
        }
    } else if (sp.kind == 2 && sp.inner != MPI_REQUEST_NULL) {
        int f = 0;
        const int rc = PMPI_Test(&sp.inner, &f, MPI_STATUS_IGNORE);
        if (rc != MPI_SUCCESS) d.rc = rc;
        if (!f) return false;
        sp.inner = MPI_REQUEST_NULL;
    }
    {
        std::lock_guard<std::mutex> g(g_mtx);
        g_pending.erase(d.sub);
        const int rc = finish(sp);       // kind 1: the stream is idle; kind 2: the copy back (local)
        if (rc != MPI_SUCCESS) d.rc = rc;
    }
    PMPI_Grequest_complete(d.sub);
    PMPI_Wait(&d.sub, MPI_STATUS_IGNORE);
    return true;
}

static int complete_ours(MPI_Request* request, MPI_Status* status, bool* ours)
{
    pending p;
    std::vector<pending> earlier;
    {
        std::lock_guard<std::mutex> g(g_mtx);
        auto it = g_pending.find(*request);
        *ours = it != g_pending.end();
        if (!*ours) return MPI_SUCCESS;
        p = it->second;
        g_pending.erase(it);
        if (p.kind == 3) {
            // collectives of a communicator are carried out in the order they were ISSUED, on every rank -
            // whatever order the application completes its requests in
            for (auto& q : g_pending) {
                if (q.second.kind == 3 && q.second.call->comm == p.call->comm && q.second.call->seq < p.call->seq &&
                    !q.second.call->done) {
                    earlier.push_back(q.second);
                }
            }
        } else {
            finish(p);
        }
    }
    int rc = MPI_SUCCESS;
    if (p.kind == 3) {
        std::sort(earlier.begin(), earlier.end(),
                  [](const pending& a, const pending& b) { return a.call->seq < b.call->seq; });
        for (auto& q : earlier) run_deferred(q);      // (their requests find the work done when they are completed)
        run_deferred(p);
        rc = p.call->rc;
    }
    PMPI_Grequest_complete(*request);
    int rc2 = PMPI_Wait(request, status);
    return rc != MPI_SUCCESS ? rc : rc2;
}

int MPI_Wait(MPI_Request* request, MPI_Status* status)
{
    if (request && *request != MPI_REQUEST_NULL) {
        bool ours = false;
        int rc = complete_ours(request, status, &ours);
        if (ours) return rc;
    }
    return PMPI_Wait(request, status);
}

int MPI_Test(MPI_Request* request, int* flag, MPI_Status* status)
{
    if (request && *request != MPI_REQUEST_NULL) {
        // A deferred call (and the deferred calls issued before it on its communicator: collectives are carried
        // out in the order of issue on every rank) is STARTED here once its agreement has arrived and looked at
        // again by later calls: MPI_Test returns flag = 0 until the started exchange has ended.  It never waits
        // for a peer - a rank that polls MPI_Test and serves point-to-point messages in between stays live
        // (tests/dropin/mpi_layer_test.cpp "MPI_Test does not wait for the peers").
        std::vector<std::shared_ptr<deferred_call>> chain;
        bool deferred = false;
        {
            std::lock_guard<std::mutex> g(g_mtx);
            auto it = g_pending.find(*request);
            if (it != g_pending.end() && it->second.kind == 3 && !it->second.call->done) {
                deferred = true;
                const auto mine = it->second.call;
                std::vector<pending*> order;
                for (auto& q : g_pending) {
                    if (q.second.kind == 3 && q.second.call->comm == mine->comm && q.second.call->seq <= mine->seq &&
                        !q.second.call->done) {
                        order.push_back(&q.second);
                    }
                }
                std::sort(order.begin(), order.end(),
                          [](const pending* a, const pending* b) { return a->call->seq < b->call->seq; });
                for (pending* q : order) {
                    if (q->inner != MPI_REQUEST_NULL) {
                        int arrived = 0;
                        PMPI_Test(&q->inner, &arrived, MPI_STATUS_IGNORE);      // (sets inner to null when it has)
                        if (!arrived) {
                            *flag = 0;
                            return MPI_SUCCESS;
                        }
                    }
                    chain.push_back(q->call);
                }
            }
        }
        if (deferred) {
            for (auto& d : chain) {
                start_deferred(*d);
                if (!started_has_ended(*d)) {
                    *flag = 0;
                    return MPI_SUCCESS;
                }
                deferred_ended(*d);
            }
        } else {
            // a started exchange of ours (rccl / staged): look, do not wait
            pending sp;
            bool ours = false;
            {
                std::lock_guard<std::mutex> g(g_mtx);
                auto it = g_pending.find(*request);
                ours = it != g_pending.end();
                if (ours) sp = it->second;
            }
            if (ours && sp.kind == 1) {
                int done = 0;
                if (gkoc_stream_query(stream(), &done) == GKOC_OK && !done) {
                    int f = 0;
                    PMPI_Iprobe(MPI_ANY_SOURCE, MPI_ANY_TAG, MPI_COMM_WORLD, &f, MPI_STATUS_IGNORE);
                    *flag = 0;
                    return MPI_SUCCESS;
                }
            } else if (ours && sp.kind == 2 && sp.inner != MPI_REQUEST_NULL) {
                int f = 0;
                PMPI_Request_get_status(sp.inner, &f, MPI_STATUS_IGNORE);       // (does not free the request)
                if (!f) {
                    *flag = 0;
                    return MPI_SUCCESS;
                }
            }
        }
        bool ours = false;
        int rc = complete_ours(request, status, &ours);   // everything it would wait for has ended
        if (ours) {
            *flag = 1;
            return rc;
        }
    }
    return PMPI_Test(request, flag, status);
}

int MPI_Waitall(int count, MPI_Request requests[], MPI_Status statuses[])
{
    for (int i = 0; i < count; ++i) {
        if (requests[i] == MPI_REQUEST_NULL) continue;
        bool ours = false;
        complete_ours(&requests[i], statuses == MPI_STATUSES_IGNORE ? MPI_STATUS_IGNORE : &statuses[i], &ours);
    }
    return PMPI_Waitall(count, requests, statuses);
}

int MPI_Request_free(MPI_Request* request)
{
    if (request && *request != MPI_REQUEST_NULL) {
        bool ours = false;
        int rc = complete_ours(request, MPI_STATUS_IGNORE, &ours);
        if (ours) return rc;
    }
    return PMPI_Request_free(request);
}

}  // extern "C"
