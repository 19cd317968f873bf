// SPDX-License-Identifier: BSD-3-Clause
// Row-partitioned CG / PipeCg + block-Jacobi(8) on the 27-pt Laplacian, one process per GPU,
// driven from C++ through the C ABI of include/gko_cdna4.h only (no Ginkgo, no Python, no MPI,
// no HIP headers): the host side of one iteration is a dozen gkoc_* calls, ~1-2 us each.
//
// What experimental::distributed::{Matrix, Vector} + solver::{Cg, PipeCg} do in the reference
// (core/distributed/matrix.cpp:300-509, vector.cpp:473-592, core/solver/cg.cpp:93-181,
// pipe_cg.cpp:95-297), restated for a z-slab partition:
//   * every rank generates its planes on the device, splits them into the local CSR block and the
//     non-local row list (gkoc_dist_split_*), and needs one plane from each z-neighbour;
//   * apply: grouped RCCL send/recv of the boundary planes on a side stream, straight out of the
//     vector (zero-copy displacements) || local SpMV, then the boundary rows;
//   * cg:      all-reduce [<r,z>, ||r||^2] (one message) and <p,q> per iteration;
//   * pipe_cg: ONE all-reduce [<r,z>, <w,z>, ||r||^2] per iteration, travelling on the side
//              stream while m = M^-1 w and n = A m run (gkoc_comm_all_reduce_begin / _end);
//   * the criterion kernel's flags are read `lag` iterations later from pinned memory (the step
//     kernels are masked by stop_status, so x is the same as with lag 0), the same number of
//     iterations later on every rank, so no collective is left unmatched.
//
// Rank / world size come from the launcher's environment (torchrun: RANK, WORLD_SIZE, LOCAL_RANK;
// mpiexec: PMI_RANK, PMI_SIZE); the RCCL communicator id travels through a file (GKOC_ID_FILE,
// default /tmp/gkoc_comm_id.<MASTER_PORT or ppid>), which is all one node needs.
// `mirror`: ONE process plays rank 0 of a 2-slab run of the z-mirror-symmetric problem (rhs = 1):
// the plane its peer would send is the plane it sends itself, and every global sum is twice the
// local one - the complete multi-rank code path, RCCL calls included, on a single GPU.
//
// Build:  g++ -O2 -std=c++17 -Iinclude examples/native_dist_cg.cpp -Lginkgo_amd/lib -lgko_cdna4
//             -Wl,-rpath,'$ORIGIN/../ginkgo_amd/lib' -o examples/native_dist_cg
// Run:    [launcher] examples/native_dist_cg [grid=64] [max_iters=1000] [reduction=1e-10]
//                    [cg|pipe_cg] [lag=4] [mirror] [dump=<file>]
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <deque>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gko_cdna4.h"

#define CK(call)                                                                         \
    do {                                                                                 \
        if (int rc_ = (call)) {                                                          \
            throw std::runtime_error(std::string(#call) + " -> " + std::to_string(rc_) + \
                                     ": " + gkoc_last_error());                          \
        }                                                                                \
    } while (0)

template <typename T>
struct dev_array {
    T* p = nullptr;
    size_t n = 0;
    explicit dev_array(size_t count, int role = GKOC_MEM_VECTOR) : n(count)
    {
        void* q = nullptr;
        CK(gkoc_malloc_role(&q, sizeof(T) * (count ? count : 1), role));
        p = static_cast<T*>(q);
    }
    dev_array(const dev_array&) = delete;
    ~dev_array() { gkoc_free(p); }
    void upload(const std::vector<T>& h) { CK(gkoc_memcpy_h2d(p, h.data(), sizeof(T) * h.size(), nullptr)); }
    std::vector<T> download() const
    {
        std::vector<T> h(n);
        CK(gkoc_memcpy_d2h(h.data(), p, sizeof(T) * n, nullptr));
        return h;
    }
};

// include/ginkgo/core/preconditioner/jacobi.hpp:589-627 with max_block_stride = 64
static gkoc_jacobi_scheme storage_scheme(uint32_t max_block_size)
{
    uint32_t p2 = 1;
    while (p2 < max_block_size) p2 *= 2;
    const uint32_t group_size = 64 / p2;
    uint32_t gp = 0;
    while ((1u << gp) < group_size) ++gp;
    gkoc_jacobi_scheme s;
    s.block_offset = max_block_size;
    s.group_offset = int64_t(max_block_size) * group_size * max_block_size;
    s.group_power = gp;
    return s;
}

static int env_int(const char* a, const char* b, int dflt)
{
    const char* v = getenv(a);
    if (!v && b) v = getenv(b);
    return v ? atoi(v) : dflt;
}

// rank 0 writes the 128-byte communicator id, the others wait for the complete file
static void exchange_id(int rank, unsigned char* id)
{
    std::string path;
    if (const char* f = getenv("GKOC_ID_FILE")) {
        path = f;
    } else if (const char* port = getenv("MASTER_PORT")) {
        path = std::string("/tmp/gkoc_comm_id.") + port;
    } else {
        path = "/tmp/gkoc_comm_id." + std::to_string(rank == 0 ? getpid() : getppid());
    }
    if (rank == 0) {
        const std::string tmp = path + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, GKOC_COMM_ID_BYTES, f) != GKOC_COMM_ID_BYTES) {
            throw std::runtime_error("cannot write " + tmp);
        }
        fclose(f);
        if (rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot rename " + tmp);
        return;
    }
    for (int tries = 0; tries < 6000; ++tries) {
        if (FILE* f = fopen(path.c_str(), "rb")) {
            const size_t got = fread(id, 1, GKOC_COMM_ID_BYTES, f);
            fclose(f);
            if (got == GKOC_COMM_ID_BYTES) return;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    throw std::runtime_error("no communicator id in " + path);
}

// the mailbox transport (gkoc_comm_ipc_*): every rank publishes the 64-byte handle of its window in
// <path>.<rank> and reads the others' - ranks may share one GPU (RCCL refuses that)
static std::string rendezvous_path()
{
    if (const char* f = getenv("GKOC_ID_FILE")) return f;
    if (const char* port = getenv("MASTER_PORT")) return std::string("/tmp/gkoc_comm_id.") + port;
    return "/tmp/gkoc_comm_id." + std::to_string(getppid());
}

static void exchange_handles(int rank, int world, const unsigned char* mine, std::vector<unsigned char>& all)
{
    const std::string base = rendezvous_path() + ".ipc.";
    {
        const std::string path = base + std::to_string(rank), tmp = path + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(mine, 1, GKOC_COMM_IPC_HANDLE_BYTES, f) != GKOC_COMM_IPC_HANDLE_BYTES) {
            throw std::runtime_error("cannot write " + tmp);
        }
        fclose(f);
        if (rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot rename " + tmp);
    }
    all.assign(size_t(world) * GKOC_COMM_IPC_HANDLE_BYTES, 0);
    for (int r = 0; r < world; ++r) {
        const std::string path = base + std::to_string(r);
        bool ok = false;
        for (int tries = 0; tries < 6000 && !ok; ++tries) {
            if (FILE* f = fopen(path.c_str(), "rb")) {
                ok = fread(&all[size_t(r) * GKOC_COMM_IPC_HANDLE_BYTES], 1, GKOC_COMM_IPC_HANDLE_BYTES, f) ==
                     GKOC_COMM_IPC_HANDLE_BYTES;
                fclose(f);
            }
            if (!ok) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!ok) throw std::runtime_error("no window handle in " + path);
    }
}

int main(int argc, char** argv)
try {
    const int64_t grid = argc > 1 ? atoll(argv[1]) : 64;
    const int64_t max_iters = argc > 2 ? atoll(argv[2]) : 1000;
    const double reduction = argc > 3 ? atof(argv[3]) : 1e-10;
    const bool pipe = argc > 4 && !strcmp(argv[4], "pipe_cg");
    int lag = argc > 5 ? atoi(argv[5]) : 4;
    bool mirror = false;
    std::string dump;
    bool use_ipc = getenv("GKOC_EXAMPLE_TRANSPORT") && !strcmp(getenv("GKOC_EXAMPLE_TRANSPORT"), "ipc");
    for (int i = 6; i < argc; ++i) {
        if (!strcmp(argv[i], "mirror")) mirror = true;
        if (!strcmp(argv[i], "ipc")) use_ipc = true;
        if (!strncmp(argv[i], "dump=", 5)) dump = argv[i] + 5;
    }
    if (lag < 0) lag = 0;
    if (lag > 12) lag = 12;
    const uint32_t bs = 8;

    // ---- who am I
    int real_rank = env_int("RANK", "PMI_RANK", 0);
    int real_world = env_int("WORLD_SIZE", "PMI_SIZE", 1);
    const int local_rank = env_int("LOCAL_RANK", "MPI_LOCALRANKID", real_rank);
    int n_dev = 0;
    CK(gkoc_get_num_devices(&n_dev));
    CK(gkoc_set_device(local_rank % (n_dev > 0 ? n_dev : 1)));
    if (mirror && real_world != 1) throw std::runtime_error("mirror needs exactly one process");
    // logical partition: in mirror mode this process is rank 0 of 2
    const int rank = real_rank, world = mirror ? 2 : real_world;
    // Partition::build_from_global_size_uniform on planes: the first (grid % world) get one more
    const int64_t plane = grid * grid;
    auto plane_begin = [&](int r) { return (grid / world) * r + std::min<int64_t>(r, grid % world); };
    const int64_t z0 = plane_begin(rank), z1 = plane_begin(rank + 1);
    const int64_t n = (z1 - z0) * plane, n_global = grid * plane;
    const int64_t lo = z0 * plane, hi = z1 * plane;
    if (n < 2 * plane && world > 1) throw std::runtime_error("fewer than two planes per rank");

    gkoc_stream_t s = nullptr, side = nullptr;
    if (!getenv("GKOC_EXAMPLE_NULL_STREAM")) CK(gkoc_stream_create(&s));   // else: the NULL stream
    CK(gkoc_stream_create_high_priority(&side));   // collectives must get onto the device beside the SpMV

    // ---- communicator (RCCL), id through a file
    gkoc_comm_t comm = nullptr;
    if (world > 1 && use_ipc && !mirror) {
        // the library's own transport: mailboxes in peer-mapped device memory, one window per rank
        unsigned char mine[GKOC_COMM_IPC_HANDLE_BYTES] = {};
        std::vector<unsigned char> all;
        CK(gkoc_comm_ipc_create(&comm, real_world, real_rank, 0, mine));
        exchange_handles(real_rank, real_world, mine, all);
        const int rc = gkoc_comm_ipc_connect(comm, all.data());
        if (rc == GKOC_E_NOT_SUPPORTED) {
            // refused on EVERY rank alike (the cards say: different devices, a window in plain memory): RCCL
            if (real_rank == 0) fprintf(stderr, "mailbox transport refused (%s): RCCL\n", gkoc_last_error());
            CK(gkoc_comm_destroy(comm));
            comm = nullptr;
            use_ipc = false;
        } else {
            CK(rc);
        }
    }
    if (world > 1 && comm == nullptr) {      // (mirror: a real one-rank RCCL communicator)
        unsigned char id[GKOC_COMM_ID_BYTES] = {};
        CK(gkoc_comm_load_rccl(getenv("GKOC_RCCL_PATH")));
        if (real_rank == 0) CK(gkoc_comm_unique_id(id));
        if (real_world > 1) exchange_id(real_rank, id);
        CK(gkoc_comm_create(&comm, real_world, real_rank, id));
    }
    const int n_peers = mirror ? 1 : real_world;   // entries of the count arrays

    // ---- this rank's planes, split into local block + non-local row list
    dev_array<int32_t> row_ptrs(n + 1, GKOC_MEM_INDICES);
    int64_t nnz = 0;
    CK(gkoc_stencil_row_ptrs_i32(s, 3, grid, 0, z0, z1 - z0, row_ptrs.p, &nnz));
    int64_t n_halo = 0, nnz_l = 0, nnz_nl = 0, n_nl_rows = 0;
    dev_array<int32_t> local_ptrs(n + 1, GKOC_MEM_INDICES);
    std::unique_ptr<dev_array<int32_t>> local_cols, nl_rows, nl_ptrs, nl_cols, recv_gidx;
    std::unique_ptr<dev_array<double>> local_vals, nl_vals;
    {
        dev_array<int32_t> cols(nnz, GKOC_MEM_INDICES);
        dev_array<double> vals(nnz, GKOC_MEM_VALUES);
        CK(gkoc_stencil_fill_f64_i32(s, 3, grid, 0, z0, z1 - z0, row_ptrs.p, cols.p, vals.p));
        dev_array<int32_t> col_map(n_global + 1), nl_full(n + 1);
        CK(gkoc_dist_split_count_i32(s, n, row_ptrs.p, cols.p, lo, hi, n_global, col_map.p,
                                     local_ptrs.p, nl_full.p, &n_halo, &nnz_l, &nnz_nl, &n_nl_rows));
        local_cols.reset(new dev_array<int32_t>(nnz_l, GKOC_MEM_INDICES));
        local_vals.reset(new dev_array<double>(nnz_l, GKOC_MEM_VALUES));
        nl_rows.reset(new dev_array<int32_t>(n_nl_rows));
        nl_ptrs.reset(new dev_array<int32_t>(n_nl_rows + 1));
        nl_cols.reset(new dev_array<int32_t>(nnz_nl));
        nl_vals.reset(new dev_array<double>(nnz_nl));
        recv_gidx.reset(new dev_array<int32_t>(n_halo));
        CK(gkoc_dist_split_fill_f64_i32(s, n, row_ptrs.p, cols.p, vals.p, lo, hi, n_global, col_map.p,
                                        local_ptrs.p, nl_full.p, local_cols->p, local_vals->p,
                                        nl_rows->p, nl_ptrs->p, nl_cols->p, nl_vals->p, recv_gidx->p));
        CK(gkoc_stream_synchronize(s));
    }
    const bool has_lower = rank > 0, has_upper = rank < world - 1;
    if (n_halo != plane * (int64_t(has_lower) + int64_t(has_upper))) {
        throw std::runtime_error("unexpected halo size");
    }
    // exchange plan: halo slots ascend with the global row, i.e. [plane below | plane above];
    // what a neighbour needs from us is our first / last plane, contiguous in the vector
    std::vector<int64_t> send_counts(n_peers, 0), send_displs(n_peers, 0), recv_counts(n_peers, 0);
    if (mirror) {
        // our only neighbour (rank 1) is played by ourselves: peer index 0 of the 1-rank comm
        send_counts[0] = recv_counts[0] = plane;
        send_displs[0] = n - plane;
    } else {
        if (has_lower) { send_counts[rank - 1] = recv_counts[rank - 1] = plane; send_displs[rank - 1] = 0; }
        if (has_upper) { send_counts[rank + 1] = recv_counts[rank + 1] = plane; send_displs[rank + 1] = n - plane; }
    }
    dev_array<double> halo(n_halo);

    // ---- block-Jacobi(8) of the local block (the blocks never cross ranks: plane % 8 == 0 here
    // or not, find_blocks only sees the local block)
    dev_array<int32_t> block_ptrs(n + 1, GKOC_MEM_INDICES);
    int64_t num_blocks = 0;
    CK(gkoc_jacobi_find_blocks_f64_i32(s, n, local_ptrs.p, local_cols->p, bs, &num_blocks, block_ptrs.p));
    const gkoc_jacobi_scheme scheme = storage_scheme(bs);
    const int64_t gsize = int64_t(1) << scheme.group_power;
    dev_array<double> blocks(size_t((num_blocks + gsize - 1) / gsize * scheme.group_offset), GKOC_MEM_INDICES);
    CK(gkoc_memset(blocks.p, 0, sizeof(double) * blocks.n, s));
    CK(gkoc_jacobi_generate_f64_i32(s, n, local_ptrs.p, local_cols->p, local_vals->p, num_blocks, bs,
                                    scheme, block_ptrs.p, blocks.p, nullptr));

    // ---- vectors, scalars, workspaces
    // z, the preconditioner's output, in another memory class than its inputs (blocks: class of
    // the indices, r: class of the vectors)
    dev_array<double> b(n), x(n), r(n), z(n, GKOC_MEM_VALUES), p(n), q(n);
    std::unique_ptr<dev_array<double>> w, m, nn, f, g;
    if (pipe) {
        for (auto* v : {&w, &m, &nn, &f, &g}) v->reset(new dev_array<double>(n));
    }
    // two scalar groups that swap roles as (rho, prev_rho): cg [rho, ||r||^2, -], pipe_cg
    // [rho, delta, ||r||^2] - what one iteration all-reduces is adjacent
    dev_array<double> sc(16);
    double *grp_a = sc.p, *grp_b = sc.p + 3, *beta = sc.p + 6, *tau0 = sc.p + 7,
           *neg_one = sc.p + 9, *two = sc.p + 10, *tmp = sc.p + 11, *beta2 = sc.p + 12;
    sc.upload({0, 0, 0, 1, 0, 0, 0, 0, 1, -1, 2, 0, 0, 0, 0, 0});
    dev_array<uint8_t> stop(1);
    const size_t red_bytes = gkoc_reduction_workspace_bytes(n, 1, sizeof(double));
    const size_t x_bytes = gkoc_x_workspace_bytes(n, sizeof(double));
    dev_array<char> red_ws(red_bytes), x_ws(x_bytes);
    constexpr int NSLOT = 16;
    uint8_t* flags_host = nullptr;
    {
        void* q_ = nullptr;
        CK(gkoc_malloc_host(&q_, 2 * NSLOT));
        flags_host = static_cast<uint8_t*>(q_);
    }
    std::vector<gkoc_event_t> events(NSLOT);
    for (auto& e : events) CK(gkoc_event_create(&e));
    CK(gkoc_fill_array_f64(s, b.p, n, 1.0));
    CK(gkoc_fill_array_f64(s, x.p, n, 0.0));

    // where the host's time goes (printed with GKOC_EXAMPLE_TRACE=1)
    enum { T_ALLREDUCE, T_EXCHANGE, T_CHECK, T_COUNT };
    double spent[T_COUNT] = {};
    struct scoped {
        double& acc;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~scoped() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    };
    // ---- the distributed pieces
    auto all_reduce = [&](double* buf, int64_t cnt) {
        if (world == 1) return;
        scoped t{spent[T_ALLREDUCE]};
        CK(gkoc_comm_all_reduce_sum(comm, s, buf, cnt, sizeof(double)));
        if (mirror) CK(gkoc_dense_scale_f64(s, 1, cnt, two, 1, buf, cnt));
    };
    double* pending_reduce = nullptr;
    int64_t pending_cnt = 0;
    auto all_reduce_begin = [&](double* buf, int64_t cnt) {
        if (world == 1) return;
        scoped t{spent[T_ALLREDUCE]};
        CK(gkoc_comm_all_reduce_begin(comm, s, side, buf, cnt, sizeof(double)));
        pending_reduce = buf;
        pending_cnt = cnt;
    };
    auto all_reduce_end = [&]() {
        if (world == 1) return;
        scoped t{spent[T_ALLREDUCE]};
        CK(gkoc_comm_all_reduce_end(comm, s));
        if (mirror) CK(gkoc_dense_scale_f64(s, 1, pending_cnt, two, 1, pending_reduce, pending_cnt));
    };
    // y = A[owned rows, :] v   (distributed::Matrix::apply_impl, matrix.cpp:450-509)
    // dot != nullptr: also the LOCAL part of <v, y> (fused into the local SpMV, the boundary rows'
    // share next to their update)
    dev_array<double> nl_dot_ws(size_t((n_nl_rows + 63) / 64 + 1));
    auto dist_apply = [&](const double* v, double* y, double* dot = nullptr, bool started = false) {
        if (world > 1 && !started) {
            scoped t{spent[T_EXCHANGE]};
            CK(gkoc_comm_exchange_begin(comm, s, side, v, send_counts.data(), send_displs.data(),
                                        halo.p, recv_counts.data(), sizeof(double)));
        }
        if (dot) {
            CK(gkoc_x_csr_spmv_dot_f64_i32(s, n, local_ptrs.p, local_cols->p, local_vals->p, v, y, dot, x_ws.p,
                                           x_bytes));
        } else {
            CK(gkoc_csr_spmv_f64_i32(s, n, n, local_ptrs.p, local_cols->p, local_vals->p, v, 1, y, 1, 1));
        }
        if (world > 1) {
            {
                scoped t{spent[T_EXCHANGE]};
                CK(gkoc_comm_exchange_end(comm, s));
            }
            if (dot) {
                CK(gkoc_x_csr_rowlist_spmv_add_dot_f64_i32(s, n_nl_rows, nl_rows->p, nl_ptrs->p, nl_cols->p,
                                                           nl_vals->p, halo.p, y, v, dot, nl_dot_ws.p,
                                                           sizeof(double) * nl_dot_ws.n));
            } else {
                CK(gkoc_csr_rowlist_spmv_add_f64_i32(s, n_nl_rows, nl_rows->p, nl_ptrs->p, nl_cols->p,
                                                     nl_vals->p, halo.p, 1, y, 1, 1));
            }
        }
    };
    auto precond = [&](const double* src, double* dst) {
        CK(gkoc_jacobi_simple_apply_f64_i32(s, num_blocks, bs, scheme, block_ptrs.p, blocks.p, src, 1, dst, 1, 1));
    };
    auto local_dot = [&](const double* u, const double* v, double* out) {
        CK(gkoc_dense_compute_dot_f64(s, n, 1, u, 1, v, 1, out, red_ws.p, red_bytes));
    };
    auto local_sqnorm = [&](const double* u, double* out) {
        CK(gkoc_dense_compute_squared_norm2_f64(s, n, 1, u, 1, out, red_ws.p, red_bytes));
    };

    struct pending_check { int64_t it; int slot; };
    std::deque<pending_check> pending;
    int next_slot = 0;
    // criterion on the SQUARED norm (ImplicitResidualNorm's kernel: sqrt(tau) <= factor * tau0)
    auto check_begin = [&](int64_t iteration, const double* tau_sq) {
        scoped t{spent[T_CHECK]};
        const int slot = next_slot;
        next_slot = (next_slot + 1) % NSLOT;
        // the kernel writes its two flags straight into pinned host memory
        // (no event behind it: an event record is a barrier packet that idles the device for
        // 6-7 us; the host sets the slot to 0xFF and polls it when it wants the answer)
        volatile uint8_t* df = flags_host + 2 * slot;
        df[0] = df[1] = 0xFF;
        CK(gkoc_implicit_residual_norm_f64(s, 1, tau_sq, tau0, reduction, 2, 1, stop.p,
                                           const_cast<uint8_t*>(df), nullptr, nullptr));
        pending.push_back({iteration, slot});
    };
    double wait_seconds = 0;   // time the host spent WAITING for criterion flags (not enqueueing)
    auto drain = [&](int64_t upto, int64_t& stop_it) {
        while (!pending.empty() && pending.front().it <= upto) {
            const pending_check c = pending.front();
            pending.pop_front();
            const auto t0 = std::chrono::steady_clock::now();
            volatile uint8_t* df = flags_host + 2 * c.slot;
            for (long spins = 0; df[0] == 0xFF || df[1] == 0xFF; ++spins) {
                if (spins == 100000) CK(gkoc_stream_synchronize(s));   // visible at the latest then
                if (spins > 200000) throw std::runtime_error("criterion flags did not arrive");
            }
            wait_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (df[0] != 0) { stop_it = c.it; return true; }
        }
        return false;
    };

    CK(gkoc_device_synchronize());
    double host_seconds = 0;   // time spent enqueueing (excludes waiting for flags and the final sync)
    const auto t_start = std::chrono::steady_clock::now();
    // baseline: ||b|| (ResidualNorm with rhs_norm)
    local_sqnorm(b.p, tau0);
    all_reduce(tau0, 1);
    CK(gkoc_dense_compute_sqrt_f64(s, 1, tau0));
    double *cur = grp_a, *prev = grp_b;
    int64_t it = 0;
    if (!pipe) {
        // ------------------------------------------------------------------ CG (cg.cpp:93-181)
        CK(gkoc_cg_initialize_f64(s, n, 1, b.p, 1, r.p, 1, z.p, 1, p.p, 1, q.p, 1, prev, cur, stop.p));
        dist_apply(x.p, q.p);                                   // r = b - A x
        CK(gkoc_dense_add_scaled_f64(s, n, 1, neg_one, 1, q.p, 1, r.p, 1));
        CK(gkoc_fill_array_f64(s, q.p, n, 0.0));
        bool have_next = false;   // z, <r,z> and ||r||^2 of the coming iteration are already there
        it = -1;
        for (;;) {
            if (!have_next) {
                // z = M^-1 r, rho = <r,z> (one kernel) and ||r||^2
                CK(gkoc_x_jacobi_simple_apply_dot_f64_i32(s, num_blocks, n, bs, scheme, block_ptrs.p, blocks.p,
                                                          r.p, z.p, cur, x_ws.p, x_bytes));
                local_sqnorm(r.p, cur + 1);
            }
            all_reduce(cur, 2);                                 // one message: [<r,z>, ||r||^2]
            ++it;
            int64_t stop_it = it;
            if (it >= max_iters) {
                drain(it, stop_it);
                it = stop_it;
                break;
            }
            check_begin(it, cur + 1);
            if (drain(it - lag, stop_it)) { it = stop_it; break; }
            CK(gkoc_cg_step_1_f64(s, n, 1, p.p, 1, z.p, 1, cur, prev, stop.p));
            if (n >= (int64_t(1) << 22)) {
                dist_apply(p.p, q.p, beta);                     // q = A p and the local <p, q>
            } else {
                // small local parts: the plain SpMV and a separate dot are faster
                dist_apply(p.p, q.p);
                local_dot(p.p, q.p, beta);
            }
            all_reduce(beta, 1);
            // x += t p, r -= t q, z = M^-1 r_new in ONE kernel; <r,z> and ||r||^2 of the new
            // vectors go into the group that is `cur` next
            CK(gkoc_x_cg_step_2_jacobi_apply_f64_i32(s, num_blocks, n, bs, scheme, block_ptrs.p, blocks.p, x.p, r.p,
                                                     p.p, q.p, beta, cur, stop.p, z.p, prev, prev + 1, 0, x_ws.p,
                                                     x_bytes));
            have_next = true;
            std::swap(cur, prev);
        }
    } else {
        // ------------------------------------------------------------ PipeCg (pipe_cg.cpp:95-297)
        CK(gkoc_pipe_cg_initialize_1_f64(s, n, 1, b.p, 1, r.p, 1, prev, stop.p));   // r = b, prev_rho = 1
        dist_apply(x.p, q.p);
        CK(gkoc_dense_add_scaled_f64(s, n, 1, neg_one, 1, q.p, 1, r.p, 1));
        precond(r.p, z.p);
        dist_apply(z.p, w->p);
        precond(w->p, m->p);
        dist_apply(m->p, nn->p);
        local_dot(r.p, z.p, cur);
        local_dot(w->p, z.p, cur + 1);
        local_sqnorm(r.p, cur + 2);
        all_reduce(cur, 3);
        int64_t stop_it = 0;
        check_begin(0, cur + 2);
        if (!drain(0, stop_it)) {
            CK(gkoc_pipe_cg_initialize_2_f64(s, n, 1, p.p, 1, q.p, 1, f->p, 1, g->p, 1, beta, z.p, 1, w->p, 1,
                                             m->p, 1, nn->p, 1, cur + 1));
            // The first step_1 (+ its three sums) and m = M^-1 w stand alone; from then on step_2 of
            // iteration k, step_1 of k + 1, the sums and m = M^-1 w are ONE kernel, and the all-reduce
            // of the sums starts together with the halo exchange of n = A m behind one fork / join
            // (gkoc_comm_all_reduce_exchange_begin): both travel while the local SpMV runs.
            double* betas[2] = {beta, beta2};
            CK(gkoc_x_pipe_cg_step_1_dots_f64(s, n, x.p, r.p, z.p, w->p, p.p, q.p, f->p, g->p, cur, betas[0],
                                              stop.p, prev, x_ws.p, x_bytes));
            precond(w->p, m->p);
            for (;;) {
                const int parity = int(it & 1);
                if (world > 1) {
                    {
                        scoped t{spent[T_ALLREDUCE]};
                        CK(gkoc_comm_all_reduce_exchange_begin(comm, s, side, prev, 3, sizeof(double), m->p,
                                                               send_counts.data(), send_displs.data(), halo.p,
                                                               recv_counts.data(), sizeof(double)));
                    }
                    dist_apply(m->p, nn->p, nullptr, true);
                    if (mirror) CK(gkoc_dense_scale_f64(s, 1, 3, two, 1, prev, 3));
                } else {
                    dist_apply(m->p, nn->p);
                }
                std::swap(cur, prev);   // cur = the new [rho, delta, ||r||^2], prev = the old rho
                ++it;
                stop_it = it;
                if (it >= max_iters) {
                    drain(it, stop_it);
                    it = stop_it;
                    break;
                }
                check_begin(it, cur + 2);
                if (drain(it - lag, stop_it)) { it = stop_it; break; }
                CK(gkoc_x_pipe_cg_steps_jacobi_f64_i32(s, num_blocks, n, bs, scheme, block_ptrs.p, blocks.p, x.p,
                                                       r.p, z.p, w->p, p.p, q.p, f->p, g->p, m->p, nn->p, prev,
                                                       cur, cur + 1, betas[parity], betas[1 - parity], stop.p,
                                                       prev, x_ws.p, x_bytes, nullptr));
            }
        }
    }
    host_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() -
                   wait_seconds;
    CK(gkoc_device_synchronize());
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();

    // ---- true residual ||b - A x|| / ||b|| over all ranks, local checksum of x
    CK(gkoc_dense_copy_f64(s, n, 1, b.p, 1, r.p, 1));
    dist_apply(x.p, q.p);
    CK(gkoc_dense_add_scaled_f64(s, n, 1, neg_one, 1, q.p, 1, r.p, 1));
    local_sqnorm(r.p, tmp);
    all_reduce(tmp, 1);
    CK(gkoc_device_synchronize());
    const std::vector<double> hs = sc.download();
    const double tr = std::sqrt(hs[tmp - sc.p]), bn = hs[tau0 - sc.p];
    const std::vector<uint8_t> hstop = stop.download();
    const std::vector<double> hx = x.download();
    double xsum = 0;
    for (double v : hx) xsum += v;
    if (!dump.empty()) {
        const std::string path = dump + "." + std::to_string(rank);
        FILE* fo = fopen(path.c_str(), "wb");
        if (!fo || fwrite(hx.data(), sizeof(double), hx.size(), fo) != hx.size()) {
            throw std::runtime_error("cannot write " + path);
        }
        fclose(fo);
    }
    printf("{\"rank\": %d, \"world\": %d, \"mirror\": %s, \"grid\": %lld, \"n_local\": %lld, \"solver\": \"%s\", "
           "\"lag\": %d, \"iterations\": %lld, \"converged\": %s, \"true_rel_residual\": %.6e, "
           "\"x_sum_local\": %.17g, \"us_per_iteration\": %.2f, \"host_us_per_iteration\": %.2f}\n",
           rank, world, mirror ? "true" : "false", (long long)grid, (long long)n, pipe ? "pipe_cg" : "cg", lag,
           (long long)it, (hstop[0] & 0x80) ? "true" : "false", tr / bn, xsum,
           seconds * 1e6 / double(it > 0 ? it : 1), host_seconds * 1e6 / double(it > 0 ? it : 1));
    if (getenv("GKOC_EXAMPLE_TRACE")) {
        const double per = 1e6 / double(it > 0 ? it : 1);
        fprintf(stderr, "host us / iteration: all-reduce calls %.1f, exchange calls %.1f, criterion + flag copy %.1f, "
                "waiting for flags %.1f, everything else (kernel launches) %.1f\n",
                spent[T_ALLREDUCE] * per, spent[T_EXCHANGE] * per, spent[T_CHECK] * per, wait_seconds * per,
                (host_seconds - spent[T_ALLREDUCE] - spent[T_EXCHANGE] - spent[T_CHECK]) * per);
    }
    for (auto& e : events) gkoc_event_destroy(e);
    gkoc_free_host(flags_host);
    if (comm && use_ipc && !mirror) {
        // nobody unmaps its window while a peer may still write into it: a last all-reduce is behind every
        // rank's exchanges in stream order, and its own stores have arrived when it completes
        dev_array<double> last(1);
        CK(gkoc_comm_all_reduce_sum(comm, s, last.p, 1, sizeof(double)));
        CK(gkoc_stream_synchronize(s));
        uint32_t st = 0;
        CK(gkoc_comm_status(comm, &st));
        if (st != 0) throw std::runtime_error("the mailbox transport stopped waiting for a peer (status " + std::to_string(st) + ")");
    }
    if (comm) gkoc_comm_destroy(comm);
    return 0;
} catch (const std::exception& e) {
    fprintf(stderr, "native_dist_cg: %s\n", e.what());
    return 1;
}
