// SPDX-License-Identifier: BSD-3-Clause
// CG + block-Jacobi on the 27-pt Laplacian, driven from C++ through the C ABI of
// include/gko_cdna4.h only (no Ginkgo, no Python, no HIP headers): the host-side
// loop of core/solver/cg.cpp:93-181 with
//   * the reference kernel sequence (mode "plain"), or
//   * the fused producer+reduction kernels and the asynchronous criterion check
//     (mode "fused", default): the criterion kernel of iteration k is read
//     `lag` iterations later from pinned memory; cg::step_1/step_2 are masked by
//     stop_status, so the result is bit-identical to lag 0;
//   * the same with two consecutive iterations captured as a hipGraph and
//     replayed (mode "graph"): for launch-bound system sizes.
// Build (g++ is enough, the device code lives in libgko_cdna4.so):
//   g++ -O2 -std=c++17 -Iinclude examples/native_cg.cpp -Lginkgo_amd/lib \
//       -lgko_cdna4 -Wl,-rpath,'$ORIGIN/../ginkgo_amd/lib' -o examples/native_cg
// Run:  examples/native_cg [grid=64] [max_iters=1000] [reduction=1e-10] [plain|fused|graph] [lag=4]
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "gko_cdna4.h"

#define CK(call)                                                              \
    do {                                                                      \
        if (int rc_ = (call)) {                                               \
            throw std::runtime_error(std::string(#call) + " -> " +            \
                                     std::to_string(rc_) + ": " +             \
                                     gkoc_last_error());                      \
        }                                                                     \
    } while (0)

template <typename T>
struct dev_array {
    T* p = nullptr;
    size_t n = 0;
    explicit dev_array(size_t count) : n(count)
    {
        void* q = nullptr;
        CK(gkoc_malloc(&q, sizeof(T) * (count ? count : 1)));
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

int main(int argc, char** argv)
try {
    const int64_t grid = argc > 1 ? atoll(argv[1]) : 64;
    const int64_t max_iters = argc > 2 ? atoll(argv[2]) : 1000;
    const double reduction = argc > 3 ? atof(argv[3]) : 1e-10;
    const bool fused = !(argc > 4 && !strcmp(argv[4], "plain"));
    const bool use_graph = argc > 4 && !strcmp(argv[4], "graph");
    int lag = fused ? (argc > 5 ? atoi(argv[5]) : 4) : 0;
    if (use_graph && lag < 1) lag = 1;
    const uint32_t bs = 8;
    const int64_t n = grid * grid * grid;
    gkoc_stream_t s = nullptr;   // default stream; graph capture needs an explicit one
    if (use_graph) CK(gkoc_stream_create(&s));

    // ---- system matrix (benchmark/utils/stencil_matrix.hpp semantics), on the device
    dev_array<int32_t> row_ptrs(n + 1);
    int64_t nnz = 0;
    CK(gkoc_stencil_row_ptrs_i32(s, 3, grid, 0, 0, grid, row_ptrs.p, &nnz));
    dev_array<int32_t> cols(nnz);
    dev_array<double> vals(nnz);
    CK(gkoc_stencil_fill_f64_i32(s, 3, grid, 0, 0, grid, row_ptrs.p, cols.p, vals.p));

    // ---- block-Jacobi(8): find_blocks + generate (jacobi.cpp:328-404)
    dev_array<int32_t> block_ptrs(n + 1);
    int64_t num_blocks = 0;
    CK(gkoc_jacobi_find_blocks_f64_i32(s, n, row_ptrs.p, cols.p, bs, &num_blocks, block_ptrs.p));
    const gkoc_jacobi_scheme scheme = storage_scheme(bs);
    const int64_t gsize = int64_t(1) << scheme.group_power;
    dev_array<double> blocks(size_t((num_blocks + gsize - 1) / gsize * scheme.group_offset));
    CK(gkoc_memset(blocks.p, 0, sizeof(double) * blocks.n, s));
    CK(gkoc_jacobi_generate_f64_i32(s, n, row_ptrs.p, cols.p, vals.p, num_blocks, bs, scheme,
                                    block_ptrs.p, blocks.p, nullptr));

    // ---- vectors, device-resident scalars, workspaces
    dev_array<double> b(n), x(n), r(n), z(n), p(n), q(n);
    dev_array<double> sc(8);   // rho, prev_rho, beta, tau, tau0, one, neg_one
    double *rho = sc.p, *prev_rho = sc.p + 1, *beta = sc.p + 2, *tau = sc.p + 3,
           *tau0 = sc.p + 4, *one = sc.p + 5, *neg_one = sc.p + 6;
    sc.upload({0, 1, 0, 0, 0, 1, -1, 0});
    dev_array<uint8_t> stop(1);
    const size_t red_bytes = gkoc_reduction_workspace_bytes(n, 1, sizeof(double));
    const size_t x_bytes = gkoc_x_workspace_bytes(n, sizeof(double));
    dev_array<char> red_ws(red_bytes), x_ws(x_bytes);
    constexpr int NSLOT = 16;
    dev_array<uint8_t> flags_dev(2 * NSLOT);
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
    CK(gkoc_device_synchronize());

    const auto t_start = std::chrono::steady_clock::now();
    // r = b, z = p = q = 0, rho = 0, prev_rho = 1, stop.reset()   (cg::initialize)
    CK(gkoc_cg_initialize_f64(s, n, 1, b.p, 1, r.p, 1, z.p, 1, p.p, 1, q.p, 1, prev_rho, rho, stop.p));
    // r = b - A x
    CK(gkoc_csr_advanced_spmv_f64_i32(s, n, n, neg_one, row_ptrs.p, cols.p, vals.p, x.p, 1, one, r.p, 1, 1));
    CK(gkoc_dense_compute_norm2_f64(s, n, 1, b.p, 1, tau0, red_ws.p, red_bytes));  // rhs_norm baseline

    struct pending_check { int64_t it; int slot; };
    std::deque<pending_check> pending;
    int next_slot = 0;
    bool have_tau = false;
    int64_t it = -1;
    auto check_done = [&](const pending_check& c) {
        CK(gkoc_event_synchronize(events[c.slot]));
        return flags_host[2 * c.slot] != 0;
    };
    auto precond_and_rho = [&](double* rho_) {
        if (fused) {
            CK(gkoc_x_jacobi_simple_apply_dot_f64_i32(s, num_blocks, n, bs, scheme, block_ptrs.p, blocks.p,
                                                      r.p, z.p, rho_, x_ws.p, x_bytes));
        } else {
            CK(gkoc_jacobi_simple_apply_f64_i32(s, num_blocks, bs, scheme, block_ptrs.p, blocks.p, r.p, 1, z.p, 1, 1));
            CK(gkoc_dense_compute_dot_f64(s, n, 1, r.p, 1, z.p, 1, rho_, red_ws.p, red_bytes));
        }
    };
    auto update = [&](double* rho_, double* prev_rho_) {
        CK(gkoc_cg_step_1_f64(s, n, 1, p.p, 1, z.p, 1, rho_, prev_rho_, stop.p));
        CK(gkoc_csr_spmv_f64_i32(s, n, n, row_ptrs.p, cols.p, vals.p, p.p, 1, q.p, 1, 1));
        CK(gkoc_dense_compute_dot_f64(s, n, 1, p.p, 1, q.p, 1, beta, red_ws.p, red_bytes));
        if (fused) {
            CK(gkoc_x_cg_step_2_norm_f64(s, n, x.p, r.p, p.p, q.p, beta, rho_, stop.p, tau, 1, x_ws.p, x_bytes));
        } else {
            CK(gkoc_cg_step_2_f64(s, n, 1, x.p, 1, r.p, 1, p.p, 1, q.p, 1, beta, rho_, stop.p));
        }
    };
    // enqueue the copy of a criterion kernel's flags into pinned memory + its event
    auto publish = [&](int64_t iteration, const uint8_t* dev_flags) {
        const int slot = next_slot;
        next_slot = (next_slot + 1) % NSLOT;
        CK(gkoc_memcpy_d2h(flags_host + 2 * slot, dev_flags, 2, s));
        CK(gkoc_event_record(events[slot], s));
        pending.push_back({iteration, slot});
    };
    auto drain = [&](int64_t upto, int64_t& stop_it) {
        while (!pending.empty() && pending.front().it <= upto) {
            const pending_check c = pending.front();
            pending.pop_front();
            if (check_done(c)) { stop_it = c.it; return true; }
        }
        return false;
    };
    gkoc_graph_t graph = nullptr;
    dev_array<uint8_t> gflags(4);    // flags of the two captured iterations
    for (;;) {
        if (use_graph && have_tau && it + 2 < max_iters) {
            if (!graph) {
                // two iterations: the roles of rho / prev_rho are back after two swaps
                CK(gkoc_stream_begin_capture(s));
                double *rk = rho, *pk = prev_rho;
                for (int k = 0; k < 2; ++k) {
                    precond_and_rho(rk);
                    CK(gkoc_residual_norm_f64(s, 1, tau, tau0, reduction, 2, 1, stop.p, gflags.p + 2 * k,
                                              nullptr, nullptr));
                    update(rk, pk);
                    std::swap(rk, pk);
                }
                CK(gkoc_stream_end_capture(s, &graph));
            }
            CK(gkoc_graph_launch(graph, s));
            publish(it + 1, gflags.p);
            publish(it + 2, gflags.p + 2);
            it += 2;
            int64_t stop_it = it;
            if (drain(it - lag, stop_it)) { it = stop_it; break; }
            continue;
        }
        precond_and_rho(rho);
        ++it;
        bool stopped = false;
        int64_t stop_it = it;
        if (it >= max_iters) {                                  // stop::Iteration
            stopped = true;
            drain(it, stop_it);
            pending.clear();
        } else {
            if (!have_tau) CK(gkoc_dense_compute_norm2_f64(s, n, 1, r.p, 1, tau, red_ws.p, red_bytes));
            if (lag == 0) {                                     // lock step, as the reference
                int allc = 0, chg = 0;
                CK(gkoc_residual_norm_f64(s, 1, tau, tau0, reduction, 2, 1, stop.p, flags_dev.p, &allc, &chg));
                stopped = allc != 0;
            } else {                                            // stop::ResidualNorm, read `lag` iterations later
                uint8_t* df = flags_dev.p + 2 * next_slot;
                CK(gkoc_residual_norm_f64(s, 1, tau, tau0, reduction, 2, 1, stop.p, df, nullptr, nullptr));
                publish(it, df);
                stopped = drain(it - lag, stop_it);
            }
        }
        if (stopped) { it = stop_it; break; }
        update(rho, prev_rho);
        have_tau = fused;
        std::swap(rho, prev_rho);
    }
    if (graph) CK(gkoc_graph_destroy(graph));
    CK(gkoc_device_synchronize());
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();

    // true residual ||b - A x|| / ||b|| with the plain kernels
    CK(gkoc_dense_copy_f64(s, n, 1, b.p, 1, r.p, 1));
    CK(gkoc_csr_advanced_spmv_f64_i32(s, n, n, neg_one, row_ptrs.p, cols.p, vals.p, x.p, 1, one, r.p, 1, 1));
    CK(gkoc_dense_compute_norm2_f64(s, n, 1, r.p, 1, tau, red_ws.p, red_bytes));
    CK(gkoc_device_synchronize());   // the downloads below use the NULL stream
    const std::vector<double> hs = sc.download();
    const double tr = hs[tau - sc.p], bn = hs[tau0 - sc.p];
    const std::vector<uint8_t> hstop = stop.download();
    double xsum = 0;
    {
        const std::vector<double> hx = x.download();
        for (double v : hx) xsum += v;
    }
    printf("{\"grid\": %lld, \"n\": %lld, \"nnz\": %lld, \"mode\": \"%s\", \"lag\": %d, \"iterations\": %lld, "
           "\"converged\": %s, \"true_rel_residual\": %.6e, \"x_sum\": %.17g, \"us_per_iteration\": %.2f}\n",
           (long long)grid, (long long)n, (long long)nnz, use_graph ? "graph" : (fused ? "fused" : "plain"), lag, (long long)it,
           (hstop[0] & 0x80) ? "true" : "false", tr / bn, xsum, seconds * 1e6 / double(it > 0 ? it : 1));
    for (auto& e : events) gkoc_event_destroy(e);
    gkoc_free_host(flags_host);
    return 0;
} catch (const std::exception& e) {
    fprintf(stderr, "native_cg: %s\n", e.what());
    return 1;
}
