// Where does the backend's allocator put Ginkgo's arrays when NOTHING states their role?
// gko::HipExecutor::raw_alloc has no role argument; csrc/arena.hip places by what it can see (sizes
// of the live allocations, vectors the kernels have been seen to write).  Three allocation orders
// through the UNMODIFIED Ginkgo API, each printing the memory class of every array and the time of
// the kernel the placement matters for; `ok` lines are what tests/test_arena_roles_gpu.py asserts:
//   vectors-first   b and x are allocated BEFORE the matrix (the first allocation of a process has
//                   nothing to be compared with)
//   gmres-basis     Gmres(30): the Krylov basis (31 n values = 4.2 GB at 256^3) is larger than the
//                   matrix' values, yet every iteration WRITES it (the SpMV's output is a row block
//                   of it)
//   five-point      5-pt 2-D Laplacian: 5 nonzeros per row, vectors are a fifth of the values
//   arena_roles_test <scenario> [grid]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <vector>

#include <ginkgo/core/base/array.hpp>
#include <ginkgo/core/base/executor.hpp>
#include <ginkgo/core/base/timer.hpp>
#include <ginkgo/core/log/logger.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/preconditioner/jacobi.hpp>
#include <ginkgo/core/solver/gmres.hpp>
#include <ginkgo/core/stop/iteration.hpp>

#include "gko_cdna4.h"

using vt = double;
using it = gko::int32;
using Csr = gko::matrix::Csr<vt, it>;
using Dense = gko::matrix::Dense<vt>;

static int class_of(const void* p)
{
    int c = -1;
    gkoc_arena_class_of(p, &c);
    return c;
}

template <typename F>
static double time_ms(std::shared_ptr<const gko::Executor> exec, int reps, F f)
{
    auto timer = gko::Timer::create_for_executor(exec);
    f();
    f();
    exec->synchronize();
    auto t0 = timer->create_time_point();
    auto t1 = timer->create_time_point();
    timer->record(t0);
    for (int i = 0; i < reps; ++i) f();
    timer->record(t1);
    timer->wait(t1);
    return std::chrono::duration<double, std::milli>(timer->difference_async(t0, t1)).count() / reps;
}

// every allocation of the executor, as Ginkgo reports it
struct alloc_log : gko::log::Logger {
    mutable std::mutex m;
    mutable std::vector<std::pair<gko::uintptr, gko::size_type>> seen;
    void on_allocation_completed(const gko::Executor*, const gko::size_type& bytes,
                                 const gko::uintptr& location) const override
    {
        std::lock_guard<std::mutex> g(m);
        seen.emplace_back(location, bytes);
    }
    alloc_log() : gko::log::Logger(gko::log::Logger::allocation_completed_mask) {}
};

static std::shared_ptr<Csr> stencil(std::shared_ptr<const gko::HipExecutor> hip, int nd, gko::int64 grid)
{
    gko::size_type n = 1;
    for (int d = 0; d < nd; ++d) n *= grid;
    gko::array<it> row_ptrs(hip, n + 1);
    int64_t nnz = 0;
    if (gkoc_stencil_row_ptrs_i32(hip->get_stream(), nd, grid, 0, 0, grid, row_ptrs.get_data(), &nnz)) std::exit(2);
    gko::array<it> cols(hip, nnz);
    gko::array<vt> vals(hip, nnz);
    if (gkoc_stencil_fill_f64_i32(hip->get_stream(), nd, grid, 0, 0, grid, row_ptrs.get_const_data(),
                                  cols.get_data(), vals.get_data())) {
        std::exit(2);
    }
    return gko::share(Csr::create(hip, gko::dim<2>{n, n}, std::move(vals), std::move(cols), std::move(row_ptrs)));
}

int main(int argc, char** argv)
{
    const std::string scenario = argc > 1 ? argv[1] : "vectors-first";
    const gko::int64 grid = argc > 2 ? std::atoll(argv[2]) : (scenario == "five-point" ? 4096 : 256);
    auto ref = gko::ReferenceExecutor::create();
    auto hip = gko::HipExecutor::create(0, ref);
    const int nd = scenario == "five-point" ? 2 : 3;
    gko::size_type n = 1;
    for (int d = 0; d < nd; ++d) n *= grid;
    std::shared_ptr<Csr> a;
    std::shared_ptr<Dense> b, x;
    if (scenario == "vectors-first") {
        b = gko::share(Dense::create(hip, gko::dim<2>{n, 1}));
        x = gko::share(Dense::create(hip, gko::dim<2>{n, 1}));
        a = stencil(hip, nd, grid);
    } else {
        a = stencil(hip, nd, grid);
        b = gko::share(Dense::create(hip, gko::dim<2>{n, 1}));
        x = gko::share(Dense::create(hip, gko::dim<2>{n, 1}));
    }
    b->fill(1.0);
    x->fill(0.0);
    const int cv = class_of(a->get_const_values()), cc = class_of(a->get_const_col_idxs()),
              cr = class_of(a->get_const_row_ptrs()), cb = class_of(b->get_const_values()),
              cx = class_of(x->get_const_values());
    std::printf("%s, %s stencil %ld^%d: n = %lu, nnz = %lu\n", scenario.c_str(), nd == 2 ? "5-pt" : "27-pt",
                long(grid), nd, static_cast<unsigned long>(n),
                static_cast<unsigned long>(a->get_num_stored_elements()));
    std::printf("memory classes: values %d, col_idxs %d, row_ptrs %d, b %d, x %d\n", cv, cc, cr, cb, cx);
    const bool arena_on = cv >= 0;
    bool ok = !arena_on || (cx != cv && cx != cc);
    std::printf("%s: the SpMV's output shares no class with values / col_idxs\n", ok ? "ok" : "MISPLACED");
    const double bytes = 12.0 * a->get_num_stored_elements() + 4.0 * (n + 1) + 16.0 * n;
    const double ms = time_ms(hip, 30, [&] { a->apply(b, x); });
    std::printf("gko::matrix::Csr::apply %8.4f ms  %8.1f GB/s\n", ms, bytes / ms / 1e6);
    int failures = ok ? 0 : 1;

    if (scenario == "gmres-basis") {
        auto log = std::make_shared<alloc_log>();
        hip->add_logger(log);
        auto gm = gko::solver::Gmres<vt>::build()
                      .with_krylov_dim(30u)
                      .with_criteria(gko::stop::Iteration::build().with_max_iters(30u))
                      .with_preconditioner(gko::preconditioner::Jacobi<vt, it>::build().with_max_block_size(8u))
                      .on(hip)
                      ->generate(a);
        x->fill(0.0);
        gm->apply(b, x);   // allocates the workspace
        hip->synchronize();
        hip->remove_logger(log);
        const gko::size_type basis_bytes = 31 * n * sizeof(vt);
        int cbasis = -2;
        for (auto& s : log->seen) {
            if (s.second == basis_bytes) cbasis = class_of(reinterpret_cast<const void*>(s.first));
        }
        std::printf("Krylov basis (%.2f GB): class %d\n", basis_bytes / 1e9, cbasis);
        ok = !arena_on || (cbasis >= 0 && cbasis != cv && cbasis != cc);
        std::printf("%s: the Krylov basis (written by every SpMV) shares no class with values / col_idxs\n",
                    ok ? "ok" : "MISPLACED");
        failures += ok ? 0 : 1;
        x->fill(0.0);
        hip->synchronize();
        auto t0 = std::chrono::steady_clock::now();
        gm->apply(b, x);
        hip->synchronize();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("gko::solver::Gmres(30) + Jacobi(8): 30 iterations, %8.4f ms/iteration\n", s * 1e3 / 30);
    }
    int64_t nvec = 0, misplaced = 0;
    gkoc_arena_role_stats(&nvec, &misplaced);
    std::printf("allocator: %ld vector sizes learnt from kernels, %ld of them found next to matrix arrays\n",
                long(nvec), long(misplaced));
    return failures;
}
