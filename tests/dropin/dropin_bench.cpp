// Throughput of the UNMODIFIED Ginkgo core on this backend, through Ginkgo's
// public API only: gko::matrix::Csr::apply and gko::solver::Cg with
// gko::preconditioner::Jacobi(8) on gko::HipExecutor (= shim + libgko_cdna4.so),
// timed with Ginkgo's own gko::Timer.  The 27-pt grid^3 matrix is written into
// Ginkgo's device arrays by the backend's generator (assembling 449 M entries
// through matrix_data on the host would take minutes); everything after that is
// Ginkgo code calling gko::kernels::hip::* symbols.
//   dropin_bench [grid=256] [reps=50] [cg_iters=100] [--json]
// --json: only Csr::apply and Cg + Jacobi(8), and one JSON object as the last line (bench.py's
// `ginkgo_api` object reads it)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>
#include <vector>

#include <ginkgo/core/base/array.hpp>
#include <ginkgo/core/base/executor.hpp>
#include <ginkgo/core/base/timer.hpp>
#include <ginkgo/core/log/convergence.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/matrix/ell.hpp>
#include <ginkgo/core/matrix/sellp.hpp>
#include <ginkgo/core/preconditioner/jacobi.hpp>
#include <ginkgo/core/solver/cg.hpp>
#include <ginkgo/core/solver/gmres.hpp>
#include <ginkgo/core/stop/combined.hpp>
#include <ginkgo/core/stop/iteration.hpp>
#include <ginkgo/core/stop/residual_norm.hpp>

#include "gko_cdna4.h"

using vt = double;
using it = gko::int32;
using Csr = gko::matrix::Csr<vt, it>;
using Dense = gko::matrix::Dense<vt>;

template <typename F>
static double time_ms(std::shared_ptr<const gko::Executor> exec, int reps, F f, int warm = 2)
{
    auto timer = gko::Timer::create_for_executor(exec);
    for (int i = 0; i < warm; ++i) f();
    exec->synchronize();
    auto t0 = timer->create_time_point();
    auto t1 = timer->create_time_point();
    timer->record(t0);
    for (int i = 0; i < reps; ++i) f();
    timer->record(t1);
    timer->wait(t1);
    return std::chrono::duration<double, std::milli>(timer->difference_async(t0, t1)).count() / reps;
}

int main(int argc, char** argv)
{
    const gko::int64 grid = argc > 1 ? std::atoll(argv[1]) : 256;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 50;
    const int cg_iters = argc > 3 ? std::atoi(argv[3]) : 100;
    const bool json = argc > 4 && std::strcmp(argv[4], "--json") == 0;
    auto ref = gko::ReferenceExecutor::create();
    auto hip = gko::HipExecutor::create(0, ref);
    const gko::size_type n = grid * grid * grid;
    std::cout << hip->get_description() << std::endl;

    gko::array<it> row_ptrs(hip, n + 1);
    int64_t nnz = 0;
    if (gkoc_stencil_row_ptrs_i32(hip->get_stream(), 3, grid, 0, 0, grid, row_ptrs.get_data(), &nnz)) return 1;
    gko::array<it> cols(hip, nnz);
    gko::array<vt> vals(hip, nnz);
    if (gkoc_stencil_fill_f64_i32(hip->get_stream(), 3, grid, 0, 0, grid, row_ptrs.get_const_data(),
                                  cols.get_data(), vals.get_data())) return 1;
    auto a = gko::share(Csr::create(hip, gko::dim<2>{n, n}, std::move(vals), std::move(cols),
                                    std::move(row_ptrs)));
    std::cout << "27-pt " << grid << "^3: n = " << n << ", nnz = " << a->get_num_stored_elements()
              << ", strategy " << a->get_strategy()->get_name() << std::endl;

    auto b_host = Dense::create(ref, gko::dim<2>{n, 1});
    std::mt19937_64 rng(42);
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    for (gko::size_type i = 0; i < n; ++i) b_host->at(i, 0) = dist(rng);
    auto b = gko::clone(hip, b_host);
    auto x = Dense::create(hip, gko::dim<2>{n, 1});

    int cv = -1, cc = -1, cr = -1, cb = -1, cx = -1;
    {
        // where the backend's allocator (HipAllocator -> csrc/arena.hip) put Ginkgo's arrays
        gkoc_arena_class_of(a->get_const_values(), &cv);
        gkoc_arena_class_of(a->get_const_col_idxs(), &cc);
        gkoc_arena_class_of(a->get_const_row_ptrs(), &cr);
        gkoc_arena_class_of(b->get_const_values(), &cb);
        gkoc_arena_class_of(x->get_const_values(), &cx);
        std::printf("memory classes: values %d, col_idxs %d, row_ptrs %d, b %d, x %d\n", cv, cc, cr, cb, cx);
    }
    const double bytes = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    // 30 untimed applies first: an idle MI355X needs ~20 ms of load to reach its clocks (the first fifteen
    // launches of a fresh process take 1.06 ms, the rest 0.97: profiles/r05_ginkgo_api_timeline.txt) - the
    // same warm-up bench.py's native line gets
    double ms = time_ms(hip, reps, [&] { a->apply(b, x); }, 30);
    const double csr_ms = ms;
    std::printf("gko::matrix::Csr::apply      %8.4f ms  %8.1f GB/s  (%.1f %% of 8 TB/s)\n", ms,
                bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
    if (!json) {
        auto ell = gko::matrix::Ell<vt, it>::create(hip);
        double conv = time_ms(hip, 2, [&] { a->convert_to(ell); });
        ms = time_ms(hip, reps, [&] { ell->apply(b, x); });
        const double eb = 12.0 * ell->get_num_stored_elements() + 16.0 * n;
        std::printf("gko::matrix::Ell::apply      %8.4f ms  %8.1f GB/s  (%.1f %%)   [Csr->Ell convert_to %.2f ms]\n", ms,
                    eb / ms / 1e6, eb / ms / 1e6 / 80.0, conv);
    }
    if (!json) {
        auto sp = gko::matrix::Sellp<vt, it>::create(hip);
        double conv = time_ms(hip, 2, [&] { a->convert_to(sp); });
        ms = time_ms(hip, reps, [&] { sp->apply(b, x); });
        const double sb = 12.0 * sp->get_num_stored_elements() + 16.0 * n;
        std::printf("gko::matrix::Sellp::apply    %8.4f ms  %8.1f GB/s  (%.1f %%)   [Csr->Sellp convert_to %.2f ms]\n", ms,
                    sb / ms / 1e6, sb / ms / 1e6 / 80.0, conv);
    }

    // CG + block-Jacobi(8), fixed iteration count (configs[2] without convergence effects)
    auto t_gen = std::chrono::steady_clock::now();
    auto solver = gko::solver::Cg<vt>::build()
                      .with_criteria(gko::stop::Iteration::build().with_max_iters(cg_iters),
                                     gko::stop::ResidualNorm<vt>::build().with_reduction_factor(1e-30))
                      .with_preconditioner(gko::preconditioner::Jacobi<vt, it>::build().with_max_block_size(8u))
                      .on(hip)
                      ->generate(a);
    hip->synchronize();
    const double gen_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_gen).count();
    auto logger = gko::share(gko::log::Convergence<vt>::create());
    solver->add_logger(logger);
    auto rhs = Dense::create(hip, gko::dim<2>{n, 1});
    rhs->fill(1.0);
    x->fill(0.0);
    solver->apply(rhs, x);   // warm-up
    x->fill(0.0);
    hip->synchronize();
    auto t0 = std::chrono::steady_clock::now();
    solver->apply(rhs, x);
    hip->synchronize();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const auto iters = logger->get_num_iterations();
    std::printf("gko::solver::Cg + Jacobi(8)  %lu iterations, %8.4f ms/iteration, %8.1f it/s   [generate %.3f s]\n",
                static_cast<unsigned long>(iters), s * 1e3 / iters, iters / s, gen_s);
    if (json) {
        int64_t fused = 0;
        gkoc_tune_get(GKOC_TUNE_DEFERRED_FUSION, &fused);
        std::printf("{\"n\": %lu, \"nnz\": %lld, \"csr_apply_ms\": %.5f, \"cg_iterations\": %lu, "
                    "\"cg_ms_per_iter\": %.5f, \"cg_iters_per_s\": %.2f, \"fused_across_calls\": %d, "
                    "\"memory_classes\": {\"values\": %d, \"col_idxs\": %d, \"row_ptrs\": %d, \"b\": %d, "
                    "\"x\": %d}}\n",
                    static_cast<unsigned long>(n), static_cast<long long>(nnz), csr_ms,
                    static_cast<unsigned long>(iters), s * 1e3 / iters, iters / s, int(fused), cv, cc, cr, cb,
                    cx);
        return 0;
    }
    {
        // GMRES(30) + block-Jacobi(8), two restart cycles (modified Gram-Schmidt, Ginkgo's default)
        auto gm = gko::solver::Gmres<vt>::build()
                      .with_krylov_dim(30u)
                      .with_criteria(gko::stop::Iteration::build().with_max_iters(60u),
                                     gko::stop::ResidualNorm<vt>::build().with_reduction_factor(1e-30))
                      .with_preconditioner(gko::preconditioner::Jacobi<vt, it>::build().with_max_block_size(8u))
                      .on(hip)
                      ->generate(a);
        auto glog = gko::share(gko::log::Convergence<vt>::create());
        gm->add_logger(glog);
        x->fill(0.0);
        gm->apply(rhs, x);   // warm-up
        x->fill(0.0);
        hip->synchronize();
        auto g0 = std::chrono::steady_clock::now();
        gm->apply(rhs, x);
        hip->synchronize();
        const double gs = std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
        const auto gi = glog->get_num_iterations();
        std::printf("gko::solver::Gmres(30) + Jacobi(8)  %lu iterations, %8.4f ms/iteration, %8.1f it/s\n",
                    static_cast<unsigned long>(gi), gs * 1e3 / gi, gi / gs);
    }
    return 0;
}
