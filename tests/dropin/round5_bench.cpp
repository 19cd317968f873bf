// Timings of what round 5 added and never timed (missing 5 / next 7), through Ginkgo's public
// API on gko::HipExecutor (the drop-in backend), 27-pt grid^3:
//   * gko::solver::CbGmres<complex<double>> with the Krylov basis kept (keep) or stored as complex<float>
//     (reduce1), next to CbGmres<double> (the tuned real path) - per-iteration time of one restart cycle;
//   * gko::preconditioner::Jacobi(8) apply with adaptive (autodetected) storage on a float matrix and with full
//     storage on a complex<double> matrix (the generic adaptive kernels, csrc/jacobi.hip's last part) next to the
//     double fast path.
// Reference for what is replaced: common/cuda_hip/solver/cb_gmres_kernels.cpp:723-1060,
// common/cuda_hip/preconditioner/jacobi_generate_kernels.instantiate.cpp:80-276.
//   round5_bench [grid=256] [krylov_dim=30] [what = all | cbd-keep | cbd-reduce1 | cbc | jacobi]
#include <chrono>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include <ginkgo/core/base/array.hpp>
#include <ginkgo/core/base/executor.hpp>
#include <ginkgo/core/base/timer.hpp>
#include <ginkgo/core/log/convergence.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/preconditioner/jacobi.hpp>
#include <ginkgo/core/solver/cb_gmres.hpp>
#include <ginkgo/core/stop/combined.hpp>
#include <ginkgo/core/stop/iteration.hpp>
#include <ginkgo/core/stop/residual_norm.hpp>

#include "gko_cdna4.h"

using it = gko::int32;
using cd = std::complex<double>;

template <typename F>
static double time_ms(std::shared_ptr<const gko::Executor> exec, int reps, F f, int warm = 3)
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

template <typename V>
static void cb_gmres(const char* name, std::shared_ptr<const gko::Executor> hip,
                     std::shared_ptr<gko::matrix::Csr<V, it>> a, gko::solver::cb_gmres::storage_precision sp,
                     const char* sp_name, unsigned kd, double basis_bytes_per_value, double mat_bytes)
{
    using Dense = gko::matrix::Dense<V>;
    const auto n = a->get_size()[0];
    auto solver = gko::solver::CbGmres<V>::build()
                      .with_krylov_dim(kd)
                      .with_storage_precision(sp)
                      .with_criteria(gko::stop::Iteration::build().with_max_iters(kd),
                                     gko::stop::ResidualNorm<V>::build().with_reduction_factor(
                                         gko::remove_complex<V>(1e-30)))
                      .on(hip)
                      ->generate(a);
    auto rhs = Dense::create(hip, gko::dim<2>{n, 1});
    rhs->fill(V(1.0));
    auto x = Dense::create(hip, gko::dim<2>{n, 1});
    x->fill(V(0.0));
    solver->apply(rhs, x);   // warm-up (allocates the basis)
    x->fill(V(0.0));
    hip->synchronize();
    auto t0 = std::chrono::steady_clock::now();
    solver->apply(rhs, x);
    hip->synchronize();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // bytes of an average iteration of the cycle (classical Gram-Schmidt, no re-orthogonalisation): the product
    // (matrix + 2 vectors), one pass over next_krylov and k + 1 basis vectors for the dots, one for the
    // update, the new basis vector written: k + 1 averaged over k = 0 .. kd - 1 is (kd + 1) / 2
    const double vb = sizeof(V);
    const double per_iter = mat_bytes + 2 * vb * n + 2 * ((kd + 1) / 2.0) * basis_bytes_per_value * n + 4 * vb * n +
                            basis_bytes_per_value * n;
    std::printf("%-34s %-8s %2u iterations  %8.3f ms/iteration  %6.1f %% of 8 TB/s (%.2f GB / iteration)\n", name,
                sp_name, kd, s * 1e3 / kd, per_iter / (s / kd) / 8e12 * 100, per_iter / 1e9);
}

int main(int argc, char** argv)
{
    const gko::int64 grid = argc > 1 ? std::atoll(argv[1]) : 256;
    const unsigned kd = argc > 2 ? unsigned(std::atoi(argv[2])) : 30u;
    const std::string what = argc > 3 ? argv[3] : "all";
    auto want = [&](const char* w) { return what == "all" || what == w; };
    auto ref = gko::ReferenceExecutor::create();
    auto hip = gko::HipExecutor::create(0, ref);
    const gko::size_type n = grid * grid * grid;
    gko::array<it> row_ptrs(hip, n + 1);
    int64_t nnz = 0;
    if (gkoc_stencil_row_ptrs_i32(hip->get_stream(), 3, grid, 0, 0, grid, row_ptrs.get_data(), &nnz)) return 1;
    gko::array<it> cols(hip, nnz);
    gko::array<double> vals(hip, nnz);
    if (gkoc_stencil_fill_f64_i32(hip->get_stream(), 3, grid, 0, 0, grid, row_ptrs.get_const_data(),
                                  cols.get_data(), vals.get_data())) return 1;
    std::printf("27-pt %ld^3: n = %lu, nnz = %lld\n", long(grid), (unsigned long)n, (long long)nnz);
    // the same matrix with complex<double> and float values (values converted on the device)
    gko::array<cd> vals_c(hip, nnz);
    gko::array<float> vals_f(hip, nnz);
    {
        auto real = gko::matrix::Dense<double>::create(hip, gko::dim<2>{gko::size_type(nnz), 1},
                                                       gko::make_array_view(hip, nnz, vals.get_data()), 1);
        auto cplx = gko::matrix::Dense<cd>::create(hip, gko::dim<2>{gko::size_type(nnz), 1},
                                                   gko::make_array_view(hip, nnz, vals_c.get_data()), 1);
        real->make_complex(cplx);
        auto flt = gko::matrix::Dense<float>::create(hip, gko::dim<2>{gko::size_type(nnz), 1},
                                                     gko::make_array_view(hip, nnz, vals_f.get_data()), 1);
        real->convert_to(flt);
    }
    auto a_d = gko::share(gko::matrix::Csr<double, it>::create(hip, gko::dim<2>{n, n}, vals, cols, row_ptrs));
    auto a_c = gko::share(gko::matrix::Csr<cd, it>::create(hip, gko::dim<2>{n, n}, std::move(vals_c), cols, row_ptrs));
    auto a_f = gko::share(gko::matrix::Csr<float, it>::create(hip, gko::dim<2>{n, n}, std::move(vals_f), cols, row_ptrs));
    hip->synchronize();

    using sp = gko::solver::cb_gmres::storage_precision;
    const double mat_d = 12.0 * nnz + 4.0 * (n + 1), mat_c = 20.0 * nnz + 4.0 * (n + 1);
    if (want("cbd-keep")) cb_gmres<double>("CbGmres<double>", hip, a_d, sp::keep, "keep", kd, 8, mat_d);
    if (want("cbd-reduce1")) cb_gmres<double>("CbGmres<double>", hip, a_d, sp::reduce1, "reduce1", kd, 4, mat_d);
    if (want("cbc")) cb_gmres<cd>("CbGmres<complex<double>>", hip, a_c, sp::keep, "keep", kd, 16, mat_c);
    if (want("cbc")) cb_gmres<cd>("CbGmres<complex<double>>", hip, a_c, sp::reduce1, "reduce1", kd, 8, mat_c);
    if (!want("jacobi")) return 0;

    // block-Jacobi(8) apply: double fast path, float adaptive (autodetect), complex<double> full storage
    auto time_jacobi = [&](const char* name, auto a, auto factory, double block_bytes_per_row) {
        using V = typename std::decay_t<decltype(*a)>::value_type;
        using Dense = gko::matrix::Dense<V>;
        auto t0 = std::chrono::steady_clock::now();
        auto jac = factory.on(hip)->generate(a);
        hip->synchronize();
        const double gen = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        auto b = Dense::create(hip, gko::dim<2>{n, 1});
        b->fill(V(1.0));
        auto x = Dense::create(hip, gko::dim<2>{n, 1});
        const double ms = time_ms(hip, 20, [&] { jac->apply(b, x); });
        const double bytes = block_bytes_per_row * n + 2.0 * sizeof(V) * n;
        std::printf("%-50s apply %8.1f us  %6.1f %% of 8 TB/s (%.2f GB)   [generate %.3f s]\n", name, ms * 1e3,
                    bytes / (ms * 1e-3) / 8e12 * 100, bytes / 1e9, gen);
    };
    time_jacobi("Jacobi<double>(8), full storage (fast path)", a_d,
                gko::preconditioner::Jacobi<double, it>::build().with_max_block_size(8u), 64.0);
    time_jacobi("Jacobi<double>(8), adaptive (autodetect)", a_d,
                gko::preconditioner::Jacobi<double, it>::build().with_max_block_size(8u).with_storage_optimization(
                    gko::precision_reduction::autodetect()), 16.0);
    time_jacobi("Jacobi<float>(8), full storage", a_f,
                gko::preconditioner::Jacobi<float, it>::build().with_max_block_size(8u), 32.0);
    time_jacobi("Jacobi<float>(8), adaptive (autodetect)", a_f,
                gko::preconditioner::Jacobi<float, it>::build().with_max_block_size(8u).with_storage_optimization(
                    gko::precision_reduction::autodetect()), 16.0);
    time_jacobi("Jacobi<complex<double>>(8), full storage", a_c,
                gko::preconditioner::Jacobi<cd, it>::build().with_max_block_size(8u), 128.0);
    time_jacobi("Jacobi<complex<double>>(8), adaptive (autodetect)", a_c,
                gko::preconditioner::Jacobi<cd, it>::build().with_max_block_size(8u).with_storage_optimization(
                    gko::precision_reduction::autodetect()), 32.0);
    return 0;
}
