// TEST INFRASTRUCTURE.  Ginkgo's OWN distributed classes (unmodified core built with
// GINKGO_BUILD_MPI=1, oracle/build_ref_mpi.py) on this backend, launched with
// `mpiexec -n <ranks>`; all ranks share GPU 0 of the test box.  Restates the equivalence checks of
// test/mpi/distributed/matrix.cpp (apply / advanced apply vs ReferenceExecutor),
// test/mpi/distributed/vector.cpp (compute_dot / compute_norm2 / compute_squared_norm2, add_scaled,
// scale) and test/mpi/solver/solver.cpp (distributed Cg with a Schwarz(block-Jacobi)
// preconditioner) for gko::experimental::distributed::{Matrix, Vector}<double, int32, int64>:
//   core/distributed/matrix.cpp:450-509 (apply_impl: local SpMV || halo exchange, non-local SpMV),
//   core/distributed/vector.cpp:473-592 (reductions + all_reduce).
// Every comparison is HipExecutor (gko-cdna4 kernels) against ReferenceExecutor on the SAME
// partitioned data; MPICH is not GPU-aware, so Ginkgo stages the halo through the host itself.
//   mpiexec -n 2 mpi_dist_test [grid=24]
#include <mpi.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <random>
#include <vector>

#include <ginkgo/ginkgo.hpp>

#include "rccl_communicator.hpp"

using vt = double;
using lit = gko::int32;
using git = gko::int64;
using dist_mtx = gko::experimental::distributed::Matrix<vt, lit, git>;
using dist_vec = gko::experimental::distributed::Vector<vt>;
using part_type = gko::experimental::distributed::Partition<lit, git>;
using dense = gko::matrix::Dense<vt>;

static int failures = 0;
static int g_rank = 0;

// present when the program is linked with the GPU-aware-MPI layer (gko_binding/mpi_rccl.cpp)
extern "C" void gkoc_mpi_stats(long* out7) __attribute__((weak));

static void check(bool ok, const char* what, double value = 0)
{
    int all_ok = ok ? 1 : 0;
    MPI_Allreduce(MPI_IN_PLACE, &all_ok, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
    if (g_rank == 0) std::printf("%-68s %s  (%.3e)\n", what, all_ok ? "PASSED" : "FAILED", value);
    if (!all_ok) ++failures;
}

// max |a - b| / max |b| over the LOCAL parts of two distributed vectors (b on the host)
static double local_rel_diff(const dist_vec* on_device, const dist_vec* on_host)
{
    auto ref = gko::ReferenceExecutor::create();
    auto d = gko::clone(ref, on_device->get_local_vector());
    auto h = on_host->get_local_vector();
    double diff = 0, scale = 0;
    for (gko::size_type i = 0; i < h->get_size()[0]; ++i) {
        for (gko::size_type j = 0; j < h->get_size()[1]; ++j) {
            diff = std::max(diff, std::abs(d->at(i, j) - h->at(i, j)));
            scale = std::max(scale, std::abs(h->at(i, j)));
        }
    }
    double both[2] = {diff, scale};
    MPI_Allreduce(MPI_IN_PLACE, both, 2, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    return both[1] > 0 ? both[0] / both[1] : both[0];
}

int main(int argc, char** argv)
{
    const gko::experimental::mpi::environment env(argc, argv);
    const gko::experimental::mpi::communicator comm(MPI_COMM_WORLD);
    g_rank = comm.rank();
    const int grid = argc > 1 ? std::atoi(argv[1]) : 24;
    const git n = git(grid) * grid * grid;
    auto ref = gko::ReferenceExecutor::create();
    auto hip = gko::HipExecutor::create(0, ref);
    if (g_rank == 0) {
        std::cout << comm.size() << " ranks, " << hip->get_description() << ", 27-pt " << grid
                  << "^3, GPU-aware MPI: " << gko::experimental::mpi::is_gpu_aware() << std::endl;
    }

    // rows by contiguous ranges (z-slabs), 27-point stencil with global column indices:
    // benchmark/utils/stencil_matrix.hpp:264-453 (diag 26, off-diagonals -1)
    auto partition = gko::share(part_type::build_from_global_size_uniform(ref, comm.size(), n));
    const auto lo = partition->get_range_bounds()[g_rank], hi = partition->get_range_bounds()[g_rank + 1];
    gko::matrix_data<vt, git> a_data{gko::dim<2>(n, n)};
    gko::matrix_data<vt, git> b_data{gko::dim<2>(n, 2)}, x_data{gko::dim<2>(n, 2)};
    std::mt19937_64 rng(42);
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    std::vector<double> bvals(2 * n), xvals(2 * n);
    for (auto& v : bvals) v = dist(rng);   // same sequence on every rank
    for (auto& v : xvals) v = dist(rng);
    for (git row = lo; row < hi; ++row) {
        const git ix = row % grid, iy = (row / grid) % grid, iz = row / (git(grid) * grid);
        for (int dz = -1; dz <= 1; ++dz) {
            for (int dy = -1; dy <= 1; ++dy) {
                for (int dx = -1; dx <= 1; ++dx) {
                    const git x = ix + dx, y = iy + dy, z = iz + dz;
                    if (x < 0 || y < 0 || z < 0 || x >= grid || y >= grid || z >= grid) continue;
                    const git col = (z * grid + y) * grid + x;
                    a_data.nonzeros.emplace_back(row, col, col == row ? 26.0 : -1.0);
                }
            }
        }
        for (int j = 0; j < 2; ++j) {
            b_data.nonzeros.emplace_back(row, j, bvals[2 * row + j]);
            x_data.nonzeros.emplace_back(row, j, xvals[2 * row + j]);
        }
    }
    auto a_host = gko::share(dist_mtx::create(ref, comm));
    auto b_host = gko::share(dist_vec::create(ref, comm));
    auto x_host = gko::share(dist_vec::create(ref, comm));
    a_host->read_distributed(a_data, partition);
    b_host->read_distributed(b_data, partition);
    x_host->read_distributed(x_data, partition);
    auto a = gko::share(dist_mtx::create(hip, comm));
    auto b = gko::share(dist_vec::create(hip, comm));
    auto x = gko::share(dist_vec::create(hip, comm));
    a->copy_from(a_host);
    b->copy_from(b_host);
    x->copy_from(x_host);
    {
        auto nl = gko::as<gko::matrix::Csr<vt, lit>>(a->get_non_local_matrix());
        long cols = nl->get_size()[1], total = 0;
        MPI_Allreduce(&cols, &total, 1, MPI_LONG, MPI_SUM, MPI_COMM_WORLD);
        check(comm.size() == 1 || total > 0, "the non-local blocks have halo columns", double(total));
    }

    // ---- read_distributed ON THE DEVICE (core/distributed/matrix.cpp:300-381, vector.cpp): partition
    // built on the HipExecutor, device_matrix_data sorted there, local / non-local split, index map and
    // the vector's local part by this backend's kernels (csrc/dist_setup.hip) - against the host read
    {
        auto part_dev = gko::share(part_type::build_from_global_size_uniform(hip, comm.size(), n));
        auto a_dev = gko::share(dist_mtx::create(hip, comm));
        auto b_dev = gko::share(dist_vec::create(hip, comm));
        a_dev->read_distributed(a_data, part_dev);
        b_dev->read_distributed(b_data, part_dev);
        auto same_csr = [&](const gko::LinOp* p, const gko::LinOp* q) {
            auto u = gko::clone(ref, gko::as<gko::matrix::Csr<vt, lit>>(p));
            auto v = gko::as<gko::matrix::Csr<vt, lit>>(q);
            if (u->get_size() != v->get_size() || u->get_num_stored_elements() != v->get_num_stored_elements()) {
                return false;
            }
            bool same = true;
            for (gko::size_type i = 0; same && i <= u->get_size()[0]; ++i) {
                same = u->get_const_row_ptrs()[i] == v->get_const_row_ptrs()[i];
            }
            for (gko::size_type k = 0; same && k < u->get_num_stored_elements(); ++k) {
                same = u->get_const_col_idxs()[k] == v->get_const_col_idxs()[k] &&
                       u->get_const_values()[k] == v->get_const_values()[k];
            }
            return same;
        };
        check(same_csr(a_dev->get_local_matrix().get(), a_host->get_local_matrix().get()),
              "read_distributed on the HipExecutor: local block identical to the host read", 0.0);
        check(same_csr(a_dev->get_non_local_matrix().get(), a_host->get_non_local_matrix().get()),
              "read_distributed on the HipExecutor: non-local block (index map) identical", 0.0);
        double dv = local_rel_diff(b_dev.get(), b_host.get());
        check(dv == 0.0, "Vector::read_distributed on the HipExecutor: local part identical", dv);
        auto x_dev = gko::share(dist_vec::create(hip, comm));
        x_dev->copy_from(x_host);
        a_host->apply(b_host, x_host);
        a_dev->apply(b_dev, x_dev);
        dv = local_rel_diff(x_dev.get(), x_host.get());
        check(dv <= 1e-14, "apply of the matrix read on the device: hip vs reference", dv);
    }

    // ---- Matrix::apply / advanced apply (test/mpi/distributed/matrix.cpp)
    a_host->apply(b_host, x_host);
    a->apply(b, x);
    double d = local_rel_diff(x.get(), x_host.get());
    check(d <= 1e-14, "distributed::Matrix::apply, 2 right-hand sides: hip vs reference", d);
    auto alpha = gko::initialize<dense>({2.0}, ref), beta = gko::initialize<dense>({-0.5}, ref);
    auto dalpha = gko::clone(hip, alpha), dbeta = gko::clone(hip, beta);
    a_host->apply(alpha, b_host, beta, x_host);
    a->apply(dalpha, b, dbeta, x);
    d = local_rel_diff(x.get(), x_host.get());
    check(d <= 1e-14, "distributed::Matrix::apply(alpha, b, beta, x): hip vs reference", d);

    // ---- this backend's CollectiveCommunicator (ginkgo_amd/gko_binding/rccl_communicator.hpp) inside
    // Ginkgo's RowGatherer (test/mpi/distributed/row_gatherer.cpp): gather the halo planes of b.
    // RCCL carries device buffers when every rank has its own GPU; on this box the ranks share one
    // and Ginkgo hands over host-staged buffers, which take the class's MPI path.
    {
        using gatherer = gko::experimental::distributed::RowGatherer<lit>;
        using imap_t = gko::experimental::distributed::index_map<lit, git>;
        const git plane = git(grid) * grid;
        std::vector<git> halo;
        for (git gi = std::max<git>(lo - plane, 0); gi < lo; ++gi) halo.push_back(gi);
        for (git gi = hi; gi < std::min<git>(hi + plane, n); ++gi) halo.push_back(gi);
        gko::array<git> halo_arr(ref, halo.begin(), halo.end());
        imap_t imap(ref, partition, g_rank, halo_arr);
        std::shared_ptr<const gko::experimental::mpi::CollectiveCommunicator> templ =
            std::make_shared<gko::cdna4::RcclCommunicator>(comm, 0);
        const bool rccl = static_cast<const gko::cdna4::RcclCommunicator*>(templ.get())->uses_rccl();
        auto mpi_exec = gko::experimental::mpi::requires_host_buffer(hip, comm)
                            ? std::shared_ptr<const gko::Executor>(ref)
                            : std::shared_ptr<const gko::Executor>(hip);
        auto rg = gatherer::create(hip, templ->create_with_same_type(comm, &imap), imap);
        auto out = dist_vec::create(mpi_exec, comm, gko::dim<2>{rg->get_size()[0], 2},
                                    gko::dim<2>{halo.size(), 2});
        rg->apply_async(b, out).wait();
        hip->synchronize();
        auto got = gko::clone(ref, out->get_local_vector());
        double worst = 0;
        // remote indices come sorted by owning rank, then by global index = the order of `halo`
        for (gko::size_type i = 0; i < halo.size(); ++i) {
            for (int j = 0; j < 2; ++j) worst = std::max(worst, std::abs(got->at(i, j) - bvals[2 * halo[i] + j]));
        }
        check(worst == 0.0, rccl ? "RowGatherer with RcclCommunicator (RCCL transport): halo of b"
                                 : "RowGatherer with RcclCommunicator (MPI path: ranks share the GPU)",
              worst);
    }

    // ---- Vector reductions and updates (test/mpi/distributed/vector.cpp)
    auto res_h = dense::create(ref, gko::dim<2>{1, 2}), res_d = dense::create(hip, gko::dim<2>{1, 2});
    auto rel = [&](const dense* dev, const dense* host) {
        auto c = gko::clone(ref, dev);
        double m = 0;
        for (int j = 0; j < 2; ++j) {
            m = std::max(m, std::abs(c->at(0, j) - host->at(0, j)) / std::max(std::abs(host->at(0, j)), 1e-300));
        }
        return m;
    };
    b_host->compute_dot(x_host, res_h);
    b->compute_dot(x, res_d);
    check(rel(res_d.get(), res_h.get()) <= 1e-13, "distributed::Vector::compute_dot (all-reduced)", rel(res_d.get(), res_h.get()));
    b_host->compute_norm2(res_h);
    b->compute_norm2(res_d);
    check(rel(res_d.get(), res_h.get()) <= 1e-13, "distributed::Vector::compute_norm2", rel(res_d.get(), res_h.get()));
    b_host->compute_squared_norm2(res_h);
    b->compute_squared_norm2(res_d);
    check(rel(res_d.get(), res_h.get()) <= 1e-13, "distributed::Vector::compute_squared_norm2", rel(res_d.get(), res_h.get()));
    x_host->add_scaled(alpha, b_host);
    x->add_scaled(dalpha, b);
    x_host->scale(beta);
    x->scale(dbeta);
    d = local_rel_diff(x.get(), x_host.get());
    check(d == 0.0, "distributed::Vector::add_scaled + scale: bit-identical local parts", d);

    // ---- distributed Cg + Schwarz(block-Jacobi(8)) (test/mpi/solver/solver.cpp, examples/distributed-solver)
    auto solve = [&](std::shared_ptr<const gko::Executor> exec, std::shared_ptr<dist_mtx> mat,
                     const dist_vec* rhs, dist_vec* sol) {
        using schwarz = gko::experimental::distributed::preconditioner::Schwarz<vt, lit, git>;
        auto logger = gko::share(gko::log::Convergence<vt>::create());
        auto solver =
            gko::solver::Cg<vt>::build()
                .with_preconditioner(
                    schwarz::build()
                        .with_local_solver(gko::preconditioner::Jacobi<vt, lit>::build().with_max_block_size(8u))
                        .on(exec))
                .with_criteria(gko::stop::Iteration::build().with_max_iters(500u),
                               gko::stop::ResidualNorm<vt>::build().with_reduction_factor(1e-10))
                .on(exec)
                ->generate(mat);
        solver->add_logger(logger);
        solver->apply(rhs, sol);
        return int(logger->get_num_iterations());
    };
    gko::matrix_data<vt, git> ones_data{gko::dim<2>(n, 1)}, zero_data{gko::dim<2>(n, 1)};
    for (git row = lo; row < hi; ++row) {
        ones_data.nonzeros.emplace_back(row, 0, 1.0);
        zero_data.nonzeros.emplace_back(row, 0, 0.0);
    }
    auto rhs_host = dist_vec::create(ref, comm), sol_host = dist_vec::create(ref, comm);
    rhs_host->read_distributed(ones_data, partition);
    sol_host->read_distributed(zero_data, partition);
    auto rhs = dist_vec::create(hip, comm), sol = dist_vec::create(hip, comm);
    rhs->copy_from(rhs_host);
    sol->copy_from(sol_host);
    const int it_host = solve(ref, a_host, rhs_host.get(), sol_host.get());
    const int it_dev = solve(hip, a, rhs.get(), sol.get());
    check(std::abs(it_host - it_dev) <= 1, "distributed Cg + Schwarz(Jacobi(8)): iteration count hip vs reference",
          double(it_dev - it_host));
    d = local_rel_diff(sol.get(), sol_host.get());
    check(d <= 1e-8, "distributed Cg solution: hip vs reference", d);
    // true residual through the distributed operator on the device
    auto one = gko::initialize<dense>({1.0}, hip), neg = gko::initialize<dense>({-1.0}, hip);
    auto r = dist_vec::create(hip, comm);
    r->copy_from(rhs);
    a->apply(neg, sol, one, r);
    auto rn = dense::create(hip, gko::dim<2>{1, 1});
    r->compute_norm2(rn);
    const double resn = gko::clone(ref, rn)->at(0, 0) / std::sqrt(double(n));
    check(resn <= 1e-9, "distributed Cg: relative true residual on the device", resn);

    if (gkoc_mpi_stats && g_rank == 0) {
        // which way the device buffers of Ginkgo's MPI calls went (libgkoc_mpi_rccl.so)
        long st[7];
        gkoc_mpi_stats(st);
        std::printf("gkoc_mpi routes: all-reduce rccl %ld staged %ld, all-to-all-v rccl %ld staged %ld, other staged "
                    "%ld, bytes through the host %ld, host calls %ld\n",
                    st[0], st[1], st[2], st[3], st[4], st[5], st[6]);
    }
    if (g_rank == 0) {
        std::printf("%s: %d iterations on hip, %d on reference\n", failures ? "FAILED" : "ALL PASSED", it_dev,
                    it_host);
    }
    return failures ? 1 : 0;
}
