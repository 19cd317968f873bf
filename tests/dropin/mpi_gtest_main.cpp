// main() for Ginkgo's own MPI tests (test/mpi/**/*.cpp, compiled unmodified) on this backend.
// The reference's core/test/gtest/ginkgo_mpi_main.cpp wraps GoogleTest's event listeners to merge
// the ranks' output; the GoogleTest stand-in of this repository (tests/dropin/gtest_shim) has no
// listener interface, so this file does the part that matters itself: MPI is initialised before
// the tests run, every rank runs every test, ranks other than 0 keep quiet, and the exit status is
// the MAXIMUM over the ranks (a test that fails on one rank fails the run).
#include <cstdio>
#include <cstdlib>
#include <string>

#include <mpi.h>

#include <gtest/gtest.h>

#include "core/test/gtest/environments.hpp"


int ResourceEnvironment::omp_threads = 0;
int ResourceEnvironment::cuda_device_id = 0;
int ResourceEnvironment::hip_device_id = 0;
int ResourceEnvironment::sycl_device_id = 0;


int main(int argc, char** argv)
{
    int provided = 0;
    MPI_Init_thread(&argc, &argv, MPI_THREAD_SERIALIZED, &provided);
    int rank = 0, size = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    ::testing::InitGoogleTest(&argc, argv);
    ::testing::AddGlobalTestEnvironment(new ResourceEnvironment);
    ::testing::AddGlobalTestEnvironment(new DeviceEnvironment(rank));
    if (rank != 0) {
        // one report, from rank 0; the other ranks' failures reach it through the exit status
        // (GKOC_TEST_RANK_LOG=<prefix>: their output goes to <prefix>.rank<k>.log instead)
        const char* prefix = std::getenv("GKOC_TEST_RANK_LOG");
        std::string to = "/dev/null";
        if (prefix && *prefix) to = std::string(prefix) + ".rank" + std::to_string(rank) + ".log";
        if (!std::freopen(to.c_str(), "w", stdout)) return 2;
    }
    int result = RUN_ALL_TESTS();
    int worst = 0;
    MPI_Allreduce(&result, &worst, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
    if (rank == 0 && worst != result) {
        std::printf("[  FAILED  ] on a rank other than 0 (exit status %d)\n", worst);
    }
    MPI_Finalize();
    return worst;
}
