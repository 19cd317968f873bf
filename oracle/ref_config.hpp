// TEST INFRASTRUCTURE -- build configuration header for compiling the
// *unmodified* reference (Ginkgo 1.12.0, /root/reference) into oracle/_ref.
//
// Ginkgo's CMake normally generates <ginkgo/config.hpp> from
// include/ginkgo/config.hpp.in; we do not run the reference's build system
// (oracle/build_ref.py drives g++ directly), so this hand-written header
// plays that role.  It is installed as oracle/_ref/include/ginkgo/config.hpp.
//
// Choices: CPU-only oracle (reference + omp kernels), no MPI / PAPI / TAU /
// METIS / HWLOC, no half / bfloat16 instantiations (keeps the exported
// symbol set equal to the one SURVEY.md section 8(b) counted), and the HIP
// platform flag set to AMD ("HCC") so that Csr strategy thresholds
// (include/ginkgo/core/matrix/csr.hpp:532-535) and the Jacobi storage scheme
// (include/ginkgo/core/preconditioner/jacobi.hpp:589-627) see an AMD device
// when our backend provides the HipExecutor.
#ifndef GKO_INCLUDE_CONFIG_H
#define GKO_INCLUDE_CONFIG_H

#define GKO_VERSION_MAJOR 1
#define GKO_VERSION_MINOR 12
#define GKO_VERSION_PATCH 0
#define GKO_VERSION_TAG "develop"
#define GKO_VERSION_STR 1, 12, 0
#define GINKGO_VERSION_TAG_DEPRECATED 0

#define GKO_VERBOSE_LEVEL 1
#define GKO_HAVE_CXXABI_H
#define GKO_SIZE_T_IS_UINT64_T

#define GINKGO_HIP_PLATFORM_HCC 1
#define GINKGO_HIP_PLATFORM_NVCC 0
#define GINKGO_DPCPP_MAJOR_VERSION 0
#define GINKGO_DPCPP_MINOR_VERSION 0

#define GKO_HAVE_PAPI_SDE 0
#define GKO_HAVE_TAU 0
#define GKO_HAVE_VTUNE 0
#define GKO_HAVE_METIS 0
#define GKO_HAVE_ROCTX 0
#define GINKGO_BUILD_MPI 0
#define GINKGO_HAVE_GPU_AWARE_MPI 0
#define GKO_HAVE_HWLOC 0
#define GINKGO_ENABLE_HALF 0
#define GINKGO_ENABLE_BFLOAT16 0
#define GINKGO_HAVE_OPENMPI_PRE_4_1_X 0

#endif  // GKO_INCLUDE_CONFIG_H
