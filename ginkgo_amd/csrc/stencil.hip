// Device-side generator for the benchmark stencil matrices (workload synthesis,
// not a solver kernel): 5/9-pt 2-D and 7/27-pt 3-D Laplacians with the exact
// entries, values and ordering of the reference's generator
// benchmark/utils/stencil_matrix.hpp:68-238 (2-D) and :264-453 (3-D):
// lexicographic x-fastest numbering, diagonal = (#stencil points - 1),
// off-diagonals = -1, entries emitted in (dz, dy, dx) order = ascending column.
//
// Generating on the device avoids assembling 5.4 GB on the host and pushing it
// over PCIe for the 256^3 configuration.  A `z-slab` variant produces the rows
// of planes [z0, z0+nz) of the global grid with GLOBAL column indices, used by
// the row-partitioned multi-GPU path.  Index arrays are integer-exact vs
// the CPU restatement of the same generator (tests/test_spmv_gpu.py).
#include "common.hpp"
#include "scan.hpp"

namespace gkoc {
namespace {

struct stencil_desc {
    int nd;
    int restricted;
    int64_t g;       // global points per dimension
    int64_t z0, nz;  // owned planes (3-D) or rows of the 2-D grid (nd == 2: y)
};

__device__ __forceinline__ bool keep(const stencil_desc& d, int dz, int dy, int dx)
{
    return !d.restricted || ((dz == 0) + (dy == 0) + (dx == 0) >= 2);
}

// local row -> (ix, iy, iz) with the slab's slowest coordinate offset by z0
__device__ __forceinline__ void coords(const stencil_desc& d, int64_t row,
                                       int64_t& ix, int64_t& iy, int64_t& iz)
{
    ix = row % d.g;
    const int64_t t = row / d.g;
    if (d.nd == 2) {
        iy = t + d.z0;
        iz = 0;
    } else {
        iy = t % d.g;
        iz = t / d.g + d.z0;
    }
}

template <typename I>
__global__ __launch_bounds__(256) void stencil_count_kernel(stencil_desc d,
                                                            int64_t n_local,
                                                            I* row_ptrs)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row > n_local) return;
    if (row == n_local) {
        row_ptrs[row] = 0;
        return;
    }
    int64_t ix, iy, iz;
    coords(d, row, ix, iy, iz);
    const int64_t gz = d.nd == 3 ? d.g : 1;
    int cnt = 0;
    for (int dz = (d.nd == 3 ? -1 : 0); dz <= (d.nd == 3 ? 1 : 0); ++dz) {
        for (int dy = -1; dy <= 1; ++dy) {
            for (int dx = -1; dx <= 1; ++dx) {
                if (!keep(d, dz, dy, dx)) continue;
                const int64_t jx = ix + dx, jy = iy + dy, jz = iz + dz;
                cnt += (jx >= 0 && jx < d.g && jy >= 0 && jy < d.g && jz >= 0 &&
                        jz < gz);
            }
        }
    }
    row_ptrs[row] = I(cnt);
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void stencil_fill_kernel(
    stencil_desc d, int64_t n_local, const I* __restrict__ row_ptrs,
    I* __restrict__ cols, T* __restrict__ vals)
{
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n_local) return;
    int64_t ix, iy, iz;
    coords(d, row, ix, iy, iz);
    const int64_t gz = d.nd == 3 ? d.g : 1;
    const int64_t grow = ix + iy * d.g + iz * d.g * d.g;
    int npts = 0;
    for (int dz = (d.nd == 3 ? -1 : 0); dz <= (d.nd == 3 ? 1 : 0); ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) npts += keep(d, dz, dy, dx);
    const T diag = T(npts - 1);
    int64_t k = row_ptrs[row];
    for (int dz = (d.nd == 3 ? -1 : 0); dz <= (d.nd == 3 ? 1 : 0); ++dz) {
        for (int dy = -1; dy <= 1; ++dy) {
            for (int dx = -1; dx <= 1; ++dx) {
                if (!keep(d, dz, dy, dx)) continue;
                const int64_t jx = ix + dx, jy = iy + dy, jz = iz + dz;
                if (jx >= 0 && jx < d.g && jy >= 0 && jy < d.g && jz >= 0 &&
                    jz < gz) {
                    const int64_t col = jx + jy * d.g + jz * d.g * d.g;
                    cols[k] = I(col);
                    vals[k] = col == grow ? diag : T(-1);
                    ++k;
                }
            }
        }
    }
}

inline int check_desc(int nd, int64_t g, int64_t z0, int64_t nz)
{
    GKOC_REQUIRE(nd == 2 || nd == 3, GKOC_E_INVALID, "nd must be 2 or 3");
    GKOC_REQUIRE(g >= 1 && z0 >= 0 && nz >= 0 && z0 + nz <= g, GKOC_E_INVALID,
                 "bad stencil slab");
    return GKOC_OK;
}

}  // namespace
}  // namespace gkoc

using namespace gkoc;

#define GKOC_DEF_STENCIL_PTRS(I, IN)                                           \
    extern "C" int gkoc_stencil_row_ptrs_##IN(                                 \
        gkoc_stream_t s, int nd, int64_t g, int restricted, int64_t z0,        \
        int64_t nz, I* row_ptrs, int64_t* nnz_host)                            \
    {                                                                          \
        int rc = check_desc(nd, g, z0, nz);                                    \
        if (rc != GKOC_OK) return rc;                                          \
        const int64_t n_local = nz * (nd == 3 ? g * g : g);                    \
        stencil_desc d{nd, restricted, g, z0, nz};                             \
        stencil_count_kernel<I>                                                \
            <<<dim3(unsigned(ceildiv(n_local + 1, 256))), dim3(256), 0,        \
               as_stream(s)>>>(d, n_local, row_ptrs);                          \
        GKOC_LAUNCH_OK();                                                      \
        rc = device_exclusive_scan<I>(as_stream(s), row_ptrs, n_local + 1);    \
        if (rc != GKOC_OK) return rc;                                          \
        if (nnz_host) {                                                        \
            I h = 0;                                                           \
            GKOC_HIP(hipMemcpyAsync(&h, row_ptrs + n_local, sizeof(I),         \
                                    hipMemcpyDeviceToHost, as_stream(s)));     \
            GKOC_HIP(hipStreamSynchronize(as_stream(s)));                      \
            *nnz_host = int64_t(h);                                            \
        }                                                                      \
        return GKOC_OK;                                                        \
    }
GKOC_DEF_STENCIL_PTRS(int32_t, i32)
GKOC_DEF_STENCIL_PTRS(int64_t, i64)

#define GKOC_DEF_STENCIL_FILL(T, TN, I, IN)                                    \
    extern "C" int gkoc_stencil_fill_##TN##_##IN(                              \
        gkoc_stream_t s, int nd, int64_t g, int restricted, int64_t z0,        \
        int64_t nz, const I* row_ptrs, I* cols, T* vals)                       \
    {                                                                          \
        int rc = check_desc(nd, g, z0, nz);                                    \
        if (rc != GKOC_OK) return rc;                                          \
        const int64_t n_local = nz * (nd == 3 ? g * g : g);                    \
        if (n_local == 0) return GKOC_OK;                                      \
        stencil_desc d{nd, restricted, g, z0, nz};                             \
        stencil_fill_kernel<T, I>                                              \
            <<<dim3(unsigned(ceildiv(n_local, 256))), dim3(256), 0,            \
               as_stream(s)>>>(d, n_local, row_ptrs, cols, vals);              \
        GKOC_LAUNCH_OK();                                                      \
        return GKOC_OK;                                                        \
    }
GKOC_DEF_STENCIL_FILL(double, f64, int32_t, i32)
GKOC_DEF_STENCIL_FILL(double, f64, int64_t, i64)
GKOC_DEF_STENCIL_FILL(float, f32, int32_t, i32)
GKOC_DEF_STENCIL_FILL(float, f32, int64_t, i64)
