// place_lab4 (development tool): does gkoc_arena_probe see the memory classes that the
// SpMV sees?  One 160 GiB VMM chunk; probe x at a few positions against y at every GiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "gko_cdna4.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define GK(x) do { int r_ = (x); if (r_) { printf("gkoc error %d %s at %d\n", r_, gkoc_last_error(), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv)
{
    const size_t GiB = size_t(1) << 30;
    const size_t chunk_gib = argc > 1 ? atoll(argv[1]) : 160;
    const size_t chunk = chunk_gib * GiB;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    void* va;
    CK(hipMemAddressReserve(&va, chunk, size_t(64) * GiB, nullptr, 0));
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap(va, chunk, 0, h, 0));
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, chunk, &acc, 1));
    char* base = (char*)va;
    struct cfg { int read_kb, write_b; size_t x_bytes; };
    const cfg cfgs[] = {{32, 1024, 2 * GiB}, {16, 1024, 1 * GiB}, {8, 1024, 1 * GiB}, {32, 4096, 2 * GiB}, {64, 1024, 2 * GiB}, {16, 1024, GiB / 4}};
    for (const cfg& c : cfgs) {
        for (size_t xg : {size_t(0), size_t(70), size_t(130)}) {
            printf("read %d KiB + write %d B per wave, x = %zu MiB at %zu GiB; y at k GiB [us]:\n", c.read_kb,
                   c.write_b, c.x_bytes >> 20, xg);
            for (size_t k = 0; k + 3 <= chunk_gib; k += 3) {
                size_t yg = k;
                if (yg >= xg && yg < xg + 2) yg = xg + 2;   // same chunk region but not overlapping x
                int64_t ns;
                GK(gkoc_arena_probe(base + xg * GiB, c.x_bytes, base + yg * GiB, c.read_kb, c.write_b, 2, &ns));
                printf(" %.0f", ns / 1e3);
            }
            printf("\n");
        }
    }
    return 0;
}
