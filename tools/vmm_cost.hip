// vmm_cost (development tool): what do the steps of the arena's granule search cost?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gko_cdna4.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t GiB = size_t(1) << 30;
    CK(hipFree(nullptr));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t fr, tot;
    CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GiB of %.1f GiB\n", fr / double(GiB), tot / double(GiB));
    void* ref;
    CK(hipMalloc(&ref, 8 * GiB));
    for (size_t gib : {size_t(1), size_t(4), size_t(8), size_t(8), size_t(32), size_t(64)}) {
        const size_t sz = gib * GiB;
        void* va;
        double t0 = now();
        CK(hipMemAddressReserve(&va, sz, sz, nullptr, 0));
        double t1 = now();
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, sz, &prop, 0));
        double t2 = now();
        CK(hipMemMap(va, sz, 0, h, 0));
        CK(hipMemSetAccess(va, sz, &acc, 1));
        double t3 = now();
        int64_t ns = 0;
        if (gib >= 4) gkoc_arena_probe(ref, 2 * GiB, va, 32, 1024, 2, &ns);
        double t4 = now();
        CK(hipMemUnmap(va, sz));
        double t5 = now();
        CK(hipMemRelease(h));
        double t6 = now();
        CK(hipMemAddressFree(va, sz));
        void* p;
        double t7 = now();
        CK(hipMalloc(&p, sz));
        double t8 = now();
        CK(hipFree(p));
        double t9 = now();
        printf("%3zu GiB: reserve %.4f create %.4f map+access %.4f probe(3 launches) %.4f (%.0f us) unmap %.4f release %.4f | hipMalloc %.4f hipFree %.4f s\n",
               gib, t1 - t0, t2 - t1, t3 - t2, t4 - t3, ns / 1e3, t5 - t4, t6 - t5, t8 - t7, t9 - t8);
    }
    // consecutive 1 GiB physical handles of a fresh process: which class (vs the first one)?
    {
        const size_t n_gr = 200;
        void* va;
        CK(hipMemAddressReserve(&va, n_gr * GiB, GiB, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs;
        double t0 = now();
        printf("consecutive 1 GiB handles, S = same class as handle 0, d = different, per handle (create+map+2 probes):\n");
        double tc = 0, tp = 0;
        for (size_t i = 0; i < n_gr; ++i) {
            hipMemGenericAllocationHandle_t h;
            double a0 = now();
            if (hipMemCreate(&h, GiB, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            CK(hipMemMap((char*)va + i * GiB, GiB, 0, h, 0));
            CK(hipMemSetAccess((char*)va + i * GiB, GiB, &acc, 1));
            double a1 = now();
            hs.push_back(h);
            if (i == 0) { printf("0"); continue; }
            int64_t n_self, n_cross;
            // reference: handle 0 read, written into its own last 32 MiB (same class by construction)
            gkoc_arena_probe(va, GiB - (64 << 20), (char*)va + GiB - (32 << 20), 32, 1024, 2, &n_self);
            gkoc_arena_probe(va, GiB - (64 << 20), (char*)va + i * GiB, 32, 1024, 2, &n_cross);
            double a2 = now();
            tc += a1 - a0; tp += a2 - a1;
            printf("%c", double(n_cross) > 0.95 * double(n_self) ? 'S' : 'd');
            if (i % 50 == 49) printf("\n");
        }
        printf("\n   %zu handles in %.2f s (create+map %.2f s, probes %.2f s)\n", hs.size(), now() - t0, tc, tp);
        // classes relative to the first 'd' handle as well
        size_t first_d = 0;
        for (size_t i = 1; i < hs.size() && !first_d; ++i) {
            int64_t n_self, n_cross;
            gkoc_arena_probe(va, GiB - (64 << 20), (char*)va + GiB - (32 << 20), 32, 1024, 2, &n_self);
            gkoc_arena_probe(va, GiB - (64 << 20), (char*)va + i * GiB, 32, 1024, 2, &n_cross);
            if (double(n_cross) <= 0.95 * double(n_self)) first_d = i;
        }
        if (first_d) {
            printf("relative to handle %zu (S = same class as it):\n", first_d);
            char* xr = (char*)va + first_d * GiB;
            for (size_t i = 0; i < hs.size(); ++i) {
                if (i == first_d) { printf("0"); continue; }
                int64_t n_self, n_cross;
                gkoc_arena_probe(xr, GiB - (64 << 20), xr + GiB - (32 << 20), 32, 1024, 2, &n_self);
                gkoc_arena_probe(xr, GiB - (64 << 20), (char*)va + i * GiB, 32, 1024, 2, &n_cross);
                printf("%c", double(n_cross) > 0.95 * double(n_self) ? 'S' : 'd');
                if (i % 50 == 49) printf("\n");
            }
            printf("\n");
        }
        CK(hipMemUnmap(va, hs.size() * GiB));
        for (auto h : hs) CK(hipMemRelease(h));
    }
    return 0;
}
