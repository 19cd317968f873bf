// place_lab2 (development tool, round 2): which allocation scheme makes the time of
// the HBM-bound kernels independent of the allocation draw, and does the
// XCD-contiguous wave order help?  Links the product library, uses only its C ABI.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/place_lab2.hip \
//     -Lginkgo_amd/lib -lgko_cdna4 -Wl,-rpath,'$ORIGIN/../ginkgo_amd/lib' -o tools/place_lab2
// usage: place_lab2 [grid=256] [reps=8] [mode=all|pmc|quick] [trials=6]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "gko_cdna4.h"

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)
#define GK(x)                                                            \
    do {                                                                 \
        int r_ = (x);                                                    \
        if (r_ != 0) {                                                   \
            printf("gkoc error %d (%s) at %s:%d\n", r_, gkoc_last_error(), __FILE__, __LINE__); \
            exit(1);                                                     \
        }                                                                \
    } while (0)

static const size_t MiB = size_t(1) << 20, GiB = size_t(1) << 30;

struct timer {
    hipEvent_t a, b;
    timer()
    {
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
    }
    template <typename F>
    double us(int reps, F f)
    {
        f();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float t;
        CK(hipEventElapsedTime(&t, a, b));
        return double(t) / reps * 1e3;
    }
};

__global__ __launch_bounds__(256) void stream_read_kernel(int64_t nnz, const double* __restrict__ vals,
                                                          const int* __restrict__ cols,
                                                          double* __restrict__ out)
{
    double acc = 0;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < nnz; i += stride) {
        acc += vals[i] * double(cols[i]);
    }
    if (acc == 12345.678) out[0] = acc;
}

__global__ void seq_kernel(int64_t n, int step, int* out)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = int(i * step);
}

struct vmm_block {
    void* va;
    size_t size;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

static vmm_block vmm_alloc(size_t bytes, size_t va_align, size_t piece)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (piece == 0) piece = bytes;
    piece = (piece + gran - 1) / gran * gran;
    const size_t sz = (bytes + piece - 1) / piece * piece;
    vmm_block blk;
    blk.size = sz;
    CK(hipMemAddressReserve(&blk.va, sz, va_align, nullptr, 0));
    for (size_t off = 0; off < sz; off += piece) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, piece, &prop, 0));
        CK(hipMemMap((char*)blk.va + off, piece, 0, h, 0));
        blk.handles.push_back(h);
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(blk.va, sz, &acc, 1));
    return blk;
}

static void vmm_free(vmm_block& b)
{
    CK(hipMemUnmap(b.va, b.size));
    for (auto h : b.handles) CK(hipMemRelease(h));
    CK(hipMemAddressFree(b.va, b.size));
}

struct problem {
    int64_t g, n, nnz, nblk;
    int *rp, *col, *bp;
    double *val, *b, *y, *blk, *z;
};

int main(int argc, char** argv)
{
    const int64_t g = argc > 1 ? atoll(argv[1]) : 256;
    const int reps = argc > 2 ? atoi(argv[2]) : 8;
    const std::string mode = argc > 3 ? argv[3] : "all";
    const int trials = argc > 4 ? atoi(argv[4]) : 6;
    const int64_t n = g * g * g;
    timer T;
    GK(gkoc_arena_configure(0, 0, 1));  // the lab places memory itself

    // master copy (plain hipMalloc, first allocations of the process)
    problem M{};
    M.g = g;
    M.n = n;
    CK(hipMalloc(&M.rp, sizeof(int) * (n + 1)));
    GK(gkoc_stencil_row_ptrs_i32(nullptr, 3, g, 0, 0, g, M.rp, &M.nnz));
    const int64_t nnz = M.nnz;
    M.nblk = n / 8;
    CK(hipMalloc(&M.col, sizeof(int) * nnz));
    CK(hipMalloc(&M.val, sizeof(double) * nnz));
    CK(hipMalloc(&M.b, sizeof(double) * n));
    CK(hipMalloc(&M.y, sizeof(double) * n));
    CK(hipMalloc(&M.blk, sizeof(double) * 64 * M.nblk));
    CK(hipMalloc(&M.bp, sizeof(int) * (M.nblk + 1)));
    CK(hipMalloc(&M.z, sizeof(double) * n));
    GK(gkoc_stencil_fill_f64_i32(nullptr, 3, g, 0, 0, g, M.rp, M.col, M.val));
    {
        std::vector<double> hb(n);
        unsigned long long s2 = 42;
        for (int64_t i = 0; i < n; ++i) {
            s2 = s2 * 6364136223846793005ULL + 1442695040888963407ULL;
            hb[i] = double(s2 >> 11) / 9007199254740992.0 * 2 - 1;
        }
        CK(hipMemcpy(M.b, hb.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        seq_kernel<<<unsigned((M.nblk + 1 + 255) / 256), 256>>>(M.nblk + 1, 8, M.bp);
        // block storage: any finite numbers (time does not depend on the values)
        CK(hipMemcpy(M.blk, M.val, sizeof(double) * 64 * M.nblk, hipMemcpyDeviceToDevice));
        CK(hipDeviceSynchronize());
    }
    const double spmv_bytes = double(nnz) * 12 + double(n + 1) * 4 + double(n) * 16;
    const double jac_bytes = double(n) * 80 + 4.0 * (M.nblk + 1);
    const gkoc_jacobi_scheme scheme{8, 512, 3};
    std::vector<double> yref(n), ytmp(n);

    auto measure = [&](const char* tag, const problem& P, bool check) {
        double t[2], tj[2];
        for (int m = 0; m < 2; ++m) {
            GK(gkoc_tune_set(GKOC_TUNE_CSR_XCD_MAP, m));
            GK(gkoc_tune_set(GKOC_TUNE_JACOBI_XCD_MAP, m));
            t[m] = T.us(reps, [&] {
                GK(gkoc_csr_spmv_f64_i32(nullptr, n, n, P.rp, P.col, P.val, P.b, 1, P.y, 1, 1));
            });
            if (check) {
                CK(hipMemcpy(ytmp.data(), P.y, sizeof(double) * n, hipMemcpyDeviceToHost));
                if (memcmp(ytmp.data(), yref.data(), sizeof(double) * n) != 0) {
                    printf("!! result differs from the reference result (map %d)\n", m);
                }
            }
            tj[m] = T.us(reps, [&] {
                GK(gkoc_jacobi_simple_apply_f64_i32(nullptr, P.nblk, 8, scheme, P.bp, P.blk, P.y, 1,
                                                    P.z, 1, 1));
            });
        }
        const double tr = T.us(reps, [&] { stream_read_kernel<<<2048, 256>>>(nnz, P.val, P.col, P.z); });
        printf("%-34s spmv %7.1f (%4.1f%%) xcd %7.1f (%4.1f%%) | jacobi %6.1f (%4.1f%%) xcd %6.1f (%4.1f%%) | read %6.1f | val %p col %p y %p blk %p\n",
               tag, t[0], spmv_bytes / t[0] / 8e4, t[1], spmv_bytes / t[1] / 8e4, tj[0],
               jac_bytes / tj[0] / 8e4, tj[1], jac_bytes / tj[1] / 8e4, tr, (void*)P.val,
               (void*)P.col, (void*)P.y, (void*)P.blk);
        fflush(stdout);
    };
    auto fill_from_master = [&](problem& P) {
        P.g = g;
        P.n = n;
        P.nnz = nnz;
        P.nblk = M.nblk;
        CK(hipMemcpy(P.rp, M.rp, sizeof(int) * (n + 1), hipMemcpyDeviceToDevice));
        CK(hipMemcpy(P.col, M.col, sizeof(int) * nnz, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(P.val, M.val, sizeof(double) * nnz, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(P.b, M.b, sizeof(double) * n, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(P.blk, M.blk, sizeof(double) * 64 * M.nblk, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(P.bp, M.bp, sizeof(int) * (M.nblk + 1), hipMemcpyDeviceToDevice));
    };
    // place the arrays of one problem with `alloc`
    auto place = [&](problem& P, const std::function<void*(size_t)>& alloc) {
        P.val = (double*)alloc(sizeof(double) * nnz);
        P.col = (int*)alloc(sizeof(int) * nnz);
        P.rp = (int*)alloc(sizeof(int) * (n + 1));
        P.b = (double*)alloc(sizeof(double) * n);
        P.y = (double*)alloc(sizeof(double) * n);
        P.blk = (double*)alloc(sizeof(double) * 64 * M.nblk);
        P.bp = (int*)alloc(sizeof(int) * (M.nblk + 1));
        P.z = (double*)alloc(sizeof(double) * n);
        fill_from_master(P);
    };

    // reference result
    GK(gkoc_tune_set(GKOC_TUNE_CSR_XCD_MAP, 0));
    GK(gkoc_csr_spmv_f64_i32(nullptr, n, n, M.rp, M.col, M.val, M.b, 1, M.y, 1, 1));
    CK(hipMemcpy(yref.data(), M.y, sizeof(double) * n, hipMemcpyDeviceToHost));

    printf("27-pt %lld^3: n %lld nnz %lld; times in us, %% of 8 TB/s on algorithmic bytes; reps %d\n",
           (long long)g, (long long)n, (long long)nnz, reps);
    measure("master (first hipMallocs)", M, true);
    if (mode == "pmc") {
        // a few launches of each variant for the counter passes
        for (int m = 0; m < 2; ++m) {
            GK(gkoc_tune_set(GKOC_TUNE_CSR_XCD_MAP, m));
            GK(gkoc_tune_set(GKOC_TUNE_JACOBI_XCD_MAP, m));
            for (int i = 0; i < reps; ++i) {
                GK(gkoc_csr_spmv_f64_i32(nullptr, n, n, M.rp, M.col, M.val, M.b, 1, M.y, 1, 1));
                GK(gkoc_jacobi_simple_apply_f64_i32(nullptr, M.nblk, 8, scheme, M.bp, M.blk, M.y, 1,
                                                    M.z, 1, 1));
            }
        }
        CK(hipDeviceSynchronize());
        return 0;
    }
    measure("master again", M, false);

    const size_t set_bytes = size_t(8) * GiB;
    auto bump = [](char* base, size_t& off) {
        return [base, &off](size_t bytes) -> void* {
            off = (off + 2 * MiB - 1) / (2 * MiB) * (2 * MiB);
            void* p = base + off;
            off += bytes;
            return p;
        };
    };

    // S0: one hipMalloc per array (the reference's behaviour)
    if (mode != "arena" && mode != "pieces") {
        std::vector<void*> all;
        for (int t = 0; t < trials; ++t) {
            problem P{};
            place(P, [&](size_t b) {
                void* p;
                CK(hipMalloc(&p, b));
                all.push_back(p);
                return p;
            });
            char tag[64];
            snprintf(tag, 64, "S0 hipMalloc per array #%d", t);
            measure(tag, P, t == 0);
        }
        for (void* p : all) CK(hipFree(p));
    }
    // S1: one physically contiguous allocation per array
    if (mode != "arena" && mode != "pieces") {
        std::vector<void*> all;
        bool ok = true;
        for (int t = 0; t < trials && ok; ++t) {
            problem P{};
            place(P, [&](size_t b) {
                void* p = nullptr;
                if (hipExtMallocWithFlags(&p, b, hipDeviceMallocContiguous) != hipSuccess) {
                    (void)hipGetLastError();
                    ok = false;
                    CK(hipMalloc(&p, b));
                }
                all.push_back(p);
                return p;
            });
            char tag[64];
            snprintf(tag, 64, "S1 contiguous per array #%d%s", t, ok ? "" : " (flag refused)");
            measure(tag, P, t == 0);
        }
        for (void* p : all) CK(hipFree(p));
    }
    // S2: one 8 GiB hipMalloc per problem, arrays placed inside
    if (mode != "arena" && mode != "pieces") {
        std::vector<void*> all;
        for (int t = 0; t < trials; ++t) {
            char* base;
            CK(hipMalloc(&base, set_bytes));
            all.push_back(base);
            size_t off = 0;
            problem P{};
            place(P, bump(base, off));
            char tag[64];
            snprintf(tag, 64, "S2 arena hipMalloc 8 GiB #%d", t);
            measure(tag, P, t == 0);
        }
        for (void* p : all) CK(hipFree(p));
    }
    // S3: one 8 GiB VMM chunk per problem: one physical handle, VA aligned to 8 GiB
    if (mode != "arena" && mode != "pieces") {
        std::vector<vmm_block> all;
        for (int t = 0; t < trials; ++t) {
            vmm_block blk = vmm_alloc(set_bytes, set_bytes, 0);
            all.push_back(blk);
            size_t off = 0;
            problem P{};
            place(P, bump((char*)blk.va, off));
            char tag[64];
            snprintf(tag, 64, "S3 arena VMM 8 GiB, VA 8 GiB-aligned #%d", t);
            measure(tag, P, t == 0);
        }
        for (auto& b : all) vmm_free(b);
    }
    // S4: one 8 GiB physically contiguous chunk per problem
    if (mode != "arena" && mode != "pieces") {
        std::vector<void*> all;
        for (int t = 0; t < trials; ++t) {
            char* base = nullptr;
            if (hipExtMallocWithFlags((void**)&base, set_bytes, hipDeviceMallocContiguous) !=
                hipSuccess) {
                (void)hipGetLastError();
                printf("S4: contiguous 8 GiB refused\n");
                break;
            }
            all.push_back(base);
            size_t off = 0;
            problem P{};
            place(P, bump(base, off));
            char tag[64];
            snprintf(tag, 64, "S4 arena contiguous 8 GiB #%d", t);
            measure(tag, P, t == 0);
        }
        for (void* p : all) CK(hipFree(p));
    }
    // S5: VMM with 2 MiB physical pieces (smallest fragments), VA 8 GiB aligned
    if (mode == "all" || mode == "pieces") {
        for (int t = 0; t < 2; ++t) {
            vmm_block blk = vmm_alloc(set_bytes, set_bytes, 2 * MiB);
            size_t off = 0;
            problem P{};
            place(P, bump((char*)blk.va, off));
            char tag[64];
            snprintf(tag, 64, "S5 VMM 2 MiB pieces #%d", t);
            measure(tag, P, t == 0);
            vmm_free(blk);
        }
        for (size_t piece_mib : {size_t(256), size_t(1024), size_t(2048), size_t(8192)}) {
            vmm_block blk = vmm_alloc(set_bytes, set_bytes, piece_mib * MiB);
            size_t off = 0;
            problem P{};
            place(P, bump((char*)blk.va, off));
            char tag[64];
            snprintf(tag, 64, "S5c VMM %zu MiB pieces", piece_mib);
            measure(tag, P, false);
            vmm_free(blk);
        }
        for (int t = 0; t < 2; ++t) {
            vmm_block blk = vmm_alloc(set_bytes, set_bytes, 64 * MiB);
            size_t off = 0;
            problem P{};
            place(P, bump((char*)blk.va, off));
            char tag[64];
            snprintf(tag, 64, "S5b VMM 64 MiB pieces #%d", t);
            measure(tag, P, t == 0);
            vmm_free(blk);
        }
    }
    // S6: one 64 GiB hipMalloc, problems packed one after the other inside it
    if (mode != "arena" && mode != "pieces") {
        char* base;
        const size_t big = size_t(64) * GiB;
        CK(hipMalloc(&base, big));
        size_t off = 0;
        for (int t = 0; t < trials && off + set_bytes <= big; ++t) {
            problem P{};
            place(P, bump(base, off));
            char tag[64];
            snprintf(tag, 64, "S6 packed in one 64 GiB hipMalloc #%d", t);
            measure(tag, P, t == 0);
        }
        CK(hipFree(base));
    }
    // S7: the library's arena through gkoc_malloc: mode 1 (plain chunks), mode 2 (class
    // regions) with the size heuristic (Ginkgo's allocation order: values, col_idxs,
    // row_ptrs, b, y, Jacobi blocks, block pointers, z) and with stated roles
    for (int variant = 0; variant < 3 && mode != "pieces"; ++variant) {
        const int amode = variant == 0 ? 1 : 2;
        GK(gkoc_arena_configure(amode, set_bytes, 1));
        std::vector<void*> all;
        for (int t = 0; t < 3; ++t) {
            problem P{};
            int nth = 0;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            const auto t0 = std::chrono::steady_clock::now();
            place(P, [&](size_t b) {
                void* p;
                // order of place(): val, col, rp, b, y, blk, bp, z
                static const int roles[8] = {GKOC_MEM_VALUES, GKOC_MEM_INDICES, GKOC_MEM_INDICES,
                                             GKOC_MEM_VECTOR, GKOC_MEM_VECTOR, GKOC_MEM_INDICES,
                                             GKOC_MEM_VECTOR, GKOC_MEM_VECTOR};
                if (variant == 2) {
                    GK(gkoc_malloc_role(&p, b, roles[nth % 8]));
                } else {
                    GK(gkoc_malloc(&p, b));
                }
                ++nth;
                all.push_back(p);
                return p;
            });
            CK(hipDeviceSynchronize());
            const double secs =
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            int cv, cc, cy, cb, cz, cr;
            GK(gkoc_arena_class_of(P.val, &cv));
            GK(gkoc_arena_class_of(P.col, &cc));
            GK(gkoc_arena_class_of(P.y, &cy));
            GK(gkoc_arena_class_of(P.blk, &cb));
            GK(gkoc_arena_class_of(P.z, &cz));
            GK(gkoc_arena_class_of(P.rp, &cr));
            char tag[96];
            snprintf(tag, 96, "S7 arena mode %d %s #%d", amode, variant == 2 ? "roles" : "auto", t);
            measure(tag, P, t == 0);
            printf("   classes: val %d col %d rp %d y %d blk %d z %d; allocation + copies took %.3f s\n", cv,
                   cc, cr, cy, cb, cz, secs);
        }
        gkoc_arena_info info;
        GK(gkoc_arena_stats(&info));
        printf("   arena: %d classes, %lld chunks, %.2f GiB reserved, %.2f GiB used, %lld allocations, %lld probes, %lld granules walked; per class reserved %.0f/%.0f/%.0f GiB\n",
               info.num_classes, (long long)info.num_chunks, info.reserved_bytes / double(GiB),
               info.used_bytes / double(GiB), (long long)info.num_allocations, (long long)info.probes,
               (long long)info.granules_walked, info.class_reserved_bytes[0] / double(GiB),
               info.class_reserved_bytes[1] / double(GiB), info.class_reserved_bytes[2] / double(GiB));
        for (void* p : all) GK(gkoc_free(p));
        GK(gkoc_arena_trim());
        GK(gkoc_arena_configure(0, 0, 1));
    }
    measure("master at the end", M, false);
    return 0;
}
