// place_lab3 (development tool, round 2): inside ONE large device allocation the
// relative physical offsets of the arrays are under our control.  Sweep the offset
// of one array at a time (GiB steps over the chunk, then finer) and record the SpMV
// and block-Jacobi times: which distances between the read streams and the written
// vector are fast?
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/place_lab3.hip \
//     -Lginkgo_amd/lib -lgko_cdna4 -Wl,-rpath,'$ORIGIN/../ginkgo_amd/lib' -o tools/place_lab3
// usage: place_lab3 [chunk GiB = 96] [kind = vmm|malloc|contig] [reps = 4]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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
#define GK(x)                                                                                   \
    do {                                                                                        \
        int r_ = (x);                                                                           \
        if (r_ != 0) {                                                                          \
            printf("gkoc error %d (%s) at %s:%d\n", r_, gkoc_last_error(), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
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

int main(int argc, char** argv)
{
    const size_t chunk_gib = argc > 1 ? size_t(atoll(argv[1])) : 96;
    const std::string kind = argc > 2 ? argv[2] : "vmm";
    const int reps = argc > 3 ? atoi(argv[3]) : 4;
    const size_t base_gib = argc > 4 ? size_t(atoll(argv[4])) : 0;   // where the base layout sits
    const bool wide = argc > 5 && !strcmp(argv[5], "wide");          // only y / valcol / z, whole chunk
    const int64_t g = 256, n = g * g * g;
    timer T;
    GK(gkoc_arena_configure(0, 0, 1));
    GK(gkoc_tune_set(GKOC_TUNE_CSR_XCD_MAP, 0));
    GK(gkoc_tune_set(GKOC_TUNE_JACOBI_XCD_MAP, 0));

    const size_t chunk = chunk_gib * GiB;
    char* base = nullptr;
    if (kind == "vmm") {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        void* va;
        CK(hipMemAddressReserve(&va, chunk, size_t(64) * GiB, nullptr, 0));
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap(va, chunk, 0, h, 0));
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, chunk, &acc, 1));
        base = (char*)va;
    } else if (kind == "contig") {
        CK(hipExtMallocWithFlags((void**)&base, chunk, hipDeviceMallocContiguous));
    } else {
        CK(hipMalloc(&base, chunk));
    }
    printf("chunk %zu GiB (%s) at %p\n", chunk_gib, kind.c_str(), (void*)base);

    // sizes
    int64_t nnz = 0;
    int* rp0;
    CK(hipMalloc(&rp0, sizeof(int) * (n + 1)));
    GK(gkoc_stencil_row_ptrs_i32(nullptr, 3, g, 0, 0, g, rp0, &nnz));
    const int64_t nblk = n / 8;
    const size_t s_val = sizeof(double) * nnz, s_col = sizeof(int) * nnz, s_vec = sizeof(double) * n,
                 s_rp = sizeof(int) * (n + 1), s_blk = sizeof(double) * 64 * nblk;
    const double spmv_bytes = double(nnz) * 12 + double(n + 1) * 4 + double(n) * 16;
    const double jac_bytes = double(n) * 80 + 4.0 * (nblk + 1);
    const gkoc_jacobi_scheme scheme{8, 512, 3};

    // base layout (offsets in MiB): val 0, col 4096, rp 6144, b 6400, y 6656, blk 7168 (1 GiB),
    // bp 8448, z 8704; everything below 9 GiB
    struct layout {
        size_t val, col, rp, b, y, blk, bp, z;
    };
    const size_t B0 = base_gib * GiB;
    const layout L0{B0, B0 + 4096 * MiB, B0 + 6144 * MiB, B0 + 6400 * MiB, B0 + 6656 * MiB,
                    B0 + 7168 * MiB, B0 + 8448 * MiB, B0 + 8704 * MiB};
    // master data at the base layout
    {
        double* val = (double*)(base + L0.val);
        int* col = (int*)(base + L0.col);
        int* rp = (int*)(base + L0.rp);
        CK(hipMemcpy(rp, rp0, s_rp, hipMemcpyDeviceToDevice));
        GK(gkoc_stencil_fill_f64_i32(nullptr, 3, g, 0, 0, g, rp, col, val));
        std::vector<double> hb(n);
        unsigned long long s2 = 42;
        for (int64_t i = 0; i < n; ++i) {
            s2 = s2 * 6364136223846793005ULL + 1442695040888963407ULL;
            hb[i] = double(s2 >> 11) / 9007199254740992.0 * 2 - 1;
        }
        CK(hipMemcpy(base + L0.b, hb.data(), s_vec, hipMemcpyHostToDevice));
        seq_kernel<<<unsigned((nblk + 1 + 255) / 256), 256>>>(nblk + 1, 8, (int*)(base + L0.bp));
        CK(hipMemcpy(base + L0.blk, val, s_blk, hipMemcpyDeviceToDevice));
        CK(hipDeviceSynchronize());
    }
    std::vector<double> yref(n), ytmp(n);
    auto spmv = [&](const layout& L) {
        GK(gkoc_csr_spmv_f64_i32(nullptr, n, n, (int*)(base + L.rp), (int*)(base + L.col),
                                 (double*)(base + L.val), (double*)(base + L.b), 1,
                                 (double*)(base + L.y), 1, 1));
    };
    auto jac = [&](const layout& L) {
        GK(gkoc_jacobi_simple_apply_f64_i32(nullptr, nblk, 8, scheme, (int*)(base + L.bp),
                                            (double*)(base + L.blk), (double*)(base + L.b), 1,
                                            (double*)(base + L.z), 1, 1));
    };
    spmv(L0);
    CK(hipMemcpy(yref.data(), base + L0.y, s_vec, hipMemcpyDeviceToHost));
    auto move = [&](size_t from, size_t to, size_t bytes) {
        if (from != to) CK(hipMemcpy(base + to, base + from, bytes, hipMemcpyDeviceToDevice));
    };
    auto read_us = [&]() {
        return T.us(reps, [&] {
            stream_read_kernel<<<2048, 256>>>(nnz, (double*)(base + L0.val), (int*)(base + L0.col),
                                              (double*)(base + L0.z));
        });
    };
    double base_spmv = T.us(reps, [&] { spmv(L0); });
    double base_jac = T.us(reps, [&] { jac(L0); });
    printf("base layout: spmv %.1f us (%.1f%%), jacobi %.1f us (%.1f%%), read %.1f us\n", base_spmv,
           spmv_bytes / base_spmv / 8e4, base_jac, jac_bytes / base_jac / 8e4, read_us());

    const size_t first_free = 9 * GiB;
    auto sweep = [&](const char* what, size_t start, size_t step, int count) {
        printf("-- sweep %s: start %.3f GiB step %.3f GiB; columns: offset GiB, spmv us, jacobi us (base now: spmv/jac)\n",
               what, start / double(GiB), step / double(GiB));
        for (int k = 0; k < count; ++k) {
            const size_t off = start + step * k;
            layout L = L0;
            const size_t bytes = !strcmp(what, "col")      ? s_col
                                 : !strcmp(what, "val")    ? s_val
                                 : !strcmp(what, "blk")    ? s_blk
                                 : !strcmp(what, "valcol") ? 4096 * MiB + s_col
                                                           : s_vec;
            if (off + bytes > chunk) break;
            if (off < B0 + 9 * GiB && off + bytes > B0) continue;   // overlaps the base set
            if (!strcmp(what, "y")) {
                L.y = off;
            } else if (!strcmp(what, "z")) {
                L.z = off;
            } else if (!strcmp(what, "col")) {
                L.col = off;
                move(L0.col, off, s_col);
            } else if (!strcmp(what, "val")) {
                L.val = off;
                move(L0.val, off, s_val);
            } else if (!strcmp(what, "b")) {
                L.b = off;
                move(L0.b, off, s_vec);
            } else if (!strcmp(what, "blk")) {
                L.blk = off;
                move(L0.blk, off, s_blk);
            } else if (!strcmp(what, "valcol")) {
                // the whole matrix moves, vectors stay
                L.val = off;
                L.col = off + 4096 * MiB;
                move(L0.val, L.val, s_val);
                move(L0.col, L.col, s_col);
            }
            const double ts = T.us(reps, [&] { spmv(L); });
            const double tj = T.us(reps, [&] { jac(L); });
            if (k % 8 == 0) {
                CK(hipMemcpy(ytmp.data(), base + L.y, s_vec, hipMemcpyDeviceToHost));
                if (memcmp(ytmp.data(), yref.data(), s_vec) != 0) printf("!! wrong result\n");
                base_spmv = T.us(reps, [&] { spmv(L0); });
                base_jac = T.us(reps, [&] { jac(L0); });
                printf("   %8.3f  %7.1f  %6.1f   (base %7.1f / %6.1f)\n", off / double(GiB), ts, tj,
                       base_spmv, base_jac);
            } else {
                printf("   %8.3f  %7.1f  %6.1f\n", off / double(GiB), ts, tj);
            }
        }
        fflush(stdout);
    };
    if (wide) {
        sweep("y", 0, GiB, int(chunk_gib));
        sweep("z", 0, GiB, int(chunk_gib));
        sweep("valcol", 0, GiB, int(chunk_gib));
        sweep("col", 0, GiB, int(chunk_gib));
        printf("read at the end %.1f us\n", read_us());
        return 0;
    }
    const int ng = int(chunk_gib) - 9 - 6;
    sweep("y", first_free, GiB, ng);
    sweep("col", first_free, GiB, ng);
    sweep("val", first_free, GiB, ng);
    sweep("valcol", first_free, GiB, ng);
    sweep("z", first_free, GiB, ng);
    sweep("blk", first_free, GiB, ng);
    sweep("b", first_free, GiB, ng);
    // finer steps
    sweep("y", first_free, 64 * MiB, 48);
    sweep("y", first_free, 2 * MiB, 40);
    sweep("col", first_free, 64 * MiB, 48);
    printf("read at the end %.1f us\n", read_us());
    return 0;
}
