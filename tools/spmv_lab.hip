// spmv_lab: measurement harness for CSR SpMV kernel variants on the 27-pt
// Laplacian (development tool, not part of the library).  Builds the matrix
// on the device with the library's generator, times each variant with HIP
// events and prints algorithmic GB/s; checks every variant against the
// production kernel's output.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude \
//       -Iginkgo_amd/csrc tools/spmv_lab.hip ginkgo_amd/csrc/runtime.hip \
//       ginkgo_amd/csrc/stencil.hip -o tools/spmv_lab
//   tools/spmv_lab [grid=256] [reps=20]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../ginkgo_amd/csrc/csr_spmv.hip"  // production kernel + C ABI entry
#include "lab_kernels.hpp"

using namespace gkoc;

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e),    \
                   __FILE__, __LINE__);                                \
            exit(1);                                                   \
        }                                                              \
    } while (0)

// ---------------------------------------------------------------- ceilings
__global__ __launch_bounds__(256) void stream_read_kernel(
    int64_t nnz, const double* __restrict__ vals, const int* __restrict__ cols,
    double* __restrict__ out)
{
    double acc = 0;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < nnz; i += stride) {
        acc += vals[i] * double(cols[i]);
    }
    if (acc == 12345.678) out[0] = acc;
}

__global__ __launch_bounds__(256) void stream_read16_kernel(
    int64_t n16, const double2* __restrict__ a, const int4* __restrict__ b,
    int64_t nb, double* __restrict__ out)
{
    double acc = 0;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += stride) {
        const double2 v = a[i];
        acc += v.x + v.y;
        if (i < nb) {
            const int4 c = b[i];
            acc += double(c.x + c.y + c.z + c.w);
        }
    }
    if (acc == 12345.678) out[0] = acc;
}

__global__ __launch_bounds__(256) void copy16_kernel(int64_t n16,
                                                     const double2* __restrict__ a,
                                                     double2* __restrict__ b)
{
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += stride) {
        b[i] = a[i];
    }
}

// every wave streams its own contiguous 1728-nonzero range (the access pattern
// of the row-segment kernels) with all of its loads in flight at once
template <int NPW>
__global__ __launch_bounds__(64) void stream_private_kernel(
    int64_t nnz, const double* __restrict__ vals, const int* __restrict__ cols,
    double* __restrict__ out)
{
    const int lane = threadIdx.x;
    const int64_t base = int64_t(blockIdx.x) * NPW;
    constexpr int IT = (NPW + 255) / 256;
    double2 v0[IT], v1[IT];
    int4 c[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int64_t k = base + i * 256 + lane * 4;
        const bool in = i * 256 + lane * 4 < NPW && k + 4 <= nnz;
        const int64_t kk = in ? k : base;
        v0[i] = *reinterpret_cast<const double2*>(vals + kk);
        v1[i] = *reinterpret_cast<const double2*>(vals + kk + 2);
        c[i] = *reinterpret_cast<const int4*>(cols + kk);
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        acc += v0[i].x + v0[i].y + v1[i].x + v1[i].y + double(c[i].x + c[i].y + c[i].z + c[i].w);
    }
    if (acc == 12345.678) out[0] = acc;
}

// same, plus the SpMV's output pattern: each wave stores 64 consecutive doubles
template <int NPW, int MODE>
__global__ __launch_bounds__(64) void stream_private_store_kernel(
    int64_t nnz, const double* __restrict__ vals, const int* __restrict__ cols,
    const double* __restrict__ bvec, const int* __restrict__ rp, double* __restrict__ out)
{
    const int lane = threadIdx.x;
    const int64_t base = int64_t(blockIdx.x) * NPW;
    constexpr int IT = (NPW + 255) / 256;
    double2 v0[IT], v1[IT];
    int4 c[IT];
    int rpv = 0;
    if (MODE & 2) rpv = rp[int64_t(blockIdx.x) * 64 + lane];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int64_t k = base + i * 256 + lane * 4;
        const bool in = i * 256 + lane * 4 < NPW && k + 4 <= nnz;
        const int64_t kk = in ? k : 0;
        v0[i] = *reinterpret_cast<const double2*>(vals + kk);
        v1[i] = *reinterpret_cast<const double2*>(vals + kk + 2);
        c[i] = *reinterpret_cast<const int4*>(cols + kk);
    }
    double acc = double(rpv);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        acc += v0[i].x + v0[i].y + v1[i].x + v1[i].y + double(c[i].x + c[i].y + c[i].z + c[i].w);
    }
    if (MODE & 4) acc += bvec[int64_t(blockIdx.x) * 64 + lane];
    constexpr int FL = (MODE >> 4) & 15;   // store flavour
    constexpr int CL = MODE >> 8;          // clustering: one wave in 2^CL stores 2^CL * 512 B
    if (CL > 0) {
        if (MODE & 1) {
            if ((blockIdx.x & ((1 << CL) - 1)) == 0) {
                double* d0 = out + int64_t(blockIdx.x) * 64;
                if (FL == 9) {
                    // the same bytes with 16 B per lane (one 1 KB store instruction each)
                    for (int i = 2 * lane; i < (64 << CL); i += 128) {
                        *reinterpret_cast<double2*>(d0 + i) = make_double2(acc, acc);
                    }
                } else {
                    for (int i = lane; i < (64 << CL); i += 64) d0[i] = acc;
                }
            }
        } else if (acc == 12345.678) out[0] = acc;
        return;
    }
    double* dst = out + int64_t(blockIdx.x) * 64 + lane;
    if (MODE & 1) {
        if (FL == 0) *dst = acc;
        else if (FL == 1) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(dst), "v"(acc) : "memory");
        else if (FL == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(dst), "v"(acc) : "memory");
        else if (FL == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(acc) : "memory");
        else if (FL == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(acc) : "memory");
        else if (FL == 5) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(acc) : "memory");
        else if (FL == 6) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(dst), "v"(acc) : "memory");
        else if (FL == 7) asm volatile("global_store_dwordx2 %0, %1, off sc0 nt" ::"v"(dst), "v"(acc) : "memory");
    } else if (acc == 12345.678) out[0] = acc;
}

// four waves per block, each streams its own range; with SYNC the four 512 B
// stores of a block are issued together after a barrier (2 KB contiguous)
template <int NPW, bool SYNC>
__global__ __launch_bounds__(256) void stream_private_store_block4_kernel(
    int64_t nnz, const double* __restrict__ vals, const int* __restrict__ cols,
    double* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int64_t base = wave * NPW;
    constexpr int IT = (NPW + 255) / 256;
    double2 v0[IT], v1[IT];
    int4 c[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int64_t k = base + i * 256 + lane * 4;
        const bool in = i * 256 + lane * 4 < NPW && k + 4 <= nnz;
        const int64_t kk = in ? k : 0;
        v0[i] = *reinterpret_cast<const double2*>(vals + kk);
        v1[i] = *reinterpret_cast<const double2*>(vals + kk + 2);
        c[i] = *reinterpret_cast<const int4*>(cols + kk);
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        acc += v0[i].x + v0[i].y + v1[i].x + v1[i].y + double(c[i].x + c[i].y + c[i].z + c[i].w);
    }
    if (SYNC) __syncthreads();
    out[wave * 64 + lane] = acc;
}

// ------------------------------------------------- classical (Ginkgo-like)
// SUB lanes per row, shuffle reduction (common/cuda_hip csr classical idea)
template <int SUB>
__global__ __launch_bounds__(256) void csr_classical_kernel(
    int64_t n_rows, const int* __restrict__ row_ptrs, const int* __restrict__ cols,
    const double* __restrict__ vals, const double* __restrict__ b,
    double* __restrict__ c)
{
    const int64_t gid = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t row = gid / SUB;
    const int sl = threadIdx.x % SUB;
    if (row >= n_rows) return;
    double sum = 0;
    for (int64_t k = row_ptrs[row] + sl; k < row_ptrs[row + 1]; k += SUB) {
        sum += vals[k] * b[cols[k]];
    }
#pragma unroll
    for (int off = SUB / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (sl == 0) c[row] = sum;
}

// ----------------------------- variant: LDS-staged matrix, row-ordered gather
// stage val/col of the 64-row segment in LDS (coalesced), then lane = row walks
// its own entries: the gather of b is row-ordered (consecutive lanes hit
// consecutive b entries for stencil-like matrices).
template <bool REMAP, int UNROLL>
__global__ __launch_bounds__(64) void csr_spmv_stage_kernel(
    int64_t n_rows, int64_t n_segments, const int* __restrict__ row_ptrs,
    const int* __restrict__ cols, const double* __restrict__ vals,
    const double* __restrict__ b, double* __restrict__ c)
{
    constexpr int CAP = 1792;
    __shared__ double lv[CAP];
    __shared__ int lc[CAP];
    const int lane = threadIdx.x;
    const int64_t seg = REMAP ? xcd_band_remap(blockIdx.x, n_segments) : int64_t(blockIdx.x);
    const int64_t r0 = seg * 64;
    const int64_t row = r0 + lane;
    const bool valid = row < n_rows;
    const int64_t r_last = (r0 + 64 < n_rows) ? r0 + 64 : n_rows;
    const int64_t rs = row_ptrs[valid ? row : r_last];
    const int64_t re = row_ptrs[valid ? row + 1 : r_last];
    const int64_t k0 = __shfl(rs, 0, 64);
    const int64_t k1 = __shfl(re, int(r_last - r0 - 1), 64);
    double sum = 0;
    for (int64_t t0 = k0; t0 < k1; t0 += CAP) {
        const int64_t t1 = (t0 + CAP < k1) ? t0 + CAP : k1;
        for (int64_t base = t0; base < t1; base += 64 * UNROLL) {
            double v[UNROLL];
            int cc[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                int64_t k = base + u * 64 + lane;
                k = k < t1 ? k : t1 - 1;
                v[u] = vals[k];
                cc[u] = cols[k];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t k = base + u * 64 + lane;
                if (k < t1) {
                    lv[k - t0] = v[u];
                    lc[k - t0] = cc[u];
                }
            }
        }
        wave_lds_sync();
        const int64_t a = rs > t0 ? rs : t0;
        const int64_t e = re < t1 ? re : t1;
        int64_t k = a;
        for (; k + 4 <= e; k += 4) {
            double xv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = b[lc[k + u - t0]];
                vv[u] = lv[k + u - t0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sum += vv[u] * xv[u];
        }
        for (; k < e; ++k) sum += lv[k - t0] * b[lc[k - t0]];
        wave_lds_sync();
    }
    if (valid) c[row] = sum;
}

struct timer {
    hipEvent_t a, b;
    timer()
    {
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
    }
    template <typename F>
    double ms(int reps, F f)
    {
        f();
        f();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float t;
        CK(hipEventElapsedTime(&t, a, b));
        return double(t) / reps;
    }
};



int main(int argc, char** argv)
{
    const int64_t g = argc > 1 ? atoll(argv[1]) : 256;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const bool pmc_mode = argc > 3 && !strcmp(argv[3], "pmc");
    const int64_t n = g * g * g;
    int* row_ptrs;
    CK(hipMalloc(&row_ptrs, sizeof(int) * (n + 1)));
    int64_t nnz = 0;
    if (gkoc_stencil_row_ptrs_i32(nullptr, 3, g, 0, 0, g, row_ptrs, &nnz)) {
        printf("generator failed: %s\n", gkoc_last_error());
        return 1;
    }
    int* cols;
    double *vals, *x, *y, *yref;
    CK(hipMalloc(&cols, sizeof(int) * nnz));
    CK(hipMalloc(&vals, sizeof(double) * nnz));
    CK(hipMalloc(&x, sizeof(double) * n));
    CK(hipMalloc(&y, sizeof(double) * n));
    CK(hipMalloc(&yref, sizeof(double) * n));
    if (gkoc_stencil_fill_f64_i32(nullptr, 3, g, 0, 0, g, row_ptrs, cols, vals)) return 1;
    std::vector<double> hx(n);
    unsigned long long s = 42;
    for (int64_t i = 0; i < n; ++i) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        hx[i] = double(s >> 11) / 9007199254740992.0 * 2 - 1;
    }
    CK(hipMemcpy(x, hx.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    const double bytes = double(nnz) * 12 + double(n + 1) * 4 + double(n) * 16;
    printf("27-pt %ld^3: n=%ld nnz=%ld algorithmic bytes=%.3f GB\n", (long)g, (long)n,
           (long)nnz, bytes / 1e9);
    timer T;
    const int64_t nseg = (n + 63) / 64;
    auto report = [&](const char* name, double ms, double by) {
        printf("%-44s %8.4f ms  %8.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms,
               by / ms / 1e6, by / ms / 1e6 / 80.0);
        fflush(stdout);
    };
    std::vector<double> href(n), hy(n);
    auto check = [&](const char* name, bool exact) {
        CK(hipMemcpy(hy.data(), y, sizeof(double) * n, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        double maxerr = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (hy[i] != href[i]) ++bad;
            const double e = fabs(hy[i] - href[i]);
            if (e > maxerr) maxerr = e;
        }
        printf("    check %-34s %s (mismatching=%ld maxerr=%.3e)\n", name,
               (exact ? bad == 0 : maxerr < 1e-11) ? "OK" : "FAILED", (long)bad, maxerr);
    };

    // ceilings
    if (!pmc_mode) {
        const int64_t n16 = nnz / 2, nb = nnz / 4;
        double ms = T.ms(reps, [&] {
            stream_read16_kernel<<<2048, 256>>>(n16, (const double2*)vals,
                                                (const int4*)cols, nb, y);
        });
        report("ceiling: read val+col, 16B loads", ms, double(nnz) * 12);
        ms = T.ms(reps, [&] { stream_read_kernel<<<2048, 256>>>(nnz, vals, cols, y); });
        report("ceiling: read val+col, 8B+4B loads", ms, double(nnz) * 12);
        ms = T.ms(reps, [&] {
            stream_read_kernel<<<8192, 256>>>(nnz, vals, cols, y);
        });
        report("ceiling: read val+col, 8B+4B, 8192 blocks", ms, double(nnz) * 12);
        ms = T.ms(reps, [&] {
            stream_private_kernel<1728><<<unsigned(nnz / 1728), 64>>>(nnz, vals, cols, y);
        });
        report("ceiling: per-wave private 1728-nnz ranges", ms, double(nnz) * 12);
        ms = T.ms(reps, [&] {
            stream_private_kernel<864><<<unsigned(nnz / 864), 64>>>(nnz, vals, cols, y);
        });
        report("ceiling: per-wave private 864-nnz ranges", ms, double(nnz) * 12);
        ms = T.ms(reps, [&] {
            stream_private_kernel<3456><<<unsigned(nnz / 3456), 64>>>(nnz, vals, cols, y);
        });
        report("ceiling: per-wave private 3456-nnz ranges", ms, double(nnz) * 12);
        const int64_t c16 = nnz / 4;  // copy half of vals into the other half
        ms = T.ms(reps, [&] {
            copy16_kernel<<<2048, 256>>>(c16, (const double2*)vals,
                                         (double2*)(vals + nnz / 2));
        });
        report("ceiling: copy 16B (read+write)", ms, double(c16) * 32);
        CK(hipDeviceSynchronize());
        // restore vals
        if (gkoc_stencil_fill_f64_i32(nullptr, 3, g, 0, 0, g, row_ptrs, cols, vals)) return 1;
    }

    // production kernel (through the C ABI entry point)
    double ms = T.ms(reps, [&] {
        gkoc_csr_spmv_f64_i32(nullptr, n, n, row_ptrs, cols, vals, x, 1, yref, 1, 1);
    });
    report("PRODUCTION gkoc_csr_spmv_f64_i32", ms, bytes);
    CK(hipMemcpy(href.data(), yref, sizeof(double) * n, hipMemcpyDeviceToHost));
    // the output allocation changes the time of every variant
    // by up to 15 %.  Variants are compared on the FASTEST of 40 candidate
    // output buffers ("Y+") and, for contrast, on the slowest ("Y-").
    double* y_fast = y;
    double* y_slow = y;
    if (!pmc_mode) {
        double tmin = 1e9, tmax = 0;
        for (int k = 0; k < 40; ++k) {
            double* yc;
            CK(hipMalloc(&yc, sizeof(double) * n));
            const double t = T.ms(4, [&] {
                gkoc_csr_spmv_f64_i32(nullptr, n, n, row_ptrs, cols, vals, x, 1, yc, 1, 1);
            });
            if (t < tmin) { tmin = t; y_fast = yc; }
            if (t > tmax) { tmax = t; y_slow = yc; }
        }
        printf("output-buffer candidates: fastest %.4f ms, slowest %.4f ms\n", tmin, tmax);
        fflush(stdout);
    }

    if (!pmc_mode) {
        // minimal reproducer: private stream reads (+ row_ptrs, + b[row]) + the y store
        const unsigned nwv = unsigned(n / 64);
        for (int which = 0; which < 2; ++which) {
            double* yy = which ? y_slow : y_fast;
            const char* tag = which ? "Y-" : "Y+";
            char nm[96];
            ms = T.ms(reps, [&] { stream_private_store_kernel<1728, 0><<<nwv, 64>>>(nnz, vals, cols, x, row_ptrs, yy); });
            snprintf(nm, 96, "repro %s: stream only", tag); report(nm, ms, double(nnz) * 12);
            ms = T.ms(reps, [&] { stream_private_store_kernel<1728, 1><<<nwv, 64>>>(nnz, vals, cols, x, row_ptrs, yy); });
            snprintf(nm, 96, "repro %s: stream + y store", tag); report(nm, ms, double(nnz) * 12 + 8.0 * n);
            ms = T.ms(reps, [&] { stream_private_store_kernel<1728, 3><<<nwv, 64>>>(nnz, vals, cols, x, row_ptrs, yy); });
            snprintf(nm, 96, "repro %s: stream + rp + y store", tag); report(nm, ms, double(nnz) * 12 + 12.0 * n);
            ms = T.ms(reps, [&] { stream_private_store_kernel<1728, 7><<<nwv, 64>>>(nnz, vals, cols, x, row_ptrs, yy); });
            snprintf(nm, 96, "repro %s: stream + rp + b[row] + y store", tag); report(nm, ms, double(nnz) * 12 + 20.0 * n);
#define REPRO_FL(FL, NAME)                                                                  \
    ms = T.ms(reps, [&] { stream_private_store_kernel<1728, 1 + 16 * FL><<<nwv, 64>>>(nnz, vals, cols, x, row_ptrs, yy); }); \
    snprintf(nm, 96, "repro %s: stream + y store [" NAME "]", tag); report(nm, ms, double(nnz) * 12 + 8.0 * n);
            REPRO_FL(1, "nt")
            ms = T.ms(reps, [&] { stream_private_store_block4_kernel<1728, false><<<nwv / 4, 256>>>(nnz, vals, cols, yy); });
            snprintf(nm, 96, "repro %s: 4 waves/block, 512 B each, no barrier", tag); report(nm, ms, double(nnz) * 12 + 8.0 * n);
            ms = T.ms(reps, [&] { stream_private_store_block4_kernel<1728, true><<<nwv / 4, 256>>>(nnz, vals, cols, yy); });
            snprintf(nm, 96, "repro %s: 4 waves/block, barrier, 2 KB together", tag); report(nm, ms, double(nnz) * 12 + 8.0 * n);
            REPRO_FL(16, "1 wave in 2 stores 1 KB")
            REPRO_FL(25, "1 in 2: 1 KB, one dwordx4 store")
            REPRO_FL(41, "1 in 4: 2 KB, two dwordx4 stores")
            REPRO_FL(32, "1 in 4: 2 KB")
            REPRO_FL(48, "1 in 8: 4 KB")
            REPRO_FL(64, "1 in 16: 8 KB")
            REPRO_FL(96, "1 in 64: 32 KB")
            REPRO_FL(128, "1 in 256: 128 KB")
            ms = T.ms(reps, [&] {
                gkoc_csr_spmv_f64_i32(nullptr, n, n, row_ptrs, cols, vals, x, 1, yy, 1, 1);
            });
            snprintf(nm, 96, "repro %s: full SpMV", tag); report(nm, ms, bytes);
        }
    }
    if (pmc_mode) {
        ms = T.ms(reps, [&] { stream_read_kernel<<<2048, 256>>>(nnz, vals, cols, y); });
        report("ceiling: read val+col, 8B+4B loads", ms, double(nnz) * 12);
        goto pmc_runs;
    }
#define RUN_WAVE(REMAP, U)                                                       \
    {                                                                            \
        ms = T.ms(reps, [&] {                                                    \
            csr_spmv_wave_kernel<double, int, false, REMAP, U>                   \
                <<<dim3(unsigned(nseg)), dim3(64)>>>(n, nseg, row_ptrs, cols,    \
                                                     vals, x, 1, y, 1, 1,        \
                                                     nullptr, nullptr);          \
        });                                                                      \
        report("wave kernel remap=" #REMAP " unroll=" #U, ms, bytes);            \
        check("wave " #REMAP " " #U, true);                                      \
    }
    RUN_WAVE(false, 9)
    RUN_WAVE(true, 9)

#define RUN_STAGE(REMAP, U)                                                      \
    {                                                                            \
        ms = T.ms(reps, [&] {                                                    \
            csr_spmv_stage_kernel<REMAP, U><<<dim3(unsigned(nseg)), dim3(64)>>>( \
                n, nseg, row_ptrs, cols, vals, x, y);                            \
        });                                                                      \
        report("stage kernel (row-ordered gather) remap=" #REMAP " u=" #U, ms,   \
               bytes);                                                           \
        check("stage " #REMAP " " #U, true);                                     \
    }
    RUN_STAGE(false, 9)

#define RUN_CLASSICAL(SUB)                                                       \
    {                                                                            \
        ms = T.ms(reps, [&] {                                                    \
            csr_classical_kernel<SUB>                                            \
                <<<dim3(unsigned((n * SUB + 255) / 256)), dim3(256)>>>(          \
                    n, row_ptrs, cols, vals, x, y);                              \
        });                                                                      \
        report("classical subwave=" #SUB, ms, bytes);                            \
        check("classical " #SUB, false);                                         \
    }

#define RUN_PIPE(ROWS, E, U, RING, SPW, ABL)                                     \
    {                                                                            \
        const int64_t nsg = (n + ROWS - 1) / ROWS;                               \
        const int64_t spw = SPW > 0 ? SPW : (nsg + (-SPW) * 256 - 1) / ((-SPW) * 256); \
        const int64_t nw = (nsg + spw - 1) / spw;                                \
        ms = T.ms(reps, [&] {                                                    \
            csr_spmv_pipe_kernel<double, int, false, ROWS, E, U, RING, ABL>      \
                <<<dim3(unsigned(nw)), dim3(64)>>>(n, nsg, spw, row_ptrs, cols,  \
                                                   vals, x, 1, y, 1, 1, nullptr, \
                                                   nullptr);                     \
        });                                                                      \
        report("pipe rows=" #ROWS " E=" #E " U=" #U " ring=" #RING " spw=" #SPW  \
               " abl=" #ABL, ms, bytes);                                         \
        if (!(ABL & 3)) check("pipe", true);                                    \
    }
    // SPW < 0: persistent, -SPW waves per CU
    if (false) {
    pmc_runs:
        RUN_PIPE(32, 4, 1, 1024, 2, 0)
        RUN_PIPE(32, 4, 1, 1024, 2, 4)
        RUN_PIPE(32, 4, 1, 1024, 2, 8)
        RUN_PIPE(32, 4, 1, 1024, 2, 12)
        return 0;
    }
#define RUN_PIPE2(ROWS, E, U, RING, GB, SPW, ABL)                                \
    {                                                                            \
        const int64_t nsg = (n + ROWS - 1) / ROWS;                               \
        const int64_t spw = SPW;                                                 \
        const int64_t nw = (nsg + spw - 1) / spw;                                \
        ms = T.ms(reps, [&] {                                                    \
            csr_spmv_pipe2_kernel<double, int, false, ROWS, E, U, RING, GB, ABL> \
                <<<dim3(unsigned(nw)), dim3(64)>>>(n, nsg, spw, row_ptrs, cols,  \
                                                   vals, x, 1, y, 1, 1, nullptr, \
                                                   nullptr);                     \
        });                                                                      \
        report("pipe2 rows=" #ROWS " E=" #E " U=" #U " ring=" #RING " gb=" #GB   \
               " spw=" #SPW " abl=" #ABL, ms, bytes);                            \
        if (!(ABL & 3)) check("pipe2", true);                                    \
    }
#define RUN_PIPE3(ROWS, E, U, RING, WPS, SPW, ABL)                               \
    {                                                                            \
        const int64_t nsg = (n + ROWS - 1) / ROWS;                               \
        const int64_t spw = SPW;                                                 \
        const int64_t nw = (nsg + spw - 1) / spw;                                \
        ms = T.ms(reps, [&] {                                                    \
            csr_spmv_pipe3_kernel<double, int, false, ROWS, E, U, RING, WPS, ABL> \
                <<<dim3(unsigned(nw)), dim3(64)>>>(n, nsg, spw, row_ptrs, cols,  \
                                                   vals, x, 1, y, 1, 1, nullptr, \
                                                   nullptr);                     \
        });                                                                      \
        report("pipe3 rows=" #ROWS " E=" #E " U=" #U " ring=" #RING " wps=" #WPS \
               " spw=" #SPW " abl=" #ABL, ms, bytes);                            \
        if (!((ABL) & 3)) check("pipe3", true);                                    \
    }
#define RUN_PAIR(ROWS, E, U, NSETS, RING, GB, SPW, ABL)                          \
    {                                                                            \
        const int64_t nsg = (n + ROWS - 1) / ROWS;                               \
        const int64_t spw = SPW;                                                 \
        const int64_t nw = (nsg + spw - 1) / spw;                                \
        ms = T.ms(reps, [&] {                                                    \
            csr_spmv_pair_kernel<double, int, ROWS, E, U, NSETS, RING, GB, ABL>  \
                <<<dim3(unsigned(nw)), dim3(128)>>>(n, nsg, spw, row_ptrs, cols, \
                                                    vals, x, y);                 \
        });                                                                      \
        report("pair rows=" #ROWS " E=" #E " U=" #U " sets=" #NSETS " ring=" #RING \
               " gb=" #GB " spw=" #SPW " abl=" #ABL, ms, bytes);                 \
        if (!(ABL & 3)) check("pair", true);                                     \
    }
#define RUN_PIPE5(RING, ABL)                                                     \
    {                                                                            \
        const int64_t nsg = (n + 63) / 64;                                       \
        const int64_t nw = (nsg + 1) / 2;                                        \
        ms = T.ms(reps, [&] {                                                    \
            csr_spmv_pipe5_kernel<RING, ABL>                                     \
                <<<dim3(unsigned(nw)), dim3(64)>>>(n, nsg, 2, row_ptrs, cols,    \
                                                   vals, x, y);                  \
        });                                                                      \
        report("pipe5 (3 sets, counted waits) ring=" #RING " abl=" #ABL, ms, bytes); \
        if (!((ABL) & 3)) check("pipe5", true);                                  \
    }
    for (int rep = 0; rep < 2; ++rep) {
    y = rep == 0 ? y_fast : y_slow;
    printf("--- variants on the %s output buffer\n", rep == 0 ? "FASTEST (Y+)" : "SLOWEST (Y-)");
    RUN_PIPE3(64, 4, 1, 1024, 1, 2, 0x2000)
    RUN_PIPE3(64, 4, 1, 1024, 1, 1, 0x1000)
    RUN_PIPE3(64, 4, 1, 1024, 1, 1, 0)
    RUN_PIPE3(32, 4, 1, 1024, 1, 2, 0x2000)
    RUN_PIPE3(32, 4, 1, 1024, 1, 1, 0)
    RUN_PIPE3(64, 4, 1, 1024, 1, 4, 0x4000)
    RUN_PIPE3(64, 4, 1, 1024, 1, 2, 0x2000)
    }
    // ELL / SELL-P through the library entry points (formats built on device)
    {
        int64_t k = 0;
        gkoc_compute_max_row_nnz_i32(nullptr, n, row_ptrs, &k);
        int* ecols;
        double* evals;
        CK(hipMalloc(&ecols, sizeof(int) * k * n));
        CK(hipMalloc(&evals, sizeof(double) * k * n));
        gkoc_csr_convert_to_ell_f64_i32(nullptr, n, row_ptrs, cols, vals, k, n, ecols, evals);
        ms = T.ms(reps, [&] { gkoc_ell_spmv_f64_i32(nullptr, n, n, k, n, ecols, evals, x, 1, y, 1, 1); });
        report("ELL spmv (library), bytes = 12*k*n + 16n", ms, double(k) * n * 12 + 16.0 * n);
        check("ell", true);
        CK(hipFree(ecols));
        CK(hipFree(evals));
        const int64_t ns = (n + 63) / 64;
        uint64_t *sets, *lens;
        CK(hipMalloc(&sets, 8 * (ns + 1)));
        CK(hipMalloc(&lens, 8 * ns));
        gkoc_sellp_compute_slice_sets_i32(nullptr, n, 64, 1, row_ptrs, sets, lens);
        uint64_t total = 0;
        CK(hipMemcpy(&total, sets + ns, 8, hipMemcpyDeviceToHost));
        int* scols;
        double* svals;
        CK(hipMalloc(&scols, sizeof(int) * total * 64));
        CK(hipMalloc(&svals, sizeof(double) * total * 64));
        gkoc_csr_convert_to_sellp_f64_i32(nullptr, n, 64, row_ptrs, cols, vals, sets, scols, svals);
        ms = T.ms(reps, [&] {
            gkoc_sellp_spmv_f64_i32(nullptr, n, n, 64, sets, lens, scols, svals, x, 1, y, 1, 1);
        });
        report("SELL-P spmv (library), bytes = 12*stored + 16n", ms, double(total) * 64 * 12 + 16.0 * n);
        check("sellp", true);
    }
    RUN_CLASSICAL(8)
    return 0;
}
