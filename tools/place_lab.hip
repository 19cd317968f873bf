// place_lab: how does the placement of the CSR arrays and the vectors in
// device memory change SpMV time?  (development tool)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude \
//     -Iginkgo_amd/csrc tools/place_lab.hip ginkgo_amd/csrc/runtime.hip \
//     ginkgo_amd/csrc/stencil.hip -o tools/place_lab
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define GKOC_LAB_TIMESTAMPS 1
__device__ unsigned long long* gkoc_lab_ts;
#include "../ginkgo_amd/csrc/csr_spmv.hip"

using namespace gkoc;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__,  \
                   __LINE__);                                                  \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

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

// dependent-load chain with a fixed byte stride: exposes translation latency
__global__ void chain_init_kernel(int64_t* buf, int64_t n_steps, int64_t stride_elems)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n_steps) buf[i * stride_elems] = ((i + 1) % n_steps) * stride_elems;
}
__global__ void chain_walk_kernel(const int64_t* buf, int64_t n_steps, int64_t* out)
{
    int64_t idx = 0;
    for (int64_t i = 0; i < n_steps; ++i) idx = __builtin_nontemporal_load(buf + idx);
    out[0] = idx;
}

// every lane touches one pseudo-random 4 KiB page of the buffer per step
__global__ __launch_bounds__(256) void tlb_stress_kernel(const double* buf, int64_t n_pages,
                                                         int steps, double* out)
{
    unsigned long long s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    double acc = 0;
    for (int i = 0; i < steps; ++i) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        const int64_t page = int64_t((s >> 20) % (unsigned long long)n_pages);
        acc += __builtin_nontemporal_load(buf + page * 512 + ((s >> 8) & 511));
    }
    if (acc == 1.2345e300) out[0] = acc;
}
__global__ __launch_bounds__(256) void tlb_stress_write_kernel(double* buf, int64_t n_pages,
                                                               int steps)
{
    unsigned long long s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    for (int i = 0; i < steps; ++i) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        const int64_t page = int64_t((s >> 20) % (unsigned long long)n_pages);
        buf[page * 512 + ((s >> 8) & 511)] = double(i);
    }
}

int main(int argc, char** argv)
{
    const int64_t g = argc > 1 ? atoll(argv[1]) : 256;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int64_t n = g * g * g;
    const size_t MB = size_t(1) << 20;
    timer T;
    // ---- A: separate hipMallocs
    int* rp0;
    CK(hipMalloc(&rp0, sizeof(int) * (n + 1)));
    int64_t nnz = 0;
    if (gkoc_stencil_row_ptrs_i32(nullptr, 3, g, 0, 0, g, rp0, &nnz)) return 1;
    const double bytes = double(nnz) * 12 + double(n + 1) * 4 + double(n) * 16;
    auto up = [&](size_t b) { return (b + 2 * MB - 1) / (2 * MB) * (2 * MB); };
    const size_t s_val = up(sizeof(double) * nnz), s_col = up(sizeof(int) * nnz),
                 s_rp = up(sizeof(int) * (n + 1)), s_vec = up(sizeof(double) * n);
    if (argc > 3 && !strcmp(argv[3], "scan")) {
        const int nbuf = argc > 4 ? atoi(argv[4]) : 100;
        int* cols;
        double *vals, *b;
        CK(hipMalloc(&cols, sizeof(int) * nnz));
        CK(hipMalloc(&vals, sizeof(double) * nnz));
        CK(hipMalloc(&b, sizeof(double) * n));
        if (gkoc_stencil_fill_f64_i32(nullptr, 3, g, 0, 0, g, rp0, cols, vals)) return 1;
        std::vector<double> hb(n);
        unsigned long long s2 = 42;
        for (int64_t i = 0; i < n; ++i) {
            s2 = s2 * 6364136223846793005ULL + 1442695040888963407ULL;
            hb[i] = double(s2 >> 11) / 9007199254740992.0 * 2 - 1;
        }
        CK(hipMemcpy(b, hb.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        printf("rp %p cols %p vals %p b %p\n", rp0, cols, vals, b);
        if (argc > 5 && !strcmp(argv[5], "trace")) {
            // progress of one launch in time: end time stamp of every wave
            std::vector<double*> ys(nbuf);
            std::vector<double> tms(nbuf);
            for (int k = 0; k < nbuf; ++k) {
                CK(hipMalloc(&ys[k], sizeof(double) * n));
                tms[k] = T.ms(5, [&] {
                    gkoc_csr_spmv_f64_i32(nullptr, n, n, rp0, cols, vals, b, 1, ys[k], 1, 1);
                });
            }
            int kmin = 0, kmax = 0;
            for (int k = 0; k < nbuf; ++k) {
                if (tms[k] < tms[kmin]) kmin = k;
                if (tms[k] > tms[kmax]) kmax = k;
            }
            for (int which = 0; which < 2; ++which) {
                double* y = ys[which ? kmax : kmin];
                const int64_t pages = int64_t(sizeof(double)) * n / 4096;
                double tr = T.ms(3, [&] { tlb_stress_kernel<<<1024, 256>>>(y, pages, 64, ys[(kmin + 1) % nbuf]); });
                double tw = T.ms(3, [&] { tlb_stress_write_kernel<<<1024, 256>>>(y, pages, 64); });
                printf("%s y %p: random-page reads %.1f us, random-page writes %.1f us (16.8 M accesses each)\n",
                       which ? "SLOWEST" : "FASTEST", (void*)y, tr * 1e3, tw * 1e3);
            }
            for (int k = 0; k < nbuf; ++k) printf("%.0f ", tms[k] * 1e3);
            printf("\n");
            const int64_t nseg = (n + 31) / 32, nw = (nseg + 1) / 2;
            unsigned long long* ts;
            CK(hipMalloc(&ts, 8 * nw));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(gkoc_lab_ts), &ts, sizeof(ts)));
            std::vector<unsigned long long> h(nw);
            for (int which = 0; which < 2; ++which) {
                double* y = ys[which ? kmax : kmin];
                CK(hipMemset(ts, 0, 8 * nw));
                for (int rep = 0; rep < 2; ++rep) {
                    csr_spmv_pipe3_kernel<double, int, false, 32, 4, 1, 1024, 1, 16>
                        <<<dim3(unsigned(nw)), dim3(64)>>>(n, nseg, 2, rp0, cols, vals, b, 1, y, 1,
                                                           1, nullptr, nullptr);
                    CK(hipDeviceSynchronize());
                }
                CK(hipMemcpy(h.data(), ts, 8 * nw, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, t1 = 0;
                for (auto v : h) {
                    if (v && v < t0) t0 = v;
                    if (v > t1) t1 = v;
                }
                printf("%s y (%.1f us by events): span of wave end stamps %.1f us (100 MHz clock)\n",
                       which ? "SLOWEST" : "FASTEST", (which ? tms[kmax] : tms[kmin]) * 1e3,
                       (t1 - t0) / 100.0);
                // time needed for each 1/32 of the waves (by median end stamp)
                const int NC = 32;
                std::vector<double> med(NC);
                for (int cidx = 0; cidx < NC; ++cidx) {
                    std::vector<unsigned long long> v(h.begin() + nw * cidx / NC,
                                                      h.begin() + nw * (cidx + 1) / NC);
                    std::sort(v.begin(), v.end());
                    med[cidx] = (v[v.size() / 2] - t0) / 100.0;
                }
                printf("   median end time of each 1/32 of the rows [us]:");
                for (int cidx = 0; cidx < NC; ++cidx) printf(" %.0f", med[cidx]);
                printf("\n   increments [us]:");
                for (int cidx = 1; cidx < NC; ++cidx) printf(" %.1f", med[cidx] - med[cidx - 1]);
                printf("\n");
            }
            return 0;
        }
        if (argc > 5 && !strcmp(argv[5], "factor")) {
            struct slot { int *r, *c; double *v, *b, *y; };
            std::vector<slot> S(nbuf);
            for (int k = 0; k < nbuf; ++k) {
                slot& q = S[k];
                CK(hipMalloc(&q.r, sizeof(int) * (n + 1)));
                CK(hipMalloc(&q.c, sizeof(int) * nnz));
                CK(hipMalloc(&q.v, sizeof(double) * nnz));
                CK(hipMalloc(&q.b, sizeof(double) * n));
                CK(hipMalloc(&q.y, sizeof(double) * n));
                CK(hipMemcpy(q.r, rp0, sizeof(int) * (n + 1), hipMemcpyDeviceToDevice));
                CK(hipMemcpy(q.c, cols, sizeof(int) * nnz, hipMemcpyDeviceToDevice));
                CK(hipMemcpy(q.v, vals, sizeof(double) * nnz, hipMemcpyDeviceToDevice));
                CK(hipMemcpy(q.b, b, sizeof(double) * n, hipMemcpyDeviceToDevice));
            }
            auto tm = [&](const slot& q) {
                return 1e3 * T.ms(4, [&] {
                    gkoc_csr_spmv_f64_i32(nullptr, n, n, q.r, q.c, q.v, q.b, 1, q.y, 1, 1);
                });
            };
            printf("slot : own arrays | slot0 with this slot's rp | cols | vals | b | y | vals+cols | b+y\n");
            for (int k = 0; k < nbuf; ++k) {
                slot a = S[0];
                const double own = tm(S[k]);
                a = S[0]; a.r = S[k].r; const double tr = tm(a);
                a = S[0]; a.c = S[k].c; const double tc = tm(a);
                a = S[0]; a.v = S[k].v; const double tv = tm(a);
                a = S[0]; a.b = S[k].b; const double tb = tm(a);
                a = S[0]; a.y = S[k].y; const double ty = tm(a);
                a = S[0]; a.v = S[k].v; a.c = S[k].c; const double tvc = tm(a);
                a = S[0]; a.b = S[k].b; a.y = S[k].y; const double tby = tm(a);
                printf("%2d : %7.1f | %7.1f | %7.1f | %7.1f | %7.1f | %7.1f | %7.1f | %7.1f\n", k, own,
                       tr, tc, tv, tb, ty, tvc, tby);
                fflush(stdout);
            }
            return 0;
        }
        if (argc > 5 && !strcmp(argv[5], "depth")) {
            // allocate the whole problem again and again (previous copies stay
            // allocated): time vs. depth into the device memory
            const double* cv = vals;
            const int* cc = cols;
            const int* cr = rp0;
            const double* cb = b;
            for (int k = 0; k < nbuf; ++k) {
                int *c2, *r2;
                double *v2, *b2, *y2;
                CK(hipMalloc(&r2, sizeof(int) * (n + 1)));
                CK(hipMalloc(&c2, sizeof(int) * nnz));
                CK(hipMalloc(&v2, sizeof(double) * nnz));
                CK(hipMalloc(&b2, sizeof(double) * n));
                CK(hipMalloc(&y2, sizeof(double) * n));
                CK(hipMemcpy(r2, cr, sizeof(int) * (n + 1), hipMemcpyDeviceToDevice));
                CK(hipMemcpy(c2, cc, sizeof(int) * nnz, hipMemcpyDeviceToDevice));
                CK(hipMemcpy(v2, cv, sizeof(double) * nnz, hipMemcpyDeviceToDevice));
                CK(hipMemcpy(b2, cb, sizeof(double) * n, hipMemcpyDeviceToDevice));
                double ms = T.ms(5, [&] {
                    gkoc_csr_spmv_f64_i32(nullptr, n, n, r2, c2, v2, b2, 1, y2, 1, 1);
                });
                double ms2 = T.ms(5, [&] { stream_read_kernel<<<2048, 256>>>(nnz, v2, c2, y2); });
                // mixed: new matrix with the first vectors, first matrix with new vectors
                double ms3 = T.ms(5, [&] {
                    gkoc_csr_spmv_f64_i32(nullptr, n, n, r2, c2, v2, cb, 1, y2, 1, 1);
                });
                double ms4 = T.ms(5, [&] {
                    gkoc_csr_spmv_f64_i32(nullptr, n, n, cr, cc, cv, b2, 1, y2, 1, 1);
                });
                printf("copy %2d (%.1f GB deep) vals %p : spmv %7.1f us | read %6.1f us | new matrix+old b %7.1f | old matrix+new b,y %7.1f\n",
                       k, (k + 1) * 5.75, (void*)v2, ms * 1e3, ms2 * 1e3, ms3 * 1e3, ms4 * 1e3);
                fflush(stdout);
            }
            return 0;
        }
        if (nbuf < 0) {
            // slide y through one big allocation
            const size_t GiB = size_t(1) << 30;
            const size_t pool_sz = size_t(-nbuf) * GiB;
            char* yp;
            CK(hipMalloc(&yp, pool_sz));
            const size_t step = (argc > 5 ? size_t(atoll(argv[5])) : 64) * MB;
            printf("y pool %p %zu GiB, step %zu MiB\n", yp, pool_sz >> 30, step >> 20);
            for (size_t off = 0; off + sizeof(double) * n <= pool_sz; off += step) {
                double* y = (double*)(yp + off);
                double ms = T.ms(4, [&] {
                    gkoc_csr_spmv_f64_i32(nullptr, n, n, rp0, cols, vals, b, 1, y, 1, 1);
                });
                printf("off %6zu MiB %7.1f us\n", off >> 20, ms * 1e3);
            }
            return 0;
        }
        for (int k = 0; k < nbuf; ++k) {
            double* y;
            CK(hipMalloc(&y, sizeof(double) * n));
            double ms = T.ms(5, [&] {
                gkoc_csr_spmv_f64_i32(nullptr, n, n, rp0, cols, vals, b, 1, y, 1, 1);
            });
            // translation probe: dependent chain at 4 KiB / 64 KiB / 2 MiB stride
            double lat[3];
            const int64_t strides[3] = {4096, 65536, int64_t(2) << 20};
            static int64_t* sink = nullptr;
            if (!sink) CK(hipMalloc(&sink, 64));
            for (int t = 0; t < 3; ++t) {
                const int64_t se = strides[t] / 8;
                const int64_t steps = int64_t(sizeof(double)) * n / strides[t];
                chain_init_kernel<<<unsigned((steps + 255) / 256), 256>>>((int64_t*)y, steps, se);
                const int rounds = t == 0 ? 1 : (t == 1 ? 8 : 64);
                double w = T.ms(1, [&] {
                    chain_walk_kernel<<<1, 1>>>((const int64_t*)y, steps * rounds, sink);
                });
                lat[t] = w * 1e6 / double(steps * rounds);
            }
            printf("y[%3d] %p  %7.1f us   chain ns/step: 4K %6.1f  64K %6.1f  2M %6.1f\n", k,
                   (void*)y, ms * 1e3, lat[0], lat[1], lat[2]);
        }
        return 0;
    }
    const size_t slack = 1024 * MB;
    const size_t pool_bytes = s_val + s_col + s_rp + 4 * s_vec + slack;
    char* pool;
    CK(hipMalloc(&pool, pool_bytes));
    printf("pool %p  %.2f GB; val %.1f MB col %.1f MB vec %.1f MB\n", pool,
           pool_bytes / 1e9, s_val / 1e6, s_col / 1e6, s_vec / 1e6);
    std::vector<double> hx(n);
    unsigned long long s = 42;
    for (int64_t i = 0; i < n; ++i) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        hx[i] = double(s >> 11) / 9007199254740992.0 * 2 - 1;
    }
    auto run = [&](const char* tag, size_t o_val, size_t o_col, size_t o_rp,
                   size_t o_b, size_t o_y, bool regen) {
        double* vals = (double*)(pool + o_val);
        int* cols = (int*)(pool + o_col);
        int* rp = (int*)(pool + o_rp);
        double* b = (double*)(pool + o_b);
        double* y = (double*)(pool + o_y);
        if (regen) {
            CK(hipMemcpy(rp, rp0, sizeof(int) * (n + 1), hipMemcpyDeviceToDevice));
            if (gkoc_stencil_fill_f64_i32(nullptr, 3, g, 0, 0, g, rp, cols, vals)) exit(1);
        }
        CK(hipMemcpy(b, hx.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        double ms = T.ms(reps, [&] {
            gkoc_csr_spmv_f64_i32(nullptr, n, n, rp, cols, vals, b, 1, y, 1, 1);
        });
        double ms2 = T.ms(reps, [&] { stream_read_kernel<<<2048, 256>>>(nnz, vals, cols, y); });
        printf("%-34s val@%7.1f col@%7.1f rp@%7.1f b@%7.1f y@%7.1f MB : spmv %7.4f ms %6.1f GB/s (%4.1f%%) | read %7.4f ms %6.1f GB/s\n",
               tag, o_val / 1e6, o_col / 1e6, o_rp / 1e6, o_b / 1e6, o_y / 1e6, ms,
               bytes / ms / 1e6, bytes / ms / 1e6 / 80.0, ms2, double(nnz) * 12 / ms2 / 1e6);
        fflush(stdout);
    };
    if (argc > 3 && !strcmp(argv[3], "soak")) {
        // time series: does the chip slow down as it heats up / hits its power cap?
        size_t o_col = s_val, o_rp = s_val + s_col, o_b = o_rp + s_rp, o_y = o_b + s_vec;
        double* vals = (double*)(pool);
        int* cols = (int*)(pool + o_col);
        int* rp = (int*)(pool + o_rp);
        double* b = (double*)(pool + o_b);
        double* y = (double*)(pool + o_y);
        CK(hipMemcpy(rp, rp0, sizeof(int) * (n + 1), hipMemcpyDeviceToDevice));
        if (gkoc_stencil_fill_f64_i32(nullptr, 3, g, 0, 0, g, rp, cols, vals)) exit(1);
        CK(hipMemcpy(b, hx.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        const int chunks = argc > 4 ? atoi(argv[4]) : 60;
        double tot = 0;
        for (int c = 0; c < chunks; ++c) {
            double ms = T.ms(200, [&] {
                gkoc_csr_spmv_f64_i32(nullptr, n, n, rp, cols, vals, b, 1, y, 1, 1);
            });
            double ms2 = T.ms(50, [&] { stream_read_kernel<<<2048, 256>>>(nnz, vals, cols, y); });
            tot += ms * 201 + ms2 * 51;
            printf("t=%7.2f s  spmv %7.4f ms (%4.1f%%)  read %7.4f ms %6.1f GB/s\n", tot / 1e3, ms,
                   bytes / ms / 1e6 / 80.0, ms2, double(nnz) * 12 / ms2 / 1e6);
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "vmm")) {
        // virtual-memory API: choose the VA alignment ourselves
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        printf("VMM granularity (recommended) = %zu\n", gran);
        auto vmm_alloc = [&](size_t bytes, size_t align) -> void* {
            const size_t sz = (bytes + gran - 1) / gran * gran;
            void* va = nullptr;
            CK(hipMemAddressReserve(&va, sz, align, nullptr, 0));
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, sz, &prop, 0));
            CK(hipMemMap(va, sz, 0, h, 0));
            hipMemAccessDesc acc{};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, sz, &acc, 1));
            return va;
        };
        // baseline arrays by hipMalloc
        int* cols;
        double *vals, *b;
        CK(hipMalloc(&cols, sizeof(int) * nnz));
        CK(hipMalloc(&vals, sizeof(double) * nnz));
        CK(hipMalloc(&b, sizeof(double) * n));
        if (gkoc_stencil_fill_f64_i32(nullptr, 3, g, 0, 0, g, rp0, cols, vals)) return 1;
        CK(hipMemcpy(b, hx.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        auto tm = [&](const char* tag, const int* rp, const int* c, const double* v,
                      const double* bb, double* y) {
            double ms = T.ms(reps, [&] {
                gkoc_csr_spmv_f64_i32(nullptr, n, n, rp, c, v, bb, 1, y, 1, 1);
            });
            printf("%-46s y=%p : %7.4f ms (%4.1f%%)\n", tag, (void*)y, ms, bytes / ms / 1e6 / 80.0);
            fflush(stdout);
        };
        for (int k = 0; k < 4; ++k) {
            double* y;
            CK(hipMalloc(&y, sizeof(double) * n));
            tm("hipMalloc matrix+b, hipMalloc y", rp0, cols, vals, b, y);
        }
        for (size_t al : {size_t(2) << 20, size_t(128) << 20, size_t(1) << 30}) {
            for (int k = 0; k < 3; ++k) {
                double* y = (double*)vmm_alloc(sizeof(double) * n, al);
                char tag[96];
                snprintf(tag, 96, "hipMalloc matrix+b, VMM y align %zu MB", al >> 20);
                tm(tag, rp0, cols, vals, b, y);
            }
        }
        // everything through VMM, 1 GB aligned
        {
            const size_t al = size_t(1) << 30;
            double* v2 = (double*)vmm_alloc(sizeof(double) * nnz, al);
            int* c2 = (int*)vmm_alloc(sizeof(int) * nnz, al);
            int* r2 = (int*)vmm_alloc(sizeof(int) * (n + 1), al);
            double* b2 = (double*)vmm_alloc(sizeof(double) * n, al);
            CK(hipMemcpy(r2, rp0, sizeof(int) * (n + 1), hipMemcpyDeviceToDevice));
            CK(hipMemcpy(v2, vals, sizeof(double) * nnz, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(c2, cols, sizeof(int) * nnz, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(b2, b, sizeof(double) * n, hipMemcpyDeviceToDevice));
            for (int k = 0; k < 3; ++k) {
                double* y = (double*)vmm_alloc(sizeof(double) * n, al);
                tm("all VMM 1 GB aligned", r2, c2, v2, b2, y);
            }
            double* y;
            CK(hipMalloc(&y, sizeof(double) * n));
            tm("VMM matrix+b, hipMalloc y", r2, c2, v2, b2, y);
        }
        // many hipMalloc'ed y of different sizes
        for (size_t extra : {size_t(0), size_t(1) << 20, size_t(2) << 20, size_t(64) << 20,
                             size_t(128) << 20, size_t(896) << 20}) {
            double* y;
            CK(hipMalloc(&y, sizeof(double) * n + extra));
            char tag[96];
            snprintf(tag, 96, "hipMalloc y of 128 MiB + %zu MiB", extra >> 20);
            tm(tag, rp0, cols, vals, b, y);
        }
        return 0;
    }
    // layout 0: [val][col][rp][b][y]
    size_t o_val = 0, o_col = s_val, o_rp = s_val + s_col, o_b = o_rp + s_rp,
           o_y = o_b + s_vec;
    run("packed val,col,rp,b,y", o_val, o_col, o_rp, o_b, o_y, true);
    run("packed (again)", o_val, o_col, o_rp, o_b, o_y, false);
    // y sweep (2 MB steps)
    for (int k = 1; k <= 12; ++k) {
        char tag[64];
        snprintf(tag, 64, "y += %d*2MB", k);
        run(tag, o_val, o_col, o_rp, o_b, o_y + size_t(k) * 2 * MB, false);
    }
    for (int k = 1; k <= 8; ++k) {
        char tag[64];
        snprintf(tag, 64, "y += %d*32MB", k);
        run(tag, o_val, o_col, o_rp, o_b, o_y + size_t(k) * 32 * MB, false);
    }
    for (size_t d : {size_t(256), size_t(4096), size_t(65536), size_t(1) << 18, MB}) {
        char tag[64];
        snprintf(tag, 64, "y += %zu B", d);
        run(tag, o_val, o_col, o_rp, o_b, o_y + d, false);
    }
    // b sweep: put b after y region
    for (int k = 0; k <= 8; ++k) {
        char tag[64];
        snprintf(tag, 64, "b after y, += %d*2MB", k);
        run(tag, o_val, o_col, o_rp, o_y + s_vec + size_t(k) * 2 * MB, o_y, false);
    }
    // vectors first, then matrix
    {
        size_t ob = 0, oy = s_vec, ov = 2 * s_vec, oc = ov + s_val, orp = oc + s_col;
        run("packed b,y,val,col,rp", ov, oc, orp, ob, oy, true);
        for (int k = 1; k <= 6; ++k) {
            char tag[64];
            snprintf(tag, 64, "  col += %d*2MB", k);
            run(tag, ov, oc + size_t(k) * 2 * MB, orp + size_t(k) * 2 * MB, ob, oy, true);
        }
        size_t oc2 = 2 * s_vec, ov2 = oc2 + s_col;
        run("packed b,y,col,val,rp", ov2, oc2, orp, ob, oy, true);
    }
    // separate allocations, as Ginkgo / torch would do
    {
        int* cols;
        double *vals, *b, *y;
        CK(hipMalloc(&cols, sizeof(int) * nnz));
        CK(hipMalloc(&vals, sizeof(double) * nnz));
        CK(hipMalloc(&b, sizeof(double) * n));
        CK(hipMalloc(&y, sizeof(double) * n));
        if (gkoc_stencil_fill_f64_i32(nullptr, 3, g, 0, 0, g, rp0, cols, vals)) return 1;
        CK(hipMemcpy(b, hx.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        double ms = T.ms(reps, [&] {
            gkoc_csr_spmv_f64_i32(nullptr, n, n, rp0, cols, vals, b, 1, y, 1, 1);
        });
        printf("separate hipMallocs: rp %p cols %p vals %p b %p y %p : %7.4f ms %6.1f GB/s (%4.1f%%)\n",
               rp0, cols, vals, b, y, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
        // several y candidates
        for (int k = 0; k < 6; ++k) {
            double* y2;
            CK(hipMalloc(&y2, sizeof(double) * n + size_t(k) * 2 * MB));
            ms = T.ms(reps, [&] {
                gkoc_csr_spmv_f64_i32(nullptr, n, n, rp0, cols, vals, b, 1, y2, 1, 1);
            });
            printf("  another y %p : %7.4f ms (%4.1f%%)\n", y2, ms, bytes / ms / 1e6 / 80.0);
        }
    }
    return 0;
}
