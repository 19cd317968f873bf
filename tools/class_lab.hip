// class_lab (development tool, round 4): how does the driver hand out the three memory classes of
// an MI355X, what does a physical handle of 1 .. 64 GiB cost, and what does a big
// handle consist of?  The arena's search (csrc/arena.hip) is designed on these answers.
//
//   class_lab survey [n_small=24]   fresh-process survey: n_small 1 GiB handles, then 2, 4, ... 64, 64
//                                   GiB handles, all HELD; per handle the time of create / map /
//                                   first use and the class of every GiB of it; then release costs
//   class_lab starve [max=140]      reproduce the state of the box that broke round 3's search: walk
//                                   1 GiB handles until one class shows a run of >= 17 (its >= 32 GiB
//                                   block), release THAT class, keep the granules of the other two
//                                   (they now have no free block below their big ones), print READY
//                                   and sleep until killed
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/class_lab tools/class_lab.hip
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);          \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

constexpr size_t MiB = size_t(1) << 20, GiB = size_t(1) << 30;

// every wave streams a private 32 KiB piece of x and writes 1 KiB of y
__global__ __launch_bounds__(64) void probe_kernel(const uint4* __restrict__ x, int loads,
                                                   uint4* __restrict__ y)
{
    const int lane = threadIdx.x;
    const uint4* xp = x + (size_t(blockIdx.x) * loads) * 64 + lane;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i + 4 <= loads; i += 4) {
        const uint4 a = xp[(i + 0) * 64], b = xp[(i + 1) * 64], c = xp[(i + 2) * 64], d = xp[(i + 3) * 64];
        acc.x += a.x ^ b.x ^ c.x ^ d.x;
        acc.y += a.y ^ b.y ^ c.y ^ d.y;
        acc.z += a.z ^ b.z ^ c.z ^ d.z;
        acc.w += a.w ^ b.w ^ c.w ^ d.w;
    }
    y[size_t(blockIdx.x) * 64 + lane] = acc;
}

__global__ void touch_kernel(uint32_t* p, size_t stride_words, size_t n)
{
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i * stride_words] = 1;
}

constexpr size_t x_bytes = GiB - 32 * MiB;       // read [32 MiB, 1 GiB) of the reference granule
constexpr size_t y_bytes = x_bytes / 32;         // 31 MiB written

static double probe_us(const char* x, char* y, int reps = 2)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    const unsigned waves = unsigned(x_bytes / (32 * 1024));
    for (int r = 0; r <= reps; ++r) {
        CK(hipEventRecord(a, nullptr));
        probe_kernel<<<waves, 64>>>(reinterpret_cast<const uint4*>(x), 32, reinterpret_cast<uint4*>(y));
        CK(hipEventRecord(b, nullptr));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (r > 0 && ms < best) best = ms;
    }
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return best * 1e3;
}

struct classes {
    std::vector<char*> ref;     // base of a 1 GiB range known to be of class k
    // class of the GiB at p: known id, or a new one (registered with p as its reference)
    int of(char* p, bool reg = true)
    {
        for (size_t k = 0; k < ref.size(); ++k) {
            const double t_ref = probe_us(ref[k] + 32 * MiB, ref[k]);
            const double t_new = probe_us(ref[k] + 32 * MiB, p);
            if (t_new > 0.95 * t_ref) return int(k);
        }
        if (reg) ref.push_back(p);
        return int(ref.size()) - (reg ? 1 : 0);
    }
};

static hipMemAllocationProp g_prop;
static hipMemAccessDesc g_acc;

struct handle {
    hipMemGenericAllocationHandle_t h;
    char* va;
    size_t bytes;
    double t_create, t_map, t_touch;
};

static bool make(size_t bytes, handle* out)
{
    void* va;
    CK(hipMemAddressReserve(&va, bytes, GiB, nullptr, 0));
    const double t0 = now();
    hipError_t e = hipMemCreate(&out->h, bytes, &g_prop, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        printf("  hipMemCreate(%zu GiB) failed: %s\n", bytes / GiB, hipGetErrorString(e));
        return false;
    }
    const double t1 = now();
    CK(hipMemMap(va, bytes, 0, out->h, 0));
    CK(hipMemSetAccess(va, bytes, &g_acc, 1));
    const double t2 = now();
    // first use: one word per 2 MiB page
    const size_t n = bytes / (2 * MiB);
    touch_kernel<<<unsigned((n + 255) / 256), 256>>>(static_cast<uint32_t*>(va), 2 * MiB / 4, n);
    CK(hipDeviceSynchronize());
    const double t3 = now();
    out->va = static_cast<char*>(va);
    out->bytes = bytes;
    out->t_create = t1 - t0;
    out->t_map = t2 - t1;
    out->t_touch = t3 - t2;
    return true;
}

static std::string class_string(classes& C, const handle& h)
{
    std::string s;
    for (size_t g = 0; g < h.bytes / GiB; ++g) s += char('A' + C.of(h.va + g * GiB));
    return s;
}

static void free_info(const char* what)
{
    size_t fr, tot;
    CK(hipMemGetInfo(&fr, &tot));
    printf("[%s] free %.2f GiB of %.2f GiB\n", what, fr / double(GiB), tot / double(GiB));
}

static int survey(int n_small)
{
    free_info("start");
    classes C;
    std::vector<handle> hs;
    double t_all = now();
    printf("== %d handles of 1 GiB, held (class letters in order of discovery)\n", n_small);
    std::string seq;
    double tc = 0, tm = 0, tt = 0;
    for (int i = 0; i < n_small; ++i) {
        handle h;
        if (!make(GiB, &h)) break;
        hs.push_back(h);
        tc += h.t_create;
        tm += h.t_map;
        tt += h.t_touch;
        seq += char('A' + C.of(h.va));
    }
    printf("   %s\n   create %.1f ms, map+access %.1f ms, first use %.1f ms per handle; whole step %.2f s\n",
           seq.c_str(), tc / n_small * 1e3, tm / n_small * 1e3, tt / n_small * 1e3, now() - t_all);
    printf("== growing handles, held\n");
    for (size_t gib : {size_t(2), size_t(4), size_t(8), size_t(16), size_t(32), size_t(64), size_t(64)}) {
        handle h;
        if (!make(gib * GiB, &h)) continue;
        hs.push_back(h);
        const double t0 = now();
        const std::string s = class_string(C, h);
        printf("  %2zu GiB: create %.4f s, map+access %.4f s, first use %.4f s, classify %.3f s | %s\n", gib,
               h.t_create, h.t_map, h.t_touch, now() - t0, s.c_str());
    }
    free_info("all held");
    // a slice of a big handle mapped by offset?
    {
        const handle& big = hs.back();
        void* va;
        CK(hipMemAddressReserve(&va, GiB, GiB, nullptr, 0));
        hipError_t e = hipMemMap(va, GiB, big.bytes / 2, big.h, 0);
        printf("== hipMemMap(1 GiB at offset %zu GiB of a %zu GiB handle): %s\n", big.bytes / 2 / GiB,
               big.bytes / GiB, hipGetErrorString(e));
        if (e == hipSuccess) {
            CK(hipMemSetAccess(va, GiB, &g_acc, 1));
            printf("   class of the slice: %c (the handle's GiB there: %c)\n",
                   'A' + C.of(static_cast<char*>(va), false), 'A' + C.of(big.va + big.bytes / 2, false));
            CK(hipMemUnmap(va, GiB));
        } else {
            (void)hipGetLastError();
        }
    }
    // release: largest first, time each
    printf("== release (unmap + hipMemRelease + sync)\n");
    for (size_t i = hs.size(); i-- > size_t(n_small);) {
        const double t0 = now();
        CK(hipMemUnmap(hs[i].va, hs[i].bytes));
        CK(hipMemRelease(hs[i].h));
        CK(hipDeviceSynchronize());
        printf("  %2zu GiB: %.4f s\n", hs[i].bytes / GiB, now() - t0);
    }
    free_info("big ones released");
    // the same sizes again: cleared or not?
    printf("== 64 GiB and 16 GiB again (released memory)\n");
    for (size_t gib : {size_t(64), size_t(16)}) {
        handle h;
        if (!make(gib * GiB, &h)) continue;
        printf("  %2zu GiB: create %.4f s, map+access %.4f s, first use %.4f s | %s\n", gib, h.t_create, h.t_map,
               h.t_touch, class_string(C, h).c_str());
        CK(hipMemUnmap(h.va, h.bytes));
        CK(hipMemRelease(h.h));
    }
    printf("total %.2f s\n", now() - t_all);
    return 0;
}

static int starve(int max_walk)
{
    free_info("start");
    classes C;
    std::vector<handle> hs;
    std::vector<int> cls;
    int run = 0, run_cls = -1;
    std::string seq;
    for (int i = 0; i < max_walk; ++i) {
        handle h;
        if (!make(GiB, &h)) break;
        const int c = C.of(h.va);
        hs.push_back(h);
        cls.push_back(c);
        seq += char('A' + c);
        run = c == run_cls ? run + 1 : 1;
        run_cls = c;
        if (run >= 17 && C.ref.size() >= 3) break;
    }
    printf("walk: %s\n", seq.c_str());
    if (run < 17) printf("no run of 17 within %d granules: releasing the class of the last run anyway\n", max_walk);
    // the reference granule of the released class must go as well: release everything of run_cls
    size_t kept = 0;
    for (size_t i = 0; i < hs.size(); ++i) {
        if (cls[i] == run_cls) {
            CK(hipMemUnmap(hs[i].va, hs[i].bytes));
            CK(hipMemRelease(hs[i].h));
        } else {
            ++kept;
        }
    }
    CK(hipDeviceSynchronize());
    printf("released class %c (%zu granules), holding %zu granules of the other two\n", 'A' + run_cls,
           hs.size() - kept, kept);
    free_info("holding");
    printf("READY\n");
    fflush(stdout);
    for (;;) sleep(1000);
    return 0;
}

int main(int argc, char** argv)
{
    CK(hipFree(nullptr));
    g_prop = hipMemAllocationProp{};
    g_prop.type = hipMemAllocationTypePinned;
    g_prop.location.type = hipMemLocationTypeDevice;
    g_prop.location.id = 0;
    g_acc = hipMemAccessDesc{};
    g_acc.location = g_prop.location;
    g_acc.flags = hipMemAccessFlagsProtReadWrite;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const std::string mode = argc > 1 ? argv[1] : "survey";
    if (mode == "survey") return survey(argc > 2 ? atoi(argv[2]) : 24);
    if (mode == "starve") return starve(argc > 2 ? atoi(argv[2]) : 140);
    printf("usage: class_lab survey [n_small] | starve [max_walk]\n");
    return 2;
}
