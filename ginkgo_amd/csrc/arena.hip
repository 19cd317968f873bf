// Device-memory arena of libgko_cdna4: the policy behind gkoc_malloc / gkoc_free,
// i.e. behind HipExecutor::raw_alloc / raw_free and HipAllocator::allocate /
// deallocate of the Ginkgo binding (reference: hip/base/memory.hip.cpp,
// hip/base/executor.hip.cpp:95-112 - one hipMalloc / hipFree per request).
//
// Why the backend places memory itself (measurements: DESIGN.md 3.2,
// profiles/r02_placement/).  The 288 GiB of an MI355X consist of three MEMORY
// CLASSES of 96 GiB (in all likelihood the three ranks of the 12-high HBM3E
// stacks; the free-block sizes of the driver's allocator say they are the three
// contiguous thirds of the physical address space).  A kernel whose
// read streams and written stream live in the SAME class is about 11 % slower than
// one whose output lives in another class (CSR SpMV on the 27-pt 256^3 Laplacian:
// 1.127 ms against 1.005 ms; values and column indices in two different classes as
// well: 0.985 ms; block-Jacobi apply 246 against 226 us).  With one hipMalloc per
// array the class of every array is an accident of the allocation history - the
// "allocation lottery" of round 1.  The arena removes the accident:
//   * it owns three REGIONS, one per class: a virtual address range backed by
//     1 GiB physical granules (hipMemCreate + hipMemMap) of that class only;
//   * the class of a fresh granule is MEASURED with a 0.2 ms probe (every wavefront
//     streams a private piece of a region's first granule and writes a short piece
//     of the new one: same class = 10 % slower; the reference is the same launch
//     writing into a reserved 32 MiB target of the region itself).  A fresh device
//     hands out all three classes within the first few GiB; granules of a class
//     nobody asked for yet wait in a pool (GKOC_ARENA_SPARE_MB, default 16384);
//   * requests are placed by role: matrix values -> class 0, index arrays ->
//     class 1, vectors (everything a kernel writes) -> class 2.  gkoc_malloc has no
//     role argument (Ginkgo's raw_alloc has none): it takes requests of at least a
//     quarter of the largest live request for matrix arrays (values / indices by
//     load) and smaller ones for vectors; gkoc_malloc_role states the role.
// Requests below 1 MiB come from plain 64 MiB chunks (they live in the caches).
//
// Modes (gkoc_arena_configure or the environment variable GKOC_ARENA):
//   0  off: hipMalloc / hipFree per request (the reference's behaviour)
//   1  plain chunks from hipMalloc, first fit, no classes
//   2  class regions as described (default ON THE PART THEY WERE MEASURED ON: gfx950 with at
//      least 200 GiB of device memory in one partition; any other device starts in mode 1
//      unless GKOC_ARENA=2 asks for it)
// Chunk size of mode 1: GKOC_ARENA_CHUNK_MB (default 8192); granule size of mode 2:
// GKOC_ARENA_GRANULE_MB (default and minimum 1024, power of two).  Fresh device
// memory costs about 30 ms per GiB whoever asks for it (the driver clears it).
// Bounds (a search must never eat the device): one search for a granule of a wanted class
// creates at most GKOC_ARENA_MAX_WALK granules (default 128: the driver hands out a class in runs
// of up to 64 GiB and more - 83 granules were needed on one box - about 4 s in the worst
// case, ONCE per process, and only on the part the layout was measured on); if the three classes
// do not show up within that bound the arena settles for the classes it has (two: vectors apart
// from the matrix arrays; one: mode 1).  At most GKOC_ARENA_SPARE_MB (default
// 8192) wait in the pools; gkoc_arena_trim() releases the pools AND the trailing free granules
// of every region (their addresses are retired, never mapped again).
// Peer access: granules are mapped for the owning device only; GKOC_ARENA_PEER_ACCESS=1 also
// grants read/write access to every device that can reach it (hipDeviceCanAccessPeer).  Without
// it arena memory must not be handed to P2P copies or hipIpc (RCCL stages user buffers through
// its own, so the distributed path does not need it).
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <map>
#include <string>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace gkoc {
namespace {

constexpr size_t MiB = size_t(1) << 20, GiB = size_t(1) << 30;
constexpr size_t small_limit = MiB;          // below: small pool
constexpr size_t small_chunk = 64 * MiB;
constexpr int max_classes = 3;

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- a range of device memory with a first-fit free list --------------------
struct span {
    char* base = nullptr;
    size_t size = 0;                          // usable bytes from base
    std::map<size_t, size_t> free_list;       // offset -> size
    std::map<size_t, size_t> used;            // offset -> size
    size_t in_use = 0;

    void* take(size_t bytes, size_t align)
    {
        for (auto it = free_list.begin(); it != free_list.end(); ++it) {
            const size_t off = it->first, sz = it->second;
            const size_t a = round_up(reinterpret_cast<uintptr_t>(base) + off, align) -
                             reinterpret_cast<uintptr_t>(base);
            if (a + bytes > off + sz) continue;
            free_list.erase(it);
            if (a > off) free_list[off] = a - off;
            if (a + bytes < off + sz) free_list[a + bytes] = off + sz - (a + bytes);
            used[a] = bytes;
            in_use += bytes;
            return base + a;
        }
        return nullptr;
    }
    bool owns(const void* p) const
    {
        return static_cast<const char*>(p) >= base && static_cast<const char*>(p) < base + size;
    }
    // false if p is not the start of an allocation
    bool give(void* p)
    {
        const size_t off = size_t(static_cast<char*>(p) - base);
        auto u = used.find(off);
        if (u == used.end()) return false;
        size_t sz = u->second;
        used.erase(u);
        in_use -= sz;
        add_free(off, sz);
        return true;
    }
    void add_free(size_t off, size_t sz)
    {
        auto nxt = free_list.lower_bound(off);
        if (nxt != free_list.end() && off + sz == nxt->first) {
            sz += nxt->second;
            nxt = free_list.erase(nxt);
        }
        if (nxt != free_list.begin()) {
            auto prv = std::prev(nxt);
            if (prv->first + prv->second == off) {
                prv->second += sz;
                return;
            }
        }
        free_list[off] = sz;
    }
    // bytes of the free block that ends at `size` (0 if the end is in use)
    size_t free_tail() const
    {
        if (free_list.empty()) return 0;
        auto last = std::prev(free_list.end());
        return last->first + last->second == size ? last->second : 0;
    }
};

struct region : span {
    size_t reserved = 0;                                    // bytes of virtual address space
    std::map<size_t, size_t> req;                           // offset -> bytes the caller asked for
    size_t retired = 0;        // bytes of [0, size) whose granules went back to the driver (trim)
    size_t tail_retired = 0;   // the part of them that sits at the very end of [0, size)
    std::vector<hipMemGenericAllocationHandle_t> granules;  // mapped back to back from base
};

struct device_arena {
    std::vector<span*> small;     // hipMalloc'ed 64 MiB chunks (modes 1, 2)
    std::vector<span*> plain;     // mode 1 chunks
    // mode 2
    bool ready = false, no_more_classes = false, failed = false, gated = false;
    int n_cls = 0;
    region reg[max_classes];
    // Virtual addresses for classifying candidates.  Every address is used for ONE
    // mapping only: on this system (ROCm 7.2) a kernel that accesses an address which
    // was unmapped and mapped to another physical handle still reaches the OLD memory
    // (stale translation; tools/vmm_tlb.hip, profiles/r02_placement/vmm_remap_stale_translation.txt)
    char* scratch = nullptr;      // next unused address of the current window
    size_t scratch_left = 0;
    hipStream_t stream = nullptr;
    std::vector<hipMemGenericAllocationHandle_t> spare[max_classes];   // classified, unmapped
    // role heuristic of gkoc_malloc: sizes (as requested) of arrays that kernels have been seen
    // to WRITE as vectors (gkoc_arena_note_vector), newest last, a handful at most
    std::vector<size_t> vector_sizes;
    // ... and of arrays the SpMV kernels have been handed as MATRIX arrays (gkoc_arena_note_matrix): a
    // request of such a size is a matrix array although it may be a multiple of a vector's size
    std::vector<size_t> matrix_sizes;
    // a size that kernels WRITE as a vector and that an SpMV is handed as a matrix array (nnz == n: a diagonal
    // or permutation matrix, ELL with one entry per row): after the second change of mind it stays a vector -
    // such a matrix moves as many bytes as a vector, and the notes would otherwise flip the class of every
    // n-vector allocated in between
    std::vector<std::pair<size_t, int>> contested_sizes;
    int64_t misplaced = 0;        // gkoc_arena_note_vector found a written vector next to matrix arrays
    int64_t probes = 0, walked = 0, classified = 0, search_ns = 0, retried = 0;
    bool surveyed = false;        // the one search for all three classes has run (survey_classes)
    // a search for class k that came back empty: what the driver reported free afterwards.  Until it
    // reports a granule more than that, another search would walk the same handles to the same end.
    size_t exhausted_free[max_classes] = {};      // that number + 1; 0 = no mark
    // the searches of this device have used up their wall-clock budget (GKOC_ARENA_SURVEY_MS): no
    // further walk; a region that needs a granule takes the driver's next one UNCLASSIFIED
    bool budget_spent = false;
    int64_t unclassified = 0;     // granules mapped into a region without a probe (after the budget)
};

std::mutex g_mtx;
int g_mode = -1;            // -1: read the environment on first use
size_t g_chunk_bytes = 0;
size_t g_granule_bytes = 0, g_spare_bytes = 0;
int g_sync_free = 1;
int g_verbose = 0;
int g_peer_access = 0;
int g_max_walk = 0;         // granules one search may create; 0 = bounded by the free memory only
int g_max_classes = 3;      // GKOC_ARENA_MAX_CLASSES: stop the survey at fewer classes (tests)
// GKOC_ARENA_SURVEY_MS: wall-clock budget of ALL searches of a device in this process (survey + the
// directed walks that extend a region), 0 = bounded by the free memory only.  On a device whose memory
// has been used the driver clears every handle it creates (30 - 60 ms per GiB), and a class that owns
// the allocator's first 150 GiB costs 150 such handles to get past: BENCH_r05 paid 9.8 s for the third
// class, which is worth 3.3 % of the SpMV.  With the budget spent the survey accepts the classes it has
// (two: matrix arrays | everything kernels write) and later granules are mapped unclassified.
int g_survey_ms = 1500;
bool g_mode_from_env = false;
device_arena g_arena[64];

void read_env_locked()
{
    if (g_mode >= 0) return;
    const char* m = std::getenv("GKOC_ARENA");
    g_mode = m ? std::atoi(m) : 2;
    g_mode_from_env = m != nullptr;
    if (g_mode < 0 || g_mode > 2) g_mode = 2;
    const char* c = std::getenv("GKOC_ARENA_CHUNK_MB");
    const long long mb = c ? std::atoll(c) : 8192;
    g_chunk_bytes = size_t(mb > 2 ? mb : 2) * MiB;
    const char* gm = std::getenv("GKOC_ARENA_GRANULE_MB");
    g_granule_bytes = GiB;
    while (gm && g_granule_bytes < size_t(std::atoll(gm)) * MiB) g_granule_bytes *= 2;
    const char* sp = std::getenv("GKOC_ARENA_SPARE_MB");
    g_spare_bytes = size_t(sp ? std::atoll(sp) : 8192) * MiB;
    const char* pa = std::getenv("GKOC_ARENA_PEER_ACCESS");
    g_peer_access = pa ? std::atoi(pa) : 0;
    const char* lim = std::getenv("GKOC_ARENA_MAX_WALK");
    g_max_walk = lim ? std::atoi(lim) : 0;
    if (g_max_walk < 0) g_max_walk = 0;
    const char* mc = std::getenv("GKOC_ARENA_MAX_CLASSES");
    g_max_classes = mc ? std::atoi(mc) : max_classes;
    if (g_max_classes < 1 || g_max_classes > max_classes) g_max_classes = max_classes;
    const char* sm = std::getenv("GKOC_ARENA_SURVEY_MS");
    g_survey_ms = sm ? std::atoi(sm) : 1500;
    if (g_survey_ms < 0) g_survey_ms = 0;
    const char* sf = std::getenv("GKOC_ARENA_SYNC_FREE");
    g_sync_free = sf ? std::atoi(sf) : 1;
    const char* v = std::getenv("GKOC_ARENA_VERBOSE");
    g_verbose = v ? std::atoi(v) : 0;
}

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return dev;
}

size_t granule_bytes() { return g_granule_bytes; }

// the first probe_target bytes of every region are never handed out: the probe
// writes there to get the "same class" reference time
constexpr size_t probe_target = 32 * MiB;

// ---- memory-class probe ----------------------------------------------------
// The access pattern that shows the classes, reduced to its essentials: every
// wavefront streams a private contiguous piece of `x` (read only, any content) and
// then stores a short contiguous piece of `y`.  x must be well beyond the 256 MiB
// memory-side cache (2 GiB: 0.40 ms same class, 0.36 ms different classes).
__global__ __launch_bounds__(64) void arena_probe_kernel(const uint4* __restrict__ x,
                                                         int loads_per_wave,
                                                         uint4* __restrict__ y,
                                                         int stores_per_wave)
{
    const int lane = threadIdx.x;
    const uint4* xp = x + (size_t(blockIdx.x) * loads_per_wave) * 64 + lane;
    uint4 acc = make_uint4(0, 0, 0, 0);
    int i = 0;
    for (; i + 4 <= loads_per_wave; i += 4) {
        const uint4 a = xp[(i + 0) * 64];
        const uint4 b = xp[(i + 1) * 64];
        const uint4 c = xp[(i + 2) * 64];
        const uint4 d = xp[(i + 3) * 64];
        acc.x += a.x ^ b.x ^ c.x ^ d.x;
        acc.y += a.y ^ b.y ^ c.y ^ d.y;
        acc.z += a.z ^ b.z ^ c.z ^ d.z;
        acc.w += a.w ^ b.w ^ c.w ^ d.w;
    }
    for (; i < loads_per_wave; ++i) {
        const uint4 a = xp[i * 64];
        acc.x += a.x;
        acc.y += a.y;
        acc.z += a.z;
        acc.w += a.w;
    }
    uint4* yp = y + (size_t(blockIdx.x) * stores_per_wave) * 64 + lane;
    for (int k = 0; k < stores_per_wave; ++k) yp[k * 64] = acc;
}

// time of one probe launch in ns (best of `reps`), on `st`
int probe_ns(hipStream_t st, const void* x, size_t x_bytes, void* y, int read_kb, int write_b,
             int reps, int64_t* ns)
{
    const int loads = read_kb;   // 64 lanes * 16 B = 1 KiB per wave load
    const int stores = write_b / 1024 > 0 ? write_b / 1024 : 1;
    const size_t waves = x_bytes / (size_t(read_kb) * 1024);
    GKOC_REQUIRE(waves > 0 && waves < (size_t(1) << 31), GKOC_E_INVALID, "probe range");
    hipEvent_t a, b;
    GKOC_HIP(hipEventCreate(&a));
    GKOC_HIP(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r <= reps; ++r) {
        GKOC_HIP(hipEventRecord(a, st));
        arena_probe_kernel<<<dim3(unsigned(waves)), dim3(64), 0, st>>>(
            static_cast<const uint4*>(x), loads, static_cast<uint4*>(y), stores);
        GKOC_HIP(hipEventRecord(b, st));
        GKOC_HIP(hipEventSynchronize(b));
        float ms = 0;
        GKOC_HIP(hipEventElapsedTime(&ms, a, b));
        if (r > 0 && ms < best) best = ms;   // launch 0 warms up
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ns = int64_t(double(best) * 1e6);
    return GKOC_OK;
}

constexpr int probe_read_kb = 32, probe_write_b = 1024;
// read the first granule of a region behind its probe target (992 MiB at 1 GiB
// granules: 0.21 ms same class, 0.19 ms different); writes 31 MiB
constexpr size_t probe_x_bytes = GiB - probe_target;

// ---- physical granules -------------------------------------------------------
hipMemAllocationProp granule_prop(int dev)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    return prop;
}

hipError_t map_granule(char* va, size_t bytes, hipMemGenericAllocationHandle_t h, int dev)
{
    hipError_t e = hipMemMap(va, bytes, 0, h, 0);
    if (e != hipSuccess) return e;
    std::vector<hipMemAccessDesc> acc(1);
    acc[0].location.type = hipMemLocationTypeDevice;
    acc[0].location.id = dev;
    acc[0].flags = hipMemAccessFlagsProtReadWrite;
    if (g_peer_access) {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess) n_dev = 0;
        for (int p = 0; p < n_dev; ++p) {
            int can = 0;
            if (p != dev && hipDeviceCanAccessPeer(&can, p, dev) == hipSuccess && can) {
                hipMemAccessDesc d = acc[0];
                d.location.id = p;
                acc.push_back(d);
            }
        }
        (void)hipGetLastError();
    }
    e = hipMemSetAccess(va, bytes, acc.data(), acc.size());
    if (e != hipSuccess && acc.size() > 1) {
        // the peers refused: the owner alone, as without the option
        (void)hipGetLastError();
        e = hipMemSetAccess(va, bytes, acc.data(), 1);
    }
    if (e != hipSuccess) (void)hipMemUnmap(va, bytes);
    return e;
}

// The verdict of one comparison: t(read class k, write cand) / t(read class k, write class k's own
// target).  Same class: ~1.00, another class: ~0.90 (lab4_probe_vs_known_classes.txt).  Between
// the bands nothing is decided.
constexpr double ratio_same = 0.965, ratio_other = 0.935;

// class of the granule mapped at `cand`: an existing class id, A.n_cls for "none of the known
// classes", -1 if the probe failed or stayed contradictory.
// Round 4: the candidate is measured against EVERY known class and the verdict must be one-hot -
// exactly one class "same", the others clearly "other" (or none "same": a new class, which is
// only believed when a second, longer measurement says so again).  Round 3 stopped at the first
// class whose ratio exceeded 0.95: tools/class_lab.hip shows isolated wrong verdicts of that rule
// inside 16 - 64 GiB blocks of one class (about 3 % of the granules), and one such granule in a
// region puts whatever lands on it into the wrong class.
int classify_at(device_arena& A, char* cand)
{
    int new_votes = 0;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const int reps = 2 + 2 * attempt;
        int same = -1, n_same = 0, n_unsure = 0;
        for (int k = 0; k < A.n_cls; ++k) {
            const char* x = A.reg[k].base + probe_target;
            int64_t t_ref = 0, t_new = 0;
            if (probe_ns(A.stream, x, probe_x_bytes, A.reg[k].base, probe_read_kb, probe_write_b, reps, &t_ref) ||
                probe_ns(A.stream, x, probe_x_bytes, cand, probe_read_kb, probe_write_b, reps, &t_new)) {
                return -1;
            }
            A.probes += 2 * (reps + 1);
            const double r = double(t_new) / double(t_ref);
            if (g_verbose > 1) {
                fprintf(stderr, "[gkoc arena]   probe vs class %d: %.0f us, reference %.0f us (%.3f)\n", k,
                        t_new / 1e3, t_ref / 1e3, r);
            }
            if (r >= ratio_same) {
                same = k;
                ++n_same;
            } else if (r > ratio_other) {
                ++n_unsure;
            }
        }
        if (n_unsure == 0 && n_same == 1) return same;
        if (n_unsure == 0 && n_same == 0 && A.n_cls < max_classes) {
            if (++new_votes >= 2 || A.n_cls == 0) return A.n_cls;
            continue;       // a new class: measure once more before a region is founded on it
        }
        ++A.retried;
    }
    return -1;
}

// region of the next class = [probe target | free space] on granule h
bool start_class(device_arena& A, int dev, hipMemGenericAllocationHandle_t h)
{
    const size_t gr = granule_bytes();
    region& R = A.reg[A.n_cls];
    if (map_granule(R.base, gr, h, dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    R.granules.push_back(h);
    R.add_free(probe_target, gr - probe_target);
    R.size = gr;
    ++A.n_cls;
    return true;
}

// ---- the search -------------------------------------------------------------------------
// What the driver does (measured, tools/class_lab.hip, profiles/r04_class_survey.txt): the three
// classes are contiguous thirds of the physical address space, the driver's buddy allocator
// serves a 1 GiB request from the SMALLEST free block that holds it, so consecutive handles walk
// through the free blocks in order of size - 1, 2, 4 ... 64 GiB - and every block lies in one
// class (only the 64 GiB block across the 96 GiB mark does not).  Which class owns the small
// blocks is an accident of what else lives on the device: on the builder's boxes all three show up
// within 13 - 33 handles, on the box that timed round 3 one class owned the first 95 GiB and
// round 3's walk (128 handles at most, every one of them probed) gave up with two classes.
// hipMemCreate costs nothing measurable at any size on clean memory (< 0.1 ms for 64 GiB); what a
// step costs is mapping the handle and the probe launches (about 5 ms with three known classes).
// So the walk (1) is bounded by the free memory, not by a count: up to three quarters of what is
// free when it starts; (2) GALLOPS: handles 0 .. 15 are all classified, then every 2nd up to 32,
// every 4th up to 64, every 8th up to 128, every 16th beyond - runs get longer as the blocks get
// bigger, and a class that is absent from the small blocks can only begin with a long run; (3)
// when a late hit ends it, the next handles are classified as well and pooled, because the
// region that needed the class will need it again and a second walk would have to come as far.
// Skipped handles are released unclassified when the walk ends; nothing is released during a
// walk (the driver would hand the same block out again).  A device walk of 288 handles costs
// about 60 classifications = 0.3 s.
int walk_stride(int step)
{
    return step < 16 ? 1 : step < 32 ? 2 : step < 64 ? 4 : step < 128 ? 8 : 16;
}

// Walks fresh granules until the goal is reached: want >= 0 (a known class): its pool holds
// n_want granules; want == -1: all classes are known (the survey; a granule of a class nobody
// has met founds its region, start_class).  hipSuccess if reached.
// Processes that share one device (MPI ranks of a test box, eight bench ranks under gloo) take turns:
// a walk holds its handles until it ends and times its probes, eight of them at once starve each other
// of memory and spoil each other's timings (a 64^3 test with 8 ranks on one GPU: minutes).  An advisory
// lock on a file per device; whoever cannot get it within 20 s walks anyway, a process that dies
// releases it with its descriptor.  One process per GPU - production - never waits.
struct walk_turn {
    int fd = -1;
    explicit walk_turn(int dev)
    {
        const char* dir = std::getenv("TMPDIR");
        char path[512];
        // keyed by the PHYSICAL device (PCI bus id): under HIP_VISIBLE_DEVICES two processes call the same GPU
        // by different ordinals and different GPUs by the same one
        char bus[32] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess) {
            (void)hipGetLastError();
            snprintf(bus, sizeof(bus), "dev%d", dev);
        }
        for (char* c = bus; *c; ++c) {
            if (*c == ':' || *c == '/' || *c == '.') *c = '_';
        }
        snprintf(path, sizeof(path), "%s/gkoc_arena_%s_uid%u.lock", dir && *dir ? dir : "/tmp", bus,
                 unsigned(getuid()));
        fd = ::open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
        if (fd < 0) return;
        for (int tries = 0; tries < 2000; ++tries) {
            if (::flock(fd, LOCK_EX | LOCK_NB) == 0) return;
            ::usleep(10000);
        }
        ::close(fd);      // did not get it: go on without
        fd = -1;
    }
    ~walk_turn()
    {
        if (fd >= 0) {
            (void)::flock(fd, LOCK_UN);
            ::close(fd);
        }
    }
    walk_turn(const walk_turn&) = delete;
    walk_turn& operator=(const walk_turn&) = delete;
};

hipError_t walk(device_arena& A, int dev, int want, int n_want)
{
    const walk_turn turn(dev);
    const size_t gr = granule_bytes();
    const hipMemAllocationProp prop = granule_prop(dev);
    std::vector<hipMemGenericAllocationHandle_t> skipped;
    auto t_start = std::chrono::steady_clock::now();
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        free_b = size_t(64) * GiB;
    }
    int64_t max_steps = int64_t(free_b / 4 * 3 / gr);
    if (g_max_walk > 0 && max_steps > g_max_walk) max_steps = g_max_walk;
    if (max_steps < 1) max_steps = 1;
    auto reached = [&] {
        if (want < 0) return A.n_cls >= g_max_classes;
        if (want >= A.n_cls) return false;
        return int(A.spare[want].size()) >= n_want;
    };
    hipError_t result = hipErrorOutOfMemory;
    int tail = -1;       // > 0: a late hit, this many more handles are classified one by one
    // the budget is checked BEFORE a step against what the dearest step so far has cost (creating a
    // handle of used memory is one driver call of 30 - 60 ms), so that the total stays below it
    const int64_t budget_ns = int64_t(g_survey_ms) * 1000000;
    int64_t dearest_step_ns = 0, t_prev_ns = 0;
    auto elapsed_ns = [&] {
        return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_start)
            .count();
    };
    for (int64_t step = 0; step < max_steps; ++step) {
        if (reached() && (want >= 0 || tail <= 0)) {
            result = hipSuccess;
            break;
        }
        if (budget_ns > 0) {
            const int64_t now_ns = elapsed_ns();
            dearest_step_ns = std::max(dearest_step_ns, now_ns - t_prev_ns);
            t_prev_ns = now_ns;
            if (A.budget_spent || A.search_ns + now_ns + dearest_step_ns > budget_ns) {
                A.budget_spent = true;
                if (g_verbose) {
                    fprintf(stderr, "[gkoc arena] search budget of %d ms spent after %lld granules (%d classes)\n",
                            g_survey_ms, (long long)A.walked, A.n_cls);
                }
                break;
            }
        }
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, gr, &prop, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            result = e;
            break;
        }
        ++A.walked;
        if (tail < 0 && step % walk_stride(int(step)) != 0) {
            skipped.push_back(h);
            continue;
        }
        int cls = 0;
        if (A.n_cls > 0) {
            if (A.scratch_left < gr) {
                // a fresh window of never-used addresses (1024 candidates)
                void* va = nullptr;
                e = hipMemAddressReserve(&va, 1024 * gr, gr, nullptr, 0);
                if (e != hipSuccess) {
                    (void)hipMemRelease(h);
                    result = e;
                    break;
                }
                A.scratch = static_cast<char*>(va);
                A.scratch_left = 1024 * gr;
            }
            char* cand = A.scratch;
            A.scratch += gr;
            A.scratch_left -= gr;
            e = map_granule(cand, gr, h, dev);
            if (e != hipSuccess) {
                (void)hipMemRelease(h);
                result = e;
                break;
            }
            cls = classify_at(A, cand);
            ++A.classified;
            (void)hipMemUnmap(cand, gr);
        }
        if (g_verbose) {
            fprintf(stderr, "[gkoc arena] granule %lld: class %d (want %d)\n", (long long)A.walked, cls, want);
        }
        const bool hit = want >= 0 ? cls == want : (cls == A.n_cls && cls < g_max_classes);
        if (cls >= 0 && cls < A.n_cls) {
            A.spare[cls].push_back(h);
        } else if (cls == A.n_cls && cls < g_max_classes && start_class(A, dev, h)) {
            // a class nobody has met yet: its region starts with this granule
        } else {
            (void)hipMemRelease(h);     // failed or contradictory probe (or a class beyond the cap)
        }
        if (hit && tail < 0 && walk_stride(int(step)) > 1) tail = 2 * (n_want > 0 ? n_want : 1) + 2;
        if (tail > 0 && --tail == 0) tail = -1;
    }
    if (result != hipSuccess && reached()) result = hipSuccess;
    for (auto h : skipped) (void)hipMemRelease(h);
    A.search_ns += elapsed_ns();
    return result;
}

// the pools hold at most GKOC_ARENA_SPARE_MB: cut them back, largest first
void trim_pools(device_arena& A)
{
    const size_t gr = granule_bytes();
    for (;;) {
        size_t total = 0;
        int big = 0;
        for (int k = 0; k < max_classes; ++k) {
            total += A.spare[k].size() * gr;
            if (A.spare[k].size() > A.spare[big].size()) big = k;
        }
        if (total <= g_spare_bytes || A.spare[big].empty()) break;
        (void)hipMemRelease(A.spare[big].back());
        A.spare[big].pop_back();
    }
}

// a granule of the KNOWN class `want`, from the pool or from the driver; n_want: how many the
// caller is going to ask for in a row (a late hit of the walk pools that many)
hipError_t acquire_granule(device_arena& A, int dev, int want, int n_want, hipMemGenericAllocationHandle_t* out)
{
    if (A.spare[want].empty() && A.budget_spent) {
        // no time left to look for the class: the driver's next granule, whatever it is (what one
        // hipMalloc per array - the reference's behaviour - gets for every array)
        const hipMemAllocationProp prop = granule_prop(dev);
        const hipError_t e = hipMemCreate(out, granule_bytes(), &prop, 0);
        if (e == hipSuccess) ++A.unclassified;
        return e;
    }
    if (A.spare[want].empty()) {
        size_t free_b = 0, total_b = 0;
        if (A.exhausted_free[want] != 0) {
            // the last search for this class found none (do not walk again - up to three
            // quarters of the free memory in handles - until memory has come back to the driver)
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
                free_b + 1 < A.exhausted_free[want] + granule_bytes()) {
                return hipErrorOutOfMemory;
            }
            (void)hipGetLastError();
            A.exhausted_free[want] = 0;
        }
        const hipError_t e = walk(A, dev, want, n_want > 0 ? n_want : 1);
        if (A.spare[want].empty() && A.budget_spent) return acquire_granule(A, dev, want, n_want, out);
        if (A.spare[want].empty()) {
            // what the search pooled of the other classes goes back before the mark is taken
            trim_pools(A);
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) A.exhausted_free[want] = free_b + 1;
            (void)hipGetLastError();
            return e != hipSuccess ? e : hipErrorOutOfMemory;
        }
    }
    *out = A.spare[want].front();
    A.spare[want].erase(A.spare[want].begin());
    return hipSuccess;
}

// map `count` more granules of its class at the end of region `cls`
hipError_t extend_region(device_arena& A, int dev, int cls, size_t count)
{
    const size_t gr = granule_bytes();
    region& R = A.reg[cls];
    for (size_t i = 0; i < count; ++i) {
        if (R.size + gr > R.reserved) return hipErrorOutOfMemory;
        hipMemGenericAllocationHandle_t h;
        hipError_t e = acquire_granule(A, dev, cls, int(count - i), &h);
        if (e != hipSuccess) return e;
        e = map_granule(R.base + R.size, gr, h, dev);
        if (e != hipSuccess) {
            (void)hipMemRelease(h);
            return e;
        }
        R.granules.push_back(h);
        R.add_free(R.size, gr);
        R.size += gr;
        R.tail_retired = 0;     // retired addresses in front of the new granule stay dead
    }
    trim_pools(A);
    return hipSuccess;
}

hipError_t classes_init(device_arena& A, int dev)
{
    const size_t gr = granule_bytes();
    hipError_t e = hipStreamCreateWithFlags(&A.stream, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    void* va = nullptr;
    size_t total = 0, free_b = 0;
    (void)hipMemGetInfo(&free_b, &total);
    // twice the device memory: addresses retired by gkoc_arena_trim are never used again
    const size_t reserve = 2 * round_up(total ? total : size_t(288) * GiB, gr);
    for (int k = 0; k < max_classes; ++k) {
        e = hipMemAddressReserve(&va, reserve, gr, nullptr, 0);
        if (e != hipSuccess) return e;
        A.reg[k].base = static_cast<char*>(va);
        A.reg[k].reserved = reserve;
        A.reg[k].size = 0;
    }
    A.ready = true;
    return hipSuccess;
}

// The classes of this device, found ONCE, in one walk, before anything is placed in a second
// class (round 3 found them one by one as the roles asked for them: when the third did not show
// up, the index arrays already sat in class 1 and the vectors had to join them - BENCH_r03).
// With fewer than three classes the roles are mapped onto what exists (class_for_role).
void survey_classes(device_arena& A, int dev)
{
    if (A.surveyed) return;
    A.surveyed = true;
    (void)walk(A, dev, -1, 0);
    (void)hipGetLastError();
    trim_pools(A);
    if (g_verbose) {
        fprintf(stderr, "[gkoc arena] survey: %d class(es), %lld granules created, %lld classified, %.0f ms\n",
                A.n_cls, (long long)A.walked, (long long)A.classified, A.search_ns / 1e6);
    }
}

// Does a request of `bytes` look like a vector (or a block of vectors: a Krylov basis, a
// multi-vector) of a system whose vectors the kernels have already been seen to write?  Multiples
// up to 64 only: the value / index arrays of a matrix with a constant number of entries per row
// (k n values of 8 bytes) are exact multiples of its n-vector as well.
bool vector_shaped(const device_arena& A, size_t bytes)
{
    // what the binding KNOWS beats what the size suggests: the values of an ELL matrix with k entries per
    // row are k n values - a multiple of the n-vector - and were classified as vectors once an SpMV
    // output had been seen
    for (size_t m : A.matrix_sizes) {
        if (m == bytes) return false;
    }
    for (size_t v : A.vector_sizes) {
        if (v > 0 && bytes % v == 0 && bytes / v <= 64) return true;
    }
    return false;
}

// Where a request goes.  An explicit role has its class (values 0, indices 1, vectors 2).
// gkoc_malloc / HipExecutor::raw_alloc know no role (Ginkgo's raw_alloc has no such argument); what
// matters is that arrays kernels WRITE do not share a class with the big read-only streams of the
// kernels that write them.  Rule, evaluated against the allocations that are live
// NOW (so that what was allocated first is re-judged once the matrix shows up):
//   * a live or requested array is "small" (a vector) if it is vector-shaped (see above) or smaller
//     than a quarter of the largest live one, "big" (a matrix array) otherwise;
//   * a big request goes to the class that holds the fewest small bytes, then the fewest big ones;
//   * a small request goes to the class that holds the fewest big bytes, then the most small ones.
// A matrix first: values 0, indices 1, vectors 2 (as with explicit roles).  Vectors first (b, x
// before A): they spread over two classes while nothing else is known, the matrix arrays then
// share the third one - no written stream next to a matrix array.  A Krylov basis (a multiple of
// a vector the kernels have written) joins the vectors although it is the largest request.
// ncls: the number of classes to choose from (3 while the device has not been surveyed).  With two
// classes the matrix arrays share class 0 and everything kernels write gets class 1 (values and
// indices in one class cost 2 %, a written vector next to either costs 6 - 11 %: DESIGN.md 3.2);
// with one class there is nothing to choose.
int class_for_role(const device_arena& A, int role, size_t bytes, int ncls)
{
    if (ncls <= 1) return 0;
    if (role != GKOC_MEM_AUTO) {
        if (ncls == 2) return role == GKOC_MEM_VECTOR ? 1 : 0;
        return role == GKOC_MEM_VALUES ? 0 : role == GKOC_MEM_INDICES ? 1 : 2;
    }
    size_t largest = bytes;
    for (int k = 0; k < max_classes; ++k) {
        for (const auto& u : A.reg[k].req) largest = std::max(largest, u.second);
    }
    auto is_small = [&](size_t b) { return vector_shaped(A, b) || b * 4 < largest; };
    size_t big_b[max_classes] = {}, small_b[max_classes] = {};
    for (int k = 0; k < ncls; ++k) {
        for (const auto& u : A.reg[k].req) (is_small(u.second) ? small_b[k] : big_b[k]) += u.second;
    }
    int best = 0;
    if (!is_small(bytes)) {
        for (int k = 1; k < ncls; ++k) {
            if (small_b[k] < small_b[best] || (small_b[k] == small_b[best] && big_b[k] < big_b[best])) best = k;
        }
    } else {
        // (ties: the highest class, where explicit roles put vectors)
        best = ncls - 1;
        for (int k = ncls - 2; k >= 0; --k) {
            if (big_b[k] < big_b[best] || (big_b[k] == big_b[best] && small_b[k] > small_b[best])) best = k;
        }
    }
    return best;
}

int span_list_malloc(std::vector<span*>& list, size_t chunk_bytes, void** ptr, size_t need,
                     size_t align)
{
    for (span* c : list) {
        if (c->size - c->in_use < need) continue;
        if (void* p = c->take(need, align)) {
            *ptr = p;
            return GKOC_OK;
        }
    }
    span* c = new span;
    const size_t want = need > chunk_bytes ? round_up(need, 2 * MiB) : chunk_bytes;
    void* base = nullptr;
    hipError_t e = hipMalloc(&base, want);
    size_t got = want;
    if (e != hipSuccess && want > round_up(need, 2 * MiB)) {
        (void)hipGetLastError();
        got = round_up(need, 2 * MiB);
        e = hipMalloc(&base, got);
    }
    if (e != hipSuccess) {
        delete c;
        return hip_fail(e, "arena chunk allocation", __FILE__, __LINE__);
    }
    c->base = static_cast<char*>(base);
    c->size = got;
    c->free_list[0] = got;
    list.push_back(c);
    *ptr = c->take(need, align);
    return *ptr ? GKOC_OK : GKOC_E_INVALID;
}

// 1 = freed, 0 = not ours, <0 = error
int span_list_free(std::vector<span*>& list, void* ptr, size_t regular_size, bool release_empty_odd)
{
    for (size_t i = 0; i < list.size(); ++i) {
        span* c = list[i];
        if (!c->owns(ptr)) continue;
        if (!c->give(ptr)) return GKOC_E_INVALID;
        if (release_empty_odd && c->in_use == 0 && c->size != regular_size) {
            (void)hipFree(c->base);
            delete c;
            list.erase(list.begin() + i);
        }
        return 1;
    }
    return 0;
}

}  // namespace

int arena_malloc(void** ptr, size_t bytes, int role)
{
    *ptr = nullptr;
    if (bytes == 0) return GKOC_OK;
    std::lock_guard<std::mutex> g(g_mtx);
    read_env_locked();
    if (g_mode == 0) {
        GKOC_HIP(hipMalloc(ptr, bytes));
        return GKOC_OK;
    }
    const int dev = current_device();
    device_arena& A = g_arena[dev];
    const size_t need = round_up(bytes, 256);
    if (bytes < small_limit) return span_list_malloc(A.small, small_chunk, ptr, need, 256);
    const size_t align = 2 * MiB;
    if (g_mode == 2 && !A.gated) {
        // the class layout was measured on MI355X (gfx950, 288 GiB, one memory partition): on
        // anything else class regions are opt-in (GKOC_ARENA=2 / gkoc_arena_configure)
        A.gated = true;
        if (!g_mode_from_env) {
            hipDeviceProp_t prop;
            const bool known = hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                               std::string(prop.gcnArchName).rfind("gfx950", 0) == 0 &&
                               prop.totalGlobalMem >= size_t(200) * GiB;
            (void)hipGetLastError();
            if (!known) A.failed = true;
        }
    }
    if (g_mode == 1 || A.failed) {
        return span_list_malloc(A.plain, g_chunk_bytes, ptr, need, align);
    }
    if (!A.ready) {
        const hipError_t e = classes_init(A, dev);
        if (e != hipSuccess) {
            // no virtual-memory API on this system: plain chunks
            (void)hipGetLastError();
            A.failed = true;
            return span_list_malloc(A.plain, g_chunk_bytes, ptr, need, align);
        }
    }
    if (A.n_cls == 0) {
        // the first granule is class 0 by definition
        hipMemGenericAllocationHandle_t h;
        const hipMemAllocationProp prop = granule_prop(dev);
        if (hipMemCreate(&h, granule_bytes(), &prop, 0) != hipSuccess || !start_class(A, dev, h)) {
            (void)hipGetLastError();
            A.failed = true;
            return span_list_malloc(A.plain, g_chunk_bytes, ptr, need, align);
        }
        ++A.walked;
    }
    // a request that wants a second class sends the survey out (once); afterwards the roles are
    // mapped onto the classes that exist
    if (!A.surveyed && class_for_role(A, role, bytes, max_classes) >= A.n_cls) survey_classes(A, dev);
    int cls = class_for_role(A, role, bytes, A.surveyed ? A.n_cls : max_classes);
    if (cls >= A.n_cls) cls = A.n_cls - 1;
    const size_t gr = granule_bytes();
    for (int attempt = 0; attempt < max_classes; ++attempt) {
        region& R = A.reg[cls];
        if (void* p = R.take(need, align)) {
            *ptr = p;
            R.req[size_t(static_cast<char*>(p) - R.base)] = bytes;
            return GKOC_OK;
        }
        const size_t tail = R.tail_retired ? 0 : R.free_tail();
        const size_t count = tail >= need + 2 * MiB ? 1 : (need + 2 * MiB - tail + gr - 1) / gr;
        if (extend_region(A, dev, cls, count) == hipSuccess) {
            if (void* p = R.take(need, align)) {
                *ptr = p;
                R.req[size_t(static_cast<char*>(p) - R.base)] = bytes;
                return GKOC_OK;
            }
        }
        (void)hipGetLastError();
        // this class is exhausted: any other class is better than failing
        cls = (cls + 1) % A.n_cls;
    }
    set_last_error("arena: out of device memory for a request of %zu bytes", bytes);
    return static_cast<int>(hipErrorOutOfMemory);
}

static int arena_free_impl(void* ptr, bool sync)
{
    if (!ptr) return GKOC_OK;
    {
        std::unique_lock<std::mutex> g(g_mtx);
        for (int dev = 0; dev < 64; ++dev) {
            device_arena& A = g_arena[dev];
            bool ours = false;
            for (span* c : A.small) ours = ours || c->owns(ptr);
            for (span* c : A.plain) ours = ours || c->owns(ptr);
            for (int k = 0; k < A.n_cls; ++k) ours = ours || A.reg[k].owns(ptr);
            if (!ours) continue;
            if (sync && g_sync_free) {
                // hipFree semantics: nothing enqueued earlier still uses the block
                g.unlock();
                GKOC_HIP(hipDeviceSynchronize());
                g.lock();
            }
            int r = span_list_free(A.small, ptr, small_chunk, false);
            if (r == 0) r = span_list_free(A.plain, ptr, g_chunk_bytes, true);
            for (int k = 0; r == 0 && k < A.n_cls; ++k) {
                if (A.reg[k].owns(ptr)) {
                    r = A.reg[k].give(ptr) ? 1 : GKOC_E_INVALID;
                    if (r == 1) A.reg[k].req.erase(size_t(static_cast<char*>(ptr) - A.reg[k].base));
                }
            }
            if (r < 0) {
                set_last_error("gkoc_free: %p is inside the arena but not the start of an "
                               "allocation", ptr);
                return r;
            }
            return GKOC_OK;
        }
    }
    GKOC_HIP(hipFree(ptr));
    return GKOC_OK;
}

int arena_free(void* ptr) { return arena_free_impl(ptr, true); }

// is ptr inside memory the arena hands out (any mode, any device)?
bool arena_owns(const void* ptr)
{
    std::lock_guard<std::mutex> g(g_mtx);
    for (int dev = 0; dev < 64; ++dev) {
        const device_arena& A = g_arena[dev];
        for (const span* c : A.small) if (c->owns(ptr)) return true;
        for (const span* c : A.plain) if (c->owns(ptr)) return true;
        for (int k = 0; k < A.n_cls; ++k) if (A.reg[k].owns(ptr)) return true;
    }
    return false;
}

// ---- stream-ordered scratch ---------------------------------------------------
namespace {
struct pending_free {
    void* ptr;
    hipEvent_t done;
};
std::mutex g_scratch_mtx;
std::vector<pending_free> g_pending;

void reclaim_scratch(bool wait)
{
    std::vector<pending_free> ready;
    {
        std::lock_guard<std::mutex> g(g_scratch_mtx);
        for (size_t i = 0; i < g_pending.size();) {
            const hipError_t e = wait ? hipEventSynchronize(g_pending[i].done)
                                      : hipEventQuery(g_pending[i].done);
            if (e == hipSuccess) {
                ready.push_back(g_pending[i]);
                g_pending.erase(g_pending.begin() + i);
            } else {
                (void)hipGetLastError();
                ++i;
            }
        }
    }
    for (auto& p : ready) {
        (void)hipEventDestroy(p.done);
        (void)arena_free_impl(p.ptr, false);
    }
}
}  // namespace

int scratch_malloc(hipStream_t, void** ptr, size_t bytes)
{
    reclaim_scratch(false);
    int rc = arena_malloc(ptr, bytes ? bytes : 1, GKOC_MEM_VECTOR);
    if (rc != GKOC_OK) {
        // memory may be held by scratch whose stream has not caught up yet
        reclaim_scratch(true);
        rc = arena_malloc(ptr, bytes ? bytes : 1, GKOC_MEM_VECTOR);
    }
    return rc;
}

int scratch_free(hipStream_t st, void* ptr)
{
    if (!ptr) return GKOC_OK;
    hipEvent_t ev;
    GKOC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    GKOC_HIP(hipEventRecord(ev, st));
    std::lock_guard<std::mutex> g(g_scratch_mtx);
    g_pending.push_back({ptr, ev});
    return GKOC_OK;
}

}  // namespace gkoc

using namespace gkoc;

extern "C" {

int gkoc_malloc_role(void** ptr, size_t bytes, int role)
{
    GKOC_REQUIRE(ptr, GKOC_E_INVALID, "ptr == NULL");
    GKOC_REQUIRE(role >= GKOC_MEM_AUTO && role <= GKOC_MEM_VECTOR, GKOC_E_INVALID, "unknown role");
    return arena_malloc(ptr, bytes, role);
}

int gkoc_arena_configure(int mode, size_t chunk_bytes, int sync_on_free)
{
    GKOC_REQUIRE(mode >= 0 && mode <= 2, GKOC_E_INVALID, "arena mode must be 0..2");
    std::lock_guard<std::mutex> g(g_mtx);
    read_env_locked();
    g_mode = mode;
    g_mode_from_env = true;     // an explicit request counts like the environment variable
    if (chunk_bytes) g_chunk_bytes = round_up(chunk_bytes, 2 * MiB);
    g_sync_free = sync_on_free ? 1 : 0;
    return GKOC_OK;
}

int gkoc_arena_stats(gkoc_arena_info* info)
{
    GKOC_REQUIRE(info, GKOC_E_INVALID, "info == NULL");
    std::lock_guard<std::mutex> g(g_mtx);
    read_env_locked();
    const device_arena& A = g_arena[current_device()];
    *info = gkoc_arena_info{};
    info->mode = g_mode;
    info->chunk_bytes = int64_t(g_mode == 2 ? granule_bytes() : g_chunk_bytes);
    for (int k = 0; k < max_classes; ++k) info->spare_bytes += int64_t(A.spare[k].size() * granule_bytes());
    info->num_classes = A.n_cls;
    info->probes = A.probes;
    info->granules_walked = A.walked;
    info->granules_classified = A.classified;
    info->search_ns = A.search_ns;
    info->probe_retries = A.retried;
    info->surveyed = A.surveyed ? 1 : 0;
    info->search_budget_ms = g_survey_ms;
    info->search_budget_spent = A.budget_spent ? 1 : 0;
    info->granules_unclassified = A.unclassified;
    auto add = [&](const span* c) {
        info->num_chunks += 1;
        info->reserved_bytes += int64_t(c->size);
        info->used_bytes += int64_t(c->in_use);
        info->num_allocations += int64_t(c->used.size());
    };
    for (const span* c : A.small) add(c);
    for (const span* c : A.plain) add(c);
    for (int k = 0; k < A.n_cls; ++k) {
        const region& R = A.reg[k];
        info->num_chunks += int64_t(R.granules.size());
        info->reserved_bytes += int64_t(R.size - R.retired);
        info->used_bytes += int64_t(R.in_use);
        info->num_allocations += int64_t(R.used.size());
        info->class_reserved_bytes[k] = int64_t(R.size - R.retired);
        info->class_used_bytes[k] = int64_t(R.in_use);
    }
    return GKOC_OK;
}

int gkoc_arena_class_of(const void* ptr, int* cls)
{
    GKOC_REQUIRE(cls, GKOC_E_INVALID, "cls == NULL");
    std::lock_guard<std::mutex> g(g_mtx);
    const device_arena& A = g_arena[current_device()];
    *cls = -1;
    for (int k = 0; k < A.n_cls; ++k) {
        if (A.reg[k].owns(ptr)) *cls = k;
    }
    return GKOC_OK;
}

// A kernel has written the array that holds ptr as a VECTOR (the output of an SpMV): requests of
// that size, or of a multiple of it (a Krylov basis, several right-hand sides), are vectors from
// now on whatever their size relative to the matrix (class_for_role).  Called by the Ginkgo
// binding; cheap (a map look-up under the arena's lock).
int gkoc_arena_note_vector(const void* ptr)
{
    if (!ptr) return GKOC_OK;
    std::lock_guard<std::mutex> g(g_mtx);
    device_arena& A = g_arena[current_device()];
    for (int k = 0; k < A.n_cls; ++k) {
        region& R = A.reg[k];
        if (!R.owns(ptr)) continue;
        const size_t off = size_t(static_cast<const char*>(ptr) - R.base);
        auto it = R.req.upper_bound(off);
        if (it == R.req.begin()) return GKOC_OK;
        --it;
        const size_t bytes = it->second;
        if (off >= it->first + bytes) return GKOC_OK;
        for (size_t v : A.vector_sizes) {
            if (v == bytes) return GKOC_OK;
        }
        // a written vector that sits in a class with arrays at least four times its size
        for (const auto& u : R.req) {
            if (u.second >= 4 * bytes) {
                ++A.misplaced;
                break;
            }
        }
        if (A.vector_sizes.size() >= 8) A.vector_sizes.erase(A.vector_sizes.begin());
        A.vector_sizes.push_back(bytes);
        for (auto& cs : A.contested_sizes) {
            if (cs.first != bytes || cs.second <= 2) continue;
            // settled as a vector (see contested_sizes): the matrix note of this size goes
            for (size_t i = 0; i < A.matrix_sizes.size(); ++i) {
                if (A.matrix_sizes[i] == bytes) {
                    A.matrix_sizes.erase(A.matrix_sizes.begin() + i);
                    break;
                }
            }
        }
        return GKOC_OK;
    }
    return GKOC_OK;
}

// The array at `ptr` is a matrix array (values / column indices handed to an SpMV kernel): later
// requests of its size are matrix arrays for class_for_role, whatever multiple of a vector they are -
// e.g. the next matrix of the same shape, or this one re-created after a conversion.
int gkoc_arena_note_matrix(const void* ptr)
{
    if (!ptr) return GKOC_OK;
    std::lock_guard<std::mutex> g(g_mtx);
    device_arena& A = g_arena[current_device()];
    for (int k = 0; k < A.n_cls; ++k) {
        region& R = A.reg[k];
        if (!R.owns(ptr)) continue;
        const size_t off = size_t(static_cast<const char*>(ptr) - R.base);
        auto it = R.req.upper_bound(off);
        if (it == R.req.begin()) return GKOC_OK;
        --it;
        const size_t bytes = it->second;
        if (off >= it->first + bytes) return GKOC_OK;
        for (size_t m : A.matrix_sizes) {
            if (m == bytes) return GKOC_OK;
        }
        for (size_t v : A.vector_sizes) {
            if (v != bytes) continue;
            // noted as a written vector before: a mistake the first time (k n values seen before any SpMV),
            // a matrix with a vector's size if it keeps coming back
            int* flips = nullptr;
            for (auto& cs : A.contested_sizes) {
                if (cs.first == bytes) flips = &cs.second;
            }
            if (flips == nullptr) {
                if (A.contested_sizes.size() >= 16) A.contested_sizes.erase(A.contested_sizes.begin());
                A.contested_sizes.emplace_back(bytes, 0);
                flips = &A.contested_sizes.back().second;
            }
            if (++*flips > 2) return GKOC_OK;      // stays a vector
        }
        if (A.matrix_sizes.size() >= 16) A.matrix_sizes.erase(A.matrix_sizes.begin());
        A.matrix_sizes.push_back(bytes);
        // (a size noted as a vector's earlier by mistake - k n values seen before any SpMV - is none)
        for (size_t i = 0; i < A.vector_sizes.size(); ++i) {
            if (A.vector_sizes[i] == bytes) {
                A.vector_sizes.erase(A.vector_sizes.begin() + i);
                break;
            }
        }
        return GKOC_OK;
    }
    return GKOC_OK;
}

int gkoc_arena_role_stats(int64_t* n_vector_sizes, int64_t* misplaced_vectors)
{
    std::lock_guard<std::mutex> g(g_mtx);
    const device_arena& A = g_arena[current_device()];
    if (n_vector_sizes) *n_vector_sizes = int64_t(A.vector_sizes.size());
    if (misplaced_vectors) *misplaced_vectors = A.misplaced;
    return GKOC_OK;
}

int gkoc_arena_probe(const void* x, size_t x_bytes, void* y, int read_kb_per_wave,
                     int write_bytes_per_wave, int reps, int64_t* ns)
{
    GKOC_REQUIRE(x && y && ns && read_kb_per_wave > 0 && write_bytes_per_wave >= 1024 && reps > 0,
                 GKOC_E_INVALID, "bad probe arguments");
    return probe_ns(nullptr, x, x_bytes, y, read_kb_per_wave, write_bytes_per_wave, reps, ns);
}

int gkoc_arena_trim(void)
{
    std::lock_guard<std::mutex> g(g_mtx);
    const int dev = current_device();
    device_arena& A = g_arena[dev];
    for (auto* list : {&A.small, &A.plain}) {
        for (size_t i = 0; i < list->size();) {
            span* c = (*list)[i];
            if (c->in_use == 0) {
                (void)hipFree(c->base);
                delete c;
                list->erase(list->begin() + i);
            } else {
                ++i;
            }
        }
    }
    for (int k = 0; k < max_classes; ++k) {
        for (auto h : A.spare[k]) (void)hipMemRelease(h);
        A.spare[k].clear();
    }
    // (without the synchronisation of gkoc_free a kernel in flight may still touch a block that
    // was handed back a moment ago: nothing is unmapped under it)
    if (!g_sync_free) (void)hipDeviceSynchronize();
    // trailing free granules of the regions go back to the driver.  Their addresses are RETIRED
    // (the range stays inside the region, neither free nor used: a later extension maps fresh
    // addresses behind it), because an address that was unmapped must not be mapped to other
    // memory again on this system (stale translations, see device_arena::scratch).
    const size_t gr = granule_bytes();
    for (int k = 0; k < A.n_cls; ++k) {
        region& R = A.reg[k];
        while (R.granules.size() > 1 && !R.free_list.empty()) {
            const size_t live_end = R.size - R.tail_retired;
            auto last = std::prev(R.free_list.end());
            if (last->first + last->second != live_end || last->second < gr) break;
            const size_t off = live_end - gr;
            if (hipMemUnmap(R.base + off, gr) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            if (last->first < off) {
                last->second = off - last->first;
            } else {
                R.free_list.erase(last);
            }
            (void)hipMemRelease(R.granules.back());
            R.granules.pop_back();
            R.tail_retired += gr;
            R.retired += gr;
        }
    }
    return GKOC_OK;
}

}  // extern "C"
