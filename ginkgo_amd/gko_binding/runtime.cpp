// gko::HipExecutor runtime services on top of the C ABI.
// Replaces the definitions of core/device_hooks/hip_hooks.cpp:21-252 (stubs) /
// hip/base/{executor,memory,stream,timer,scoped_device_id,roctx}.hip.cpp (stock
// backend).  Interface: include/ginkgo/core/base/{executor,memory,stream,timer,
// scoped_device_id_guard}.hpp.
#include <chrono>
#include <cstring>

#include <ginkgo/core/base/event.hpp>
#include <ginkgo/core/base/memory.hpp>
#include <ginkgo/core/base/scoped_device_id_guard.hpp>
#include <ginkgo/core/base/stream.hpp>
#include <ginkgo/core/base/timer.hpp>
#include <ginkgo/core/base/version.hpp>
#include <ginkgo/core/log/profiler_hook.hpp>

#include "shim_common.hpp"

namespace gko {
namespace cdna4 {

// RAII device switch (the role of detail::hip_scoped_device_id_guard)
class device_guard : public ::gko::detail::generic_scoped_device_id_guard {
public:
    explicit device_guard(int device_id) : original_{-1}, changed_{false}
    {
        GKOC_CALL(gkoc_get_device(&original_));
        if (original_ != device_id) {
            GKOC_CALL(gkoc_set_device(device_id));
            changed_ = true;
        }
    }
    ~device_guard() override
    {
        if (changed_) gkoc_set_device(original_);  // must not throw
    }

private:
    int original_;
    bool changed_;
};

}  // namespace cdna4


version version_info::get_hip_version() noexcept
{
    return {GKOC_VERSION_MAJOR, GKOC_VERSION_MINOR, 0, "gko-cdna4 (gfx950)"};
}


// ---------------------------------------------------------------- allocators
// HipAllocator = the library's arena (csrc/arena.hip): device memory in regions of
// different memory classes, requests placed by size (Ginkgo's raw_alloc carries no
// role): matrix-sized arrays and vector-sized arrays do not share a class, which is
// worth 11 % on the SpMV and the block-Jacobi apply
void* HipAllocator::allocate(size_type num_bytes)
{
    void* p = nullptr;
    const int role = cdna4::alloc_role_hint;
    cdna4::alloc_role_hint = 0;
    if (role != 0) {
        GKOC_CALL(gkoc_malloc_role(&p, num_bytes, role));
    } else {
        GKOC_CALL(gkoc_malloc(&p, num_bytes));
    }
    return p;
}

void HipAllocator::deallocate(void* dev_ptr) { gkoc_free(dev_ptr); }

// stream-ordered / unified / pinned-host allocation modes are outside the hot
// path; they allocate plain device memory through the same entry point so that
// user code selecting them keeps working (no host access to such buffers).
HipAsyncAllocator::HipAsyncAllocator(GKO_HIP_STREAM_STRUCT* stream) : stream_{stream} {}

void* HipAsyncAllocator::allocate(size_type num_bytes)
{
    void* p = nullptr;
    GKOC_CALL(gkoc_malloc(&p, num_bytes));
    return p;
}

void HipAsyncAllocator::deallocate(void* dev_ptr)
{
    gkoc_stream_synchronize(reinterpret_cast<gkoc_stream_t>(stream_));
    gkoc_free(dev_ptr);
}

bool HipAsyncAllocator::check_environment(int, GKO_HIP_STREAM_STRUCT* stream) const
{
    return stream == stream_;
}

HipUnifiedAllocator::HipUnifiedAllocator(int device_id)
    : HipUnifiedAllocator{device_id, 1u}
{}

HipUnifiedAllocator::HipUnifiedAllocator(int device_id, unsigned int flags)
    : device_id_{device_id}, flags_{flags}
{}

void* HipUnifiedAllocator::allocate(size_type num_bytes)
{
    void* p = nullptr;
    GKOC_CALL(gkoc_malloc_managed(&p, num_bytes, flags_));
    return p;
}

void HipUnifiedAllocator::deallocate(void* dev_ptr) { gkoc_free(dev_ptr); }

bool HipUnifiedAllocator::check_environment(int device_id, GKO_HIP_STREAM_STRUCT*) const
{
    return device_id == device_id_;
}

HipHostAllocator::HipHostAllocator(int device_id) : device_id_{device_id} {}

void* HipHostAllocator::allocate(size_type num_bytes)
{
    void* p = nullptr;
    GKOC_CALL(gkoc_malloc_host(&p, num_bytes));
    return p;
}

void HipHostAllocator::deallocate(void* ptr) { gkoc_free_host(ptr); }

bool HipHostAllocator::check_environment(int device_id, GKO_HIP_STREAM_STRUCT*) const
{
    return device_id == device_id_;
}


// ------------------------------------------------------------------ executor
std::shared_ptr<HipExecutor> HipExecutor::create(
    int device_id, std::shared_ptr<Executor> master, bool,
    allocation_mode alloc_mode, GKO_HIP_STREAM_STRUCT* stream)
{
    // same mapping as the stock backend (hip/base/executor.hip.cpp:27-42);
    // 1 / 2 = hipMemAttachGlobal / hipMemAttachHost
    std::shared_ptr<HipAllocatorBase> alloc;
    switch (alloc_mode) {
    case allocation_mode::device:
        alloc = std::make_shared<HipAllocator>();
        break;
    case allocation_mode::unified_global:
        alloc = std::make_shared<HipUnifiedAllocator>(device_id, 1u);
        break;
    case allocation_mode::unified_host:
        alloc = std::make_shared<HipUnifiedAllocator>(device_id, 2u);
        break;
    default:
        throw ::gko::NotSupported(__FILE__, __LINE__, __func__, "allocation_mode");
    }
    return create(device_id, std::move(master), std::move(alloc), stream);
}

std::shared_ptr<HipExecutor> HipExecutor::create(
    int device_id, std::shared_ptr<Executor> master,
    std::shared_ptr<HipAllocatorBase> alloc, GKO_HIP_STREAM_STRUCT* stream)
{
    if (!alloc->check_environment(device_id, stream)) {
        throw Error{__FILE__, __LINE__,
                    "Allocator uses incorrect stream or device ID."};
    }
    return std::shared_ptr<HipExecutor>(
        new HipExecutor(device_id, std::move(master), std::move(alloc), stream));
}

void HipExecutor::populate_exec_info(const machine_topology*)
{
    // NUMA / PCI affinity lookups need hwloc, which this build does not use
}

int HipExecutor::get_num_devices()
{
    int n = 0;
    GKOC_CALL(gkoc_get_num_devices(&n));
    return n;
}

void HipExecutor::set_gpu_property()
{
    if (this->get_device_id() < 0 || this->get_device_id() >= get_num_devices()) {
        return;
    }
    gkoc_device_info info;
    GKOC_CALL(gkoc_get_device_info(this->get_device_id(), &info));
    auto& ei = this->get_exec_info();
    ei.num_computing_units = info.num_cu;
    ei.major = info.major;
    ei.minor = info.minor;
    ei.max_workgroup_size = info.max_threads_per_block;
    ei.max_workitem_sizes = {info.max_threads_per_block,
                             info.max_threads_per_block, 64};
    ei.num_pu_per_cu = 4;                   // SIMDs per CU on CDNA4
    ei.max_subgroup_size = info.wave_size;  // 64: sizes the Jacobi storage scheme
}

void HipExecutor::init_handles()
{
    // no hipBLAS / hipSPARSE handles: nothing on the hot path uses them
}

void* HipExecutor::raw_alloc(size_type num_bytes) const
{
    cdna4::device_guard g(this->get_device_id());
    return alloc_->allocate(num_bytes);
}

void HipExecutor::raw_free(void* ptr) const noexcept
{
    try {
        cdna4::device_guard g(this->get_device_id());
        cdna4::launch_deferred();   // a held kernel may still have to read or write ptr
        cdna4::forget_learned_if(ptr);   // ... and what a solve has shown may name what goes away
        alloc_->deallocate(ptr);
    } catch (...) {
    }
}

void OmpExecutor::raw_copy_to(const HipExecutor* dest, size_type num_bytes,
                              const void* src_ptr, void* dest_ptr) const
{
    if (num_bytes > 0) {
        cdna4::device_guard g(dest->get_device_id());
        cdna4::launch_deferred();
        auto s = reinterpret_cast<gkoc_stream_t>(dest->get_stream());
        GKOC_CALL(gkoc_memcpy_h2d(dest_ptr, src_ptr, num_bytes, s));
        GKOC_CALL(gkoc_stream_synchronize(s));
    }
}

void HipExecutor::raw_copy_to(const OmpExecutor*, size_type num_bytes,
                              const void* src_ptr, void* dest_ptr) const
{
    if (num_bytes > 0) {
        cdna4::device_guard g(this->get_device_id());
        cdna4::launch_deferred();
        GKOC_CALL(gkoc_memcpy_d2h(dest_ptr, src_ptr, num_bytes,
                                  reinterpret_cast<gkoc_stream_t>(this->get_stream())));
    }
}

void HipExecutor::raw_copy_to(const HipExecutor*, size_type num_bytes,
                              const void* src_ptr, void* dest_ptr) const
{
    if (num_bytes > 0) {
        cdna4::device_guard g(this->get_device_id());
        cdna4::launch_deferred();
        auto s = reinterpret_cast<gkoc_stream_t>(this->get_stream());
        GKOC_CALL(gkoc_memcpy_d2d(dest_ptr, src_ptr, num_bytes, s));
        GKOC_CALL(gkoc_stream_synchronize(s));
    }
}

void HipExecutor::raw_copy_to(const CudaExecutor* dest, size_type, const void*,
                              void*) const GKO_NOT_SUPPORTED(dest);

void HipExecutor::raw_copy_to(const DpcppExecutor* dest, size_type, const void*,
                              void*) const GKO_NOT_SUPPORTED(dest);

void HipExecutor::synchronize() const
{
    cdna4::device_guard g(this->get_device_id());
    cdna4::launch_deferred();
    GKOC_CALL(gkoc_stream_synchronize(
        reinterpret_cast<gkoc_stream_t>(this->get_stream())));
}

scoped_device_id_guard HipExecutor::get_scoped_device_id_guard() const
{
    return {this, this->get_device_id()};
}

scoped_device_id_guard::scoped_device_id_guard(const HipExecutor*, int device_id)
    : scope_(std::make_unique<cdna4::device_guard>(device_id))
{}

std::string HipExecutor::get_description() const
{
    gkoc_device_info info;
    std::memset(&info, 0, sizeof(info));
    gkoc_get_device_info(this->get_device_id(), &info);
    return "HipExecutor (gko-cdna4) on device " +
           std::to_string(this->get_device_id()) + " (" + info.arch +
           ") with host " + this->get_master()->get_description();
}

std::string HipError::get_error(int64 error_code)
{
    return "HIP / gko-cdna4 error " + std::to_string(error_code);
}


// ------------------------------------------------------------------- streams
hip_stream::hip_stream() : stream_{nullptr}, device_id_{-1} {}

hip_stream::hip_stream(int device_id) : stream_{nullptr}, device_id_{device_id}
{
    cdna4::device_guard g(device_id_);
    gkoc_stream_t s = nullptr;
    GKOC_CALL(gkoc_stream_create(&s));
    stream_ = reinterpret_cast<GKO_HIP_STREAM_STRUCT*>(s);
}

hip_stream::~hip_stream()
{
    if (stream_) {
        try {
            cdna4::device_guard g(device_id_);
            gkoc_stream_destroy(reinterpret_cast<gkoc_stream_t>(stream_));
        } catch (...) {
        }
    }
}

hip_stream::hip_stream(hip_stream&& other)
    : stream_{std::exchange(other.stream_, nullptr)},
      device_id_{std::exchange(other.device_id_, -1)}
{}

GKO_HIP_STREAM_STRUCT* hip_stream::get() const { return stream_; }


// -------------------------------------------------------------------- timers
HipTimer::HipTimer(std::shared_ptr<const HipExecutor> exec)
    : device_id_{exec->get_device_id()}, stream_{exec->get_stream()}
{}

void HipTimer::init_time_point(time_point& time)
{
    time.type_ = time_point::type::hip;
    cdna4::device_guard g(device_id_);
    gkoc_event_t e = nullptr;
    GKOC_CALL(gkoc_event_create(&e));
    time.data_.hip_event = reinterpret_cast<GKO_HIP_EVENT_STRUCT*>(e);
}

void HipTimer::record(time_point& time)
{
    cdna4::device_guard g(device_id_);
    cdna4::launch_deferred();
    GKOC_CALL(gkoc_event_record(time.data_.hip_event,
                                reinterpret_cast<gkoc_stream_t>(stream_)));
}

void HipTimer::wait(time_point& time)
{
    cdna4::device_guard g(device_id_);
    GKOC_CALL(gkoc_event_synchronize(time.data_.hip_event));
}

std::chrono::nanoseconds HipTimer::difference_async(const time_point& start,
                                                    const time_point& stop)
{
    cdna4::device_guard g(device_id_);
    GKOC_CALL(gkoc_event_synchronize(stop.data_.hip_event));
    int64_t ns = 0;
    GKOC_CALL(gkoc_event_elapsed_ns(start.data_.hip_event, stop.data_.hip_event, &ns));
    return std::chrono::nanoseconds{ns};
}


namespace kernels {
namespace hip {

void reset_device(int) {}

void destroy_event(GKO_HIP_EVENT_STRUCT* event) { gkoc_event_destroy(event); }

std::string get_device_name(int device_id)
{
    gkoc_device_info info;
    GKOC_CALL(gkoc_get_device_info(device_id, &info));
    return info.arch;
}

namespace event {

namespace {
// an event on the executor's stream: recorded at construction (core/base/event_kernels.hpp;
// used by distributed::RowGatherer / Matrix to order the halo exchange after the pack kernel)
class stream_event : public ::gko::detail::Event {
public:
    explicit stream_event(std::shared_ptr<const HipExecutor> exec) : exec_{std::move(exec)}
    {
        cdna4::device_guard g(exec_->get_device_id());
        GKOC_CALL(gkoc_event_create(&ev_));
        cdna4::launch_deferred();
        GKOC_CALL(gkoc_event_record(ev_, reinterpret_cast<gkoc_stream_t>(exec_->get_stream())));
    }
    ~stream_event()
    {
        cdna4::device_guard g(exec_->get_device_id());
        gkoc_event_destroy(ev_);
    }
    void synchronize() const override
    {
        cdna4::device_guard g(exec_->get_device_id());
        GKOC_CALL(gkoc_event_synchronize(ev_));
    }

private:
    std::shared_ptr<const HipExecutor> exec_;
    gkoc_event_t ev_{};
};
}  // namespace

void record_event(std::shared_ptr<const HipExecutor> exec,
                  std::shared_ptr<const ::gko::detail::Event>& event)
{
    event = std::make_shared<stream_event>(std::move(exec));
}

}  // namespace event

}  // namespace hip
}  // namespace kernels


namespace log {

// ROCTX ranges: bound at run time by the library (gkoc_range_push / pop), no-ops without a
// roctx library on the system
void begin_roctx(const char* name, profile_event_category) { gkoc_range_push(name); }
void end_roctx(const char*, profile_event_category) { gkoc_range_pop(); }

}  // namespace log
}  // namespace gko
