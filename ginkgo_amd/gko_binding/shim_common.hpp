// Ginkgo-side binding of libgko_cdna4.so: shared helpers.
//
// This directory is the shim INTEGRATION.md describes: it is compiled against
// the UNMODIFIED Ginkgo headers (public include/ + the core/**_kernels.hpp
// declarations) and defines, with strong linkage, the `gko::HipExecutor`
// runtime members and the hot-path `gko::kernels::hip::*` kernels by forwarding
// to the C ABI (include/gko_cdna4.h).  Linked together with Ginkgo's own stub
// object (core/device_hooks/hip_hooks.cpp, all symbols weakened) it forms a
// link-compatible replacement for libginkgo_hip.so.  No device code and no HIP
// headers are needed here: everything goes through the C ABI.
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <string>

#include <ginkgo/core/base/array.hpp>
#include <ginkgo/core/base/exception.hpp>
#include <ginkgo/core/base/executor.hpp>
#include <ginkgo/core/base/types.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/stop/stopping_status.hpp>

#include "complex_abi.hpp"

namespace gko {
namespace cdna4 {

// error convention of SURVEY.md 8(b): non-zero status -> gko::Error subclass
inline void check(int status, const char* file, int line, const char* what)
{
    if (status == GKOC_OK) return;
    const std::string msg = std::string(what) + ": " + gkoc_last_error();
    if (status == GKOC_E_NOT_SUPPORTED) {
        throw ::gko::NotSupported(file, line, what, gkoc_last_error());
    }
    if (status == GKOC_E_OVERFLOW) {
        throw ::gko::OverflowError(file, line, gkoc_last_error());
    }
    throw ::gko::HipError(file, line, msg, status);
}

#define GKOC_CALL(expr) ::gko::cdna4::check((expr), __FILE__, __LINE__, #expr)

// Fusion across calls (fusion.cpp): cg::step_2 and the block-Jacobi application that follows it
// are held back until the next call shows whether one kernel can do them together with the dot
// product.  Everything that enters the backend goes through launch_deferred() first.
// a role for the NEXT HipAllocator::allocate of this thread (GKOC_MEM_*; 0 = none): the one place where the binding
// knows what an array Ginkgo allocates is for - jacobi::generate re-homes the block array (kernels.cpp)
extern thread_local int alloc_role_hint;
extern thread_local int deferred_state;   // != 0: this thread holds something or caches a norm
void flush_deferred();
// "Nothing has entered the backend since" is a statement about ALL host threads: Executor::run may be called
// from any thread on one executor and one stream (include/ginkgo/core/base/executor.hpp:1283-1289), and what
// thread A's cg::step_2 left behind for ||r|| is void once thread B has written r.  Every entry into the
// backend, from any thread, therefore takes a number from ONE process-wide counter; a thread's by-products
// carry the number of the call that produced them, and a later call of that thread may use them only if no
// number has been handed out in between (fusion.cpp: chain_epoch).  Reading calls (dot, norm2) take no number.
extern std::atomic<uint64_t> backend_epoch;
extern thread_local uint64_t last_entry_epoch;      // the number this thread's last entry was given
inline void launch_deferred()
{
    last_entry_epoch = backend_epoch.fetch_add(1, std::memory_order_acq_rel) + 1;
    if (deferred_state != 0) flush_deferred();
}
bool hold_step_2(int vt, gkoc_stream_t s, int64_t n, void* x, void* r, const void* p, const void* q,
                 const void* beta, const void* rho, const uint8_t* stop);
bool hold_jacobi_apply(int vt, int it, gkoc_stream_t s, int64_t num_blocks, uint32_t max_bs,
                       gkoc_jacobi_scheme scheme, const void* block_ptrs, const void* blocks,
                       const void* b, int64_t n, void* z);
bool hold_sub_scaled(int vt, gkoc_stream_t s, int64_t n, const void* alpha, const void* x, void* y);
bool fused_dot(int vt, gkoc_stream_t s, int64_t n, const void* x, const void* y, void* result,
               array<char>& tmp);
bool cached_norm2(int vt, gkoc_stream_t s, int64_t n, const void* x, void* result);
// by-products (fusion.cpp): nothing held, two kernels leave ||r|| / <b, z> behind for the calls
// that would otherwise compute them with a pass of their own
bool step_2_with_norm(int vt, int dev, gkoc_stream_t s, int64_t n, void* x, void* r, const void* p,
                      const void* q, const void* beta, const void* rho, const uint8_t* stop);
bool jacobi_apply_with_dot(int vt, int it, int dev, gkoc_stream_t s, int64_t num_blocks, uint32_t max_bs,
                           gkoc_jacobi_scheme scheme, const void* block_ptrs, const void* blocks,
                           const void* b, int64_t n, void* z);
void launch_deferred_for_read(const void* result);
// round 6: csr::spmv(A, b -> c) that also leaves <b, c> where the dot product this solve has shown to follow it
// will be asked to put it (fusion.cpp); false: the caller runs the plain product
bool spmv_with_dot(int vt, int it, int dev, gkoc_stream_t s, int64_t n, const void* row_ptrs, const void* cols,
                   const void* vals, const void* b, void* c);
// round 6 (fusion.cpp): the criterion's entry runs the cg::step_1 this solve has shown to follow it
bool criterion_then_step_1(int vt, gkoc_stream_t s, const void* tau, const void* orig_tau, double goal,
                           uint8_t stopping_id, bool set_finalized, bool implicit, uint8_t* stop, uint8_t* flags,
                           int* all_converged, int* one_changed);
bool step_1_done_ahead(int vt, gkoc_stream_t s, int64_t n, void* p, const void* z, const void* rho,
                       const void* prev_rho, const uint8_t* stop);
// what the by-product mode has learned about the running solve (step_2 -> block-Jacobi application of r)
// is void: a new solve begins, memory is freed, a preconditioner is generated
void forget_learned();
void forget_learned_if(const void* freed);

// the stream of a kernel launch: what was held back is launched first
inline gkoc_stream_t stream_of(const std::shared_ptr<const HipExecutor>& exec)
{
    launch_deferred();
    return reinterpret_cast<gkoc_stream_t>(exec->get_stream());
}
// ... for the calls that take part in the fusion themselves
inline gkoc_stream_t stream_keeping_deferred(const std::shared_ptr<const HipExecutor>& exec)
{
    return reinterpret_cast<gkoc_stream_t>(exec->get_stream());
}
template <typename T>
constexpr int vt_of()
{
    return sizeof(T) == 8 ? 0 : 1;
}
template <typename I>
constexpr int it_of()
{
    return sizeof(I) == 4 ? 0 : 1;
}

template <typename T>
inline int64_t rows(const matrix::Dense<T>* d)
{
    return static_cast<int64_t>(d->get_size()[0]);
}
template <typename T>
inline int64_t cols(const matrix::Dense<T>* d)
{
    return static_cast<int64_t>(d->get_size()[1]);
}
template <typename T>
inline int64_t ld(const matrix::Dense<T>* d)
{
    return static_cast<int64_t>(d->get_stride());
}

inline uint8_t* raw(array<stopping_status>* s)
{
    return reinterpret_cast<uint8_t*>(s->get_data());
}
inline const uint8_t* raw(const array<stopping_status>* s)
{
    return reinterpret_cast<const uint8_t*>(s->get_const_data());
}
inline uint8_t* raw(stopping_status* s) { return reinterpret_cast<uint8_t*>(s); }
inline const uint8_t* raw(const stopping_status* s)
{
    return reinterpret_cast<const uint8_t*>(s);
}

}  // namespace cdna4
}  // namespace gko
