// Kernel fusion ACROSS calls of the unmodified Ginkgo core.
//
// Ginkgo's Cg (core/solver/cg.cpp:131-176) issues, back to back and with nothing in between,
//     cg::step_2(x, r, p, q, beta, rho)      x += t p, r -= t q
//     jacobi::simple_apply(r -> z)           next iteration's preconditioner application
//     dense::compute_conj_dot(r, z -> rho')
//     dense::compute_norm2(r -> tau)         stop::ResidualNorm::check_impl
// which re-read r three times and z once.  The library has all four in one kernel
// (gkoc_x_cg_step_2_jacobi_apply_*; x, r, z bit-identical to the separate kernels).  To use it
// behind an API that asks for them one at a time, the first two are HELD, not launched:
//     step_2 arrives            -> held
//     simple_apply(b == r)      -> held as well (fast-path block layout, one column)
//     compute_conj_dot(r, z)    -> ONE launch does all three and leaves ||r|| next to the workspace
//     compute_norm2(r)          -> 8-byte copy of that value
// The same for the modified Gram-Schmidt loop of Ginkgo's Gmres (core/solver/gmres.cpp:156-177), which
// alternates h_i = <v_i, w> and w -= h_i v_i:
//     sub_scaled(h_i, v_i, w)          -> held
//     compute_conj_dot(v_{i+1}, w)     -> ONE launch of gkoc_x_gmres_mgs_step_*: the update and the
//                                         next dot read w once (w bit-identical)
// EVERY other entry into the backend (any kernel: stream_of(); synchronize, copies, frees, events,
// timers: runtime.cpp) first launches what is held, unfused and in order, and forgets the cached
// norm, so nothing outside this file can observe the difference.  What can: code that obtains raw
// device pointers of a solver's INTERNAL vectors and reads them with its own launches between two
// backend calls - GKOC_TUNE_DEFERRED_FUSION = 0 (gkoc_tune_set / env GKOC_TUNE_5) switches the
// mechanism off.
#include "shim_common.hpp"

namespace gko {
namespace cdna4 {

thread_local int deferred_state = 0;

namespace {

struct held_ops {
    int stage = 0;   // 0 nothing, 1 step_2, 2 step_2 + simple_apply, 3 sub_scaled (y -= alpha x)
    int vt = 0, it = 0;
    gkoc_stream_t s = nullptr;
    int64_t n = 0;
    void *x = nullptr, *r = nullptr;
    const void *p = nullptr, *q = nullptr, *beta = nullptr, *rho = nullptr;
    const uint8_t* stop = nullptr;
    // stage 3: y -= alpha x
    const void *ss_alpha = nullptr, *ss_x = nullptr;
    void* ss_y = nullptr;
    int64_t num_blocks = 0;
    uint32_t max_bs = 0;
    gkoc_jacobi_scheme scheme{};
    const void *block_ptrs = nullptr, *blocks = nullptr;
    void* z = nullptr;
    // ||r|| left behind by the fused launch
    const void* norm_of = nullptr;
    const void* norm_at = nullptr;
    int norm_vt = 0;
    int64_t norm_n = 0;
    gkoc_stream_t norm_s = nullptr;
};

// Per thread: the calls that take part arrive back to back on the thread that runs the solver, and
// what a thread holds never outlives the solver's apply (the loop ends with the dot product and the
// criterion check).  Another thread - possibly working on another device - neither sees nor has to
// launch it.
thread_local held_ops held;

int enabled()
{
    int64_t v = 1;
    gkoc_tune_get(GKOC_TUNE_DEFERRED_FUSION, &v);
    return v != 0;
}

void publish() { deferred_state = held.stage | (held.norm_of ? 4 : 0); }

void launch_step_2(const held_ops& h)
{
    if (h.vt == 0) {
        GKOC_CALL(gkoc_cg_step_2_f64(h.s, h.n, 1, static_cast<double*>(h.x), 1, static_cast<double*>(h.r), 1,
                                     static_cast<const double*>(h.p), 1, static_cast<const double*>(h.q), 1,
                                     static_cast<const double*>(h.beta), static_cast<const double*>(h.rho),
                                     h.stop));
    } else {
        GKOC_CALL(gkoc_cg_step_2_f32(h.s, h.n, 1, static_cast<float*>(h.x), 1, static_cast<float*>(h.r), 1,
                                     static_cast<const float*>(h.p), 1, static_cast<const float*>(h.q), 1,
                                     static_cast<const float*>(h.beta), static_cast<const float*>(h.rho),
                                     h.stop));
    }
}

#define GKOC_FUSION_TYPES(_)  \
    _(0, 0, double, int32_t, f64, i32) _(0, 1, double, int64_t, f64, i64) \
    _(1, 0, float, int32_t, f32, i32) _(1, 1, float, int64_t, f32, i64)

void launch_apply(const held_ops& h)
{
#define CASE(VT, IT, T, I, TN, IN)                                                                  \
    if (h.vt == VT && h.it == IT) {                                                                 \
        GKOC_CALL(gkoc_jacobi_simple_apply_##TN##_##IN(                                             \
            h.s, h.num_blocks, h.max_bs, h.scheme, static_cast<const I*>(h.block_ptrs),             \
            static_cast<const T*>(h.blocks), static_cast<const T*>(h.r), 1, static_cast<T*>(h.z), 1, \
            1));                                                                                    \
    }
    GKOC_FUSION_TYPES(CASE)
#undef CASE
}

void launch_sub_scaled(const held_ops& h)
{
    if (h.vt == 0) {
        GKOC_CALL(gkoc_dense_sub_scaled_f64(h.s, h.n, 1, static_cast<const double*>(h.ss_alpha), 1,
                                            static_cast<const double*>(h.ss_x), 1,
                                            static_cast<double*>(h.ss_y), 1));
    } else {
        GKOC_CALL(gkoc_dense_sub_scaled_f32(h.s, h.n, 1, static_cast<const float*>(h.ss_alpha), 1,
                                            static_cast<const float*>(h.ss_x), 1,
                                            static_cast<float*>(h.ss_y), 1));
    }
}

}  // namespace

void flush_deferred()
{
    const held_ops h = held;
    held.stage = 0;
    held.norm_of = nullptr;
    publish();
    if (h.stage == 3) {
        launch_sub_scaled(h);
        return;
    }
    if (h.stage >= 1) launch_step_2(h);
    if (h.stage >= 2) launch_apply(h);
}

bool hold_sub_scaled(int vt, gkoc_stream_t s, int64_t n, const void* alpha, const void* x, void* y)
{
    // (the caller has been through stream_of(): nothing is held at this point)
    if (n <= 0 || x == y || !enabled()) return false;
    held.stage = 3;
    held.vt = vt;
    held.s = s;
    held.n = n;
    held.ss_alpha = alpha;
    held.ss_x = x;
    held.ss_y = y;
    publish();
    return true;
}

bool hold_step_2(int vt, gkoc_stream_t s, int64_t n, void* x, void* r, const void* p, const void* q,
                 const void* beta, const void* rho, const uint8_t* stop)
{
    // (the caller has been through stream_of(): nothing is held at this point)
    if (n <= 0 || !enabled()) return false;
    held.stage = 1;
    held.vt = vt;
    held.s = s;
    held.n = n;
    held.x = x;
    held.r = r;
    held.p = p;
    held.q = q;
    held.beta = beta;
    held.rho = rho;
    held.stop = stop;
    publish();
    return true;
}

bool hold_jacobi_apply(int vt, int it, gkoc_stream_t s, int64_t num_blocks, uint32_t max_bs,
                       gkoc_jacobi_scheme scheme, const void* block_ptrs, const void* blocks,
                       const void* b, int64_t n, void* z)
{
    const int64_t bo = scheme.block_offset;
    const bool fast_layout = bo >= 1 && bo <= 16 && (bo & (bo - 1)) == 0 &&
                             (bo << scheme.group_power) == 64 && int64_t(max_bs) <= bo;
    if (held.stage != 1 || held.vt != vt || held.s != s || held.n != n || held.r != b || z == b ||
        z == held.x || num_blocks <= 0 || !fast_layout ||
        !gkoc_x_cg_step_2_jacobi_apply_fits(num_blocks, n, scheme, vt == 0 ? 8 : 4)) {
        return false;
    }
    held.stage = 2;
    held.it = it;
    held.num_blocks = num_blocks;
    held.max_bs = max_bs;
    held.scheme = scheme;
    held.block_ptrs = block_ptrs;
    held.blocks = blocks;
    held.z = z;
    publish();
    return true;
}

bool fused_dot(int vt, gkoc_stream_t s, int64_t n, const void* x, const void* y, void* result,
               array<char>& tmp)
{
    if (held.stage == 3) {
        // w -= h_i v_i is held and this is <v_{i+1}, w> (either operand order): one pass over w
        const void* other = y == held.ss_y ? x : (x == held.ss_y ? y : nullptr);
        if (held.vt != vt || held.s != s || held.n != n || other == nullptr || other == held.ss_y ||
            result == held.ss_alpha || result == held.ss_y) {
            return false;   // the caller launches what is held, then its own kernel
        }
        const held_ops g = held;
        held.stage = 0;
        publish();
        const size_t work = gkoc_x_workspace_bytes(n, vt == 0 ? 8 : 4);
        bool fits = true;
        try {
            if (tmp.get_size() < work) tmp.resize_and_reset(work);
        } catch (...) {
            fits = false;
        }
        if (!fits) {
            launch_sub_scaled(g);
            return false;
        }
        if (vt == 0) {
            GKOC_CALL(gkoc_x_gmres_mgs_step_f64(g.s, g.n, static_cast<double*>(g.ss_y),
                                                static_cast<const double*>(g.ss_x),
                                                static_cast<const double*>(g.ss_alpha),
                                                static_cast<const double*>(other),
                                                static_cast<double*>(result), tmp.get_data(), work));
        } else {
            GKOC_CALL(gkoc_x_gmres_mgs_step_f32(g.s, g.n, static_cast<float*>(g.ss_y),
                                                static_cast<const float*>(g.ss_x),
                                                static_cast<const float*>(g.ss_alpha),
                                                static_cast<const float*>(other),
                                                static_cast<float*>(result), tmp.get_data(), work));
        }
        return true;
    }
    if (held.stage != 2 || held.vt != vt || held.s != s || held.n != n || held.r != x || held.z != y ||
        result == held.rho || result == held.beta) {
        return false;
    }
    const held_ops h = held;
    held.stage = 0;
    held.norm_of = nullptr;
    publish();
    // from here on nothing is held: resizing tmp may free memory, which comes back through flush_deferred()
    const size_t vsize = vt == 0 ? 8 : 4;
    const size_t work = (gkoc_x_workspace_bytes(n, vsize) + 15) / 16 * 16;
    bool ok = true;
    try {
        if (tmp.get_size() < work + 16) tmp.resize_and_reset(work + 16);
    } catch (...) {
        ok = false;
    }
    if (!ok) {
        launch_step_2(h);
        launch_apply(h);
        return false;
    }
    char* norm_at = tmp.get_data() + work;
    int rc = GKOC_E_NOT_SUPPORTED;
#define CASE(VT, IT, T, I, TN, IN)                                                                    \
    if (h.vt == VT && h.it == IT) {                                                                   \
        rc = gkoc_x_cg_step_2_jacobi_apply_##TN##_##IN(                                               \
            h.s, h.num_blocks, h.n, h.max_bs, h.scheme, static_cast<const I*>(h.block_ptrs),          \
            static_cast<const T*>(h.blocks), static_cast<T*>(h.x), static_cast<T*>(h.r),              \
            static_cast<const T*>(h.p), static_cast<const T*>(h.q), static_cast<const T*>(h.beta),    \
            static_cast<const T*>(h.rho), h.stop, static_cast<T*>(h.z), static_cast<T*>(result),      \
            reinterpret_cast<T*>(norm_at), 1, tmp.get_data(), work);                                  \
    }
    GKOC_FUSION_TYPES(CASE)
#undef CASE
    if (rc != GKOC_OK) {
        // the argument checks of the fused entry refuse BEFORE anything is launched: the held
        // kernels run one by one and the caller computes its dot product itself
        launch_step_2(h);
        launch_apply(h);
        return false;
    }
    held.norm_of = h.r;
    held.norm_at = norm_at;
    held.norm_vt = vt;
    held.norm_n = n;
    held.norm_s = s;
    publish();
    return true;
}

bool cached_norm2(int vt, gkoc_stream_t s, int64_t n, const void* x, void* result)
{
    const void* src = nullptr;
    {
            if (held.stage == 0 && held.norm_of && held.norm_of == x && held.norm_vt == vt && held.norm_n == n &&
            held.norm_s == s) {
            src = held.norm_at;
        }
    }
    if (!src) return false;
    GKOC_CALL(gkoc_memcpy_d2d(result, src, vt == 0 ? 8 : 4, s));
    // the value stays valid: nothing has touched r
    return true;
}

}  // namespace cdna4
}  // namespace gko

// for code outside the binding that launches on the executor's stream (rccl_communicator.hpp)
extern "C" void gko_cdna4_launch_deferred() { gko::cdna4::launch_deferred(); }
