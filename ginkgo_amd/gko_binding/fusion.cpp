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
//
// BY-PRODUCTS (the default, GKOC_TUNE_DEFERRED_FUSION = 0): nothing is held, every call launches at
// once - but two of the kernels leave something behind that a later call would otherwise compute
// with a pass of its own:
//     cg::step_2                   also leaves ||r|| (gkoc_x_cg_step_2_norm_*: the new r is in registers)
//     jacobi::simple_apply(b -> z) also leaves <b, z> (gkoc_x_jacobi_simple_apply_dot_*)
//     compute_conj_dot(r, z)       -> an 8-byte copy if (operands, length, stream) are those of the
//                                     apply and nothing has entered the backend since, else a reduction
//     compute_norm2(r)             -> the same for ||r||
// ANTICIPATED APPLICATION (by-product mode, round 5): Ginkgo's Cg has no exit between cg::step_2 and
// the next iteration's preconditioner application (core/solver/cg.cpp:131-176: step_2 ends the loop
// body, precond->apply(r, z) begins it), and z is dead in between (its last reader was step_1).  Once
// THIS solve has shown step_2(x, r) followed directly by jacobi::simple_apply(M, r -> z), the next
// step_2(x, r) runs the library's one-kernel form (gkoc_x_cg_step_2_jacobi_apply_*: x, r, z with the
// bits of the separate kernels, ||r|| and <r, z> left behind) - so x and r exist when step_2 returns
// exactly as before and z = M r is written EARLY; the simple_apply(M, r -> z) that follows finds its
// work done and launches nothing.  Nothing is held, nothing is launched late.  What was learned is
// forgotten at every cg::initialize, every jacobi::generate and when one of the arrays it names is
// freed, and is confirmed anew within each solve; an application that does not arrive as
// predicted finds z already holding M r - which it then overwrites.  GKOC_TUNE_ANTICIPATE = 0 switches
// it off.  Per iteration on the 27-pt 256^3 problem: 336 us instead of 139 + 273.
// x, r, z have the bits of the separate kernels and exist when the call returns, so code that reads
// a solver's vectors with its own launches between two calls (the case ADVICE r03 raised against
// holding calls back) sees what the reference would show it.  A by-product is forgotten by EVERY
// entry into the backend except the reading calls above (stream_of() -> flush_deferred()): a kernel
// that rewrites r or z is such an entry.  Per-iteration saving on the 27-pt 256^3 problem: the two
// reductions' passes over r and z, 70 of 1610 us.  GKOC_TUNE_DEFERRED_FUSION = 2: one kernel per call.
#include <cstdio>
#include <map>
#include <utility>

#include "shim_common.hpp"

namespace gko {
namespace cdna4 {

thread_local int deferred_state = 0;
thread_local int alloc_role_hint = 0;
std::atomic<uint64_t> backend_epoch{0};
thread_local uint64_t last_entry_epoch = 0;

namespace {
// the number of the last call of THIS thread that took part in a chain of by-products (see shim_common.hpp)
thread_local uint64_t chain_epoch = ~uint64_t(0);

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

// 0: by-products (default), 1: calls held back and fused (opt-in), 2: one kernel per call
int mode()
{
    int64_t v = 0;
    gkoc_tune_get(GKOC_TUNE_DEFERRED_FUSION, &v);
    return static_cast<int>(v);
}

int enabled() { return mode() == 1; }

// <b, z> left behind by the block-Jacobi application
struct dot_byproduct {
    const void* x = nullptr;
    const void* y = nullptr;
    const void* at = nullptr;
    int vt = 0;
    int64_t n = 0;
    gkoc_stream_t s = nullptr;
};
thread_local dot_byproduct bp_dot;
// how often a reduction was answered by a by-product (tests)
thread_local int64_t hits_norm = 0, hits_dot = 0;

// what this solve has shown so far: cg::step_2(x, r) followed DIRECTLY by jacobi::simple_apply(r -> z)
struct followup {
    bool valid = false;
    int vt = 0, it = 0;
    gkoc_stream_t s = nullptr;
    int64_t n = 0;
    const void *x = nullptr, *r = nullptr;
    int64_t num_blocks = 0;
    uint32_t max_bs = 0;
    gkoc_jacobi_scheme scheme{};
    const void *block_ptrs = nullptr, *blocks = nullptr;
    void* z = nullptr;
};
thread_local followup learned;
// the step_2 that was the LAST entry into the backend (a candidate for learning), and the application
// that has been done ahead of its call
struct just_stepped {
    const void *x = nullptr, *r = nullptr;
    int vt = 0;
    int64_t n = 0;
    gkoc_stream_t s = nullptr;
};
thread_local just_stepped last_step_2;
thread_local bool anticipated = false;      // z = M r of `learned` is in place, its call has not come yet
thread_local int64_t hits_apply = 0;

int anticipate_level()
{
    int64_t v = 1;
    gkoc_tune_get(GKOC_TUNE_ANTICIPATE, &v);
    return static_cast<int>(v);
}
int anticipate_enabled() { return anticipate_level() != 0; }

// ---- round 6: the NEXT step_1 behind the criterion, and results written where they are wanted ------------
// Ginkgo's Cg waits on the host for the criterion's answer and then calls cg::step_1 (core/solver/cg.cpp:148-
// 165): 29 - 37 us per iteration in which the device has nothing to do (profiles/r05_ginkgo_api_timeline.txt).
// Once THIS solve has shown the criterion followed directly by cg::step_1(p, z, rho, prev_rho), the criterion's
// entry enqueues the predicted step_1 behind its own kernel before it starts to wait (the two scalars have
// changed places, as `swap(prev_rho, rho)` at the end of the loop body says; the dot product that has just
// written rho is checked against that).  The step is masked by stop_status: if the criterion stops the column,
// p stays as it is.  The cg::step_1 that arrives finds its work done.  If the solve ends instead (iteration
// limit), p - dead from there on - has been advanced once more.
// In the same way the one-kernel step_2 + application writes <r, z> and ||r|| straight to where the dot
// product and the norm of the previous iteration were asked to put them (rho alternates between two scalars,
// the criterion's norm has one home): the two calls then launch nothing at all, not even an 8-byte copy.
// GKOC_TUNE_ANTICIPATE: 0 nothing, 1 (default) all of it, 2 only the application (round 5's behaviour).
struct step1_call {
    bool valid = false;
    int vt = 0;
    gkoc_stream_t s = nullptr;
    int64_t n = 0;
    void* p = nullptr;
    const void *z = nullptr, *rho = nullptr, *prev_rho = nullptr;
    const uint8_t* stop = nullptr;
};
thread_local step1_call learned_step1;      // the last cg::step_1 of this solve, seen directly behind the criterion
thread_local step1_call ahead;              // the step_1 the criterion's entry has run ahead of its call
thread_local bool step1_is_ahead = false;
thread_local uint64_t criterion_epoch = ~uint64_t(0) - 1;
thread_local const void* last_dot_dest = nullptr;     // where the last single-column dot product was written
thread_local const void* last_norm_of = nullptr;      // the last single-column norm: of what, where to
thread_local void* last_norm_dest = nullptr;
thread_local int64_t hits_step1 = 0;

// csr::spmv(A, p -> q) that leaves <p, q> where the dot product behind it will be asked to put it (by-product
// mode, round 6).  Ginkgo's Cg calls A->apply(p, q) and then p->compute_conj_dot(q, beta) with nothing in between
// (core/solver/cg.cpp:164-167).  Once THIS solve has shown exactly that - csr::spmv(A, b -> c) as one entry into
// the backend and compute_dot(b, c -> dest) as the next -, the next csr::spmv with the same operands runs the
// library's one-pass form (gkoc_x_csr_spmv_dot_*: c has the bits of the plain product, the sum takes the fused
// kernel's fixed tree) and writes the sum straight to `dest`; the dot product that follows finds its work done
// and launches nothing.  dest's last reader was the previous iteration's step_2, so writing it one call early
// is not observable through the solver.  Forgotten with everything else that is learned (forget_learned).
struct spmv_dot_shape {
    bool valid = false;
    int vt = 0, it = 0;
    gkoc_stream_t s = nullptr;
    int64_t n = 0;
    const void *row_ptrs = nullptr, *cols = nullptr, *vals = nullptr, *b = nullptr;
    void* c = nullptr;
    void* dest = nullptr;
};
thread_local spmv_dot_shape learned_spmv;     // confirmed in this solve
thread_local spmv_dot_shape last_spmv;        // the product that was this thread's last writing entry (a candidate)
thread_local uint64_t last_spmv_epoch = 0;
thread_local int spmv_dot_seen = 0;           // consecutive times the same (operands, dest) have been seen
thread_local int64_t hits_spmv_dot = 0;

void publish()
{
    deferred_state = held.stage | (held.norm_of ? 4 : 0) | (bp_dot.x ? 8 : 0) | (last_step_2.r ? 16 : 0) |
                     (anticipated ? 32 : 0);
}

// what this thread's earlier calls left behind is void (another entry into the backend - of any thread - has
// come between): the values are simply not used; z = M r written ahead of its call stays what it is
void drop_byproducts()
{
    if (held.norm_of == nullptr && bp_dot.x == nullptr && last_step_2.r == nullptr && !anticipated) return;
    held.norm_of = nullptr;
    bp_dot.x = nullptr;
    last_step_2.r = nullptr;
    anticipated = false;
    publish();
}

// a READING call (dot, norm2): may it use what is left behind?  It takes no number itself.
bool chain_intact_for_read()
{
    if (backend_epoch.load(std::memory_order_acquire) == chain_epoch) return true;
    drop_byproducts();
    return false;
}

// a WRITING call that takes part (the block-Jacobi application): it takes the next number; the chain
// goes on iff that is the number right behind the chain's last one
bool chain_step()
{
    const uint64_t e = backend_epoch.fetch_add(1, std::memory_order_acq_rel) + 1;
    const bool intact = e == chain_epoch + 1;
    chain_epoch = e;
    last_entry_epoch = e;
    if (!intact) drop_byproducts();
    return intact;
}

// Device memory of the by-products, one block per (device, stream) this thread has used:
// [||r|| : 64 B | <b,z> : 64 B | workspace of the step_2 pass | workspace of the apply pass].
// Kept for the life of the thread (a few MB: 8 bytes per 64 rows, twice).
struct side_block {
    char* p = nullptr;
    size_t work = 0;
};
thread_local std::map<std::pair<int, gkoc_stream_t>, side_block> side_blocks;

side_block* side_for(int dev, gkoc_stream_t s, size_t work)
{
    work = (work + 255) / 256 * 256;
    auto& b = side_blocks[{dev, s}];
    if (b.p == nullptr || b.work < work) {
        if (b.p) {
            gkoc_free(b.p);      // (synchronises: nothing in flight uses it)
            b.p = nullptr;
        }
        void* mem = nullptr;
        if (gkoc_malloc(&mem, 128 + 2 * work) != GKOC_OK) {
            b.work = 0;
            return nullptr;
        }
        b.p = static_cast<char*>(mem);
        b.work = work;
    }
    return &b;
}

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
    // z = M r was written ahead of its call and something ELSE has entered the backend: this solve does not have
    // the shape that was learned (a logger that launches, a user LinOp between the steps) - it has to show it
    // again before the next cg::step_2 writes z early
    if (anticipated) learned.valid = false;
    held.stage = 0;
    held.norm_of = nullptr;
    bp_dot.x = nullptr;
    last_step_2.r = nullptr;
    anticipated = false;
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
    // (the first thing a single-column jacobi::simple_apply does: its number in the order of entries)
    chain_step();
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

// a free: what the solve has shown is void if it points at what goes away (the arrays of the
// preconditioner and the solver's vectors are whole allocations: their data pointers are the bases);
// the temporaries a criterion or a logger frees in every iteration are none of them
void forget_learned_if(const void* freed)
{
    const followup& f = learned;
    const step1_call& c = learned_step1;
    if ((f.valid && (freed == f.blocks || freed == f.block_ptrs || freed == f.z || freed == f.r || freed == f.x)) ||
        (c.valid && (freed == c.p || freed == c.z || freed == c.rho || freed == c.prev_rho || freed == c.stop)) ||
        (freed != nullptr && (freed == last_dot_dest || freed == last_norm_dest || freed == last_norm_of))) {
        forget_learned();
    }
    // the product + dot shape names the matrix arrays, both vectors and the scalar it writes EARLY: a freed
    // scalar (the temporary result of a distributed dot product, say) must never be written again
    for (const spmv_dot_shape* t : {&learned_spmv, &last_spmv}) {
        if (t->valid && freed != nullptr &&
            (freed == t->dest || freed == t->b || freed == t->c || freed == t->vals || freed == t->cols ||
             freed == t->row_ptrs)) {
            learned_spmv.valid = false;
            last_spmv.valid = false;
            spmv_dot_seen = 0;
        }
    }
}

// ---- the criterion (one column, synchronous form) with the predicted cg::step_1 behind its kernel ------------
bool criterion_then_step_1(int vt, gkoc_stream_t s, const void* tau, const void* orig_tau, double goal,
                           uint8_t stopping_id, bool set_finalized, bool implicit, uint8_t* stop, uint8_t* flags,
                           int* all_converged, int* one_changed)
{
    // (the caller has been through stream_of(): this entry's number)
    criterion_epoch = last_entry_epoch;
    // a Combined criterion may hold TWO residual-norm criteria: the step runs behind the first one only
    if (step1_is_ahead) return false;
    const step1_call& c = learned_step1;
    // (a rank without rows - an empty part of a distributed vector - has null vectors: nothing to run ahead, and
    //  an entry that refuses them must not throw on that rank alone)
    if (!c.valid || c.vt != vt || c.s != s || c.stop != stop || c.n <= 0 || c.p == nullptr || c.z == nullptr ||
        c.rho == nullptr || c.prev_rho == nullptr || tau == nullptr || orig_tau == nullptr ||
        (anticipate_level() != 1 && anticipate_level() != 3) || mode() != 0 || last_dot_dest == nullptr ||
        last_dot_dest != c.prev_rho) {
        return false;
    }
    // this iteration's scalars: rho is where the dot product has just been written = last iteration's prev_rho
    int rc;
    if (vt == 0) {
        rc = gkoc_x_residual_norm_then_cg_step_1_f64(
            s, static_cast<const double*>(tau), static_cast<const double*>(orig_tau), goal, stopping_id,
            set_finalized ? 1 : 0, implicit ? 1 : 0, stop, flags, all_converged, one_changed, c.n,
            static_cast<double*>(c.p), static_cast<const double*>(c.z), static_cast<const double*>(c.prev_rho),
            static_cast<const double*>(c.rho));
    } else {
        rc = gkoc_x_residual_norm_then_cg_step_1_f32(
            s, static_cast<const float*>(tau), static_cast<const float*>(orig_tau), float(goal), stopping_id,
            set_finalized ? 1 : 0, implicit ? 1 : 0, stop, flags, all_converged, one_changed, c.n,
            static_cast<float*>(c.p), static_cast<const float*>(c.z), static_cast<const float*>(c.prev_rho),
            static_cast<const float*>(c.rho));
    }
    if (rc == GKOC_E_INVALID) {
        // refused before anything was launched (argument checks): the caller runs the plain criterion
        return false;
    }
    GKOC_CALL(rc);
    ahead = c;
    ahead.rho = c.prev_rho;
    ahead.prev_rho = c.rho;
    step1_is_ahead = true;
    return true;
}

// cg::step_1 (one column) arrives: has the criterion's entry done it already?  Either way this solve's
// pattern is noted (only a step_1 DIRECTLY behind the criterion counts).
bool step_1_done_ahead(int vt, gkoc_stream_t s, int64_t n, void* p, const void* z, const void* rho,
                       const void* prev_rho, const uint8_t* stop)
{
    const bool was = step1_is_ahead;
    step1_is_ahead = false;
    // this call's number in the order of entries (it writes p, whoever launches it)
    const uint64_t e = backend_epoch.fetch_add(1, std::memory_order_acq_rel) + 1;
    last_entry_epoch = e;
    step1_call now;
    now.valid = e == criterion_epoch + 1;
    now.vt = vt;
    now.s = s;
    now.n = n;
    now.p = p;
    now.z = z;
    now.rho = rho;
    now.prev_rho = prev_rho;
    now.stop = stop;
    learned_step1 = now;
    if (!was) return false;
    if (ahead.vt == vt && ahead.s == s && ahead.n == n && ahead.p == p && ahead.z == z && ahead.rho == rho &&
        ahead.prev_rho == prev_rho && ahead.stop == stop) {
        ++hits_step1;
        if (deferred_state != 0) flush_deferred();      // (what any writing entry does to this thread's by-products)
        return true;
    }
    // a step_1 other than the predicted one: p has been advanced by a step the solver did not ask for.
    // Ginkgo's Cg cannot get here (its loop has no other exit between the criterion and step_1); say so loudly.
    std::fprintf(stderr, "[gko-cdna4] cg::step_1 arrived with other operands than the step run ahead of it; "
                         "GKOC_TUNE_ANTICIPATE=2 switches the prediction off\n");
    learned_step1.valid = false;
    return false;
}

void forget_learned()
{
    learned.valid = false;
    learned_step1.valid = false;
    learned_spmv.valid = false;
    last_spmv.valid = false;
    spmv_dot_seen = 0;
    step1_is_ahead = false;
    last_dot_dest = nullptr;
    last_norm_of = nullptr;
    last_norm_dest = nullptr;
    if (last_step_2.r || anticipated) {
        last_step_2.r = nullptr;
        anticipated = false;
        publish();
    }
}

namespace {
// the one-kernel form of step_2 + the application this solve has shown to follow it
bool step_2_anticipating(int vt, int dev, gkoc_stream_t s, int64_t n, void* x, void* r, const void* p,
                         const void* q, const void* beta, const void* rho, const uint8_t* stop)
{
    const followup& f = learned;
    if (!f.valid || f.vt != vt || f.s != s || f.n != n || f.x != x || f.r != r || f.z == p || f.z == q ||
        f.z == x || f.z == r || f.z == beta || f.z == rho || !anticipate_enabled()) {
        return false;
    }
    const size_t work = (gkoc_x_workspace_bytes(n, vt == 0 ? 8 : 4) + 15) / 16 * 16;
    side_block* b = side_for(dev, s, work);
    if (!b || b->work < work) return false;
    char* norm_at = b->p;
    char* dot_at = b->p + 64;
    if (anticipate_level() == 1 || anticipate_level() == 3) {
        // <r, z> goes where the NEXT dot product will be asked to put it: the scalar that was prev_rho in this
        // iteration's step_1 (cg.cpp:176 swaps the two); ||r|| where the criterion's norm of r went last time.
        // Nothing reads either before those calls come (prev_rho's last reader was step_1).
        const step1_call& c = learned_step1;
        if (c.valid && c.vt == vt && c.s == s && c.rho == rho && c.prev_rho != nullptr && c.prev_rho != rho &&
            c.prev_rho != beta && c.prev_rho != last_norm_dest) {
            dot_at = static_cast<char*>(const_cast<void*>(c.prev_rho));
        }
        if (last_norm_of == r && last_norm_dest != nullptr && last_norm_dest != rho && last_norm_dest != beta) {
            norm_at = static_cast<char*>(last_norm_dest);
        }
    }
    int rc = GKOC_E_NOT_SUPPORTED;
#define CASE(VT, IT, T, I, TN, IN)                                                                    \
    if (vt == VT && f.it == IT) {                                                                     \
        rc = gkoc_x_cg_step_2_jacobi_apply_##TN##_##IN(                                               \
            s, f.num_blocks, n, f.max_bs, f.scheme, static_cast<const I*>(f.block_ptrs),              \
            static_cast<const T*>(f.blocks), static_cast<T*>(x), static_cast<T*>(r),                  \
            static_cast<const T*>(p), static_cast<const T*>(q), static_cast<const T*>(beta),          \
            static_cast<const T*>(rho), stop, static_cast<T*>(f.z), reinterpret_cast<T*>(dot_at),     \
            reinterpret_cast<T*>(norm_at), 1, b->p + 128, work);                                      \
    }
    GKOC_FUSION_TYPES(CASE)
#undef CASE
    if (rc != GKOC_OK) return false;     // refused before anything was launched
    held.norm_of = r;
    held.norm_at = norm_at;
    held.norm_vt = vt;
    held.norm_n = n;
    held.norm_s = s;
    bp_dot.x = r;
    bp_dot.y = f.z;
    bp_dot.at = dot_at;
    bp_dot.vt = vt;
    bp_dot.n = n;
    bp_dot.s = s;
    anticipated = true;
    last_step_2.r = nullptr;
    chain_epoch = last_entry_epoch;      // (step_2 came in through stream_of(): that entry's number)
    publish();
    return true;
}
}  // namespace

// cg::step_2 that leaves ||r_new|| behind (by-product mode; the caller has been through stream_of())
bool step_2_with_norm(int vt, int dev, gkoc_stream_t s, int64_t n, void* x, void* r, const void* p,
                      const void* q, const void* beta, const void* rho, const uint8_t* stop)
{
    if (n <= 0 || mode() != 0) return false;
    if (learned.valid && (learned.x != x || learned.r != r)) learned.valid = false;    // another solve
    if (step_2_anticipating(vt, dev, s, n, x, r, p, q, beta, rho, stop)) return true;
    const size_t work = gkoc_x_workspace_bytes(n, vt == 0 ? 8 : 4);
    side_block* b = side_for(dev, s, work);
    if (!b) return false;
    int rc;
    if (vt == 0) {
        rc = gkoc_x_cg_step_2_norm_f64(s, n, static_cast<double*>(x), static_cast<double*>(r),
                                       static_cast<const double*>(p), static_cast<const double*>(q),
                                       static_cast<const double*>(beta), static_cast<const double*>(rho), stop,
                                       reinterpret_cast<double*>(b->p), 1, b->p + 128, b->work);
    } else {
        rc = gkoc_x_cg_step_2_norm_f32(s, n, static_cast<float*>(x), static_cast<float*>(r),
                                       static_cast<const float*>(p), static_cast<const float*>(q),
                                       static_cast<const float*>(beta), static_cast<const float*>(rho), stop,
                                       reinterpret_cast<float*>(b->p), 1, b->p + 128, b->work);
    }
    if (rc != GKOC_OK) return false;    // refused before anything was launched: the plain kernel runs
    held.norm_of = r;
    held.norm_at = b->p;
    held.norm_vt = vt;
    held.norm_n = n;
    held.norm_s = s;
    // a candidate: if the NEXT entry into the backend is a block-Jacobi application of r, the solve has
    // shown its shape (jacobi_apply_with_dot learns it)
    last_step_2.x = x;
    last_step_2.r = r;
    last_step_2.vt = vt;
    last_step_2.n = n;
    last_step_2.s = s;
    chain_epoch = last_entry_epoch;      // (step_2 came in through stream_of(): that entry's number)
    publish();
    return true;
}

// jacobi::simple_apply(b -> z) that leaves <b, z> behind (by-product mode).  A reading call: what
// step_2 left behind stays valid unless z is the vector it belongs to.
bool jacobi_apply_with_dot(int vt, int it, int dev, gkoc_stream_t s, int64_t num_blocks, uint32_t max_bs,
                           gkoc_jacobi_scheme scheme, const void* block_ptrs, const void* blocks,
                           const void* b, int64_t n, void* z)
{
    if (mode() != 0 || held.stage != 0 || n <= 0 || num_blocks <= 0 || z == b) return false;
    const int64_t bo = scheme.block_offset;
    const bool fast_layout = bo >= 1 && bo <= 16 && (bo & (bo - 1)) == 0 &&
                             (bo << scheme.group_power) == 64 && int64_t(max_bs) <= bo;
    if (!fast_layout) return false;
    if (anticipated) {
        // the step_2 in front of this call has done it already (nothing has entered the backend since)
        const followup& f = learned;
        anticipated = false;
        if (f.valid && f.vt == vt && f.it == it && f.s == s && f.n == n && f.r == b && f.z == z &&
            f.num_blocks == num_blocks && f.max_bs == max_bs && f.block_ptrs == block_ptrs && f.blocks == blocks &&
            f.scheme.block_offset == scheme.block_offset && f.scheme.group_offset == scheme.group_offset &&
            f.scheme.group_power == scheme.group_power) {
            ++hits_apply;
            publish();
            return true;          // z = M r and <r, z> are in place
        }
        // not the application that was predicted: what step_2 left behind for it is void
        learned.valid = false;
        bp_dot.x = nullptr;
        publish();
    }
    if (last_step_2.r != nullptr) {
        // cg::step_2(x, r) was the last entry into the backend and this applies M to its r: the shape of
        // this solve's iteration (confirmed here, used by the next step_2)
        const just_stepped t = last_step_2;
        last_step_2.r = nullptr;
        if (t.r == b && t.vt == vt && t.n == n && t.s == s && z != t.x && anticipate_enabled() &&
            gkoc_x_cg_step_2_jacobi_apply_fits(num_blocks, n, scheme, vt == 0 ? 8 : 4)) {
            followup& f = learned;
            f.valid = true;
            f.vt = vt;
            f.it = it;
            f.s = s;
            f.n = n;
            f.x = t.x;
            f.r = t.r;
            f.num_blocks = num_blocks;
            f.max_bs = max_bs;
            f.scheme = scheme;
            f.block_ptrs = block_ptrs;
            f.blocks = blocks;
            f.z = z;
        } else {
            learned.valid = false;
        }
    }
    if (z == held.norm_of) held.norm_of = nullptr;
    bp_dot.x = nullptr;
    publish();
    const size_t work = gkoc_x_workspace_bytes(n, vt == 0 ? 8 : 4);
    side_block* sb = side_for(dev, s, work);
    if (!sb) return false;
    if (held.norm_of && held.norm_at != sb->p) {
        // (the block was re-allocated under a by-product of this stream: gone)
        held.norm_of = nullptr;
        publish();
    }
    char* dot_at = sb->p + 64;
    char* wk = sb->p + 128 + sb->work;
    int rc = GKOC_E_NOT_SUPPORTED;
#define CASE(VT, IT, T, I, TN, IN)                                                                     \
    if (vt == VT && it == IT) {                                                                        \
        rc = gkoc_x_jacobi_simple_apply_dot_##TN##_##IN(                                               \
            s, num_blocks, n, max_bs, scheme, static_cast<const I*>(block_ptrs),                       \
            static_cast<const T*>(blocks), static_cast<const T*>(b), static_cast<T*>(z),               \
            reinterpret_cast<T*>(dot_at), wk, sb->work);                                               \
    }
    GKOC_FUSION_TYPES(CASE)
#undef CASE
    if (rc != GKOC_OK) return false;    // refused before anything was launched
    bp_dot.x = b;
    bp_dot.y = z;
    bp_dot.at = dot_at;
    bp_dot.vt = vt;
    bp_dot.n = n;
    bp_dot.s = s;
    publish();
    return true;
}

// csr::spmv(A, b -> c), one column, square A (by-product mode; the caller has been through stream_of())
bool spmv_with_dot(int vt, int it, int dev, gkoc_stream_t s, int64_t n, const void* row_ptrs, const void* cols,
                   const void* vals, const void* b, void* c)
{
    last_spmv.valid = false;
    if (mode() != 0 || anticipate_level() != 1 || n <= 0 || b == c) return false;
    spmv_dot_shape now;
    now.valid = true;
    now.vt = vt;
    now.it = it;
    now.s = s;
    now.n = n;
    now.row_ptrs = row_ptrs;
    now.cols = cols;
    now.vals = vals;
    now.b = b;
    now.c = c;
    const spmv_dot_shape f = learned_spmv;
    if (f.valid && spmv_dot_seen >= 2 && f.vt == vt && f.it == it && f.s == s && f.n == n &&
        f.row_ptrs == row_ptrs && f.cols == cols && f.vals == vals && f.b == b && f.c == c && f.dest != nullptr &&
        f.dest != b && f.dest != c) {
        const size_t work = gkoc_x_workspace_bytes(n, vt == 0 ? 8 : 4);
        side_block* sb = side_for(dev, s, work);
        if (sb && sb->work >= work) {
            char* wk = sb->p + 128 + sb->work;
            int rc = GKOC_E_NOT_SUPPORTED;
#define CASE(VT, IT, T, I, TN, IN)                                                                          \
    if (vt == VT && it == IT) {                                                                             \
        rc = gkoc_x_csr_spmv_dot_##TN##_##IN(s, n, static_cast<const I*>(row_ptrs),                         \
                                             static_cast<const I*>(cols), static_cast<const T*>(vals),      \
                                             static_cast<const T*>(b), static_cast<T*>(c),                  \
                                             static_cast<T*>(f.dest), wk, sb->work);                        \
    }
            GKOC_FUSION_TYPES(CASE)
#undef CASE
            if (rc == GKOC_OK) {
                bp_dot.x = b;
                bp_dot.y = c;
                bp_dot.at = f.dest;
                bp_dot.vt = vt;
                bp_dot.n = n;
                bp_dot.s = s;
                chain_epoch = last_entry_epoch;
                publish();
                ++hits_spmv_dot;
                now.dest = f.dest;
                last_spmv = now;                 // (the dot product behind it confirms the shape again)
                last_spmv_epoch = last_entry_epoch;
                return true;
            }
        }
    }
    last_spmv = now;
    last_spmv_epoch = last_entry_epoch;
    return false;
}

// A call that only READS vectors (a reduction into `result`) is about to launch: what is held is
// launched, by-products stay unless the result overwrites something they belong to.
void launch_deferred_for_read(const void* result)
{
    if (deferred_state == 0) return;
    if (held.stage != 0 || result == held.norm_of || result == bp_dot.x || result == bp_dot.y) {
        flush_deferred();
    }
}

bool fused_dot(int vt, gkoc_stream_t s, int64_t n, const void* x, const void* y, void* result,
               array<char>& tmp)
{
    if (deferred_state != 0 && held.stage == 0) chain_intact_for_read();
    last_dot_dest = result;
    if (last_spmv.valid) {
        // is this the dot product of the operands of the product that was the last entry into the backend?
        const spmv_dot_shape t = last_spmv;
        last_spmv.valid = false;
        const bool next = backend_epoch.load(std::memory_order_acquire) == last_spmv_epoch;
        if (next && t.vt == vt && t.s == s && t.n == n && ((x == t.b && y == t.c) || (x == t.c && y == t.b)) &&
            result != x && result != y) {
            // the shape counts once it has been seen TWICE in a row with the same operands and the same place
            // for the result (a result that moves - a temporary - is never written early)
            const spmv_dot_shape& o = learned_spmv;
            const bool same = o.dest == result && o.b == t.b && o.c == t.c && o.vals == t.vals && o.cols == t.cols &&
                              o.row_ptrs == t.row_ptrs && o.n == t.n && o.vt == t.vt && o.it == t.it && o.s == t.s;
            spmv_dot_seen = same ? spmv_dot_seen + 1 : 1;
            learned_spmv = t;
            learned_spmv.dest = result;
            learned_spmv.valid = true;
        } else {
            if (learned_spmv.valid && (t.b == learned_spmv.b || t.c == learned_spmv.c)) {
                learned_spmv.valid = false;      // the product was not followed by its dot product this time
            }
            spmv_dot_seen = 0;
        }
    }
    if (held.stage == 0 && bp_dot.x != nullptr && bp_dot.vt == vt && bp_dot.s == s && bp_dot.n == n &&
        ((x == bp_dot.x && y == bp_dot.y) || (x == bp_dot.y && y == bp_dot.x)) && result != x &&
        result != y) {
        // <b, z> came with the block-Jacobi application and nothing has entered the backend since
        // (if it was written straight to `result`, there is nothing left to do)
        if (result != bp_dot.at) GKOC_CALL(gkoc_memcpy_d2d(result, bp_dot.at, vt == 0 ? 8 : 4, s));
        ++hits_dot;
        return true;
    }
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
    if (deferred_state != 0 && held.stage == 0) chain_intact_for_read();
    last_norm_of = x;
    last_norm_dest = result;
    {
            if (held.stage == 0 && held.norm_of && held.norm_of == x && held.norm_vt == vt && held.norm_n == n &&
            held.norm_s == s) {
            src = held.norm_at;
        }
    }
    if (!src) return false;
    if (src != result) GKOC_CALL(gkoc_memcpy_d2d(result, src, vt == 0 ? 8 : 4, s));
    ++hits_norm;
    // the value stays valid: nothing has touched r
    return true;
}

}  // namespace cdna4
}  // namespace gko

// for code outside the binding that launches on the executor's stream (rccl_communicator.hpp)
extern "C" void gko_cdna4_launch_deferred() { gko::cdna4::launch_deferred(); }
// reductions of the calling thread that were answered by a value another kernel had left behind
extern "C" void gko_cdna4_byproduct_hits(int64_t* norms, int64_t* dots)
{
    if (norms) *norms = gko::cdna4::hits_norm;
    if (dots) *dots = gko::cdna4::hits_dot;
}
// block-Jacobi applications of the calling thread that a preceding cg::step_2 had already done
extern "C" void gko_cdna4_anticipated_steps(int64_t* steps)
{
    if (steps) *steps = gko::cdna4::hits_step1;
}
// csr::spmv calls of the calling thread that also left <b, c> for the dot product behind them
extern "C" void gko_cdna4_spmv_dot_hits(int64_t* products)
{
    if (products) *products = gko::cdna4::hits_spmv_dot;
}
extern "C" void gko_cdna4_anticipated_applies(int64_t* applies)
{
    if (applies) *applies = gko::cdna4::hits_apply;
}
