// Ginkgo-side binding of the IDR(s) kernels (core/solver/idr_kernels.hpp:22-72) to the C ABI.
#include <complex>

#include <ginkgo/core/matrix/dense.hpp>

#include "core/solver/idr_kernels.hpp"
#include "shim_common.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace idr {

using namespace ::gko::cdna4;
using exec_t = std::shared_ptr<const HipExecutor>;

#define DEF(T, TN)                                                                                    \
    template <>                                                                                       \
    void initialize<T>(exec_t exec, const size_type nrhs, matrix::Dense<T>* m,                         \
                       matrix::Dense<T>* subspace_vectors, bool deterministic,                        \
                       array<stopping_status>* stop_status)                                           \
    {                                                                                                 \
        GKOC_CALL(gkoc_idr_initialize_##TN(stream_of(exec), nrhs, rows(m), m->get_values(), ld(m),    \
                                           cols(subspace_vectors), subspace_vectors->get_values(),    \
                                           ld(subspace_vectors), deterministic ? 1 : 0,               \
                                           raw(stop_status)));                                        \
    }                                                                                                 \
    template <>                                                                                       \
    void step_1<T>(exec_t exec, const size_type nrhs, const size_type k, const matrix::Dense<T>* m,    \
                   const matrix::Dense<T>* f, const matrix::Dense<T>* residual,                       \
                   const matrix::Dense<T>* g, matrix::Dense<T>* c, matrix::Dense<T>* v,               \
                   const array<stopping_status>* stop_status)                                         \
    {                                                                                                 \
        GKOC_CALL(gkoc_idr_step_1_##TN(stream_of(exec), rows(v), nrhs, rows(m), k,                    \
                                       m->get_const_values(), ld(m), f->get_const_values(), ld(f),    \
                                       residual->get_const_values(), ld(residual),                    \
                                       g->get_const_values(), ld(g), c->get_values(), ld(c),          \
                                       v->get_values(), ld(v), raw(stop_status)));                    \
    }                                                                                                 \
    template <>                                                                                       \
    void step_2<T>(exec_t exec, const size_type nrhs, const size_type k,                               \
                   const matrix::Dense<T>* omega, const matrix::Dense<T>* preconditioned_vector,      \
                   const matrix::Dense<T>* c, matrix::Dense<T>* u,                                    \
                   const array<stopping_status>* stop_status)                                         \
    {                                                                                                 \
        GKOC_CALL(gkoc_idr_step_2_##TN(stream_of(exec), rows(u), nrhs, rows(c), k,                    \
                                       omega->get_const_values(),                                     \
                                       preconditioned_vector->get_const_values(),                     \
                                       ld(preconditioned_vector), c->get_const_values(), ld(c),       \
                                       u->get_values(), ld(u), raw(stop_status)));                    \
    }                                                                                                 \
    template <>                                                                                       \
    void step_3<T>(exec_t exec, const size_type nrhs, const size_type k, const matrix::Dense<T>* p,    \
                   matrix::Dense<T>* g, matrix::Dense<T>* g_k, matrix::Dense<T>* u,                   \
                   matrix::Dense<T>* m, matrix::Dense<T>* f, matrix::Dense<T>*,                       \
                   matrix::Dense<T>* residual, matrix::Dense<T>* x,                                   \
                   const array<stopping_status>* stop_status)                                         \
    {                                                                                                 \
        GKOC_CALL(gkoc_idr_step_3_##TN(stream_of(exec), rows(g), nrhs, rows(m), k,                    \
                                       p->get_const_values(), ld(p), g->get_values(), ld(g),          \
                                       g_k->get_values(), ld(g_k), u->get_values(), ld(u),            \
                                       m->get_values(), ld(m), f->get_values(), ld(f),                \
                                       residual->get_values(), ld(residual), x->get_values(), ld(x),  \
                                       raw(stop_status)));                                            \
    }                                                                                                 \
    template <>                                                                                       \
    void compute_omega<T>(exec_t exec, const size_type nrhs, const remove_complex<T> kappa,           \
                          const matrix::Dense<T>* tht,                                                \
                          const matrix::Dense<remove_complex<T>>* residual_norm,                      \
                          matrix::Dense<T>* omega, const array<stopping_status>* stop_status)         \
    {                                                                                                 \
        GKOC_CALL(gkoc_idr_compute_omega_##TN(stream_of(exec), nrhs, kappa, tht->get_const_values(),  \
                                              residual_norm->get_const_values(),                      \
                                              omega->get_values(), raw(stop_status)));                \
    }
DEF(double, f64)
DEF(float, f32)
DEF(std::complex<double>, c128)
DEF(std::complex<float>, c64)
#undef DEF

}  // namespace idr
}  // namespace hip
}  // namespace kernels
}  // namespace gko
