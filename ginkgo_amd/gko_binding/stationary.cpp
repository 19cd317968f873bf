// gko::kernels::hip::ir::initialize and chebyshev::{init_update, update} forwarded to
// the C ABI (csrc/krylov_steps.hip): Ginkgo's own Ir and Chebyshev drivers
// (core/solver/ir.cpp, chebyshev.cpp) run on this backend.
#include "core/solver/chebyshev_kernels.hpp"
#include "core/solver/ir_kernels.hpp"
#include "shim_common.hpp"

namespace gko {
namespace kernels {
namespace hip {

using cdna4::cols;
using cdna4::ld;
using cdna4::raw;
using cdna4::rows;
using cdna4::stream_of;
using exec_t = std::shared_ptr<const HipExecutor>;


namespace ir {

void initialize(exec_t exec, array<stopping_status>* stop_status)
{
    GKOC_CALL(gkoc_ir_initialize(stream_of(exec), static_cast<int64_t>(stop_status->get_size()),
                                 raw(stop_status)));
}

}  // namespace ir


namespace chebyshev {

#define DEF(T, TN)                                                                          \
    template <>                                                                             \
    void init_update<T>(exec_t exec, const solver::detail::coeff_type<T> alpha,             \
                        const matrix::Dense<T>* inner_sol, matrix::Dense<T>* update_sol,    \
                        matrix::Dense<T>* output)                                           \
    {                                                                                       \
        GKOC_CALL(gkoc_chebyshev_init_update_##TN(                                          \
            stream_of(exec), rows(output), cols(output), alpha, inner_sol->get_const_values(), \
            ld(inner_sol), update_sol->get_values(), ld(update_sol), output->get_values(),  \
            ld(output)));                                                                   \
    }                                                                                       \
    template <>                                                                             \
    void update<T>(exec_t exec, const solver::detail::coeff_type<T> alpha,                  \
                   const solver::detail::coeff_type<T> beta, matrix::Dense<T>* inner_sol,   \
                   matrix::Dense<T>* update_sol, matrix::Dense<T>* output)                  \
    {                                                                                       \
        GKOC_CALL(gkoc_chebyshev_update_##TN(                                               \
            stream_of(exec), rows(output), cols(output), alpha, beta, inner_sol->get_values(), \
            ld(inner_sol), update_sol->get_values(), ld(update_sol), output->get_values(),  \
            ld(output)));                                                                   \
    }
DEF(double, f64)
DEF(float, f32)
#undef DEF

// complex values: coeff_type is complex<double>, handed over by address
#define DEF(T, TN)                                                                          \
    template <>                                                                             \
    void init_update<T>(exec_t exec, const solver::detail::coeff_type<T> alpha,             \
                        const matrix::Dense<T>* inner_sol, matrix::Dense<T>* update_sol,    \
                        matrix::Dense<T>* output)                                           \
    {                                                                                       \
        const std::complex<double> a = alpha;                                               \
        GKOC_CALL(gkoc_chebyshev_init_update_##TN(                                          \
            stream_of(exec), rows(output), cols(output), &a, inner_sol->get_const_values(), \
            ld(inner_sol), update_sol->get_values(), ld(update_sol), output->get_values(),  \
            ld(output)));                                                                   \
    }                                                                                       \
    template <>                                                                             \
    void update<T>(exec_t exec, const solver::detail::coeff_type<T> alpha,                  \
                   const solver::detail::coeff_type<T> beta, matrix::Dense<T>* inner_sol,   \
                   matrix::Dense<T>* update_sol, matrix::Dense<T>* output)                  \
    {                                                                                       \
        const std::complex<double> a = alpha, b = beta;                                     \
        GKOC_CALL(gkoc_chebyshev_update_##TN(                                               \
            stream_of(exec), rows(output), cols(output), &a, &b, inner_sol->get_values(),   \
            ld(inner_sol), update_sol->get_values(), ld(update_sol), output->get_values(),  \
            ld(output)));                                                                   \
    }
DEF(std::complex<double>, c128)
DEF(std::complex<float>, c64)
#undef DEF

}  // namespace chebyshev

}  // namespace hip
}  // namespace kernels
}  // namespace gko
