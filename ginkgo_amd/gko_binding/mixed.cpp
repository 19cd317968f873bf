// gko::kernels::hip for a core built with GINKGO_MIXED_PRECISION (CMake -DGINKGO_MIXED_PRECISION=ON,
// core/base/mixed_precision_types.hpp:15-120): the (matrix, input, output) value-type triples of
// csr::spmv / advanced_spmv and ell::spmv / advanced_spmv whose three types are NOT all the same, and
// dense::row_gather / advanced_row_gather between two precisions.  (The uniform instantiations are in
// kernels.cpp / complex.cpp / complex_solvers.cpp / conversions.cpp.)  Defined unconditionally: a core
// built without the switch never references them, a core built with it finds them - one
// libginkgo_hip.so for both (the test infrastructure builds the second flavor of the core).
// Kernels: csrc/mixed_precision.hip (gkoc_{csr,ell}_spmv_mixed_*, gkoc_dense_row_gather_mixed_*).
#include <complex>

#include <ginkgo/core/matrix/coo.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/matrix/ell.hpp>

#include "core/matrix/coo_kernels.hpp"
#include "core/matrix/csr_kernels.hpp"
#include "core/matrix/dense_kernels.hpp"
#include "core/matrix/ell_kernels.hpp"
#include "shim_common.hpp"

namespace gko {
namespace kernels {
namespace hip {

using cdna4::cols;
using cdna4::ld;
using cdna4::rows;
using cdna4::stream_of;
using exec_t = std::shared_ptr<const HipExecutor>;

namespace {
template <typename T>
struct code;
template <>
struct code<double> {
    static constexpr int value = GKOC_VT_F64;
};
template <>
struct code<float> {
    static constexpr int value = GKOC_VT_F32;
};
template <>
struct code<std::complex<double>> {
    static constexpr int value = GKOC_VT_C128;
};
template <>
struct code<std::complex<float>> {
    static constexpr int value = GKOC_VT_C64;
};
}  // namespace

// the six non-uniform triples of a pair (wide W, narrow N), as (matrix, input, output)
#define FOR_TRIPLES_OF(M, W, N, I, IN)                                                     \
    M(W, W, N, I, IN) M(W, N, W, I, IN) M(W, N, N, I, IN) M(N, W, W, I, IN) M(N, W, N, I, IN) \
        M(N, N, W, I, IN)
#define FOR_TRIPLES(M)                                                    \
    FOR_TRIPLES_OF(M, double, float, int32, i32)                          \
    FOR_TRIPLES_OF(M, double, float, int64, i64)                          \
    FOR_TRIPLES_OF(M, std::complex<double>, std::complex<float>, int32, i32) \
    FOR_TRIPLES_OF(M, std::complex<double>, std::complex<float>, int64, i64)

namespace csr {

#define DEF(MT, IT, OT, I, IN)                                                                         \
    template <>                                                                                        \
    void spmv<MT, IT, OT, I>(exec_t exec, const matrix::Csr<MT, I>* a, const matrix::Dense<IT>* b,     \
                             matrix::Dense<OT>* c)                                                     \
    {                                                                                                  \
        GKOC_CALL(gkoc_csr_spmv_mixed_##IN(stream_of(exec), code<MT>::value, code<IT>::value,          \
                                           code<OT>::value, a->get_size()[0], a->get_size()[1],        \
                                           nullptr, a->get_const_row_ptrs(), a->get_const_col_idxs(),  \
                                           a->get_const_values(), b->get_const_values(), ld(b),        \
                                           nullptr, c->get_values(), ld(c), cols(c)));                 \
    }                                                                                                  \
    template <>                                                                                        \
    void advanced_spmv<MT, IT, OT, I>(exec_t exec, const matrix::Dense<MT>* alpha,                     \
                                      const matrix::Csr<MT, I>* a, const matrix::Dense<IT>* b,         \
                                      const matrix::Dense<OT>* beta, matrix::Dense<OT>* c)             \
    {                                                                                                  \
        GKOC_CALL(gkoc_csr_spmv_mixed_##IN(                                                            \
            stream_of(exec), code<MT>::value, code<IT>::value, code<OT>::value, a->get_size()[0],      \
            a->get_size()[1], alpha->get_const_values(), a->get_const_row_ptrs(),                      \
            a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(), ld(b),              \
            beta->get_const_values(), c->get_values(), ld(c), cols(c)));                               \
    }
FOR_TRIPLES(DEF)
#undef DEF

}  // namespace csr


namespace ell {

// ell's template parameters are (input, matrix, output, index): core/matrix/ell_kernels.hpp:21-35
#define DEF(MT, IT, OT, I, IN)                                                                         \
    template <>                                                                                        \
    void spmv<IT, MT, OT, I>(exec_t exec, const matrix::Ell<MT, I>* a, const matrix::Dense<IT>* b,     \
                             matrix::Dense<OT>* c)                                                     \
    {                                                                                                  \
        GKOC_CALL(gkoc_ell_spmv_mixed_##IN(                                                            \
            stream_of(exec), code<MT>::value, code<IT>::value, code<OT>::value, a->get_size()[0],      \
            a->get_size()[1], a->get_num_stored_elements_per_row(), a->get_stride(), nullptr,          \
            a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(), ld(b), nullptr,     \
            c->get_values(), ld(c), cols(c)));                                                         \
    }                                                                                                  \
    template <>                                                                                        \
    void advanced_spmv<IT, MT, OT, I>(exec_t exec, const matrix::Dense<MT>* alpha,                     \
                                      const matrix::Ell<MT, I>* a, const matrix::Dense<IT>* b,         \
                                      const matrix::Dense<OT>* beta, matrix::Dense<OT>* c)             \
    {                                                                                                  \
        GKOC_CALL(gkoc_ell_spmv_mixed_##IN(                                                            \
            stream_of(exec), code<MT>::value, code<IT>::value, code<OT>::value, a->get_size()[0],      \
            a->get_size()[1], a->get_num_stored_elements_per_row(), a->get_stride(),                   \
            alpha->get_const_values(), a->get_const_col_idxs(), a->get_const_values(),                 \
            b->get_const_values(), ld(b), beta->get_const_values(), c->get_values(), ld(c), cols(c))); \
    }
FOR_TRIPLES(DEF)
#undef DEF

}  // namespace ell


namespace dense {

#define DEF(VT, OT, I, IN)                                                                             \
    template <>                                                                                        \
    void row_gather<VT, OT, I>(exec_t exec, const I* gather_indices, const matrix::Dense<VT>* orig,    \
                               matrix::Dense<OT>* row_collection)                                      \
    {                                                                                                  \
        GKOC_CALL(gkoc_dense_row_gather_mixed_##IN(                                                    \
            stream_of(exec), code<VT>::value, code<OT>::value, rows(row_collection), cols(orig), nullptr, \
            gather_indices, orig->get_const_values(), ld(orig), nullptr, row_collection->get_values(), \
            ld(row_collection)));                                                                      \
    }                                                                                                  \
    template <>                                                                                        \
    void advanced_row_gather<VT, OT, I>(exec_t exec, const matrix::Dense<VT>* alpha,                   \
                                        const I* gather_indices, const matrix::Dense<VT>* orig,        \
                                        const matrix::Dense<VT>* beta, matrix::Dense<OT>* row_collection) \
    {                                                                                                  \
        GKOC_CALL(gkoc_dense_row_gather_mixed_##IN(                                                    \
            stream_of(exec), code<VT>::value, code<OT>::value, rows(row_collection), cols(orig),       \
            alpha->get_const_values(), gather_indices, orig->get_const_values(), ld(orig),             \
            beta->get_const_values(), row_collection->get_values(), ld(row_collection)));              \
    }
#define FOR_PAIRS(M)                                                                              \
    M(double, float, int32, i32) M(double, float, int64, i64) M(float, double, int32, i32)        \
        M(float, double, int64, i64) M(std::complex<double>, std::complex<float>, int32, i32)     \
            M(std::complex<double>, std::complex<float>, int64, i64)                              \
                M(std::complex<float>, std::complex<double>, int32, i32)                          \
                    M(std::complex<float>, std::complex<double>, int64, i64)
FOR_PAIRS(DEF)
#undef DEF

// m = beta m + alpha I with real scalars on a complex matrix
#define DEF(T, TN)                                                                                     \
    template <>                                                                                        \
    void add_scaled_identity<T, remove_complex<T>>(exec_t exec, const matrix::Dense<remove_complex<T>>* alpha, \
                                                   const matrix::Dense<remove_complex<T>>* beta,       \
                                                   matrix::Dense<T>* mtx)                              \
    {                                                                                                  \
        GKOC_CALL(gkoc_dense_add_scaled_identity_real_##TN(stream_of(exec), rows(mtx), cols(mtx),      \
                                                           alpha->get_const_values(),                  \
                                                           beta->get_const_values(), mtx->get_values(), \
                                                           ld(mtx)));                                  \
    }
DEF(std::complex<double>, c128)
DEF(std::complex<float>, c64)
#undef DEF

}  // namespace dense


namespace coo {

#define DEF(T, TN)                                                                       \
    template <>                                                                          \
    void conj_array<T>(exec_t exec, size_type num, T* values)                            \
    {                                                                                    \
        GKOC_CALL(gkoc_conj_array_##TN(stream_of(exec), static_cast<int64_t>(num), values)); \
    }
DEF(std::complex<double>, c128)
DEF(std::complex<float>, c64)
#undef DEF

}  // namespace coo

}  // namespace hip
}  // namespace kernels
}  // namespace gko
