// gko::kernels::hip for std::complex<float / double>: the Krylov step kernels Ginkgo's solvers call
// between their products (cg, gmres / common_gmres) and the ELL / SELL-P products - forwarded to the
// complex instantiations of the same C-ABI entries the real types use (csrc/cg_stop.hip,
// csrc/gmres.hip, csrc/complex_formats.hip; value type gkoc_c128 / gkoc_c64 = the layout of
// std::complex).  The element-wise families (bicg, bicgstab, cgs, fcg, gcr, pipe_cg) are in the
// generated krylov.cpp.  Complex kernels agree with the reference to rounding, not bit for bit
// (textbook complex quotient, include/gko_cdna4.h).
#include <complex>

#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/diagonal.hpp>
#include <ginkgo/core/base/device_matrix_data.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/matrix/ell.hpp>
#include <ginkgo/core/matrix/sellp.hpp>
#include <ginkgo/core/preconditioner/jacobi.hpp>

#include "core/matrix/csr_kernels.hpp"
#include "core/matrix/diagonal_kernels.hpp"
#include "core/matrix/dense_kernels.hpp"
#include "core/matrix/ell_kernels.hpp"
#include "core/matrix/sellp_kernels.hpp"
#include "core/preconditioner/jacobi_kernels.hpp"
#include "core/solver/cg_kernels.hpp"
#include "core/solver/common_gmres_kernels.hpp"
#include "core/solver/gmres_kernels.hpp"
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

namespace {
inline gkoc_c128* px(std::complex<double>* p) { return reinterpret_cast<gkoc_c128*>(p); }
inline const gkoc_c128* px(const std::complex<double>* p) { return reinterpret_cast<const gkoc_c128*>(p); }
inline gkoc_c64* px(std::complex<float>* p) { return reinterpret_cast<gkoc_c64*>(p); }
inline const gkoc_c64* px(const std::complex<float>* p) { return reinterpret_cast<const gkoc_c64*>(p); }
}  // namespace

#define FOR_C(M) M(std::complex<double>, c128) M(std::complex<float>, c64)
#define FOR_C_I(M)                                                                            \
    M(std::complex<double>, c128, int32, i32) M(std::complex<double>, c128, int64, i64)       \
        M(std::complex<float>, c64, int32, i32) M(std::complex<float>, c64, int64, i64)

namespace cg {

#define DEF(T, TN)                                                                                  \
    template <>                                                                                     \
    void initialize<T>(exec_t exec, const matrix::Dense<T>* b, matrix::Dense<T>* r,                 \
                       matrix::Dense<T>* z, matrix::Dense<T>* p, matrix::Dense<T>* q,               \
                       matrix::Dense<T>* prev_rho, matrix::Dense<T>* rho,                           \
                       array<stopping_status>* stop_status)                                         \
    {                                                                                               \
        GKOC_CALL(gkoc_cg_initialize_##TN(stream_of(exec), rows(b), cols(b), px(b->get_const_values()), \
                                          ld(b), px(r->get_values()), ld(r), px(z->get_values()),   \
                                          ld(z), px(p->get_values()), ld(p), px(q->get_values()),   \
                                          ld(q), px(prev_rho->get_values()), px(rho->get_values()), \
                                          raw(stop_status)));                                       \
    }                                                                                               \
    template <>                                                                                     \
    void step_1<T>(exec_t exec, matrix::Dense<T>* p, const matrix::Dense<T>* z,                     \
                   const matrix::Dense<T>* rho, const matrix::Dense<T>* prev_rho,                   \
                   const array<stopping_status>* stop_status)                                       \
    {                                                                                               \
        GKOC_CALL(gkoc_cg_step_1_##TN(stream_of(exec), rows(p), cols(p), px(p->get_values()), ld(p), \
                                      px(z->get_const_values()), ld(z), px(rho->get_const_values()), \
                                      px(prev_rho->get_const_values()), raw(stop_status)));         \
    }                                                                                               \
    template <>                                                                                     \
    void step_2<T>(exec_t exec, matrix::Dense<T>* x, matrix::Dense<T>* r, const matrix::Dense<T>* p, \
                   const matrix::Dense<T>* q, const matrix::Dense<T>* beta,                         \
                   const matrix::Dense<T>* rho, const array<stopping_status>* stop_status)          \
    {                                                                                               \
        GKOC_CALL(gkoc_cg_step_2_##TN(stream_of(exec), rows(x), cols(x), px(x->get_values()), ld(x), \
                                      px(r->get_values()), ld(r), px(p->get_const_values()), ld(p), \
                                      px(q->get_const_values()), ld(q),                             \
                                      px(beta->get_const_values()), px(rho->get_const_values()),    \
                                      raw(stop_status)));                                           \
    }
FOR_C(DEF)
#undef DEF

}  // namespace cg


namespace gmres {

#define DEF(T, TN)                                                                                  \
    template <>                                                                                     \
    void restart<T>(exec_t exec, const matrix::Dense<T>* residual,                                  \
                    const matrix::Dense<remove_complex<T>>* residual_norm,                          \
                    matrix::Dense<T>* residual_norm_collection, matrix::Dense<T>* krylov_bases,     \
                    size_type* final_iter_nums)                                                     \
    {                                                                                               \
        GKOC_CALL(gkoc_gmres_restart_##TN(                                                          \
            stream_of(exec), rows(residual), cols(residual), px(residual->get_const_values()),      \
            ld(residual), residual_norm->get_const_values(),                                        \
            px(residual_norm_collection->get_values()), px(krylov_bases->get_values()),             \
            ld(krylov_bases), reinterpret_cast<uint64_t*>(final_iter_nums)));                       \
    }                                                                                               \
    template <>                                                                                     \
    void multi_axpy<T>(exec_t exec, const matrix::Dense<T>* krylov_bases, const matrix::Dense<T>* y, \
                       matrix::Dense<T>* before_preconditioner, const size_type* final_iter_nums,   \
                       stopping_status* stop_status)                                                \
    {                                                                                               \
        GKOC_CALL(gkoc_gmres_multi_axpy_##TN(                                                       \
            stream_of(exec), rows(before_preconditioner), cols(before_preconditioner),              \
            px(krylov_bases->get_const_values()), ld(krylov_bases), px(y->get_const_values()),      \
            ld(y), px(before_preconditioner->get_values()), ld(before_preconditioner),              \
            reinterpret_cast<const uint64_t*>(final_iter_nums), raw(stop_status)));                 \
    }                                                                                               \
    template <>                                                                                     \
    void multi_dot<T>(exec_t exec, const matrix::Dense<T>* krylov_bases,                            \
                      const matrix::Dense<T>* next_krylov, matrix::Dense<T>* hessenberg_col)        \
    {                                                                                               \
        const int64_t n = rows(next_krylov), k = cols(next_krylov);                                 \
        const int64_t dots = rows(hessenberg_col) - 1;                                              \
        const size_t bytes = gkoc_gmres_multi_dot_workspace_bytes(n, k, dots, sizeof(T));           \
        array<char> tmp(exec, bytes);                                                               \
        GKOC_CALL(gkoc_gmres_multi_dot_##TN(                                                        \
            stream_of(exec), n, k, dots, px(krylov_bases->get_const_values()), ld(krylov_bases),    \
            px(next_krylov->get_const_values()), ld(next_krylov), px(hessenberg_col->get_values()), \
            ld(hessenberg_col), tmp.get_data(), bytes));                                            \
        exec->synchronize(); /* tmp is released on return */                                        \
    }
FOR_C(DEF)
#undef DEF

}  // namespace gmres


namespace common_gmres {

#define DEF(T, TN)                                                                                  \
    template <>                                                                                     \
    void initialize<T>(exec_t exec, const matrix::Dense<T>* b, matrix::Dense<T>* residual,          \
                       matrix::Dense<T>* givens_sin, matrix::Dense<T>* givens_cos,                  \
                       stopping_status* stop_status)                                                \
    {                                                                                               \
        GKOC_CALL(gkoc_common_gmres_initialize_##TN(                                                \
            stream_of(exec), rows(b), cols(b), px(b->get_const_values()), ld(b),                    \
            px(residual->get_values()), ld(residual), px(givens_sin->get_values()), ld(givens_sin), \
            px(givens_cos->get_values()), ld(givens_cos), rows(givens_sin), raw(stop_status)));     \
    }                                                                                               \
    template <>                                                                                     \
    void hessenberg_qr<T>(exec_t exec, matrix::Dense<T>* givens_sin, matrix::Dense<T>* givens_cos,  \
                          matrix::Dense<remove_complex<T>>* residual_norm,                          \
                          matrix::Dense<T>* residual_norm_collection,                               \
                          matrix::Dense<T>* hessenberg_iter, size_type iter,                        \
                          size_type* final_iter_nums, const stopping_status* stop_status)           \
    {                                                                                               \
        GKOC_CALL(gkoc_common_gmres_hessenberg_qr_##TN(                                             \
            stream_of(exec), cols(givens_sin), px(givens_sin->get_values()), ld(givens_sin),        \
            px(givens_cos->get_values()), ld(givens_cos), residual_norm->get_values(),              \
            px(residual_norm_collection->get_values()), ld(residual_norm_collection),               \
            px(hessenberg_iter->get_values()), ld(hessenberg_iter), iter,                           \
            reinterpret_cast<uint64_t*>(final_iter_nums), raw(stop_status)));                       \
    }                                                                                               \
    template <>                                                                                     \
    void solve_krylov<T>(exec_t exec, const matrix::Dense<T>* residual_norm_collection,             \
                         const matrix::Dense<T>* hessenberg, matrix::Dense<T>* y,                   \
                         const size_type* final_iter_nums, const stopping_status* stop_status)      \
    {                                                                                               \
        GKOC_CALL(gkoc_common_gmres_solve_krylov_##TN(                                              \
            stream_of(exec), cols(residual_norm_collection),                                        \
            px(residual_norm_collection->get_const_values()), ld(residual_norm_collection),         \
            px(hessenberg->get_const_values()), ld(hessenberg), px(y->get_values()), ld(y),         \
            reinterpret_cast<const uint64_t*>(final_iter_nums), raw(stop_status)));                 \
    }
FOR_C(DEF)
#undef DEF

}  // namespace common_gmres


namespace ell {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void spmv<T, T, T, I>(exec_t exec, const matrix::Ell<T, I>* a, const matrix::Dense<T>* b,       \
                          matrix::Dense<T>* c)                                                      \
    {                                                                                               \
        GKOC_CALL(gkoc_ell_spmv_##TN##_##IN(                                                        \
            stream_of(exec), a->get_size()[0], a->get_size()[1], a->get_num_stored_elements_per_row(), \
            a->get_stride(), a->get_const_col_idxs(), px(a->get_const_values()),                    \
            px(b->get_const_values()), ld(b), px(c->get_values()), ld(c), cols(c)));                \
    }                                                                                               \
    template <>                                                                                     \
    void advanced_spmv<T, T, T, I>(exec_t exec, const matrix::Dense<T>* alpha,                      \
                                   const matrix::Ell<T, I>* a, const matrix::Dense<T>* b,           \
                                   const matrix::Dense<T>* beta, matrix::Dense<T>* c)               \
    {                                                                                               \
        GKOC_CALL(gkoc_ell_advanced_spmv_##TN##_##IN(                                               \
            stream_of(exec), a->get_size()[0], a->get_size()[1], a->get_num_stored_elements_per_row(), \
            a->get_stride(), px(alpha->get_const_values()), a->get_const_col_idxs(),                \
            px(a->get_const_values()), px(b->get_const_values()), ld(b),                            \
            px(beta->get_const_values()), px(c->get_values()), ld(c), cols(c)));                    \
    }
FOR_C_I(DEF)
#undef DEF

}  // namespace ell


namespace sellp {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void spmv<T, I>(exec_t exec, const matrix::Sellp<T, I>* a, const matrix::Dense<T>* b,           \
                    matrix::Dense<T>* c)                                                            \
    {                                                                                               \
        GKOC_CALL(gkoc_sellp_spmv_##TN##_##IN(                                                      \
            stream_of(exec), a->get_size()[0], a->get_size()[1], a->get_slice_size(),               \
            reinterpret_cast<const uint64_t*>(a->get_const_slice_sets()),                           \
            reinterpret_cast<const uint64_t*>(a->get_const_slice_lengths()), a->get_const_col_idxs(), \
            px(a->get_const_values()), px(b->get_const_values()), ld(b), px(c->get_values()), ld(c), \
            cols(c)));                                                                              \
    }                                                                                               \
    template <>                                                                                     \
    void advanced_spmv<T, I>(exec_t exec, const matrix::Dense<T>* alpha,                            \
                             const matrix::Sellp<T, I>* a, const matrix::Dense<T>* b,               \
                             const matrix::Dense<T>* beta, matrix::Dense<T>* c)                     \
    {                                                                                               \
        GKOC_CALL(gkoc_sellp_advanced_spmv_##TN##_##IN(                                             \
            stream_of(exec), a->get_size()[0], a->get_size()[1], a->get_slice_size(),               \
            px(alpha->get_const_values()),                                                          \
            reinterpret_cast<const uint64_t*>(a->get_const_slice_sets()),                           \
            reinterpret_cast<const uint64_t*>(a->get_const_slice_lengths()), a->get_const_col_idxs(), \
            px(a->get_const_values()), px(b->get_const_values()), ld(b),                            \
            px(beta->get_const_values()), px(c->get_values()), ld(c), cols(c)));                    \
    }
FOR_C_I(DEF)
#undef DEF

}  // namespace sellp


// block-Jacobi on complex values (csrc/jacobi.hip: GKOC_DEF_CJACOBI for uniform storage,
// GKOC_DEF_JACOBI_ADAPTIVE_ANY where the factory asked for block-wise / adaptive precision)
namespace jacobi {

namespace {
template <typename I>
gkoc_jacobi_scheme cscheme(const preconditioner::block_interleaved_storage_scheme<I>& s)
{
    return {static_cast<int64_t>(s.block_offset), static_cast<int64_t>(s.group_offset), s.group_power};
}
bool has_precisions(const array<precision_reduction>& prec)
{
    return prec.get_const_data() != nullptr && prec.get_size() != 0;
}
const uint8_t* bytes(const array<precision_reduction>& prec)
{
    return reinterpret_cast<const uint8_t*>(prec.get_const_data());
}
}  // namespace

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void find_blocks<T, I>(exec_t exec, const matrix::Csr<T, I>* system_matrix, uint32 max_block_size, \
                           size_type& num_blocks, array<I>& block_pointers)                         \
    {                                                                                               \
        int64_t nb = 0;                                                                             \
        GKOC_CALL(gkoc_jacobi_find_blocks_##TN##_##IN(                                              \
            stream_of(exec), system_matrix->get_size()[0], system_matrix->get_const_row_ptrs(),     \
            system_matrix->get_const_col_idxs(), max_block_size, &nb, block_pointers.get_data()));  \
        num_blocks = static_cast<size_type>(nb);                                                    \
    }                                                                                               \
    template <>                                                                                     \
    void generate<T, I>(exec_t exec, const matrix::Csr<T, I>* system_matrix, size_type num_blocks,  \
                        uint32 max_block_size, remove_complex<T> accuracy,                          \
                        const preconditioner::block_interleaved_storage_scheme<I>& storage_scheme,  \
                        array<remove_complex<T>>& conditioning,                                     \
                        array<precision_reduction>& block_precisions,                               \
                        const array<I>& block_pointers, array<T>& blocks)                           \
    {                                                                                               \
        if (has_precisions(block_precisions)) {                                                     \
            GKOC_CALL(gkoc_jacobi_generate_adaptive_##TN##_##IN(                                    \
                stream_of(exec), system_matrix->get_size()[0], system_matrix->get_const_row_ptrs(), \
                system_matrix->get_const_col_idxs(), px(system_matrix->get_const_values()),         \
                num_blocks, max_block_size, cscheme(storage_scheme),                                \
                block_pointers.get_const_data(), accuracy,                                          \
                reinterpret_cast<uint8_t*>(block_precisions.get_data()), conditioning.get_data(),   \
                px(blocks.get_data())));                                                            \
            return;                                                                                 \
        }                                                                                           \
        GKOC_CALL(gkoc_jacobi_generate_##TN##_##IN(                                                 \
            stream_of(exec), system_matrix->get_size()[0], system_matrix->get_const_row_ptrs(),     \
            system_matrix->get_const_col_idxs(), px(system_matrix->get_const_values()), num_blocks, \
            max_block_size, cscheme(storage_scheme), block_pointers.get_const_data(),               \
            px(blocks.get_data()), nullptr));                                                       \
    }                                                                                               \
    template <>                                                                                     \
    void simple_apply<T, I>(exec_t exec, size_type num_blocks, uint32 max_block_size,               \
                            const preconditioner::block_interleaved_storage_scheme<I>& storage_scheme, \
                            const array<precision_reduction>& block_precisions,                     \
                            const array<I>& block_pointers, const array<T>& blocks,                 \
                            const matrix::Dense<T>* b, matrix::Dense<T>* x)                         \
    {                                                                                               \
        if (has_precisions(block_precisions)) {                                                     \
            GKOC_CALL(gkoc_jacobi_apply_adaptive_##TN##_##IN(                                       \
                stream_of(exec), num_blocks, max_block_size, cscheme(storage_scheme),               \
                block_pointers.get_const_data(), px(blocks.get_const_data()),                       \
                bytes(block_precisions), nullptr, px(b->get_const_values()), ld(b), nullptr,        \
                px(x->get_values()), ld(x), cols(x)));                                              \
            return;                                                                                 \
        }                                                                                           \
        GKOC_CALL(gkoc_jacobi_simple_apply_##TN##_##IN(                                             \
            stream_of(exec), num_blocks, max_block_size, cscheme(storage_scheme),                   \
            block_pointers.get_const_data(), px(blocks.get_const_data()), px(b->get_const_values()), \
            ld(b), px(x->get_values()), ld(x), cols(x)));                                           \
    }                                                                                               \
    template <>                                                                                     \
    void apply<T, I>(exec_t exec, size_type num_blocks, uint32 max_block_size,                      \
                     const preconditioner::block_interleaved_storage_scheme<I>& storage_scheme,     \
                     const array<precision_reduction>& block_precisions,                            \
                     const array<I>& block_pointers, const array<T>& blocks,                        \
                     const matrix::Dense<T>* alpha, const matrix::Dense<T>* b,                      \
                     const matrix::Dense<T>* beta, matrix::Dense<T>* x)                             \
    {                                                                                               \
        if (has_precisions(block_precisions)) {                                                     \
            GKOC_CALL(gkoc_jacobi_apply_adaptive_##TN##_##IN(                                       \
                stream_of(exec), num_blocks, max_block_size, cscheme(storage_scheme),               \
                block_pointers.get_const_data(), px(blocks.get_const_data()),                       \
                bytes(block_precisions), px(alpha->get_const_values()), px(b->get_const_values()),  \
                ld(b), px(beta->get_const_values()), px(x->get_values()), ld(x), cols(x)));         \
            return;                                                                                 \
        }                                                                                           \
        GKOC_CALL(gkoc_jacobi_apply_##TN##_##IN(                                                    \
            stream_of(exec), num_blocks, max_block_size, cscheme(storage_scheme),                   \
            block_pointers.get_const_data(), px(blocks.get_const_data()),                           \
            px(alpha->get_const_values()), px(b->get_const_values()), ld(b),                        \
            px(beta->get_const_values()), px(x->get_values()), ld(x), cols(x)));                    \
    }
FOR_C_I(DEF)
#undef DEF

}  // namespace jacobi


// conversions on the way to and from the complex products: csr -> ell / sellp
// (csrc/formats.hip: the same staged kernels as for the real types)
namespace csr {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void convert_to_ell<T, I>(exec_t exec, const matrix::Csr<T, I>* source, matrix::Ell<T, I>* result) \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_convert_to_ell_##TN##_##IN(                                              \
            stream_of(exec), source->get_size()[0], source->get_const_row_ptrs(),                   \
            source->get_const_col_idxs(), source->get_const_values(),                               \
            result->get_num_stored_elements_per_row(), result->get_stride(), result->get_col_idxs(), \
            result->get_values()));                                                                 \
    }                                                                                               \
    template <>                                                                                     \
    void convert_to_sellp<T, I>(exec_t exec, const matrix::Csr<T, I>* source,                       \
                                matrix::Sellp<T, I>* result)                                        \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_convert_to_sellp_##TN##_##IN(                                            \
            stream_of(exec), source->get_size()[0], result->get_slice_size(),                       \
            source->get_const_row_ptrs(), source->get_const_col_idxs(), source->get_const_values(), \
            reinterpret_cast<const uint64_t*>(result->get_const_slice_sets()), result->get_col_idxs(), \
            result->get_values()));                                                                 \
    }
FOR_C_I(DEF)
#undef DEF

}  // namespace csr


namespace diagonal {

#define DEF(T, TN)                                                                                  \
    template <>                                                                                     \
    void apply_to_dense<T>(exec_t exec, const matrix::Diagonal<T>* a, const matrix::Dense<T>* b,    \
                           matrix::Dense<T>* c, bool inverse)                                       \
    {                                                                                               \
        GKOC_CALL(gkoc_diagonal_apply_to_dense_##TN(stream_of(exec), rows(b), cols(b),              \
                                                    a->get_const_values(), b->get_const_values(),   \
                                                    ld(b), c->get_values(), ld(c), inverse));       \
    }                                                                                               \
    template <>                                                                                     \
    void right_apply_to_dense<T>(exec_t exec, const matrix::Diagonal<T>* a,                         \
                                 const matrix::Dense<T>* b, matrix::Dense<T>* c)                    \
    {                                                                                               \
        GKOC_CALL(gkoc_diagonal_right_apply_to_dense_##TN(stream_of(exec), rows(b), cols(b),        \
                                                          a->get_const_values(),                    \
                                                          b->get_const_values(), ld(b),             \
                                                          c->get_values(), ld(c)));                 \
    }
FOR_C(DEF)
#undef DEF

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void fill_in_matrix_data<T, I>(exec_t exec, const device_matrix_data<T, I>& data,               \
                                   matrix::Diagonal<T>* output)                                     \
    {                                                                                               \
        GKOC_CALL(gkoc_diagonal_fill_in_matrix_data_##TN##_##IN(                                    \
            stream_of(exec), static_cast<int64_t>(data.get_num_stored_elements()),                  \
            data.get_const_row_idxs(), data.get_const_col_idxs(), data.get_const_values(),          \
            output->get_values()));                                                                 \
    }
FOR_C_I(DEF)
#undef DEF

}  // namespace diagonal


namespace dense {

#define DEF(T, TN)                                                                                  \
    template <>                                                                                     \
    void simple_apply<T>(exec_t exec, const matrix::Dense<T>* a, const matrix::Dense<T>* b,         \
                         matrix::Dense<T>* c)                                                       \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_simple_apply_##TN(stream_of(exec), rows(c), cols(c), cols(a),          \
                                               a->get_const_values(), ld(a), b->get_const_values(), \
                                               ld(b), c->get_values(), ld(c)));                     \
    }                                                                                               \
    template <>                                                                                     \
    void apply<T>(exec_t exec, const matrix::Dense<T>* alpha, const matrix::Dense<T>* a,            \
                  const matrix::Dense<T>* b, const matrix::Dense<T>* beta, matrix::Dense<T>* c)     \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_apply_##TN(stream_of(exec), rows(c), cols(c), cols(a),                 \
                                        alpha->get_const_values(), a->get_const_values(), ld(a),    \
                                        b->get_const_values(), ld(b), beta->get_const_values(),     \
                                        c->get_values(), ld(c)));                                   \
    }                                                                                               \
    template <>                                                                                     \
    void compute_sqrt<T>(exec_t exec, matrix::Dense<T>* data)                                       \
    {                                                                                               \
        for (int64_t r = 0; r < rows(data); ++r) {                                                  \
            GKOC_CALL(gkoc_dense_compute_sqrt_##TN(stream_of(exec), cols(data),                     \
                                                   data->get_values() + r * ld(data)));             \
        }                                                                                           \
    }
FOR_C(DEF)
#undef DEF

}  // namespace dense


// Jacobi::transpose() / conj_transpose() of complex blocks (needed by Bicg)
namespace jacobi {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void transpose_jacobi<T, I>(exec_t exec, size_type num_blocks, uint32,                          \
                                const array<precision_reduction>& block_precisions,                 \
                                const array<I>& block_pointers, const array<T>& blocks,             \
                                const preconditioner::block_interleaved_storage_scheme<I>& scheme,  \
                                array<T>& out_blocks)                                               \
    {                                                                                               \
        if (has_precisions(block_precisions)) {                                                     \
            GKOC_CALL(gkoc_jacobi_transpose_adaptive_##TN##_##IN(                                   \
                stream_of(exec), num_blocks, cscheme(scheme), block_pointers.get_const_data(),      \
                px(blocks.get_const_data()), bytes(block_precisions), 0,                            \
                px(out_blocks.get_data())));                                                        \
            return;                                                                                 \
        }                                                                                           \
        GKOC_CALL(gkoc_cjacobi_transpose_##TN##_##IN(stream_of(exec), num_blocks, cscheme(scheme),  \
                                                     block_pointers.get_const_data(),               \
                                                     blocks.get_const_data(), 0,                    \
                                                     out_blocks.get_data()));                       \
    }                                                                                               \
    template <>                                                                                     \
    void conj_transpose_jacobi<T, I>(exec_t exec, size_type num_blocks, uint32,                     \
                                     const array<precision_reduction>& block_precisions,            \
                                     const array<I>& block_pointers, const array<T>& blocks,        \
                                     const preconditioner::block_interleaved_storage_scheme<I>& scheme, \
                                     array<T>& out_blocks)                                          \
    {                                                                                               \
        if (has_precisions(block_precisions)) {                                                     \
            GKOC_CALL(gkoc_jacobi_transpose_adaptive_##TN##_##IN(                                   \
                stream_of(exec), num_blocks, cscheme(scheme), block_pointers.get_const_data(),      \
                px(blocks.get_const_data()), bytes(block_precisions), 1,                            \
                px(out_blocks.get_data())));                                                        \
            return;                                                                                 \
        }                                                                                           \
        GKOC_CALL(gkoc_cjacobi_transpose_##TN##_##IN(stream_of(exec), num_blocks, cscheme(scheme),  \
                                                     block_pointers.get_const_data(),               \
                                                     blocks.get_const_data(), 1,                    \
                                                     out_blocks.get_data()));                       \
    }
FOR_C_I(DEF)
#undef DEF

}  // namespace jacobi

}  // namespace hip
}  // namespace kernels
}  // namespace gko
