// gko::kernels::hip: the conversions between Dense / Csr / Coo / Ell / Sellp / Hybrid that Ginkgo's
// matrix classes run on the device, and the diagonal / transpose / 1-norm helpers, forwarded to the
// C ABI (csrc/conversions.hip).  Real value types; complex stays with Ginkgo's NotCompiled stubs.
#include <ginkgo/core/matrix/coo.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/matrix/diagonal.hpp>
#include <ginkgo/core/matrix/ell.hpp>
#include <ginkgo/core/matrix/hybrid.hpp>
#include <ginkgo/core/matrix/sellp.hpp>
#include <ginkgo/core/matrix/sparsity_csr.hpp>

#include "core/components/absolute_array_kernels.hpp"
#include "core/components/fill_array_kernels.hpp"
#include "core/factorization/factorization_kernels.hpp"
#include "core/preconditioner/jacobi_kernels.hpp"
#include "core/matrix/csr_builder.hpp"
#include "core/matrix/coo_kernels.hpp"
#include "core/matrix/csr_kernels.hpp"
#include "core/matrix/dense_kernels.hpp"
#include "core/matrix/ell_kernels.hpp"
#include "core/matrix/hybrid_kernels.hpp"
#include "core/matrix/permutation_kernels.hpp"
#include "core/matrix/scaled_permutation_kernels.hpp"
#include "core/matrix/sellp_kernels.hpp"
#include <complex>

#include <ginkgo/core/base/index_set.hpp>

#include "shim_common.hpp"

namespace gko {
namespace kernels {
namespace hip {

using cdna4::cols;
using cdna4::ld;
using cdna4::rows;
using cdna4::stream_of;
using exec_t = std::shared_ptr<const HipExecutor>;

#define FOR_VT(M) M(double, f64) M(float, f32)
#define FOR_VT_IT(M)                                                                \
    M(double, f64, int32, i32) M(double, f64, int64, i64) M(float, f32, int32, i32) \
        M(float, f32, int64, i64)
// ... and with the two complex types (gkoc_c128 / gkoc_c64 are std::complex here: shim_common.hpp)
#define FOR_AVT(M) FOR_VT(M) M(std::complex<double>, c128) M(std::complex<float>, c64)
#define FOR_AVT_IT(M)                                                                            \
    FOR_VT_IT(M)                                                                                 \
    M(std::complex<double>, c128, int32, i32) M(std::complex<double>, c128, int64, i64)          \
        M(std::complex<float>, c64, int32, i32) M(std::complex<float>, c64, int64, i64)

inline const uint64_t* u64(const size_type* p) { return reinterpret_cast<const uint64_t*>(p); }
inline uint64_t* u64(size_type* p) { return reinterpret_cast<uint64_t*>(p); }


namespace components {

#define DEF(T, TN)                                                                              \
    template <>                                                                                 \
    void inplace_absolute_array<T>(exec_t exec, T* data, size_type n)                           \
    {                                                                                           \
        GKOC_CALL(gkoc_dense_absolute_##TN(stream_of(exec), static_cast<int64_t>(n), 1, data, 1, \
                                           data, 1));                                           \
    }                                                                                           \
    template <>                                                                                 \
    void outplace_absolute_array<T>(exec_t exec, const T* in, size_type n, T* out)              \
    {                                                                                           \
        GKOC_CALL(gkoc_dense_absolute_##TN(stream_of(exec), static_cast<int64_t>(n), 1, in, 1,   \
                                           out, 1));                                            \
    }                                                                                           \
    template <>                                                                                 \
    void fill_seq_array<T>(exec_t exec, T* data, size_type n)                                   \
    {                                                                                           \
        GKOC_CALL(gkoc_fill_seq_array_##TN(stream_of(exec), data, static_cast<int64_t>(n)));    \
    }
FOR_VT(DEF)
#undef DEF

template <>
void fill_seq_array<size_type>(exec_t exec, size_type* data, size_type n)
{
    GKOC_CALL(gkoc_fill_seq_array_u64(stream_of(exec), u64(data), static_cast<int64_t>(n)));
}

}  // namespace components


namespace dense {

#define DEF(T, TN)                                                                                  \
    template <>                                                                                     \
    void compute_norm1<T>(exec_t exec, const matrix::Dense<T>* x, matrix::Dense<T>* result,         \
                          array<char>& tmp)                                                         \
    {                                                                                               \
        const auto s = stream_of(exec);                                                             \
        const size_t bytes = gkoc_reduction_workspace_bytes(rows(x), cols(x), sizeof(T));           \
        if (tmp.get_size() < bytes) tmp.resize_and_reset(bytes);                                    \
        GKOC_CALL(gkoc_dense_compute_norm1_##TN(s, rows(x), cols(x), x->get_const_values(), ld(x),  \
                                                result->get_values(), tmp.get_data(), bytes));      \
    }                                                                                               \
    template <>                                                                                     \
    void conj_transpose<T>(exec_t exec, const matrix::Dense<T>* orig, matrix::Dense<T>* trans)      \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_transpose_##TN(stream_of(exec), rows(orig), cols(orig),                \
                                            orig->get_const_values(), ld(orig),                     \
                                            trans->get_values(), ld(trans)));                       \
    }                                                                                               \
    template <>                                                                                     \
    void count_nonzeros_per_row<T, int32>(exec_t exec, const matrix::Dense<T>* source,              \
                                          int32* result)                                            \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_count_nonzeros_per_row_##TN(stream_of(exec), rows(source),             \
                                                         cols(source), source->get_const_values(),  \
                                                         ld(source), result, 4));                   \
    }                                                                                               \
    template <>                                                                                     \
    void count_nonzeros_per_row<T, int64>(exec_t exec, const matrix::Dense<T>* source,              \
                                          int64* result)                                            \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_count_nonzeros_per_row_##TN(stream_of(exec), rows(source),             \
                                                         cols(source), source->get_const_values(),  \
                                                         ld(source), result, 8));                   \
    }                                                                                               \
    template <>                                                                                     \
    void count_nonzeros_per_row<T, size_type>(exec_t exec, const matrix::Dense<T>* source,          \
                                              size_type* result)                                    \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_count_nonzeros_per_row_##TN(stream_of(exec), rows(source),             \
                                                         cols(source), source->get_const_values(),  \
                                                         ld(source), result, 8));                   \
    }
FOR_VT(DEF)
#undef DEF

#define DEF(T, TN)                                                                                  \
    template <>                                                                                     \
    void transpose<T>(exec_t exec, const matrix::Dense<T>* orig, matrix::Dense<T>* trans)           \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_transpose_##TN(stream_of(exec), rows(orig), cols(orig),                \
                                            orig->get_const_values(), ld(orig),                     \
                                            trans->get_values(), ld(trans)));                       \
    }                                                                                               \
    template <>                                                                                     \
    void extract_diagonal<T>(exec_t exec, const matrix::Dense<T>* orig,                             \
                             matrix::Diagonal<T>* diag)                                             \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_extract_diagonal_##TN(                                                 \
            stream_of(exec), static_cast<int64_t>(diag->get_size()[0]), orig->get_const_values(),   \
            ld(orig), diag->get_values()));                                                         \
    }                                                                                               \
    template <>                                                                                     \
    void add_scaled_identity<T, T>(exec_t exec, const matrix::Dense<T>* alpha,                      \
                                   const matrix::Dense<T>* beta, matrix::Dense<T>* mtx)             \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_add_scaled_identity_##TN(                                              \
            stream_of(exec), rows(mtx), cols(mtx), alpha->get_const_values(),                       \
            beta->get_const_values(), mtx->get_values(), ld(mtx)));                                 \
    }                                                                                               \
    template <>                                                                                     \
    void add_scaled_diag<T>(exec_t exec, const matrix::Dense<T>* alpha,                             \
                            const matrix::Diagonal<T>* x, matrix::Dense<T>* y)                      \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_add_scaled_diag_##TN(                                                  \
            stream_of(exec), static_cast<int64_t>(x->get_size()[0]), alpha->get_const_values(),     \
            x->get_const_values(), y->get_values(), ld(y), 0));                                     \
    }                                                                                               \
    template <>                                                                                     \
    void sub_scaled_diag<T>(exec_t exec, const matrix::Dense<T>* alpha,                             \
                            const matrix::Diagonal<T>* x, matrix::Dense<T>* y)                      \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_add_scaled_diag_##TN(                                                  \
            stream_of(exec), static_cast<int64_t>(x->get_size()[0]), alpha->get_const_values(),     \
            x->get_const_values(), y->get_values(), ld(y), 1));                                     \
    }                                                                                               \
    template <>                                                                                     \
    void compute_max_nnz_per_row<T>(exec_t exec, const matrix::Dense<T>* source,                    \
                                    size_type& result)                                              \
    {                                                                                               \
        uint64_t r = 0;                                                                             \
        GKOC_CALL(gkoc_dense_max_nnz_per_row_##TN(stream_of(exec), rows(source), cols(source),      \
                                                  source->get_const_values(), ld(source), &r));     \
        result = static_cast<size_type>(r);                                                         \
    }                                                                                               \
    template <>                                                                                     \
    void compute_slice_sets<T>(exec_t exec, const matrix::Dense<T>* source, size_type slice_size,   \
                               size_type stride_factor, size_type* slice_sets,                      \
                               size_type* slice_lengths)                                            \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_compute_slice_sets_##TN(                                               \
            stream_of(exec), rows(source), cols(source), source->get_const_values(), ld(source),    \
            static_cast<int64_t>(slice_size), static_cast<int64_t>(stride_factor),                  \
            u64(slice_sets), u64(slice_lengths)));                                                  \
    }
FOR_AVT(DEF)
#undef DEF

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void convert_to_csr<T, I>(exec_t exec, const matrix::Dense<T>* source,                          \
                              matrix::Csr<T, I>* result)                                            \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_to_csr_##TN##_##IN(                                                    \
            stream_of(exec), rows(source), cols(source), source->get_const_values(), ld(source),    \
            result->get_const_row_ptrs(), result->get_col_idxs(), result->get_values()));           \
    }                                                                                               \
    template <>                                                                                     \
    void convert_to_sparsity_csr<T, I>(exec_t exec, const matrix::Dense<T>* source,                 \
                                       matrix::SparsityCsr<T, I>* result)                           \
    {                                                                                               \
        const auto s = stream_of(exec);                                                             \
        GKOC_CALL(gkoc_dense_to_csr_##TN##_##IN(s, rows(source), cols(source),                      \
                                                source->get_const_values(), ld(source),             \
                                                result->get_const_row_ptrs(),                       \
                                                result->get_col_idxs(), nullptr));                  \
        GKOC_CALL(gkoc_fill_array_##TN(s, result->get_value(), 1, T(1)));                           \
    }
FOR_VT_IT(DEF)
#undef DEF

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void convert_to_coo<T, I>(exec_t exec, const matrix::Dense<T>* source, const int64* row_ptrs,   \
                              matrix::Coo<T, I>* result)                                            \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_to_coo_##TN##_##IN(                                                    \
            stream_of(exec), rows(source), cols(source), source->get_const_values(), ld(source),    \
            row_ptrs, result->get_row_idxs(), result->get_col_idxs(), result->get_values()));       \
    }                                                                                               \
    template <>                                                                                     \
    void convert_to_ell<T, I>(exec_t exec, const matrix::Dense<T>* source,                          \
                              matrix::Ell<T, I>* result)                                            \
    {                                                                                               \
        const int64_t k = static_cast<int64_t>(result->get_num_stored_elements_per_row());          \
        GKOC_CALL(gkoc_dense_to_ell_##TN##_##IN(                                                    \
            stream_of(exec), rows(source), cols(source), source->get_const_values(), ld(source), k, \
            k, static_cast<int64_t>(result->get_stride()), result->get_col_idxs(),                  \
            result->get_values(), nullptr, nullptr, nullptr, nullptr));                             \
    }                                                                                               \
    template <>                                                                                     \
    void convert_to_hybrid<T, I>(exec_t exec, const matrix::Dense<T>* source,                       \
                                 const int64* coo_row_ptrs, matrix::Hybrid<T, I>* result)           \
    {                                                                                               \
        const int64_t k = static_cast<int64_t>(result->get_ell_num_stored_elements_per_row());      \
        GKOC_CALL(gkoc_dense_to_ell_##TN##_##IN(                                                    \
            stream_of(exec), rows(source), cols(source), source->get_const_values(), ld(source), k, \
            k, static_cast<int64_t>(result->get_ell_stride()), result->get_ell_col_idxs(),          \
            result->get_ell_values(), coo_row_ptrs, result->get_coo_row_idxs(),                     \
            result->get_coo_col_idxs(), result->get_coo_values()));                                 \
    }                                                                                               \
    template <>                                                                                     \
    void convert_to_sellp<T, I>(exec_t exec, const matrix::Dense<T>* source,                        \
                                matrix::Sellp<T, I>* result)                                        \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_to_sellp_##TN##_##IN(                                                  \
            stream_of(exec), rows(source), cols(source), source->get_const_values(), ld(source),    \
            static_cast<int64_t>(result->get_slice_size()), u64(result->get_const_slice_sets()),    \
            result->get_col_idxs(), result->get_values()));                                         \
    }
FOR_AVT_IT(DEF)
#undef DEF

}  // namespace dense


namespace csr {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void fill_in_dense<T, I>(exec_t exec, const matrix::Csr<T, I>* source,                          \
                             matrix::Dense<T>* result)                                              \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_fill_in_dense_##TN##_##IN(                                               \
            stream_of(exec), static_cast<int64_t>(source->get_size()[0]),                           \
            source->get_const_row_ptrs(), source->get_const_col_idxs(), source->get_const_values(), \
            result->get_values(), ld(result)));                                                     \
    }                                                                                               \
    template <>                                                                                     \
    void check_diagonal_entries_exist<T, I>(exec_t exec, const matrix::Csr<T, I>* mtx,              \
                                            bool& has_all_diags)                                    \
    {                                                                                               \
        int missing = 0;                                                                            \
        const auto n = std::min(mtx->get_size()[0], mtx->get_size()[1]);                            \
        GKOC_CALL(gkoc_csr_missing_diagonal_##IN(stream_of(exec), static_cast<int64_t>(n),          \
                                                 mtx->get_const_row_ptrs(),                         \
                                                 mtx->get_const_col_idxs(), &missing));             \
        has_all_diags = missing == 0;                                                               \
    }                                                                                               \
    template <>                                                                                     \
    void add_scaled_identity<T, I>(exec_t exec, const matrix::Dense<T>* alpha,                      \
                                   const matrix::Dense<T>* beta, matrix::Csr<T, I>* mtx)            \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_add_scaled_identity_##TN##_##IN(                                         \
            stream_of(exec), static_cast<int64_t>(mtx->get_size()[0]), mtx->get_const_row_ptrs(),   \
            mtx->get_const_col_idxs(), mtx->get_values(), alpha->get_const_values(),                \
            beta->get_const_values()));                                                             \
    }
FOR_AVT_IT(DEF)
#undef DEF

}  // namespace csr


namespace coo {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void fill_in_dense<T, I>(exec_t exec, const matrix::Coo<T, I>* source,                          \
                             matrix::Dense<T>* result)                                              \
    {                                                                                               \
        GKOC_CALL(gkoc_coo_fill_in_dense_##TN##_##IN(                                               \
            stream_of(exec), static_cast<int64_t>(source->get_num_stored_elements()),               \
            source->get_const_row_idxs(), source->get_const_col_idxs(), source->get_const_values(), \
            result->get_values(), ld(result)));                                                     \
    }                                                                                               \
    template <>                                                                                     \
    void extract_diagonal<T, I>(exec_t exec, const matrix::Coo<T, I>* orig,                         \
                                matrix::Diagonal<T>* diag)                                          \
    {                                                                                               \
        GKOC_CALL(gkoc_coo_extract_diagonal_##TN##_##IN(                                            \
            stream_of(exec), static_cast<int64_t>(orig->get_num_stored_elements()),                 \
            orig->get_const_row_idxs(), orig->get_const_col_idxs(), orig->get_const_values(),       \
            diag->get_values()));                                                                   \
    }
FOR_AVT_IT(DEF)
#undef DEF

}  // namespace coo


namespace ell {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void fill_in_dense<T, I>(exec_t exec, const matrix::Ell<T, I>* source,                          \
                             matrix::Dense<T>* result)                                              \
    {                                                                                               \
        GKOC_CALL(gkoc_ell_fill_in_dense_##TN##_##IN(                                               \
            stream_of(exec), static_cast<int64_t>(source->get_size()[0]),                           \
            static_cast<int64_t>(source->get_num_stored_elements_per_row()),                        \
            static_cast<int64_t>(source->get_stride()), source->get_const_col_idxs(),               \
            source->get_const_values(), result->get_values(), ld(result)));                         \
    }                                                                                               \
    template <>                                                                                     \
    void extract_diagonal<T, I>(exec_t exec, const matrix::Ell<T, I>* orig,                         \
                                matrix::Diagonal<T>* diag)                                          \
    {                                                                                               \
        GKOC_CALL(gkoc_ell_extract_diagonal_##TN##_##IN(                                            \
            stream_of(exec), static_cast<int64_t>(diag->get_size()[0]),                             \
            static_cast<int64_t>(orig->get_num_stored_elements_per_row()),                          \
            static_cast<int64_t>(orig->get_stride()), orig->get_const_col_idxs(),                   \
            orig->get_const_values(), diag->get_values()));                                         \
    }                                                                                               \
    template <>                                                                                     \
    void count_nonzeros_per_row<T, I>(exec_t exec, const matrix::Ell<T, I>* source, I* result)      \
    {                                                                                               \
        GKOC_CALL(gkoc_ell_count_nonzeros_per_row_##IN(                                             \
            stream_of(exec), static_cast<int64_t>(source->get_size()[0]),                           \
            static_cast<int64_t>(source->get_num_stored_elements_per_row()),                        \
            static_cast<int64_t>(source->get_stride()), source->get_const_col_idxs(), result));     \
    }                                                                                               \
    template <>                                                                                     \
    void convert_to_csr<T, I>(exec_t exec, const matrix::Ell<T, I>* source,                         \
                              matrix::Csr<T, I>* result)                                            \
    {                                                                                               \
        GKOC_CALL(gkoc_ell_to_csr_##TN##_##IN(                                                      \
            stream_of(exec), static_cast<int64_t>(source->get_size()[0]),                           \
            static_cast<int64_t>(source->get_num_stored_elements_per_row()),                        \
            static_cast<int64_t>(source->get_stride()), source->get_const_col_idxs(),               \
            source->get_const_values(), result->get_const_row_ptrs(), result->get_col_idxs(),       \
            result->get_values()));                                                                 \
    }
FOR_AVT_IT(DEF)
#undef DEF

}  // namespace ell


namespace sellp {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void fill_in_dense<T, I>(exec_t exec, const matrix::Sellp<T, I>* source,                        \
                             matrix::Dense<T>* result)                                              \
    {                                                                                               \
        GKOC_CALL(gkoc_sellp_fill_in_dense_##TN##_##IN(                                             \
            stream_of(exec), static_cast<int64_t>(source->get_size()[0]),                           \
            static_cast<int64_t>(source->get_slice_size()), u64(source->get_const_slice_sets()),    \
            source->get_const_col_idxs(), source->get_const_values(), result->get_values(),         \
            ld(result)));                                                                           \
    }                                                                                               \
    template <>                                                                                     \
    void extract_diagonal<T, I>(exec_t exec, const matrix::Sellp<T, I>* orig,                       \
                                matrix::Diagonal<T>* diag)                                          \
    {                                                                                               \
        GKOC_CALL(gkoc_sellp_extract_diagonal_##TN##_##IN(                                          \
            stream_of(exec), static_cast<int64_t>(diag->get_size()[0]),                             \
            static_cast<int64_t>(orig->get_slice_size()), u64(orig->get_const_slice_sets()),        \
            orig->get_const_col_idxs(), orig->get_const_values(), diag->get_values()));             \
    }                                                                                               \
    template <>                                                                                     \
    void count_nonzeros_per_row<T, I>(exec_t exec, const matrix::Sellp<T, I>* source, I* result)    \
    {                                                                                               \
        GKOC_CALL(gkoc_sellp_count_nonzeros_per_row_##IN(                                           \
            stream_of(exec), static_cast<int64_t>(source->get_size()[0]),                           \
            static_cast<int64_t>(source->get_slice_size()), u64(source->get_const_slice_sets()),    \
            source->get_const_col_idxs(), result));                                                 \
    }                                                                                               \
    template <>                                                                                     \
    void convert_to_csr<T, I>(exec_t exec, const matrix::Sellp<T, I>* source,                       \
                              matrix::Csr<T, I>* result)                                            \
    {                                                                                               \
        GKOC_CALL(gkoc_sellp_to_csr_##TN##_##IN(                                                    \
            stream_of(exec), static_cast<int64_t>(source->get_size()[0]),                           \
            static_cast<int64_t>(source->get_slice_size()), u64(source->get_const_slice_sets()),    \
            source->get_const_col_idxs(), source->get_const_values(),                               \
            result->get_const_row_ptrs(), result->get_col_idxs(), result->get_values()));           \
    }
FOR_AVT_IT(DEF)
#undef DEF

}  // namespace sellp


namespace hybrid {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void convert_to_csr<T, I>(exec_t exec, const matrix::Hybrid<T, I>* source,                      \
                              const I* ell_row_ptrs, const I* coo_row_ptrs,                         \
                              matrix::Csr<T, I>* result)                                            \
    {                                                                                               \
        const auto ell = source->get_ell();                                                         \
        GKOC_CALL(gkoc_hybrid_to_csr_##TN##_##IN(                                                   \
            stream_of(exec), static_cast<int64_t>(source->get_size()[0]),                           \
            static_cast<int64_t>(ell->get_num_stored_elements_per_row()),                           \
            static_cast<int64_t>(ell->get_stride()), ell->get_const_col_idxs(),                     \
            ell->get_const_values(), source->get_const_coo_col_idxs(),                              \
            source->get_const_coo_values(), ell_row_ptrs, coo_row_ptrs, result->get_row_ptrs(),     \
            result->get_col_idxs(), result->get_values()));                                         \
    }
FOR_AVT_IT(DEF)
#undef DEF

}  // namespace hybrid



// ------------------------------------------------------------ permutations
namespace dense {

// one C entry point does all sixteen: (row_perm, col_perm, row_scale, col_scale, inverse)
#define PERMUTE(TN, IN, rp, cp, rs, cs, inv)                                                        \
    GKOC_CALL(gkoc_dense_permute_##TN##_##IN(stream_of(exec), rows(orig), cols(orig),               \
                                             orig->get_const_values(), ld(orig),                    \
                                             permuted->get_values(), ld(permuted), rp, cp, rs, cs,  \
                                             inv))
#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void symm_permute<T, I>(exec_t exec, const I* perm, const matrix::Dense<T>* orig,               \
                            matrix::Dense<T>* permuted)                                             \
    {                                                                                               \
        PERMUTE(TN, IN, perm, perm, nullptr, nullptr, 0);                                           \
    }                                                                                               \
    template <>                                                                                     \
    void inv_symm_permute<T, I>(exec_t exec, const I* perm, const matrix::Dense<T>* orig,           \
                                matrix::Dense<T>* permuted)                                         \
    {                                                                                               \
        PERMUTE(TN, IN, perm, perm, nullptr, nullptr, 1);                                           \
    }                                                                                               \
    template <>                                                                                     \
    void nonsymm_permute<T, I>(exec_t exec, const I* rp, const I* cp,                               \
                               const matrix::Dense<T>* orig, matrix::Dense<T>* permuted)            \
    {                                                                                               \
        PERMUTE(TN, IN, rp, cp, nullptr, nullptr, 0);                                               \
    }                                                                                               \
    template <>                                                                                     \
    void inv_nonsymm_permute<T, I>(exec_t exec, const I* rp, const I* cp,                           \
                                   const matrix::Dense<T>* orig, matrix::Dense<T>* permuted)        \
    {                                                                                               \
        PERMUTE(TN, IN, rp, cp, nullptr, nullptr, 1);                                               \
    }                                                                                               \
    template <>                                                                                     \
    void col_permute<T, I>(exec_t exec, const I* perm, const matrix::Dense<T>* orig,                \
                           matrix::Dense<T>* permuted)                                              \
    {                                                                                               \
        PERMUTE(TN, IN, nullptr, perm, nullptr, nullptr, 0);                                        \
    }                                                                                               \
    template <>                                                                                     \
    void inv_row_permute<T, I>(exec_t exec, const I* perm, const matrix::Dense<T>* orig,            \
                               matrix::Dense<T>* permuted)                                          \
    {                                                                                               \
        PERMUTE(TN, IN, perm, nullptr, nullptr, nullptr, 1);                                        \
    }                                                                                               \
    template <>                                                                                     \
    void inv_col_permute<T, I>(exec_t exec, const I* perm, const matrix::Dense<T>* orig,            \
                               matrix::Dense<T>* permuted)                                          \
    {                                                                                               \
        PERMUTE(TN, IN, nullptr, perm, nullptr, nullptr, 1);                                        \
    }                                                                                               \
    template <>                                                                                     \
    void symm_scale_permute<T, I>(exec_t exec, const T* scale, const I* perm,                       \
                                  const matrix::Dense<T>* orig, matrix::Dense<T>* permuted)         \
    {                                                                                               \
        PERMUTE(TN, IN, perm, perm, scale, scale, 0);                                               \
    }                                                                                               \
    template <>                                                                                     \
    void inv_symm_scale_permute<T, I>(exec_t exec, const T* scale, const I* perm,                   \
                                      const matrix::Dense<T>* orig, matrix::Dense<T>* permuted)     \
    {                                                                                               \
        PERMUTE(TN, IN, perm, perm, scale, scale, 1);                                               \
    }                                                                                               \
    template <>                                                                                     \
    void nonsymm_scale_permute<T, I>(exec_t exec, const T* rs, const I* rp, const T* cs,            \
                                     const I* cp, const matrix::Dense<T>* orig,                     \
                                     matrix::Dense<T>* permuted)                                    \
    {                                                                                               \
        PERMUTE(TN, IN, rp, cp, rs, cs, 0);                                                         \
    }                                                                                               \
    template <>                                                                                     \
    void inv_nonsymm_scale_permute<T, I>(exec_t exec, const T* rs, const I* rp, const T* cs,        \
                                         const I* cp, const matrix::Dense<T>* orig,                 \
                                         matrix::Dense<T>* permuted)                                \
    {                                                                                               \
        PERMUTE(TN, IN, rp, cp, rs, cs, 1);                                                         \
    }                                                                                               \
    template <>                                                                                     \
    void row_scale_permute<T, I>(exec_t exec, const T* scale, const I* perm,                        \
                                 const matrix::Dense<T>* orig, matrix::Dense<T>* permuted)          \
    {                                                                                               \
        PERMUTE(TN, IN, perm, nullptr, scale, nullptr, 0);                                          \
    }                                                                                               \
    template <>                                                                                     \
    void inv_row_scale_permute<T, I>(exec_t exec, const T* scale, const I* perm,                    \
                                     const matrix::Dense<T>* orig, matrix::Dense<T>* permuted)      \
    {                                                                                               \
        PERMUTE(TN, IN, perm, nullptr, scale, nullptr, 1);                                          \
    }                                                                                               \
    template <>                                                                                     \
    void col_scale_permute<T, I>(exec_t exec, const T* scale, const I* perm,                        \
                                 const matrix::Dense<T>* orig, matrix::Dense<T>* permuted)          \
    {                                                                                               \
        PERMUTE(TN, IN, nullptr, perm, nullptr, scale, 0);                                          \
    }                                                                                               \
    template <>                                                                                     \
    void inv_col_scale_permute<T, I>(exec_t exec, const T* scale, const I* perm,                    \
                                     const matrix::Dense<T>* orig, matrix::Dense<T>* permuted)      \
    {                                                                                               \
        PERMUTE(TN, IN, nullptr, perm, nullptr, scale, 1);                                          \
    }                                                                                               \
    template <>                                                                                     \
    void advanced_row_gather<T, T, I>(exec_t exec, const matrix::Dense<T>* alpha,                   \
                                      const I* gather_indices, const matrix::Dense<T>* orig,        \
                                      const matrix::Dense<T>* beta,                                 \
                                      matrix::Dense<T>* row_collection)                             \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_advanced_row_gather_##TN##_##IN(                                       \
            stream_of(exec), rows(row_collection), cols(orig), alpha->get_const_values(),           \
            gather_indices, orig->get_const_values(), ld(orig), beta->get_const_values(),           \
            row_collection->get_values(), ld(row_collection)));                                     \
    }
FOR_AVT_IT(DEF)
#undef DEF
#undef PERMUTE

// real value types: the real part is the matrix, the imaginary part is zero
#define DEF(T, TN)                                                                                  \
    template <>                                                                                     \
    void get_real<T>(exec_t exec, const matrix::Dense<T>* source, matrix::Dense<T>* result)         \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_copy_##TN(stream_of(exec), rows(source), cols(source),                 \
                                       source->get_const_values(), ld(source),                      \
                                       result->get_values(), ld(result)));                          \
    }                                                                                               \
    template <>                                                                                     \
    void get_imag<T>(exec_t exec, const matrix::Dense<T>* source, matrix::Dense<T>* result)         \
    {                                                                                               \
        GKOC_CALL(gkoc_dense_fill_##TN(stream_of(exec), rows(result), cols(result),                 \
                                       result->get_values(), ld(result), T(0)));                    \
    }
FOR_VT(DEF)
#undef DEF

}  // namespace dense


namespace csr {

#define PERMUTE(TN, IN, rp, rinv, cp, rs, cs, mode)                                                 \
    GKOC_CALL(gkoc_csr_permute_##TN##_##IN(                                                         \
        stream_of(exec), static_cast<int64_t>(orig->get_size()[0]), orig->get_const_row_ptrs(),     \
        orig->get_const_col_idxs(), orig->get_const_values(), rp, rinv, cp, rs, cs, mode,           \
        permuted->get_row_ptrs(), permuted->get_col_idxs(), permuted->get_values()))
#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void inv_symm_permute<T, I>(exec_t exec, const I* perm, const matrix::Csr<T, I>* orig,          \
                                matrix::Csr<T, I>* permuted)                                        \
    {                                                                                               \
        PERMUTE(TN, IN, perm, 1, perm, nullptr, nullptr, 0);                                        \
    }                                                                                               \
    template <>                                                                                     \
    void inv_nonsymm_permute<T, I>(exec_t exec, const I* rp, const I* cp,                           \
                                   const matrix::Csr<T, I>* orig, matrix::Csr<T, I>* permuted)      \
    {                                                                                               \
        PERMUTE(TN, IN, rp, 1, cp, nullptr, nullptr, 0);                                            \
    }                                                                                               \
    template <>                                                                                     \
    void row_permute<T, I>(exec_t exec, const I* perm, const matrix::Csr<T, I>* orig,               \
                           matrix::Csr<T, I>* permuted)                                             \
    {                                                                                               \
        PERMUTE(TN, IN, perm, 0, nullptr, nullptr, nullptr, 0);                                     \
    }                                                                                               \
    template <>                                                                                     \
    void inv_row_permute<T, I>(exec_t exec, const I* perm, const matrix::Csr<T, I>* orig,           \
                               matrix::Csr<T, I>* permuted)                                         \
    {                                                                                               \
        PERMUTE(TN, IN, perm, 1, nullptr, nullptr, nullptr, 0);                                     \
    }                                                                                               \
    template <>                                                                                     \
    void inv_col_permute<T, I>(exec_t exec, const I* perm, const matrix::Csr<T, I>* orig,           \
                               matrix::Csr<T, I>* permuted)                                         \
    {                                                                                               \
        PERMUTE(TN, IN, nullptr, 0, perm, nullptr, nullptr, 0);                                     \
    }                                                                                               \
    template <>                                                                                     \
    void inv_symm_scale_permute<T, I>(exec_t exec, const T* scale, const I* perm,                   \
                                      const matrix::Csr<T, I>* orig, matrix::Csr<T, I>* permuted)   \
    {                                                                                               \
        PERMUTE(TN, IN, perm, 1, perm, scale, scale, 2);                                            \
    }                                                                                               \
    template <>                                                                                     \
    void inv_nonsymm_scale_permute<T, I>(exec_t exec, const T* rs, const I* rp, const T* cs,        \
                                         const I* cp, const matrix::Csr<T, I>* orig,                \
                                         matrix::Csr<T, I>* permuted)                               \
    {                                                                                               \
        PERMUTE(TN, IN, rp, 1, cp, rs, cs, 2);                                                      \
    }                                                                                               \
    template <>                                                                                     \
    void row_scale_permute<T, I>(exec_t exec, const T* scale, const I* perm,                        \
                                 const matrix::Csr<T, I>* orig, matrix::Csr<T, I>* permuted)        \
    {                                                                                               \
        PERMUTE(TN, IN, perm, 0, nullptr, scale, nullptr, 1);                                       \
    }                                                                                               \
    template <>                                                                                     \
    void inv_row_scale_permute<T, I>(exec_t exec, const T* scale, const I* perm,                    \
                                     const matrix::Csr<T, I>* orig, matrix::Csr<T, I>* permuted)    \
    {                                                                                               \
        PERMUTE(TN, IN, perm, 1, nullptr, scale, nullptr, 2);                                       \
    }                                                                                               \
    template <>                                                                                     \
    void inv_col_scale_permute<T, I>(exec_t exec, const T* scale, const I* perm,                    \
                                     const matrix::Csr<T, I>* orig, matrix::Csr<T, I>* permuted)    \
    {                                                                                               \
        PERMUTE(TN, IN, nullptr, 0, perm, nullptr, scale, 2);                                       \
    }                                                                                               \
    template <>                                                                                     \
    void calculate_nonzeros_per_row_in_span<T, I>(exec_t exec, const matrix::Csr<T, I>* source,     \
                                                  const span& row_span, const span& col_span,       \
                                                  array<I>* row_nnz)                                \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_count_in_span_##TN##_##IN(                                               \
            stream_of(exec), static_cast<int64_t>(row_span.length()),                               \
            static_cast<int64_t>(row_span.begin), static_cast<int64_t>(col_span.begin),             \
            static_cast<int64_t>(col_span.end), source->get_const_row_ptrs(),                       \
            source->get_const_col_idxs(), row_nnz->get_data()));                                    \
    }                                                                                               \
    template <>                                                                                     \
    void compute_submatrix<T, I>(exec_t exec, const matrix::Csr<T, I>* source, gko::span row_span,  \
                                 gko::span col_span, matrix::Csr<T, I>* result)                     \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_submatrix_##TN##_##IN(                                                   \
            stream_of(exec), static_cast<int64_t>(row_span.length()),                               \
            static_cast<int64_t>(row_span.begin), static_cast<int64_t>(col_span.begin),             \
            static_cast<int64_t>(col_span.end), source->get_const_row_ptrs(),                       \
            source->get_const_col_idxs(), source->get_const_values(), result->get_const_row_ptrs(), \
            result->get_col_idxs(), result->get_values()));                                         \
    }
FOR_AVT_IT(DEF)
#undef DEF
#undef PERMUTE

}  // namespace csr


namespace csr {

// C = alpha A B + beta D and C = alpha A + beta B through triplets (csrc/conversions.hip): expand the
// contributions row by row, sort them (stable) by (row, column), add up the runs, rows -> pointers
// x *= alpha / x /= alpha over n contiguous values (one scalar), per value type
inline int scale_values(gkoc_stream_t s, int64_t n, const double* a, double* x, bool inv)
{
    return inv ? gkoc_dense_inv_scale_f64(s, n, 1, a, 1, x, 1) : gkoc_dense_scale_f64(s, n, 1, a, 1, x, 1);
}
inline int scale_values(gkoc_stream_t s, int64_t n, const float* a, float* x, bool inv)
{
    return inv ? gkoc_dense_inv_scale_f32(s, n, 1, a, 1, x, 1) : gkoc_dense_scale_f32(s, n, 1, a, 1, x, 1);
}
inline int scale_values(gkoc_stream_t s, int64_t n, const std::complex<double>* a, std::complex<double>* x, bool inv)
{
    return inv ? gkoc_cdense_inv_scale_c128(s, n, 1, a, 1, 0, x, 1) : gkoc_cdense_scale_c128(s, n, 1, a, 1, 0, x, 1);
}
inline int scale_values(gkoc_stream_t s, int64_t n, const std::complex<float>* a, std::complex<float>* x, bool inv)
{
    return inv ? gkoc_cdense_inv_scale_c64(s, n, 1, a, 1, 0, x, 1) : gkoc_cdense_scale_c64(s, n, 1, a, 1, 0, x, 1);
}

#define DEF(T, TN, I, IN)                                                                           \
    static void from_contributions_##TN##_##IN(                                                     \
        exec_t exec, const T* alpha, const matrix::Csr<T, I>* a, const matrix::Csr<T, I>* b,        \
        const T* beta, const matrix::Csr<T, I>* d, matrix::Csr<T, I>* c)                            \
    {                                                                                               \
        const auto s = stream_of(exec);                                                             \
        const auto n = static_cast<int64_t>(a->get_size()[0]);                                      \
        array<int64> offsets{exec, static_cast<size_type>(n + 1)};                                  \
        int64_t total = 0;                                                                          \
        GKOC_CALL(gkoc_csr_spgemm_count_##IN(s, n, a->get_const_row_ptrs(), a->get_const_col_idxs(), \
                                             b ? b->get_const_row_ptrs() : nullptr,                 \
                                             d ? d->get_const_row_ptrs() : nullptr,                 \
                                             offsets.get_data(), &total));                          \
        const auto nt = static_cast<size_type>(total);                                              \
        array<I> t_rows{exec, nt}, t_cols{exec, nt};                                                \
        array<T> t_vals{exec, nt};                                                                  \
        GKOC_CALL(gkoc_csr_spgemm_expand_##TN##_##IN(                                               \
            s, n, alpha, a->get_const_row_ptrs(), a->get_const_col_idxs(), a->get_const_values(),   \
            b ? b->get_const_row_ptrs() : nullptr, b ? b->get_const_col_idxs() : nullptr,           \
            b ? b->get_const_values() : nullptr, beta, d ? d->get_const_row_ptrs() : nullptr,       \
            d ? d->get_const_col_idxs() : nullptr, d ? d->get_const_values() : nullptr,             \
            offsets.get_const_data(), t_rows.get_data(), t_cols.get_data(), t_vals.get_data()));    \
        array<char> sort_work{exec, gkoc_sort_row_major_workspace_bytes(total, sizeof(T), sizeof(I))}; \
        GKOC_CALL(gkoc_sort_row_major_##TN##_##IN(s, total, t_rows.get_data(), t_cols.get_data(),   \
                                                  t_vals.get_data(), sort_work.get_data(),          \
                                                  sort_work.get_size()));                           \
        array<char> work{exec, gkoc_compact_workspace_bytes(total)};                                \
        int64_t kept = 0;                                                                           \
        GKOC_CALL(gkoc_sum_duplicates_count_##IN(s, total, t_rows.get_const_data(),                 \
                                                 t_cols.get_const_data(), work.get_data(),          \
                                                 work.get_size(), &kept));                          \
        matrix::CsrBuilder<T, I> builder{c};                                                        \
        builder.get_col_idx_array().resize_and_reset(static_cast<size_type>(kept));                 \
        builder.get_value_array().resize_and_reset(static_cast<size_type>(kept));                   \
        array<I> c_rows{exec, static_cast<size_type>(kept)};                                        \
        GKOC_CALL(gkoc_sum_duplicates_fill_##TN##_##IN(                                             \
            s, total, t_rows.get_const_data(), t_cols.get_const_data(), t_vals.get_const_data(),    \
            work.get_const_data(), c_rows.get_data(), builder.get_col_idx_array().get_data(),       \
            builder.get_value_array().get_data()));                                                 \
        GKOC_CALL(gkoc_convert_idxs_to_ptrs_##IN(s, kept, c_rows.get_const_data(), n,               \
                                                 c->get_row_ptrs()));                               \
        exec->synchronize(); /* the temporaries are released on return */                           \
    }                                                                                               \
    /* csr::scale / inv_scale (reference/matrix/csr_kernels.cpp:1342-1370): every stored value */   \
    template <>                                                                                     \
    void scale<T, I>(exec_t exec, const matrix::Dense<T>* alpha, matrix::Csr<T, I>* to_scale)       \
    {                                                                                               \
        GKOC_CALL(scale_values(stream_of(exec), static_cast<int64_t>(to_scale->get_num_stored_elements()), \
                               alpha->get_const_values(), to_scale->get_values(), false));          \
    }                                                                                               \
    template <>                                                                                     \
    void inv_scale<T, I>(exec_t exec, const matrix::Dense<T>* alpha, matrix::Csr<T, I>* to_scale)   \
    {                                                                                               \
        GKOC_CALL(scale_values(stream_of(exec), static_cast<int64_t>(to_scale->get_num_stored_elements()), \
                               alpha->get_const_values(), to_scale->get_values(), true));           \
    }                                                                                               \
    template <>                                                                                     \
    void spgemm<T, I>(exec_t exec, const matrix::Csr<T, I>* a, const matrix::Csr<T, I>* b,          \
                      matrix::Csr<T, I>* c)                                                         \
    {                                                                                               \
        from_contributions_##TN##_##IN(exec, nullptr, a, b, nullptr, nullptr, c);                   \
    }                                                                                               \
    template <>                                                                                     \
    void advanced_spgemm<T, I>(exec_t exec, const matrix::Dense<T>* alpha,                          \
                               const matrix::Csr<T, I>* a, const matrix::Csr<T, I>* b,              \
                               const matrix::Dense<T>* beta, const matrix::Csr<T, I>* d,            \
                               matrix::Csr<T, I>* c)                                                \
    {                                                                                               \
        from_contributions_##TN##_##IN(exec, alpha->get_const_values(), a, b,                       \
                                       beta->get_const_values(), d, c);                             \
    }                                                                                               \
    template <>                                                                                     \
    void spgeam<T, I>(exec_t exec, const matrix::Dense<T>* alpha, const matrix::Csr<T, I>* a,       \
                      const matrix::Dense<T>* beta, const matrix::Csr<T, I>* b,                     \
                      matrix::Csr<T, I>* c)                                                         \
    {                                                                                               \
        /* alpha A's entries come first ("D"), then beta B's: 0 + alpha a + beta b */               \
        from_contributions_##TN##_##IN(exec, beta->get_const_values(), b, nullptr,                  \
                                       alpha->get_const_values(), a, c);                            \
    }                                                                                               \
    /* the values of a product / sum whose pattern exists (Csr::multiply_reuse etc.): positions     \
     * by bisection of C's sorted rows, Ginkgo's lookup structures are not used */                  \
    template <>                                                                                     \
    void spgemm_reuse<T, I>(exec_t exec, const matrix::Csr<T, I>* a, const matrix::Csr<T, I>* b,    \
                            const matrix::csr::lookup_data<I>&, matrix::Csr<T, I>* c)               \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_spgemm_reuse_##TN##_##IN(                                                \
            stream_of(exec), static_cast<int64_t>(c->get_size()[0]), a->get_const_row_ptrs(),       \
            a->get_const_col_idxs(), a->get_const_values(), b->get_const_row_ptrs(),                \
            b->get_const_col_idxs(), b->get_const_values(), nullptr, nullptr, nullptr, nullptr,     \
            nullptr, c->get_const_row_ptrs(), c->get_const_col_idxs(), c->get_values()));           \
    }                                                                                               \
    template <>                                                                                     \
    void advanced_spgemm_reuse<T, I>(exec_t exec, const matrix::Dense<T>* alpha,                    \
                                     const matrix::Csr<T, I>* a, const matrix::Csr<T, I>* b,        \
                                     const matrix::Dense<T>* beta, const matrix::Csr<T, I>* d,      \
                                     const matrix::csr::lookup_data<I>&, matrix::Csr<T, I>* c)      \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_spgemm_reuse_##TN##_##IN(                                                \
            stream_of(exec), static_cast<int64_t>(c->get_size()[0]), a->get_const_row_ptrs(),       \
            a->get_const_col_idxs(), a->get_const_values(), b->get_const_row_ptrs(),                \
            b->get_const_col_idxs(), b->get_const_values(), alpha->get_const_values(),              \
            beta->get_const_values(), d->get_const_row_ptrs(), d->get_const_col_idxs(),             \
            d->get_const_values(), c->get_const_row_ptrs(), c->get_const_col_idxs(),                \
            c->get_values()));                                                                      \
    }                                                                                               \
    template <>                                                                                     \
    void spgeam_numeric<T, I>(exec_t exec, const matrix::Dense<T>* alpha,                           \
                              const matrix::Csr<T, I>* a, const matrix::Dense<T>* beta,             \
                              const matrix::Csr<T, I>* b, matrix::Csr<T, I>* c)                     \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_spgeam_numeric_##TN##_##IN(                                              \
            stream_of(exec), static_cast<int64_t>(c->get_size()[0]), alpha->get_const_values(),     \
            a->get_const_row_ptrs(), a->get_const_col_idxs(), a->get_const_values(),                \
            beta->get_const_values(), b->get_const_row_ptrs(), b->get_const_col_idxs(),             \
            b->get_const_values(), c->get_const_row_ptrs(), c->get_values()));                      \
    }
FOR_AVT_IT(DEF)
#undef DEF

// Ginkgo's per-row lookup structures (core/matrix/csr_lookup.hpp), built on the device with the
// reference's bits (csrc/csr_lookup.hip).  This backend's own spgemm_reuse / spgeam kernels find
// positions by bisection and do not read them; Ginkgo's factorisations would (out of scope, stubs).
#define DEF(I, IN)                                                                                  \
    template <>                                                                                     \
    void build_lookup_offsets<I>(exec_t exec, const I* row_ptrs, const I* col_idxs,                 \
                                 size_type num_rows, matrix::csr::sparsity_type allowed,            \
                                 I* storage_offsets)                                                \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_build_lookup_offsets_##IN(stream_of(exec), static_cast<int64_t>(num_rows), \
                                                     row_ptrs, col_idxs, static_cast<int>(allowed), \
                                                     storage_offsets));                             \
    }                                                                                               \
    template <>                                                                                     \
    void build_lookup<I>(exec_t exec, const I* row_ptrs, const I* col_idxs, size_type num_rows,     \
                         matrix::csr::sparsity_type allowed, const I* storage_offsets,              \
                         int64* row_desc, int32* storage)                                           \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_build_lookup_##IN(stream_of(exec), static_cast<int64_t>(num_rows),       \
                                             row_ptrs, col_idxs, static_cast<int>(allowed),         \
                                             storage_offsets, reinterpret_cast<int64_t*>(row_desc), \
                                             reinterpret_cast<int32_t*>(storage)));                 \
    }
DEF(int32, i32)
DEF(int64, i64)
#undef DEF

}  // namespace csr


namespace coo {

// conj_array for real value types: nothing to do
#define DEF(T, TN)                                                    \
    template <>                                                       \
    void conj_array<T>(exec_t exec, size_type, T*)                    \
    {                                                                 \
        cdna4::launch_deferred();                                     \
    }
FOR_VT(DEF)
#undef DEF

}  // namespace coo


namespace jacobi {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void scalar_l1<T, I>(exec_t exec, const matrix::Csr<T, I>* csr, matrix::Diagonal<T>* diag)      \
    {                                                                                               \
        GKOC_CALL(gkoc_jacobi_scalar_l1_##TN##_##IN(                                                \
            stream_of(exec), static_cast<int64_t>(csr->get_size()[0]), csr->get_const_row_ptrs(),   \
            csr->get_const_col_idxs(), csr->get_const_values(), diag->get_values()));               \
    }                                                                                               \
    template <>                                                                                     \
    void block_l1<T, I>(exec_t exec, size_type num_blocks, const array<I>& block_pointers,          \
                        matrix::Csr<T, I>* csr)                                                     \
    {                                                                                               \
        GKOC_CALL(gkoc_jacobi_block_l1_##TN##_##IN(                                                 \
            stream_of(exec), static_cast<int64_t>(num_blocks), block_pointers.get_const_data(),     \
            csr->get_const_row_ptrs(), csr->get_const_col_idxs(), csr->get_values()));              \
    }
FOR_AVT_IT(DEF)
#undef DEF

}  // namespace jacobi


namespace factorization {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void add_diagonal_elements<T, I>(exec_t exec, matrix::Csr<T, I>* mtx, bool)                     \
    {                                                                                               \
        const auto s = stream_of(exec);                                                             \
        const auto n = static_cast<int64_t>(mtx->get_size()[0]);                                    \
        array<I> shift{exec, static_cast<size_type>(n + 1)};                                        \
        int64_t missing = 0;                                                                        \
        GKOC_CALL(gkoc_csr_missing_diagonal_shift_##IN(                                             \
            s, n, static_cast<int64_t>(mtx->get_size()[1]), mtx->get_const_row_ptrs(),              \
            mtx->get_const_col_idxs(), shift.get_data(), &missing));                                \
        if (missing == 0) return;                                                                   \
        const auto new_nnz = mtx->get_num_stored_elements() + static_cast<size_type>(missing);      \
        array<T> new_values{exec, new_nnz};                                                         \
        array<I> new_cols{exec, new_nnz};                                                           \
        array<I> new_ptrs{exec, static_cast<size_type>(n + 1)};                                     \
        GKOC_CALL(gkoc_csr_add_diagonal_fill_##TN##_##IN(                                           \
            s, n, mtx->get_const_row_ptrs(), mtx->get_const_col_idxs(), mtx->get_const_values(),    \
            shift.get_const_data(), new_ptrs.get_data(), new_cols.get_data(),                       \
            new_values.get_data()));                                                                \
        exec->copy(static_cast<size_type>(n + 1), new_ptrs.get_const_data(), mtx->get_row_ptrs());  \
        matrix::CsrBuilder<T, I> builder{mtx};                                                      \
        builder.get_value_array() = std::move(new_values);                                          \
        builder.get_col_idx_array() = std::move(new_cols);                                          \
    }
FOR_AVT_IT(DEF)
#undef DEF

}  // namespace factorization


namespace permutation {

#define DEF(I, IN)                                                                                  \
    template <>                                                                                     \
    void invert<I>(exec_t exec, const I* perm, size_type size, I* out)                              \
    {                                                                                               \
        GKOC_CALL(gkoc_permutation_invert_##IN(stream_of(exec), static_cast<int64_t>(size), perm,   \
                                               out));                                               \
    }                                                                                               \
    template <>                                                                                     \
    void compose<I>(exec_t exec, const I* first, const I* second, size_type size, I* out)           \
    {                                                                                               \
        GKOC_CALL(gkoc_permutation_compose_##IN(stream_of(exec), static_cast<int64_t>(size), first, \
                                                second, out));                                      \
    }
DEF(int32, i32)
DEF(int64, i64)
#undef DEF

}  // namespace permutation


namespace scaled_permutation {

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void invert<T, I>(exec_t exec, const T* in_scale, const I* in_perm, size_type size,             \
                      T* out_scale, I* out_perm)                                                    \
    {                                                                                               \
        GKOC_CALL(gkoc_scaled_permutation_invert_##TN##_##IN(                                       \
            stream_of(exec), static_cast<int64_t>(size), in_scale, in_perm, out_scale, out_perm));  \
    }                                                                                               \
    template <>                                                                                     \
    void compose<T, I>(exec_t exec, const T* first_scale, const I* first, const T* second_scale,    \
                       const I* second, size_type size, T* out_scale, I* out_perm)                  \
    {                                                                                               \
        GKOC_CALL(gkoc_scaled_permutation_compose_##TN##_##IN(                                      \
            stream_of(exec), static_cast<int64_t>(size), first_scale, first, second_scale, second,  \
            out_scale, out_perm));                                                                  \
    }
FOR_AVT_IT(DEF)
#undef DEF

}  // namespace scaled_permutation



// csr::calculate_nonzeros_per_row_in_index_set / compute_submatrix_from_index_set
// (Csr::create_submatrix(index_set, index_set), core/matrix/csr.cpp:1446-1486) for all four value types
namespace csr {

namespace {
inline const double* abi(const double* p) { return p; }
inline double* abi(double* p) { return p; }
inline const float* abi(const float* p) { return p; }
inline float* abi(float* p) { return p; }
inline const gkoc_c128* abi(const std::complex<double>* p) { return reinterpret_cast<const gkoc_c128*>(p); }
inline gkoc_c128* abi(std::complex<double>* p) { return reinterpret_cast<gkoc_c128*>(p); }
inline const gkoc_c64* abi(const std::complex<float>* p) { return reinterpret_cast<const gkoc_c64*>(p); }
inline gkoc_c64* abi(std::complex<float>* p) { return reinterpret_cast<gkoc_c64*>(p); }
}  // namespace

#define DEF(T, TN, I, IN)                                                                           \
    template <>                                                                                     \
    void calculate_nonzeros_per_row_in_index_set<T, I>(                                             \
        exec_t exec, const matrix::Csr<T, I>* source, const gko::index_set<I>& row_index_set,       \
        const gko::index_set<I>& col_index_set, I* row_nnz)                                         \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_count_in_index_set_##TN##_##IN(                                          \
            stream_of(exec), static_cast<int64_t>(row_index_set.get_num_elems()),                   \
            static_cast<int64_t>(row_index_set.get_num_subsets()), row_index_set.get_subsets_begin(), \
            row_index_set.get_superset_indices(), static_cast<int64_t>(col_index_set.get_num_subsets()), \
            col_index_set.get_subsets_begin(), col_index_set.get_subsets_end(),                     \
            static_cast<int64_t>(col_index_set.get_size()), source->get_const_row_ptrs(),           \
            source->get_const_col_idxs(), row_nnz));                                                \
    }                                                                                               \
    template <>                                                                                     \
    void compute_submatrix_from_index_set<T, I>(                                                    \
        exec_t exec, const matrix::Csr<T, I>* source, const gko::index_set<I>& row_index_set,       \
        const gko::index_set<I>& col_index_set, matrix::Csr<T, I>* result)                          \
    {                                                                                               \
        GKOC_CALL(gkoc_csr_submatrix_from_index_set_##TN##_##IN(                                    \
            stream_of(exec), static_cast<int64_t>(row_index_set.get_num_elems()),                   \
            static_cast<int64_t>(row_index_set.get_num_subsets()), row_index_set.get_subsets_begin(), \
            row_index_set.get_superset_indices(), static_cast<int64_t>(col_index_set.get_num_subsets()), \
            col_index_set.get_subsets_begin(), col_index_set.get_subsets_end(),                     \
            col_index_set.get_superset_indices(), static_cast<int64_t>(col_index_set.get_size()),   \
            source->get_const_row_ptrs(), source->get_const_col_idxs(), abi(source->get_const_values()), \
            result->get_const_row_ptrs(), result->get_col_idxs(), abi(result->get_values())));      \
    }
DEF(double, f64, int32, i32)
DEF(double, f64, int64, i64)
DEF(float, f32, int32, i32)
DEF(float, f32, int64, i64)
DEF(std::complex<double>, c128, int32, i32)
DEF(std::complex<double>, c128, int64, i64)
DEF(std::complex<float>, c64, int32, i32)
DEF(std::complex<float>, c64, int64, i64)
#undef DEF

}  // namespace csr

}  // namespace hip
}  // namespace kernels
}  // namespace gko
