// gko::kernels::hip::{diagonal, sparsity_csr, components::{reduce_add_array, convert_precision},
// jacobi::scalar_conj}: the kernels Ginkgo's generic LinOp and solver test-suites reach next to the
// hot path (test/matrix/matrix.cpp, test/solver/solver.cpp, test/matrix/{diagonal,sparsity_csr}_
// kernels.cpp, test/components/*).  Real value types; complex instantiations stay on Ginkgo's
// NotCompiled stubs.  SparsityCsr has no values of its own: its product, transpose and sort run
// through the Csr kernels on a temporary value array filled with the matrix' one value (the
// products value * b are then the reference's, term by term).
#include <ginkgo/core/base/device_matrix_data.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/diagonal.hpp>
#include <ginkgo/core/matrix/sparsity_csr.hpp>

#include "core/components/precision_conversion_kernels.hpp"
#include "core/components/reduce_array_kernels.hpp"
#include "core/matrix/diagonal_kernels.hpp"
#include "core/matrix/sparsity_csr_kernels.hpp"
#include "core/preconditioner/jacobi_kernels.hpp"
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
#define FOR_VT_IT(M)                                                     \
    M(double, f64, int32, i32) M(double, f64, int64, i64) M(float, f32, int32, i32) \
        M(float, f32, int64, i64)


// ===================================================================== diagonal
namespace diagonal {

#define DEF(T, TN)                                                                                 \
    template <>                                                                                    \
    void apply_to_dense<T>(exec_t exec, const matrix::Diagonal<T>* a, const matrix::Dense<T>* b,   \
                           matrix::Dense<T>* c, bool inverse)                                      \
    {                                                                                              \
        GKOC_CALL(gkoc_diagonal_apply_to_dense_##TN(stream_of(exec), rows(b), cols(b),             \
                                                    a->get_const_values(), b->get_const_values(),  \
                                                    ld(b), c->get_values(), ld(c), inverse));      \
    }                                                                                              \
    template <>                                                                                    \
    void right_apply_to_dense<T>(exec_t exec, const matrix::Diagonal<T>* a,                        \
                                 const matrix::Dense<T>* b, matrix::Dense<T>* c)                   \
    {                                                                                              \
        GKOC_CALL(gkoc_diagonal_right_apply_to_dense_##TN(                                         \
            stream_of(exec), rows(b), static_cast<int64_t>(a->get_size()[1]),                      \
            a->get_const_values(), b->get_const_values(), ld(b), c->get_values(), ld(c)));         \
    }                                                                                              \
    template <>                                                                                    \
    void conj_transpose<T>(exec_t exec, const matrix::Diagonal<T>* orig,                           \
                           matrix::Diagonal<T>* trans)                                             \
    {                                                                                              \
        GKOC_CALL(gkoc_memcpy_d2d(trans->get_values(), orig->get_const_values(),                   \
                                  sizeof(T) * orig->get_size()[0], stream_of(exec)));              \
    }
FOR_VT(DEF)
#undef DEF

#define DEF(T, TN, I, IN)                                                                          \
    template <>                                                                                    \
    void apply_to_csr<T, I>(exec_t exec, const matrix::Diagonal<T>* a, const matrix::Csr<T, I>* b, \
                            matrix::Csr<T, I>* c, bool inverse)                                    \
    {                                                                                              \
        c->copy_from(b);                                                                           \
        GKOC_CALL(gkoc_diagonal_apply_to_csr_##TN##_##IN(                                          \
            stream_of(exec), static_cast<int64_t>(c->get_size()[0]), a->get_const_values(),        \
            c->get_const_row_ptrs(), c->get_values(), inverse));                                   \
    }                                                                                              \
    template <>                                                                                    \
    void right_apply_to_csr<T, I>(exec_t exec, const matrix::Diagonal<T>* a,                       \
                                  const matrix::Csr<T, I>* b, matrix::Csr<T, I>* c)                \
    {                                                                                              \
        c->copy_from(b);                                                                           \
        GKOC_CALL(gkoc_diagonal_right_apply_to_csr_##TN##_##IN(                                    \
            stream_of(exec), static_cast<int64_t>(c->get_num_stored_elements()),                   \
            a->get_const_values(), c->get_const_col_idxs(), c->get_values()));                     \
    }                                                                                              \
    template <>                                                                                    \
    void fill_in_matrix_data<T, I>(exec_t exec, const device_matrix_data<T, I>& data,              \
                                   matrix::Diagonal<T>* output)                                    \
    {                                                                                              \
        GKOC_CALL(gkoc_diagonal_fill_in_matrix_data_##TN##_##IN(                                   \
            stream_of(exec), static_cast<int64_t>(data.get_num_stored_elements()),                 \
            data.get_const_row_idxs(), data.get_const_col_idxs(), data.get_const_values(),         \
            output->get_values()));                                                                \
    }                                                                                              \
    template <>                                                                                    \
    void convert_to_csr<T, I>(exec_t exec, const matrix::Diagonal<T>* source,                      \
                              matrix::Csr<T, I>* result)                                           \
    {                                                                                              \
        GKOC_CALL(gkoc_diagonal_convert_to_csr_##TN##_##IN(                                        \
            stream_of(exec), static_cast<int64_t>(source->get_size()[0]),                          \
            source->get_const_values(), result->get_row_ptrs(), result->get_col_idxs(),            \
            result->get_values()));                                                                \
    }
FOR_VT_IT(DEF)
#undef DEF

}  // namespace diagonal


// ===================================================================== sparsity_csr
namespace sparsity_csr {

namespace {
// the one value of a SparsityCsr repeated nnz times: what the Csr kernels multiply with
template <typename T>
array<T> values_of(exec_t exec, const T* value, size_type nnz)
{
    array<T> vals{exec, nnz};
    T v{};
    exec->get_master()->copy_from(exec.get(), 1, value, &v);
    vals.fill(v);
    return vals;
}
}  // namespace

#define DEF(T, TN, I, IN)                                                                          \
    template <>                                                                                    \
    void spmv<T, T, T, I>(exec_t exec, const matrix::SparsityCsr<T, I>* a,                         \
                          const matrix::Dense<T>* b, matrix::Dense<T>* c)                          \
    {                                                                                              \
        auto vals = values_of<T>(exec, a->get_const_value(), a->get_num_nonzeros());               \
        GKOC_CALL(gkoc_csr_spmv_##TN##_##IN(                                                       \
            stream_of(exec), a->get_size()[0], a->get_size()[1], a->get_const_row_ptrs(),          \
            a->get_const_col_idxs(), vals.get_const_data(), b->get_const_values(), ld(b),          \
            c->get_values(), ld(c), cols(c)));                                                     \
        exec->synchronize(); /* vals is released on return */                                      \
    }                                                                                              \
    template <>                                                                                    \
    void advanced_spmv<T, T, T, I>(exec_t exec, const matrix::Dense<T>* alpha,                     \
                                   const matrix::SparsityCsr<T, I>* a, const matrix::Dense<T>* b,  \
                                   const matrix::Dense<T>* beta, matrix::Dense<T>* c)              \
    {                                                                                              \
        auto vals = values_of<T>(exec, a->get_const_value(), a->get_num_nonzeros());               \
        GKOC_CALL(gkoc_csr_advanced_spmv_##TN##_##IN(                                              \
            stream_of(exec), a->get_size()[0], a->get_size()[1], alpha->get_const_values(),        \
            a->get_const_row_ptrs(), a->get_const_col_idxs(), vals.get_const_data(),               \
            b->get_const_values(), ld(b), beta->get_const_values(), c->get_values(), ld(c),        \
            cols(c)));                                                                             \
        exec->synchronize();                                                                       \
    }                                                                                              \
    template <>                                                                                    \
    void fill_in_dense<T, I>(exec_t exec, const matrix::SparsityCsr<T, I>* input,                  \
                             matrix::Dense<T>* output)                                             \
    {                                                                                              \
        auto vals = values_of<T>(exec, input->get_const_value(), input->get_num_nonzeros());       \
        GKOC_CALL(gkoc_csr_fill_in_dense_##TN##_##IN(                                              \
            stream_of(exec), static_cast<int64_t>(input->get_size()[0]),                           \
            input->get_const_row_ptrs(), input->get_const_col_idxs(), vals.get_const_data(),       \
            output->get_values(), ld(output)));                                                    \
        exec->synchronize();                                                                       \
    }                                                                                              \
    template <>                                                                                    \
    void diagonal_element_prefix_sum<T, I>(exec_t exec, const matrix::SparsityCsr<T, I>* matrix,   \
                                           I* prefix_sum)                                          \
    {                                                                                              \
        const auto n = static_cast<int64_t>(matrix->get_size()[0]);                                \
        GKOC_CALL(gkoc_sparsity_csr_count_diagonal_##IN(stream_of(exec), n,                        \
                                                        matrix->get_const_row_ptrs(),              \
                                                        matrix->get_const_col_idxs(), prefix_sum)); \
        GKOC_CALL(gkoc_prefix_sum_nonnegative_##IN(stream_of(exec), prefix_sum, n + 1));           \
    }                                                                                              \
    template <>                                                                                    \
    void remove_diagonal_elements<T, I>(exec_t exec, const I* row_ptrs, const I* col_idxs,         \
                                        const I* diag_prefix_sum,                                  \
                                        matrix::SparsityCsr<T, I>* matrix)                         \
    {                                                                                              \
        GKOC_CALL(gkoc_sparsity_csr_remove_diagonal_##IN(                                          \
            stream_of(exec), static_cast<int64_t>(matrix->get_size()[0]), row_ptrs, col_idxs,      \
            diag_prefix_sum, matrix->get_row_ptrs(), matrix->get_col_idxs()));                     \
    }                                                                                              \
    template <>                                                                                    \
    void transpose<T, I>(exec_t exec, const matrix::SparsityCsr<T, I>* orig,                       \
                         matrix::SparsityCsr<T, I>* trans)                                         \
    {                                                                                              \
        const int64_t nnz = orig->get_num_nonzeros();                                              \
        array<T> vin{exec, static_cast<size_type>(nnz)}, vout{exec, static_cast<size_type>(nnz)};  \
        vin.fill(T{1});                                                                            \
        array<char> work(exec,                                                                     \
                         gkoc_csr_transpose_workspace_bytes(nnz, orig->get_size()[1], sizeof(I))); \
        GKOC_CALL(gkoc_csr_transpose_##TN##_##IN(                                                  \
            stream_of(exec), orig->get_size()[0], orig->get_size()[1], orig->get_const_row_ptrs(), \
            orig->get_const_col_idxs(), vin.get_const_data(), nnz, trans->get_row_ptrs(),          \
            trans->get_col_idxs(), vout.get_data(), work.get_data(), work.get_size()));            \
        exec->synchronize();                                                                       \
    }                                                                                              \
    template <>                                                                                    \
    void sort_by_column_index<T, I>(exec_t exec, matrix::SparsityCsr<T, I>* to_sort)               \
    {                                                                                              \
        array<T> vals{exec, to_sort->get_num_nonzeros()};                                          \
        vals.fill(T{1});                                                                           \
        GKOC_CALL(gkoc_csr_sort_by_column_index_##TN##_##IN(                                       \
            stream_of(exec), to_sort->get_size()[0], to_sort->get_const_row_ptrs(),                \
            to_sort->get_col_idxs(), vals.get_data()));                                            \
        exec->synchronize();                                                                       \
    }                                                                                              \
    template <>                                                                                    \
    void is_sorted_by_column_index<T, I>(exec_t exec, const matrix::SparsityCsr<T, I>* to_check,   \
                                         bool* is_sorted)                                          \
    {                                                                                              \
        int flag = 1;                                                                              \
        GKOC_CALL(gkoc_csr_is_sorted_by_column_index_##TN##_##IN(                                  \
            stream_of(exec), to_check->get_size()[0], to_check->get_const_row_ptrs(),              \
            to_check->get_const_col_idxs(), &flag));                                               \
        *is_sorted = flag != 0;                                                                    \
    }
FOR_VT_IT(DEF)
#undef DEF

}  // namespace sparsity_csr


// ===================================================================== components
namespace components {

#define DEF(T, TN)                                                                                 \
    template <>                                                                                    \
    void reduce_add_array<T>(exec_t exec, const array<T>& arr, array<T>& val)                      \
    {                                                                                              \
        GKOC_CALL(gkoc_reduce_add_array_##TN(stream_of(exec),                                      \
                                             static_cast<int64_t>(arr.get_size()),                 \
                                             arr.get_const_data(), val.get_data()));               \
    }
DEF(double, f64)
DEF(float, f32)
DEF(int32, i32)
DEF(int64, i64)
#undef DEF
template <>
void reduce_add_array<size_type>(exec_t exec, const array<size_type>& arr, array<size_type>& val)
{
    GKOC_CALL(gkoc_reduce_add_array_u64(stream_of(exec), static_cast<int64_t>(arr.get_size()),
                                        reinterpret_cast<const uint64_t*>(arr.get_const_data()),
                                        reinterpret_cast<uint64_t*>(val.get_data())));
}

template <>
void convert_precision<float, double>(exec_t exec, size_type size, const float* in, double* out)
{
    GKOC_CALL(gkoc_convert_precision_f32_f64(stream_of(exec), static_cast<int64_t>(size), in, out));
}
template <>
void convert_precision<double, float>(exec_t exec, size_type size, const double* in, float* out)
{
    GKOC_CALL(gkoc_convert_precision_f64_f32(stream_of(exec), static_cast<int64_t>(size), in, out));
}

}  // namespace components


// ===================================================================== jacobi::scalar_conj
namespace jacobi {

#define DEF(T, TN)                                                                                 \
    template <>                                                                                    \
    void scalar_conj<T>(exec_t exec, const array<T>& diag, array<T>& conj_diag)                    \
    {                                                                                              \
        GKOC_CALL(gkoc_memcpy_d2d(conj_diag.get_data(), diag.get_const_data(),                     \
                                  sizeof(T) * diag.get_size(), stream_of(exec)));                  \
    }
FOR_VT(DEF)
#undef DEF

}  // namespace jacobi


}  // namespace hip
}  // namespace kernels
}  // namespace gko
