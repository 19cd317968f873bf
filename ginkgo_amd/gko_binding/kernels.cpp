// gko::kernels::hip::* hot-path kernels forwarded to the C ABI.
//
// Each function is an explicit specialisation of the kernel template that
// Ginkgo core declares (core/**/*_kernels.hpp) for {double, float} x
// {int32, int64}; every other instantiation (complex, half, mixed precision)
// stays on Ginkgo's own weakened GKO_NOT_COMPILED stub.  See INTEGRATION.md.
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/diagonal.hpp>
#include <ginkgo/core/matrix/ell.hpp>
#include <ginkgo/core/matrix/sellp.hpp>
#include <ginkgo/core/preconditioner/jacobi.hpp>

#include <ginkgo/core/base/device_matrix_data.hpp>
#include <ginkgo/core/base/matrix_data.hpp>

#include "core/base/device_matrix_data_kernels.hpp"
#include "core/components/fill_array_kernels.hpp"
#include "core/components/format_conversion_kernels.hpp"
#include "core/components/prefix_sum_kernels.hpp"
#include "core/matrix/csr_kernels.hpp"
#include "core/matrix/dense_kernels.hpp"
#include "core/matrix/ell_kernels.hpp"
#include "core/matrix/sellp_kernels.hpp"
#include "core/preconditioner/jacobi_kernels.hpp"
#include "core/solver/cg_kernels.hpp"
#include "core/solver/common_gmres_kernels.hpp"
#include "core/solver/gmres_kernels.hpp"
#include "core/stop/criterion_kernels.hpp"
#include "core/stop/residual_norm_kernels.hpp"
#include <complex>

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

#define FOR_VT(M) M(double, f64) M(float, f32)
#define FOR_VT_IT(M)                                                     \
    M(double, f64, int32, i32) M(double, f64, int64, i64) M(float, f32, int32, i32) \
        M(float, f32, int64, i64)
#define FOR_IT(M) M(int32, i32) M(int64, i64)


// Ell / Sellp / Hybrid::read(device_matrix_data) hand over 64-bit row pointers; the
// converters take the matrix' own index type: narrowed into a temporary for int32 matrices
template <typename I>
struct ptrs_as {
    ptrs_as(exec_t exec, const int64* ptrs, size_type n);
    const I* get() const;
};
template <>
struct ptrs_as<int64> {
    ptrs_as(exec_t, const int64* ptrs, size_type) : p_{ptrs} {}
    const int64* get() const { return p_; }
    const int64* p_;
};
template <>
struct ptrs_as<int32> {
    ptrs_as(exec_t exec, const int64* ptrs, size_type n) : tmp_{exec, n}
    {
        GKOC_CALL(gkoc_narrow_i64_to_i32(stream_of(exec), static_cast<int64_t>(n), ptrs,
                                         tmp_.get_data()));
    }
    const int32* get() const { return tmp_.get_const_data(); }
    array<int32> tmp_;
};


// the allocator learns which arrays are vectors from what the kernels write (csrc/arena.hip,
// gkoc_arena_note_vector); one look-up per NEW output pointer of a thread
inline void note_written_vector(const void* p)
{
    static thread_local const void* last = nullptr;
    if (p != last) {
        last = p;
        gkoc_arena_note_vector(p);
    }
}

// ... and which arrays are matrix arrays from what the SpMV entries are handed (gkoc_arena_note_matrix)
inline void note_matrix_arrays(const void* values, const void* col_idxs)
{
    // the last few value arrays this thread has announced (a solver that alternates two matrices - a
    // multigrid cycle, a preconditioner with its own matrix - would otherwise take the allocator's mutex with
    // every product: ADVICE round 5)
    static thread_local const void* seen[8] = {};
    static thread_local int next = 0;
    for (const void* p : seen) {
        if (p == values) return;
    }
    seen[next] = values;
    next = (next + 1) % 8;
    gkoc_arena_note_matrix(values);
    gkoc_arena_note_matrix(col_idxs);
}

// ===================================================================== csr
namespace csr {

#define DEF(T, TN, I, IN)                                                       \
    template <>                                                                 \
    void spmv<T, T, T, I>(exec_t exec, const matrix::Csr<T, I>* a,              \
                          const matrix::Dense<T>* b, matrix::Dense<T>* c)       \
    {                                                                           \
        note_written_vector(c->get_const_values());                             \
        note_matrix_arrays(a->get_const_values(), a->get_const_col_idxs());     \
        const auto st_ = stream_of(exec);                                       \
        if (cols(c) == 1 && ld(b) == 1 && ld(c) == 1 &&                         \
            a->get_size()[0] == a->get_size()[1] &&                             \
            cdna4::spmv_with_dot(cdna4::vt_of<T>(), cdna4::it_of<I>(),          \
                                 exec->get_device_id(), st_, rows(c),           \
                                 a->get_const_row_ptrs(),                       \
                                 a->get_const_col_idxs(),                       \
                                 a->get_const_values(), b->get_const_values(),  \
                                 c->get_values())) {                            \
            return; /* the product and <b, c> for the dot product behind it */  \
        }                                                                       \
        GKOC_CALL(gkoc_csr_spmv_##TN##_##IN(                                    \
            st_, a->get_size()[0], a->get_size()[1],                            \
            a->get_const_row_ptrs(), a->get_const_col_idxs(),                   \
            a->get_const_values(), b->get_const_values(), ld(b),                \
            c->get_values(), ld(c), cols(c)));                                  \
    }                                                                           \
    template <>                                                                 \
    void advanced_spmv<T, T, T, I>(                                             \
        exec_t exec, const matrix::Dense<T>* alpha, const matrix::Csr<T, I>* a, \
        const matrix::Dense<T>* b, const matrix::Dense<T>* beta,                \
        matrix::Dense<T>* c)                                                    \
    {                                                                           \
        GKOC_CALL(gkoc_csr_advanced_spmv_##TN##_##IN(                           \
            stream_of(exec), a->get_size()[0], a->get_size()[1],                \
            alpha->get_const_values(), a->get_const_row_ptrs(),                 \
            a->get_const_col_idxs(), a->get_const_values(),                     \
            b->get_const_values(), ld(b), beta->get_const_values(),             \
            c->get_values(), ld(c), cols(c)));                                  \
    }                                                                           \
    template <>                                                                 \
    void extract_diagonal<T, I>(exec_t exec, const matrix::Csr<T, I>* orig,     \
                                matrix::Diagonal<T>* diag)                      \
    {                                                                           \
        GKOC_CALL(gkoc_csr_extract_diagonal_##TN##_##IN(                        \
            stream_of(exec), orig->get_size()[0], orig->get_size()[1],          \
            orig->get_const_row_ptrs(), orig->get_const_col_idxs(),             \
            orig->get_const_values(), diag->get_values()));                     \
    }                                                                           \
    template <>                                                                 \
    void is_sorted_by_column_index<T, I>(exec_t exec,                           \
                                         const matrix::Csr<T, I>* to_check,     \
                                         bool* is_sorted)                       \
    {                                                                           \
        int flag = 1;                                                           \
        GKOC_CALL(gkoc_csr_is_sorted_by_column_index_##TN##_##IN(               \
            stream_of(exec), to_check->get_size()[0],                           \
            to_check->get_const_row_ptrs(), to_check->get_const_col_idxs(),     \
            &flag));                                                            \
        *is_sorted = flag != 0;                                                 \
    }                                                                           \
    template <>                                                                 \
    void sort_by_column_index<T, I>(exec_t exec, matrix::Csr<T, I>* to_sort)    \
    {                                                                           \
        GKOC_CALL(gkoc_csr_sort_by_column_index_##TN##_##IN(                    \
            stream_of(exec), to_sort->get_size()[0],                            \
            to_sort->get_const_row_ptrs(), to_sort->get_col_idxs(),             \
            to_sort->get_values()));                                            \
    }                                                                           \
    template <>                                                                 \
    void convert_to_ell<T, I>(exec_t exec, const matrix::Csr<T, I>* source,     \
                              matrix::Ell<T, I>* result)                        \
    {                                                                           \
        GKOC_CALL(gkoc_csr_convert_to_ell_##TN##_##IN(                          \
            stream_of(exec), source->get_size()[0],                             \
            source->get_const_row_ptrs(), source->get_const_col_idxs(),         \
            source->get_const_values(),                                         \
            result->get_num_stored_elements_per_row(), result->get_stride(),    \
            result->get_col_idxs(), result->get_values()));                     \
    }                                                                           \
    template <>                                                                 \
    void convert_to_sellp<T, I>(exec_t exec, const matrix::Csr<T, I>* source,   \
                                matrix::Sellp<T, I>* result)                    \
    {                                                                           \
        GKOC_CALL(gkoc_csr_convert_to_sellp_##TN##_##IN(                        \
            stream_of(exec), source->get_size()[0], result->get_slice_size(),   \
            source->get_const_row_ptrs(), source->get_const_col_idxs(),         \
            source->get_const_values(),                                         \
            reinterpret_cast<const uint64_t*>(result->get_const_slice_sets()),  \
            result->get_col_idxs(), result->get_values()));                     \
    }
FOR_VT_IT(DEF)
#undef DEF

}  // namespace csr


// ===================================================================== ell
namespace ell {

#define DEF(T, TN, I, IN)                                                       \
    template <>                                                                 \
    void spmv<T, T, T, I>(exec_t exec, const matrix::Ell<T, I>* a,              \
                          const matrix::Dense<T>* b, matrix::Dense<T>* c)       \
    {                                                                           \
        note_written_vector(c->get_const_values());                             \
        note_matrix_arrays(a->get_const_values(), a->get_const_col_idxs());     \
        GKOC_CALL(gkoc_ell_spmv_##TN##_##IN(                                    \
            stream_of(exec), a->get_size()[0], a->get_size()[1],                \
            a->get_num_stored_elements_per_row(), a->get_stride(),              \
            a->get_const_col_idxs(), a->get_const_values(),                     \
            b->get_const_values(), ld(b), c->get_values(), ld(c), cols(c)));    \
    }                                                                           \
    template <>                                                                 \
    void advanced_spmv<T, T, T, I>(                                             \
        exec_t exec, const matrix::Dense<T>* alpha, const matrix::Ell<T, I>* a, \
        const matrix::Dense<T>* b, const matrix::Dense<T>* beta,                \
        matrix::Dense<T>* c)                                                    \
    {                                                                           \
        GKOC_CALL(gkoc_ell_advanced_spmv_##TN##_##IN(                           \
            stream_of(exec), a->get_size()[0], a->get_size()[1],                \
            a->get_num_stored_elements_per_row(), a->get_stride(),              \
            alpha->get_const_values(), a->get_const_col_idxs(),                 \
            a->get_const_values(), b->get_const_values(), ld(b),                \
            beta->get_const_values(), c->get_values(), ld(c), cols(c)));        \
    }
FOR_VT_IT(DEF)
#undef DEF

#define DEF(I, IN)                                                              \
    template <>                                                                 \
    void compute_max_row_nnz<I>(exec_t exec, const array<I>& row_ptrs,          \
                                size_type& max_nnz)                             \
    {                                                                           \
        int64_t m = 0;                                                          \
        GKOC_CALL(gkoc_compute_max_row_nnz_##IN(                                \
            stream_of(exec), static_cast<int64_t>(row_ptrs.get_size()) - 1,     \
            row_ptrs.get_const_data(), &m));                                    \
        max_nnz = static_cast<size_type>(m);                                    \
    }
FOR_IT(DEF)
#undef DEF

#define DEF(T, TN, I, IN)                                                       \
    template <>                                                                 \
    void fill_in_matrix_data<T, I>(exec_t exec,                                 \
                                   const device_matrix_data<T, I>& data,        \
                                   const int64* row_ptrs,                       \
                                   matrix::Ell<T, I>* output)                   \
    {                                                                           \
        const auto n = output->get_size()[0];                                   \
        ptrs_as<I> ptrs(exec, row_ptrs, n + 1);                                 \
        GKOC_CALL(gkoc_csr_convert_to_ell_##TN##_##IN(                          \
            stream_of(exec), n, ptrs.get(), data.get_const_col_idxs(),          \
            data.get_const_values(),                                            \
            output->get_num_stored_elements_per_row(), output->get_stride(),    \
            output->get_col_idxs(), output->get_values()));                     \
        exec->synchronize(); /* the temporary row pointers die here */          \
    }
FOR_VT_IT(DEF)
DEF(std::complex<double>, c128, int32, i32)
DEF(std::complex<double>, c128, int64, i64)
DEF(std::complex<float>, c64, int32, i32)
DEF(std::complex<float>, c64, int64, i64)
#undef DEF

}  // namespace ell


// =================================================================== sellp
namespace sellp {

#define DEF(T, TN, I, IN)                                                       \
    template <>                                                                 \
    void spmv<T, I>(exec_t exec, const matrix::Sellp<T, I>* a,                  \
                    const matrix::Dense<T>* b, matrix::Dense<T>* c)             \
    {                                                                           \
        note_written_vector(c->get_const_values());                             \
        note_matrix_arrays(a->get_const_values(), a->get_const_col_idxs());     \
        GKOC_CALL(gkoc_sellp_spmv_##TN##_##IN(                                  \
            stream_of(exec), a->get_size()[0], a->get_size()[1],                \
            a->get_slice_size(),                                                \
            reinterpret_cast<const uint64_t*>(a->get_const_slice_sets()),       \
            reinterpret_cast<const uint64_t*>(a->get_const_slice_lengths()),    \
            a->get_const_col_idxs(), a->get_const_values(),                     \
            b->get_const_values(), ld(b), c->get_values(), ld(c), cols(c)));    \
    }                                                                           \
    template <>                                                                 \
    void advanced_spmv<T, I>(exec_t exec, const matrix::Dense<T>* alpha,        \
                             const matrix::Sellp<T, I>* a,                      \
                             const matrix::Dense<T>* b,                         \
                             const matrix::Dense<T>* beta, matrix::Dense<T>* c) \
    {                                                                           \
        GKOC_CALL(gkoc_sellp_advanced_spmv_##TN##_##IN(                         \
            stream_of(exec), a->get_size()[0], a->get_size()[1],                \
            a->get_slice_size(), alpha->get_const_values(),                     \
            reinterpret_cast<const uint64_t*>(a->get_const_slice_sets()),       \
            reinterpret_cast<const uint64_t*>(a->get_const_slice_lengths()),    \
            a->get_const_col_idxs(), a->get_const_values(),                     \
            b->get_const_values(), ld(b), beta->get_const_values(),             \
            c->get_values(), ld(c), cols(c)));                                  \
    }
FOR_VT_IT(DEF)
#undef DEF

#define DEF(I, IN)                                                              \
    template <>                                                                 \
    void compute_slice_sets<I>(exec_t exec, const array<I>& row_ptrs,           \
                               size_type slice_size, size_type stride_factor,   \
                               size_type* slice_sets, size_type* slice_lengths) \
    {                                                                           \
        GKOC_CALL(gkoc_sellp_compute_slice_sets_##IN(                           \
            stream_of(exec), static_cast<int64_t>(row_ptrs.get_size()) - 1,     \
            slice_size, stride_factor, row_ptrs.get_const_data(),               \
            reinterpret_cast<uint64_t*>(slice_sets),                            \
            reinterpret_cast<uint64_t*>(slice_lengths)));                       \
    }
FOR_IT(DEF)
#undef DEF

#define DEF(T, TN, I, IN)                                                       \
    template <>                                                                 \
    void fill_in_matrix_data<T, I>(exec_t exec,                                 \
                                   const device_matrix_data<T, I>& data,        \
                                   const int64* row_ptrs,                       \
                                   matrix::Sellp<T, I>* output)                 \
    {                                                                           \
        const auto n = output->get_size()[0];                                   \
        ptrs_as<I> ptrs(exec, row_ptrs, n + 1);                                 \
        GKOC_CALL(gkoc_csr_convert_to_sellp_##TN##_##IN(                        \
            stream_of(exec), n, output->get_slice_size(), ptrs.get(),           \
            data.get_const_col_idxs(), data.get_const_values(),                 \
            reinterpret_cast<const uint64_t*>(output->get_const_slice_sets()),  \
            output->get_col_idxs(), output->get_values()));                     \
        exec->synchronize();                                                    \
    }
FOR_VT_IT(DEF)
DEF(std::complex<double>, c128, int32, i32)
DEF(std::complex<double>, c128, int64, i64)
DEF(std::complex<float>, c64, int32, i32)
DEF(std::complex<float>, c64, int64, i64)
#undef DEF

}  // namespace sellp


// =================================================================== dense
namespace dense {

template <typename T>
inline void* scratch(array<char>& tmp, int64_t n, int64_t nrhs, size_t& bytes)
{
    bytes = gkoc_reduction_workspace_bytes(n, nrhs, sizeof(T));
    if (tmp.get_size() < bytes) tmp.resize_and_reset(bytes);
    return tmp.get_data();
}

#define DEF(T, TN)                                                              \
    template <>                                                                 \
    void fill<T>(exec_t exec, matrix::Dense<T>* mat, T value)                   \
    {                                                                           \
        GKOC_CALL(gkoc_dense_fill_##TN(stream_of(exec), rows(mat), cols(mat),   \
                                       mat->get_values(), ld(mat), value));     \
    }                                                                           \
    template <>                                                                 \
    void copy<T, T>(exec_t exec, const matrix::Dense<T>* input,                 \
                    matrix::Dense<T>* output)                                   \
    {                                                                           \
        GKOC_CALL(gkoc_dense_copy_##TN(stream_of(exec), rows(input),            \
                                       cols(input), input->get_const_values(),  \
                                       ld(input), output->get_values(),         \
                                       ld(output)));                            \
    }                                                                           \
    template <>                                                                 \
    void inplace_absolute_dense<T>(exec_t exec, matrix::Dense<T>* source)       \
    {                                                                           \
        GKOC_CALL(gkoc_dense_absolute_##TN(                                     \
            stream_of(exec), rows(source), cols(source),                        \
            source->get_const_values(), ld(source), source->get_values(),       \
            ld(source)));                                                       \
    }                                                                           \
    template <>                                                                 \
    void outplace_absolute_dense<T>(exec_t exec, const matrix::Dense<T>* source, \
                                    matrix::Dense<T>* result)                   \
    {                                                                           \
        GKOC_CALL(gkoc_dense_absolute_##TN(                                     \
            stream_of(exec), rows(source), cols(source),                        \
            source->get_const_values(), ld(source), result->get_values(),       \
            ld(result)));                                                       \
    }                                                                           \
    template <>                                                                 \
    void simple_apply<T>(exec_t exec, const matrix::Dense<T>* a,                \
                         const matrix::Dense<T>* b, matrix::Dense<T>* c)        \
    {                                                                           \
        GKOC_CALL(gkoc_dense_simple_apply_##TN(                                 \
            stream_of(exec), rows(c), cols(c), cols(a), a->get_const_values(),  \
            ld(a), b->get_const_values(), ld(b), c->get_values(), ld(c)));      \
    }                                                                           \
    template <>                                                                 \
    void apply<T>(exec_t exec, const matrix::Dense<T>* alpha,                   \
                  const matrix::Dense<T>* a, const matrix::Dense<T>* b,         \
                  const matrix::Dense<T>* beta, matrix::Dense<T>* c)            \
    {                                                                           \
        GKOC_CALL(gkoc_dense_apply_##TN(                                        \
            stream_of(exec), rows(c), cols(c), cols(a),                         \
            alpha->get_const_values(), a->get_const_values(), ld(a),            \
            b->get_const_values(), ld(b), beta->get_const_values(),             \
            c->get_values(), ld(c)));                                           \
    }                                                                           \
    template <>                                                                 \
    void scale<T, T>(exec_t exec, const matrix::Dense<T>* alpha,                \
                     matrix::Dense<T>* x)                                       \
    {                                                                           \
        GKOC_CALL(gkoc_dense_scale_##TN(stream_of(exec), rows(x), cols(x),      \
                                        alpha->get_const_values(), cols(alpha), \
                                        x->get_values(), ld(x)));               \
    }                                                                           \
    template <>                                                                 \
    void inv_scale<T, T>(exec_t exec, const matrix::Dense<T>* alpha,            \
                         matrix::Dense<T>* x)                                   \
    {                                                                           \
        GKOC_CALL(gkoc_dense_inv_scale_##TN(                                    \
            stream_of(exec), rows(x), cols(x), alpha->get_const_values(),       \
            cols(alpha), x->get_values(), ld(x)));                              \
    }                                                                           \
    template <>                                                                 \
    void add_scaled<T, T>(exec_t exec, const matrix::Dense<T>* alpha,           \
                          const matrix::Dense<T>* x, matrix::Dense<T>* y)       \
    {                                                                           \
        GKOC_CALL(gkoc_dense_add_scaled_##TN(                                   \
            stream_of(exec), rows(x), cols(x), alpha->get_const_values(),       \
            cols(alpha), x->get_const_values(), ld(x), y->get_values(),         \
            ld(y)));                                                            \
    }                                                                           \
    template <>                                                                 \
    void sub_scaled<T, T>(exec_t exec, const matrix::Dense<T>* alpha,           \
                          const matrix::Dense<T>* x, matrix::Dense<T>* y)       \
    {                                                                           \
        if (cols(x) == 1 && cols(alpha) == 1 && ld(x) == 1 && ld(y) == 1 &&     \
            cdna4::hold_sub_scaled(cdna4::vt_of<T>(), stream_of(exec), rows(x), \
                                   alpha->get_const_values(),                   \
                                   x->get_const_values(), y->get_values())) {   \
            return; /* Gmres' modified Gram-Schmidt: fused with the next dot */ \
        }                                                                       \
        GKOC_CALL(gkoc_dense_sub_scaled_##TN(                                   \
            stream_of(exec), rows(x), cols(x), alpha->get_const_values(),       \
            cols(alpha), x->get_const_values(), ld(x), y->get_values(),         \
            ld(y)));                                                            \
    }                                                                           \
    template <>                                                                 \
    void compute_dot<T>(exec_t exec, const matrix::Dense<T>* x,                 \
                        const matrix::Dense<T>* y, matrix::Dense<T>* result,    \
                        array<char>& tmp)                                       \
    {                                                                           \
        if (cols(x) == 1 && ld(x) == 1 && ld(y) == 1 &&                         \
            cdna4::fused_dot(cdna4::vt_of<T>(),                                 \
                             cdna4::stream_keeping_deferred(exec), rows(x),     \
                             x->get_const_values(), y->get_const_values(),      \
                             result->get_values(), tmp)) {                      \
            return; /* step_2 + block-Jacobi + this dot in one launch, or the   \
                       value the block-Jacobi application left behind */        \
        }                                                                       \
        cdna4::launch_deferred_for_read(result->get_values());                  \
        size_t bytes = 0;                                                       \
        void* w = scratch<T>(tmp, rows(x), cols(x), bytes);                     \
        GKOC_CALL(gkoc_dense_compute_dot_##TN(                                  \
            cdna4::stream_keeping_deferred(exec), rows(x), cols(x),             \
            x->get_const_values(), ld(x), y->get_const_values(), ld(y),         \
            result->get_values(), w, bytes));                                   \
    }                                                                           \
    template <>                                                                 \
    void compute_dot_dispatch<T>(exec_t exec, const matrix::Dense<T>* x,        \
                                 const matrix::Dense<T>* y,                     \
                                 matrix::Dense<T>* result, array<char>& tmp)    \
    {                                                                           \
        compute_dot<T>(exec, x, y, result, tmp);                                \
    }                                                                           \
    template <>                                                                 \
    void compute_conj_dot<T>(exec_t exec, const matrix::Dense<T>* x,            \
                             const matrix::Dense<T>* y,                         \
                             matrix::Dense<T>* result, array<char>& tmp)        \
    {                                                                           \
        compute_dot<T>(exec, x, y, result, tmp);                                \
    }                                                                           \
    template <>                                                                 \
    void compute_conj_dot_dispatch<T>(exec_t exec, const matrix::Dense<T>* x,   \
                                      const matrix::Dense<T>* y,                \
                                      matrix::Dense<T>* result,                 \
                                      array<char>& tmp)                         \
    {                                                                           \
        compute_dot<T>(exec, x, y, result, tmp);                                \
    }                                                                           \
    template <>                                                                 \
    void compute_norm2<T>(exec_t exec, const matrix::Dense<T>* x,               \
                          matrix::Dense<T>* result, array<char>& tmp)           \
    {                                                                           \
        if (cols(x) == 1 &&                                                     \
            cdna4::cached_norm2(cdna4::vt_of<T>(),                              \
                                cdna4::stream_keeping_deferred(exec), rows(x),  \
                                x->get_const_values(), result->get_values())) { \
            return; /* computed by the fused launch that produced x */          \
        }                                                                       \
        cdna4::launch_deferred_for_read(result->get_values());                  \
        size_t bytes = 0;                                                       \
        void* w = scratch<T>(tmp, rows(x), cols(x), bytes);                     \
        GKOC_CALL(gkoc_dense_compute_norm2_##TN(                                \
            cdna4::stream_keeping_deferred(exec), rows(x), cols(x),             \
            x->get_const_values(), ld(x), result->get_values(), w, bytes));     \
    }                                                                           \
    template <>                                                                 \
    void compute_norm2_dispatch<T>(exec_t exec, const matrix::Dense<T>* x,      \
                                   matrix::Dense<T>* result, array<char>& tmp)  \
    {                                                                           \
        compute_norm2<T>(exec, x, result, tmp);                                 \
    }                                                                           \
    template <>                                                                 \
    void compute_squared_norm2<T>(exec_t exec, const matrix::Dense<T>* x,       \
                                  matrix::Dense<T>* result, array<char>& tmp)   \
    {                                                                           \
        size_t bytes = 0;                                                       \
        void* w = scratch<T>(tmp, rows(x), cols(x), bytes);                     \
        GKOC_CALL(gkoc_dense_compute_squared_norm2_##TN(                        \
            stream_of(exec), rows(x), cols(x), x->get_const_values(), ld(x),    \
            result->get_values(), w, bytes));                                   \
    }                                                                           \
    template <>                                                                 \
    void compute_sqrt<T>(exec_t exec, matrix::Dense<T>* data)                   \
    {                                                                           \
        /* element-wise sqrt of a (1 x nrhs) row in Ginkgo's use */             \
        for (int64_t r = 0; r < rows(data); ++r) {                              \
            GKOC_CALL(gkoc_dense_compute_sqrt_##TN(                             \
                stream_of(exec), cols(data), data->get_values() + r * ld(data))); \
        }                                                                       \
    }
FOR_VT(DEF)
#undef DEF

#define DEF(T, TN, I, IN)                                                       \
    template <>                                                                 \
    void fill_in_matrix_data<T, I>(exec_t exec,                                 \
                                   const device_matrix_data<T, I>& data,        \
                                   matrix::Dense<T>* output)                    \
    {                                                                           \
        GKOC_CALL(gkoc_dense_fill_in_matrix_data_##TN##_##IN(                   \
            stream_of(exec), data.get_num_stored_elements(),                    \
            data.get_const_row_idxs(), data.get_const_col_idxs(),               \
            data.get_const_values(), output->get_values(), ld(output)));        \
    }                                                                           \
    template <>                                                                 \
    void row_gather<T, T, I>(exec_t exec, const I* gather_indices,              \
                             const matrix::Dense<T>* orig,                      \
                             matrix::Dense<T>* row_collection)                  \
    {                                                                           \
        GKOC_CALL(gkoc_dense_row_gather_##TN##_##IN(                            \
            stream_of(exec), rows(row_collection), cols(orig), gather_indices,  \
            orig->get_const_values(), ld(orig), row_collection->get_values(),   \
            ld(row_collection)));                                               \
    }
FOR_VT_IT(DEF)
#undef DEF

// dense::copy between the two real precisions (Dense::convert_to, mixed-precision applies)
template <>
void copy<double, float>(exec_t exec, const matrix::Dense<double>* input,
                         matrix::Dense<float>* output)
{
    GKOC_CALL(gkoc_dense_convert_f64_f32(stream_of(exec), rows(input), cols(input),
                                         input->get_const_values(), ld(input),
                                         output->get_values(), ld(output)));
}
template <>
void copy<float, double>(exec_t exec, const matrix::Dense<float>* input,
                         matrix::Dense<double>* output)
{
    GKOC_CALL(gkoc_dense_convert_f32_f64(stream_of(exec), rows(input), cols(input),
                                         input->get_const_values(), ld(input),
                                         output->get_values(), ld(output)));
}

// complex Dense copies: a complex row-major matrix is a real one with twice the columns and
// twice the stride (the only complex kernels of this backend; they keep Dense::convert_to /
// clone of complex vectors working, e.g. for Ginkgo's cross-executor tests)
#define DEF(CI, RI, CO, RO, FN)                                                                  \
    template <>                                                                                  \
    void copy<CI, CO>(exec_t exec, const matrix::Dense<CI>* input, matrix::Dense<CO>* output)    \
    {                                                                                            \
        GKOC_CALL(FN(stream_of(exec), rows(input), 2 * cols(input),                              \
                     reinterpret_cast<const RI*>(input->get_const_values()), 2 * ld(input),      \
                     reinterpret_cast<RO*>(output->get_values()), 2 * ld(output)));              \
    }
DEF(std::complex<double>, double, std::complex<double>, double, gkoc_dense_copy_f64)
DEF(std::complex<float>, float, std::complex<float>, float, gkoc_dense_copy_f32)
DEF(std::complex<double>, double, std::complex<float>, float, gkoc_dense_convert_f64_f32)
DEF(std::complex<float>, float, std::complex<double>, double, gkoc_dense_convert_f32_f64)
#undef DEF

}  // namespace dense


// ====================================================================== cg
namespace cg {

#define DEF(T, TN)                                                              \
    template <>                                                                 \
    void initialize<T>(exec_t exec, const matrix::Dense<T>* b,                  \
                       matrix::Dense<T>* r, matrix::Dense<T>* z,                \
                       matrix::Dense<T>* p, matrix::Dense<T>* q,                \
                       matrix::Dense<T>* prev_rho, matrix::Dense<T>* rho,       \
                       array<stopping_status>* stop_status)                     \
    {                                                                           \
        cdna4::forget_learned(); /* a new solve confirms its shape anew */     \
        GKOC_CALL(gkoc_cg_initialize_##TN(                                      \
            stream_of(exec), rows(b), cols(b), b->get_const_values(), ld(b),    \
            r->get_values(), ld(r), z->get_values(), ld(z), p->get_values(),    \
            ld(p), q->get_values(), ld(q), prev_rho->get_values(),              \
            rho->get_values(), raw(stop_status)));                              \
    }                                                                           \
    template <>                                                                 \
    void step_1<T>(exec_t exec, matrix::Dense<T>* p, const matrix::Dense<T>* z, \
                   const matrix::Dense<T>* rho,                                 \
                   const matrix::Dense<T>* prev_rho,                            \
                   const array<stopping_status>* stop_status)                   \
    {                                                                           \
        if (cols(p) == 1 && ld(p) == 1 && ld(z) == 1 &&                         \
            cdna4::step_1_done_ahead(                                           \
                cdna4::vt_of<T>(), cdna4::stream_keeping_deferred(exec),        \
                rows(p), p->get_values(), z->get_const_values(),                \
                rho->get_const_values(), prev_rho->get_const_values(),          \
                raw(stop_status))) {                                            \
            return; /* run behind the criterion's kernel: fusion.cpp */         \
        }                                                                       \
        GKOC_CALL(gkoc_cg_step_1_##TN(                                          \
            stream_of(exec), rows(p), cols(p), p->get_values(), ld(p),          \
            z->get_const_values(), ld(z), rho->get_const_values(),              \
            prev_rho->get_const_values(), raw(stop_status)));                   \
    }                                                                           \
    template <>                                                                 \
    void step_2<T>(exec_t exec, matrix::Dense<T>* x, matrix::Dense<T>* r,       \
                   const matrix::Dense<T>* p, const matrix::Dense<T>* q,        \
                   const matrix::Dense<T>* beta, const matrix::Dense<T>* rho,   \
                   const array<stopping_status>* stop_status)                   \
    {                                                                           \
        if (cols(x) == 1 && ld(x) == 1 && ld(r) == 1 && ld(p) == 1 &&           \
            ld(q) == 1 &&                                                       \
            cdna4::hold_step_2(cdna4::vt_of<T>(), stream_of(exec), rows(x),     \
                               x->get_values(), r->get_values(),                \
                               p->get_const_values(), q->get_const_values(),    \
                               beta->get_const_values(),                        \
                               rho->get_const_values(), raw(stop_status))) {    \
            return; /* launched by the next call into the backend */           \
        }                                                                       \
        if (cols(x) == 1 && ld(x) == 1 && ld(r) == 1 && ld(p) == 1 &&           \
            ld(q) == 1 &&                                                       \
            cdna4::step_2_with_norm(                                            \
                cdna4::vt_of<T>(), exec->get_device_id(),                       \
                cdna4::stream_keeping_deferred(exec), rows(x), x->get_values(), \
                r->get_values(), p->get_const_values(), q->get_const_values(),  \
                beta->get_const_values(), rho->get_const_values(),              \
                raw(stop_status))) {                                            \
            return; /* step_2, and ||r|| for the criterion left behind */       \
        }                                                                       \
        GKOC_CALL(gkoc_cg_step_2_##TN(                                          \
            stream_of(exec), rows(x), cols(x), x->get_values(), ld(x),          \
            r->get_values(), ld(r), p->get_const_values(), ld(p),               \
            q->get_const_values(), ld(q), beta->get_const_values(),             \
            rho->get_const_values(), raw(stop_status)));                        \
    }
FOR_VT(DEF)
#undef DEF

}  // namespace cg


// =================================================================== gmres
namespace gmres {

#define DEF(T, TN)                                                              \
    template <>                                                                 \
    void restart<T>(exec_t exec, const matrix::Dense<T>* residual,              \
                    const matrix::Dense<T>* residual_norm,                      \
                    matrix::Dense<T>* residual_norm_collection,                 \
                    matrix::Dense<T>* krylov_bases, size_type* final_iter_nums) \
    {                                                                           \
        GKOC_CALL(gkoc_gmres_restart_##TN(                                      \
            stream_of(exec), rows(residual), cols(residual),                    \
            residual->get_const_values(), ld(residual),                         \
            residual_norm->get_const_values(),                                  \
            residual_norm_collection->get_values(), krylov_bases->get_values(), \
            ld(krylov_bases), reinterpret_cast<uint64_t*>(final_iter_nums)));   \
    }                                                                           \
    template <>                                                                 \
    void multi_axpy<T>(exec_t exec, const matrix::Dense<T>* krylov_bases,       \
                       const matrix::Dense<T>* y,                               \
                       matrix::Dense<T>* before_preconditioner,                 \
                       const size_type* final_iter_nums,                        \
                       stopping_status* stop_status)                            \
    {                                                                           \
        GKOC_CALL(gkoc_gmres_multi_axpy_##TN(                                   \
            stream_of(exec), rows(before_preconditioner),                       \
            cols(before_preconditioner), krylov_bases->get_const_values(),      \
            ld(krylov_bases), y->get_const_values(), ld(y),                     \
            before_preconditioner->get_values(), ld(before_preconditioner),     \
            reinterpret_cast<const uint64_t*>(final_iter_nums),                 \
            raw(stop_status)));                                                 \
    }                                                                           \
    template <>                                                                 \
    void multi_dot<T>(exec_t exec, const matrix::Dense<T>* krylov_bases,        \
                      const matrix::Dense<T>* next_krylov,                      \
                      matrix::Dense<T>* hessenberg_col)                         \
    {                                                                           \
        const int64_t n = rows(next_krylov), k = cols(next_krylov);             \
        const int64_t dots = rows(hessenberg_col) - 1;                          \
        const size_t bytes =                                                    \
            gkoc_gmres_multi_dot_workspace_bytes(n, k, dots, sizeof(T));        \
        array<char> tmp(exec, bytes);                                           \
        GKOC_CALL(gkoc_gmres_multi_dot_##TN(                                    \
            stream_of(exec), n, k, dots, krylov_bases->get_const_values(),      \
            ld(krylov_bases), next_krylov->get_const_values(),                  \
            ld(next_krylov), hessenberg_col->get_values(), ld(hessenberg_col),  \
            tmp.get_data(), bytes));                                            \
        exec->synchronize(); /* tmp is released on return */                    \
    }
FOR_VT(DEF)
#undef DEF

}  // namespace gmres

namespace common_gmres {

#define DEF(T, TN)                                                              \
    template <>                                                                 \
    void initialize<T>(exec_t exec, const matrix::Dense<T>* b,                  \
                       matrix::Dense<T>* residual, matrix::Dense<T>* givens_sin, \
                       matrix::Dense<T>* givens_cos,                            \
                       stopping_status* stop_status)                            \
    {                                                                           \
        GKOC_CALL(gkoc_common_gmres_initialize_##TN(                            \
            stream_of(exec), rows(b), cols(b), b->get_const_values(), ld(b),    \
            residual->get_values(), ld(residual), givens_sin->get_values(),     \
            ld(givens_sin), givens_cos->get_values(), ld(givens_cos),           \
            rows(givens_sin), raw(stop_status)));                               \
    }                                                                           \
    template <>                                                                 \
    void hessenberg_qr<T>(exec_t exec, matrix::Dense<T>* givens_sin,            \
                          matrix::Dense<T>* givens_cos,                         \
                          matrix::Dense<T>* residual_norm,                      \
                          matrix::Dense<T>* residual_norm_collection,           \
                          matrix::Dense<T>* hessenberg_iter, size_type iter,    \
                          size_type* final_iter_nums,                           \
                          const stopping_status* stop_status)                   \
    {                                                                           \
        GKOC_CALL(gkoc_common_gmres_hessenberg_qr_##TN(                         \
            stream_of(exec), cols(givens_sin), givens_sin->get_values(),        \
            ld(givens_sin), givens_cos->get_values(), ld(givens_cos),           \
            residual_norm->get_values(),                                        \
            residual_norm_collection->get_values(),                             \
            ld(residual_norm_collection), hessenberg_iter->get_values(),        \
            ld(hessenberg_iter), iter,                                          \
            reinterpret_cast<uint64_t*>(final_iter_nums), raw(stop_status)));   \
    }                                                                           \
    template <>                                                                 \
    void solve_krylov<T>(exec_t exec,                                           \
                         const matrix::Dense<T>* residual_norm_collection,      \
                         const matrix::Dense<T>* hessenberg,                    \
                         matrix::Dense<T>* y, const size_type* final_iter_nums, \
                         const stopping_status* stop_status)                    \
    {                                                                           \
        GKOC_CALL(gkoc_common_gmres_solve_krylov_##TN(                          \
            stream_of(exec), cols(residual_norm_collection),                    \
            residual_norm_collection->get_const_values(),                       \
            ld(residual_norm_collection), hessenberg->get_const_values(),       \
            ld(hessenberg), y->get_values(), ld(y),                             \
            reinterpret_cast<const uint64_t*>(final_iter_nums),                 \
            raw(stop_status)));                                                 \
    }
FOR_VT(DEF)
#undef DEF

}  // namespace common_gmres


// ==================================================================== stop
namespace residual_norm {

#define DEF(T, TN)                                                              \
    template <>                                                                 \
    void residual_norm<T>(exec_t exec, const matrix::Dense<T>* tau,             \
                          const matrix::Dense<T>* orig_tau,                     \
                          T rel_residual_goal, uint8 stoppingId,                \
                          bool setFinalized,                                    \
                          array<stopping_status>* stop_status,                  \
                          array<bool>* device_storage, bool* all_converged,     \
                          bool* one_changed)                                    \
    {                                                                           \
        if (device_storage->get_size() < 2) device_storage->resize_and_reset(2); \
        int allc = 0, chg = 0;                                                  \
        const auto st_ = stream_of(exec);                                       \
        if (cols(tau) == 1 &&                                                   \
            cdna4::criterion_then_step_1(                                       \
                cdna4::vt_of<T>(), st_, tau->get_const_values(),                \
                orig_tau->get_const_values(), double(rel_residual_goal),        \
                stoppingId, setFinalized, false, raw(stop_status),             \
                reinterpret_cast<uint8_t*>(device_storage->get_data()), &allc,  \
                &chg)) {                                                        \
            *all_converged = allc != 0;                                         \
            *one_changed = chg != 0;                                            \
            return; /* and the cg::step_1 that follows is under way */          \
        }                                                                       \
        GKOC_CALL(gkoc_residual_norm_##TN(                                      \
            st_, cols(tau), tau->get_const_values(),                            \
            orig_tau->get_const_values(), rel_residual_goal, stoppingId,        \
            setFinalized ? 1 : 0, raw(stop_status),                             \
            reinterpret_cast<uint8_t*>(device_storage->get_data()), &allc,      \
            &chg));                                                             \
        *all_converged = allc != 0;                                             \
        *one_changed = chg != 0;                                                \
    }
FOR_VT(DEF)
#undef DEF

}  // namespace residual_norm

namespace implicit_residual_norm {

#define DEF(T, TN)                                                              \
    template <>                                                                 \
    void implicit_residual_norm<T>(                                             \
        exec_t exec, const matrix::Dense<T>* tau,                               \
        const matrix::Dense<T>* orig_tau, T rel_residual_goal,                  \
        uint8 stoppingId, bool setFinalized,                                    \
        array<stopping_status>* stop_status, array<bool>* device_storage,       \
        bool* all_converged, bool* one_changed)                                 \
    {                                                                           \
        if (device_storage->get_size() < 2) device_storage->resize_and_reset(2); \
        int allc = 0, chg = 0;                                                  \
        const auto st_ = stream_of(exec);                                       \
        if (cols(tau) == 1 &&                                                   \
            cdna4::criterion_then_step_1(                                       \
                cdna4::vt_of<T>(), st_, tau->get_const_values(),                \
                orig_tau->get_const_values(), double(rel_residual_goal),        \
                stoppingId, setFinalized, true, raw(stop_status),             \
                reinterpret_cast<uint8_t*>(device_storage->get_data()), &allc,  \
                &chg)) {                                                        \
            *all_converged = allc != 0;                                         \
            *one_changed = chg != 0;                                            \
            return; /* and the cg::step_1 that follows is under way */          \
        }                                                                       \
        GKOC_CALL(gkoc_implicit_residual_norm_##TN(                             \
            st_, cols(tau), tau->get_const_values(),                            \
            orig_tau->get_const_values(), rel_residual_goal, stoppingId,        \
            setFinalized ? 1 : 0, raw(stop_status),                             \
            reinterpret_cast<uint8_t*>(device_storage->get_data()), &allc,      \
            &chg));                                                             \
        *all_converged = allc != 0;                                             \
        *one_changed = chg != 0;                                                \
    }
FOR_VT(DEF)
#undef DEF

}  // namespace implicit_residual_norm

namespace set_all_statuses {

void set_all_statuses(exec_t exec, uint8 stoppingId, bool setFinalized,
                      array<stopping_status>* stop_status)
{
    GKOC_CALL(gkoc_set_all_statuses(stream_of(exec),
                                    static_cast<int64_t>(stop_status->get_size()),
                                    stoppingId, setFinalized ? 1 : 0,
                                    raw(stop_status)));
}

}  // namespace set_all_statuses


// ================================================================== jacobi
namespace jacobi {

template <typename I>
inline gkoc_jacobi_scheme scheme_of(
    const preconditioner::block_interleaved_storage_scheme<I>& s)
{
    return {static_cast<int64_t>(s.block_offset),
            static_cast<int64_t>(s.group_offset), s.group_power};
}

// Precisions attached to the blocks (Jacobi::storage_optimization other than full
// precision): the adaptive entry points of the C ABI handle fixed, block-wise and
// autodetect requests alike; implemented for value type double.
template <typename T>
inline bool has_precisions(const array<precision_reduction>& prec)
{
    return prec.get_const_data() != nullptr && prec.get_size() != 0;
}

// the C ABI of block-wise / adaptive storage per value type: double has kernels of its own
// (csrc/jacobi.hip, bit-identical to the reference), float goes through the generic ones
template <typename T, typename I>
struct adaptive_abi;
template <>
struct adaptive_abi<double, int32> {
    static constexpr auto generate = gkoc_jacobi_generate_adaptive_f64_i32;
    static constexpr auto apply = gkoc_jacobi_apply_adaptive_f64_i32;
};
template <>
struct adaptive_abi<double, int64> {
    static constexpr auto generate = gkoc_jacobi_generate_adaptive_f64_i64;
    static constexpr auto apply = gkoc_jacobi_apply_adaptive_f64_i64;
};
template <>
struct adaptive_abi<float, int32> {
    static constexpr auto generate = gkoc_jacobi_generate_adaptive_f32_i32;
    static constexpr auto apply = gkoc_jacobi_apply_adaptive_f32_i32;
};
template <>
struct adaptive_abi<float, int64> {
    static constexpr auto generate = gkoc_jacobi_generate_adaptive_f32_i64;
    static constexpr auto apply = gkoc_jacobi_apply_adaptive_f32_i64;
};

void initialize_precisions(exec_t exec, const array<precision_reduction>& source,
                           array<precision_reduction>& precisions)
{
    GKOC_CALL(gkoc_jacobi_initialize_precisions(
        stream_of(exec), reinterpret_cast<const uint8_t*>(source.get_const_data()),
        static_cast<int64_t>(source.get_size()),
        reinterpret_cast<uint8_t*>(precisions.get_data()),
        static_cast<int64_t>(precisions.get_size())));
}

// The block array of a block-Jacobi preconditioner is k n values (k = the block size): to the allocator a
// multi-vector, so Ginkgo's raw_alloc put it among the vectors - and the application, which streams the blocks
// AND reads / writes vectors, had both in one memory class: the one-kernel cg::step_2 + application
// took 372 us behind the core against 337 us in the native loop, which allocates the blocks with the index
// arrays.  generate is the one call that knows what this array is, and it OVERWRITES it: an owning array of a
// class that holds vectors is re-allocated in the class of the matrix' column indices before it is filled (same
// size; a view on user memory, a small array, or GKOC_TUNE_JACOBI_REHOME = 0: left alone).
template <typename T>
void rehome_jacobi_blocks(exec_t exec, const void* col_idxs, array<T>& blocks)
{
    int64_t on = 1;
    gkoc_tune_get(GKOC_TUNE_JACOBI_REHOME, &on);
    const size_t bytes = blocks.get_size() * sizeof(T);
    if (on == 0 || !blocks.is_owning() || bytes < (size_t(16) << 20) || col_idxs == nullptr) return;
    int have = -1, want = -1;
    gkoc_arena_class_of(blocks.get_const_data(), &have);
    gkoc_arena_class_of(col_idxs, &want);
    if (have < 0 || want < 0 || have == want) return;
    try {
        cdna4::alloc_role_hint = want == 0 ? GKOC_MEM_VALUES : want == 1 ? GKOC_MEM_INDICES : GKOC_MEM_VECTOR;
        array<T> moved(exec, blocks.get_size());
        cdna4::alloc_role_hint = 0;
        blocks = std::move(moved);
    } catch (...) {
        cdna4::alloc_role_hint = 0;      // no room there: the array stays where it is
    }
}

#define DEF(T, TN, I, IN)                                                       \
    template <>                                                                 \
    void find_blocks<T, I>(exec_t exec, const matrix::Csr<T, I>* system_matrix, \
                           uint32 max_block_size, size_type& num_blocks,        \
                           array<I>& block_pointers)                            \
    {                                                                           \
        int64_t nb = 0;                                                         \
        GKOC_CALL(gkoc_jacobi_find_blocks_##TN##_##IN(                          \
            stream_of(exec), system_matrix->get_size()[0],                      \
            system_matrix->get_const_row_ptrs(),                                \
            system_matrix->get_const_col_idxs(), max_block_size, &nb,           \
            block_pointers.get_data()));                                        \
        num_blocks = static_cast<size_type>(nb);                                \
    }                                                                           \
    template <>                                                                 \
    void generate<T, I>(                                                        \
        exec_t exec, const matrix::Csr<T, I>* system_matrix,                    \
        size_type num_blocks, uint32 max_block_size, T accuracy,                \
        const preconditioner::block_interleaved_storage_scheme<I>&              \
            storage_scheme,                                                     \
        array<T>& conditioning, array<precision_reduction>& block_precisions,   \
        const array<I>& block_pointers, array<T>& blocks)                       \
    {                                                                           \
        cdna4::forget_learned(); /* the blocks change: what a solve has shown \
                                    about THESE arrays is void (any scheme) */   \
        rehome_jacobi_blocks(exec, system_matrix->get_const_col_idxs(), blocks); \
        if (has_precisions<T>(block_precisions)) {                              \
            GKOC_CALL((adaptive_abi<T, I>::generate(                            \
                stream_of(exec), system_matrix->get_size()[0],                  \
                system_matrix->get_const_row_ptrs(),                            \
                system_matrix->get_const_col_idxs(),                            \
                system_matrix->get_const_values(), num_blocks, max_block_size,  \
                scheme_of(storage_scheme), block_pointers.get_const_data(),     \
                accuracy,                                                       \
                reinterpret_cast<uint8_t*>(block_precisions.get_data()),        \
                conditioning.get_data(), blocks.get_data())));                  \
            return;                                                             \
        }                                                                       \
        GKOC_CALL(gkoc_jacobi_generate_##TN##_##IN(                             \
            stream_of(exec), system_matrix->get_size()[0],                      \
            system_matrix->get_const_row_ptrs(),                                \
            system_matrix->get_const_col_idxs(),                                \
            system_matrix->get_const_values(), num_blocks, max_block_size,      \
            scheme_of(storage_scheme), block_pointers.get_const_data(),         \
            blocks.get_data(), nullptr));                                       \
    }                                                                           \
    template <>                                                                 \
    void simple_apply<T, I>(                                                    \
        exec_t exec, size_type num_blocks, uint32 max_block_size,               \
        const preconditioner::block_interleaved_storage_scheme<I>&              \
            storage_scheme,                                                     \
        const array<precision_reduction>& block_precisions,                     \
        const array<I>& block_pointers, const array<T>& blocks,                 \
        const matrix::Dense<T>* b, matrix::Dense<T>* x)                         \
    {                                                                           \
        if (!has_precisions<T>(block_precisions) && cols(b) == 1 &&             \
            ld(b) == 1 && ld(x) == 1 &&                                         \
            cdna4::hold_jacobi_apply(                                           \
                cdna4::vt_of<T>(), cdna4::it_of<I>(),                           \
                cdna4::stream_keeping_deferred(exec), num_blocks,               \
                max_block_size, scheme_of(storage_scheme),                      \
                block_pointers.get_const_data(), blocks.get_const_data(),       \
                b->get_const_values(), rows(b), x->get_values())) {             \
            return; /* follows a held cg::step_2: see fusion.cpp */             \
        }                                                                       \
        if (!has_precisions<T>(block_precisions) && cols(b) == 1 &&             \
            ld(b) == 1 && ld(x) == 1 &&                                         \
            cdna4::jacobi_apply_with_dot(                                       \
                cdna4::vt_of<T>(), cdna4::it_of<I>(), exec->get_device_id(),    \
                cdna4::stream_keeping_deferred(exec), num_blocks,               \
                max_block_size, scheme_of(storage_scheme),                      \
                block_pointers.get_const_data(), blocks.get_const_data(),       \
                b->get_const_values(), rows(b), x->get_values())) {             \
            return; /* x = M b, and <b, x> left behind for the dot that follows */ \
        }                                                                       \
        if (has_precisions<T>(block_precisions)) {                              \
            GKOC_CALL((adaptive_abi<T, I>::apply(                               \
                stream_of(exec), num_blocks, max_block_size,                    \
                scheme_of(storage_scheme), block_pointers.get_const_data(),     \
                blocks.get_const_data(),                                        \
                reinterpret_cast<const uint8_t*>(                               \
                    block_precisions.get_const_data()),                         \
                nullptr, b->get_const_values(), ld(b), nullptr,                 \
                x->get_values(), ld(x), cols(b))));                             \
            return;                                                             \
        }                                                                       \
        GKOC_CALL(gkoc_jacobi_simple_apply_##TN##_##IN(                         \
            stream_of(exec), num_blocks, max_block_size,                        \
            scheme_of(storage_scheme), block_pointers.get_const_data(),         \
            blocks.get_const_data(), b->get_const_values(), ld(b),              \
            x->get_values(), ld(x), cols(b)));                                  \
    }                                                                           \
    template <>                                                                 \
    void apply<T, I>(                                                           \
        exec_t exec, size_type num_blocks, uint32 max_block_size,               \
        const preconditioner::block_interleaved_storage_scheme<I>&              \
            storage_scheme,                                                     \
        const array<precision_reduction>& block_precisions,                     \
        const array<I>& block_pointers, const array<T>& blocks,                 \
        const matrix::Dense<T>* alpha, const matrix::Dense<T>* b,               \
        const matrix::Dense<T>* beta, matrix::Dense<T>* x)                      \
    {                                                                           \
        if (has_precisions<T>(block_precisions)) {                              \
            GKOC_CALL((adaptive_abi<T, I>::apply(                               \
                stream_of(exec), num_blocks, max_block_size,                    \
                scheme_of(storage_scheme), block_pointers.get_const_data(),     \
                blocks.get_const_data(),                                        \
                reinterpret_cast<const uint8_t*>(                               \
                    block_precisions.get_const_data()),                         \
                alpha->get_const_values(), b->get_const_values(), ld(b),        \
                beta->get_const_values(), x->get_values(), ld(x), cols(b))));   \
            return;                                                             \
        }                                                                       \
        GKOC_CALL(gkoc_jacobi_apply_##TN##_##IN(                                \
            stream_of(exec), num_blocks, max_block_size,                        \
            scheme_of(storage_scheme), block_pointers.get_const_data(),         \
            blocks.get_const_data(), alpha->get_const_values(),                 \
            b->get_const_values(), ld(b), beta->get_const_values(),             \
            x->get_values(), ld(x), cols(b)));                                  \
    }
FOR_VT_IT(DEF)
#undef DEF

#define DEF(T, TN, I, IN)                                                       \
    static void transpose_blocks_##TN##_##IN(                                   \
        exec_t exec, size_type num_blocks, uint32 max_block_size,               \
        const array<precision_reduction>& block_precisions,                     \
        const array<I>& block_pointers, const array<T>& blocks,                 \
        const preconditioner::block_interleaved_storage_scheme<I>& scheme,      \
        array<T>& out_blocks)                                                   \
    {                                                                           \
        const auto prec = has_precisions<T>(block_precisions)                   \
                              ? reinterpret_cast<const uint8_t*>(               \
                                    block_precisions.get_const_data())          \
                              : nullptr;                                        \
        if (prec && std::is_same<T, float>::value) {                            \
            GKOC_CALL(gkoc_jacobi_transpose_adaptive_f32_##IN(                  \
                stream_of(exec), num_blocks, scheme_of(scheme),                 \
                block_pointers.get_const_data(),                                \
                reinterpret_cast<const float*>(blocks.get_const_data()), prec,  \
                0, reinterpret_cast<float*>(out_blocks.get_data())));           \
            return;                                                             \
        }                                                                       \
        GKOC_CALL(gkoc_jacobi_transpose_##TN##_##IN(                            \
            stream_of(exec), num_blocks, max_block_size, scheme_of(scheme),     \
            block_pointers.get_const_data(), blocks.get_const_data(), prec,     \
            out_blocks.get_data()));                                            \
    }                                                                           \
    template <>                                                                 \
    void transpose_jacobi<T, I>(                                                \
        exec_t exec, size_type num_blocks, uint32 max_block_size,               \
        const array<precision_reduction>& block_precisions,                     \
        const array<I>& block_pointers, const array<T>& blocks,                 \
        const preconditioner::block_interleaved_storage_scheme<I>& scheme,      \
        array<T>& out_blocks)                                                   \
    {                                                                           \
        transpose_blocks_##TN##_##IN(exec, num_blocks, max_block_size,          \
                                     block_precisions, block_pointers, blocks,  \
                                     scheme, out_blocks);                       \
    }                                                                           \
    template <>                                                                 \
    void conj_transpose_jacobi<T, I>(                                           \
        exec_t exec, size_type num_blocks, uint32 max_block_size,               \
        const array<precision_reduction>& block_precisions,                     \
        const array<I>& block_pointers, const array<T>& blocks,                 \
        const preconditioner::block_interleaved_storage_scheme<I>& scheme,      \
        array<T>& out_blocks)                                                   \
    {                                                                           \
        transpose_blocks_##TN##_##IN(exec, num_blocks, max_block_size,          \
                                     block_precisions, block_pointers, blocks,  \
                                     scheme, out_blocks);                       \
    }
FOR_VT_IT(DEF)
#undef DEF

#define DEF(T, TN)                                                              \
    template <>                                                                 \
    void invert_diagonal<T>(exec_t exec, const array<T>& diag,                  \
                            array<T>& inv_diag)                                 \
    {                                                                           \
        GKOC_CALL(gkoc_jacobi_invert_diagonal_##TN(                             \
            stream_of(exec), static_cast<int64_t>(diag.get_size()),             \
            diag.get_const_data(), inv_diag.get_data()));                       \
    }                                                                           \
    template <>                                                                 \
    void simple_scalar_apply<T>(exec_t exec, const array<T>& diag,              \
                                const matrix::Dense<T>* b, matrix::Dense<T>* x) \
    {                                                                           \
        GKOC_CALL(gkoc_jacobi_simple_scalar_apply_##TN(                         \
            stream_of(exec), rows(x), cols(x), diag.get_const_data(),           \
            b->get_const_values(), ld(b), x->get_values(), ld(x)));             \
    }                                                                           \
    template <>                                                                 \
    void scalar_apply<T>(exec_t exec, const array<T>& diag,                     \
                         const matrix::Dense<T>* alpha,                         \
                         const matrix::Dense<T>* b,                             \
                         const matrix::Dense<T>* beta, matrix::Dense<T>* x)     \
    {                                                                           \
        GKOC_CALL(gkoc_jacobi_scalar_apply_##TN(                                \
            stream_of(exec), rows(x), cols(x), diag.get_const_data(),           \
            alpha->get_const_values(), b->get_const_values(), ld(b),            \
            beta->get_const_values(), x->get_values(), ld(x)));                 \
    }
FOR_VT(DEF)
#undef DEF

// jacobi::convert_to_dense / scalar_convert_to_dense (core/preconditioner/jacobi_kernels.hpp:105-118;
// reference/preconditioner/jacobi_kernels.cpp:670-721): the preconditioner as a dense matrix.  A
// convenience of small problems - computed as M * I with the apply kernels above (every entry is a
// stored value times one plus zeros: exact), for every storage precision the apply kernels read.
namespace {
template <typename T>
std::unique_ptr<matrix::Dense<T>> identity_on(exec_t exec, size_type n)
{
    auto host = matrix::Dense<T>::create(exec->get_master(), dim<2>{n, n});
    host->fill(zero<T>());
    for (size_type i = 0; i < n; ++i) host->at(i, i) = one<T>();
    return clone(exec, host);
}
}  // namespace

#define DEF(T, TN)                                                              \
    template <>                                                                 \
    void scalar_convert_to_dense<T>(exec_t exec, const array<T>& blocks,        \
                                    matrix::Dense<T>* result)                   \
    {                                                                           \
        const auto n = result->get_size()[0];                                   \
        if (n == 0 || result->get_size()[1] == 0) return;                       \
        auto eye = matrix::Dense<T>::create(exec->get_master(), result->get_size()); \
        eye->fill(zero<T>());                                                   \
        for (size_type i = 0; i < std::min(n, result->get_size()[1]); ++i) {    \
            eye->at(i, i) = one<T>();                                           \
        }                                                                       \
        auto dev = clone(exec, eye);                                            \
        simple_scalar_apply<T>(exec, blocks, dev.get(), result);                \
        exec->synchronize(); /* the identity is released on return */           \
    }
FOR_VT(DEF)
DEF(std::complex<double>, c128)
DEF(std::complex<float>, c64)
#undef DEF

#define DEF(T, TN, I, IN)                                                       \
    template <>                                                                 \
    void convert_to_dense<T, I>(                                                \
        exec_t exec, size_type num_blocks,                                      \
        const array<precision_reduction>& block_precisions,                     \
        const array<I>& block_pointers, const array<T>& blocks,                 \
        const preconditioner::block_interleaved_storage_scheme<I>&              \
            storage_scheme,                                                     \
        T* result_values, size_type result_stride)                              \
    {                                                                           \
        const auto n = static_cast<size_type>(                                  \
            exec->copy_val_to_host(block_pointers.get_const_data() + num_blocks)); \
        if (n == 0) return;                                                     \
        auto eye = identity_on<T>(exec, n);                                     \
        auto out = matrix::Dense<T>::create(                                    \
            exec, dim<2>{n, n},                                                 \
            make_array_view(exec, (n - 1) * result_stride + n, result_values),  \
            result_stride);                                                     \
        /* (max_block_size only selects a kernel: the scheme's block_offset bounds it) */ \
        simple_apply<T, I>(exec, num_blocks,                                    \
                           static_cast<uint32>(storage_scheme.block_offset),    \
                           storage_scheme, block_precisions, block_pointers,    \
                           blocks, eye.get(), out.get());                       \
        exec->synchronize(); /* the identity is released on return */           \
    }
FOR_VT_IT(DEF)
DEF(std::complex<double>, c128, int32, i32)
DEF(std::complex<double>, c128, int64, i64)
DEF(std::complex<float>, c64, int32, i32)
DEF(std::complex<float>, c64, int64, i64)
#undef DEF

}  // namespace jacobi


// ============================================================== components
namespace components {

template <>
void fill_array<double>(exec_t exec, double* data, size_type n, double val)
{
    GKOC_CALL(gkoc_fill_array_f64(stream_of(exec), data, n, val));
}
template <>
void fill_array<float>(exec_t exec, float* data, size_type n, float val)
{
    GKOC_CALL(gkoc_fill_array_f32(stream_of(exec), data, n, val));
}
template <>
void fill_array<int32>(exec_t exec, int32* data, size_type n, int32 val)
{
    GKOC_CALL(gkoc_fill_array_i32(stream_of(exec), data, n, val));
}
template <>
void fill_array<int64>(exec_t exec, int64* data, size_type n, int64 val)
{
    GKOC_CALL(gkoc_fill_array_i64(stream_of(exec), data, n, val));
}
template <>
void fill_array<size_type>(exec_t exec, size_type* data, size_type n,
                           size_type val)
{
    GKOC_CALL(gkoc_fill_array_i64(stream_of(exec),
                                  reinterpret_cast<int64_t*>(data), n,
                                  static_cast<int64_t>(val)));
}
#define DEF(U)                                                                      \
    template <>                                                                     \
    void fill_array<U>(exec_t exec, U* data, size_type n, U val)                    \
    {                                                                               \
        GKOC_CALL(gkoc_fill_array_small(stream_of(exec), data, static_cast<int64_t>(n), \
                                        static_cast<int>(sizeof(U)),                \
                                        static_cast<uint32_t>(val)));               \
    }
DEF(bool)
DEF(char)
DEF(uint16)
DEF(uint32)
#undef DEF
template <>
void fill_seq_array<int32>(exec_t exec, int32* data, size_type n)
{
    GKOC_CALL(gkoc_fill_seq_array_i32(stream_of(exec), data, n));
}
template <>
void fill_seq_array<int64>(exec_t exec, int64* data, size_type n)
{
    GKOC_CALL(gkoc_fill_seq_array_i64(stream_of(exec), data, n));
}
template <>
void prefix_sum_nonnegative<int32>(exec_t exec, int32* counts, size_type n)
{
    GKOC_CALL(gkoc_prefix_sum_nonnegative_checked_i32(stream_of(exec), counts, n));
}
template <>
void prefix_sum_nonnegative<int64>(exec_t exec, int64* counts, size_type n)
{
    GKOC_CALL(gkoc_prefix_sum_nonnegative_checked_i64(stream_of(exec), counts, n));
}
template <>
void prefix_sum_nonnegative<size_type>(exec_t exec, size_type* counts,
                                       size_type n)
{
    GKOC_CALL(gkoc_prefix_sum_nonnegative_checked_u64(
        stream_of(exec), reinterpret_cast<uint64_t*>(counts), n));
}
template <>
void convert_ptrs_to_sizes<int32>(exec_t exec, const int32* ptrs,
                                  size_type num_blocks, size_type* sizes)
{
    GKOC_CALL(gkoc_convert_ptrs_to_sizes_i32(
        stream_of(exec), num_blocks, ptrs, reinterpret_cast<uint64_t*>(sizes)));
}
template <>
void convert_ptrs_to_sizes<int64>(exec_t exec, const int64* ptrs,
                                  size_type num_blocks, size_type* sizes)
{
    GKOC_CALL(gkoc_convert_ptrs_to_sizes_i64(
        stream_of(exec), num_blocks, ptrs, reinterpret_cast<uint64_t*>(sizes)));
}
template <>
void convert_idxs_to_ptrs<int32, int32>(exec_t exec, const int32* idxs,
                                        size_type num_idxs,
                                        size_type num_blocks, int32* ptrs)
{
    GKOC_CALL(gkoc_convert_idxs_to_ptrs_i32(stream_of(exec), num_idxs, idxs,
                                            num_blocks, ptrs));
}
template <>
void convert_idxs_to_ptrs<int64, int64>(exec_t exec, const int64* idxs,
                                        size_type num_idxs,
                                        size_type num_blocks, int64* ptrs)
{
    GKOC_CALL(gkoc_convert_idxs_to_ptrs_i64(stream_of(exec), num_idxs, idxs,
                                            num_blocks, ptrs));
}

template <>
void convert_idxs_to_ptrs<int32, int64>(exec_t exec, const int32* idxs, size_type num_idxs,
                                        size_type num_blocks, int64* ptrs)
{
    GKOC_CALL(gkoc_convert_idxs_to_ptrs_i32_i64(stream_of(exec), num_idxs, idxs, num_blocks, ptrs));
}
template <>
void convert_idxs_to_ptrs<int64, int32>(exec_t exec, const int64* idxs, size_type num_idxs,
                                        size_type num_blocks, int32* ptrs)
{
    GKOC_CALL(gkoc_convert_idxs_to_ptrs_i64_i32(stream_of(exec), num_idxs, idxs, num_blocks, ptrs));
}

#define DEF(T, TN, I, IN)                                                       \
    template <>                                                                 \
    void aos_to_soa<T, I>(exec_t exec,                                          \
                          const array<matrix_data_entry<T, I>>& in,             \
                          device_matrix_data<T, I>& out)                        \
    {                                                                           \
        static_assert(sizeof(matrix_data_entry<T, I>) ==                        \
                          (sizeof(T) > sizeof(I) ? 2 * sizeof(T)                \
                                                 : 3 * sizeof(I)) ||            \
                          sizeof(matrix_data_entry<T, I>) ==                    \
                              2 * sizeof(I) + sizeof(T),                        \
                      "unexpected matrix_data_entry layout");                   \
        GKOC_CALL(gkoc_aos_to_soa_##TN##_##IN(                                  \
            stream_of(exec), static_cast<int64_t>(in.get_size()),               \
            in.get_const_data(), out.get_row_idxs(), out.get_col_idxs(),        \
            out.get_values()));                                                 \
    }
FOR_VT_IT(DEF)
#undef DEF


// device-side assembly (core/base/device_matrix_data.cpp:106-129): the workspace lives
// until the stream has drained; remove_zeros / sum_duplicates replace the arrays only
// if entries go away, like the reference (test/base/device_matrix_data_kernels.cpp:
// "no reallocation")
struct assembly_scratch {
    exec_t exec;
    array<char> buf;
    assembly_scratch(exec_t e, size_t bytes) : exec{e}, buf{e, bytes} {}
    ~assembly_scratch() { exec->synchronize(); }
};

#define DEF(T, TN, I, IN)                                                       \
    template <>                                                                 \
    void soa_to_aos<T, I>(exec_t exec, const device_matrix_data<T, I>& in,      \
                          array<matrix_data_entry<T, I>>& out)                  \
    {                                                                           \
        GKOC_CALL(gkoc_soa_to_aos_##TN##_##IN(                                  \
            stream_of(exec),                                                    \
            static_cast<int64_t>(in.get_num_stored_elements()),                 \
            in.get_const_row_idxs(), in.get_const_col_idxs(),                   \
            in.get_const_values(), out.get_data()));                            \
    }                                                                           \
    template <>                                                                 \
    void sort_row_major<T, I>(exec_t exec, size_type num_elems, I* row_idxs,    \
                              I* col_idxs, T* values)                           \
    {                                                                           \
        const auto nnz = static_cast<int64_t>(num_elems);                       \
        assembly_scratch w(exec, gkoc_sort_row_major_workspace_bytes(           \
                                     nnz, sizeof(T), sizeof(I)));               \
        GKOC_CALL(gkoc_sort_row_major_##TN##_##IN(                              \
            stream_of(exec), nnz, row_idxs, col_idxs, values,                   \
            w.buf.get_data(), w.buf.get_size()));                               \
    }                                                                           \
    template <>                                                                 \
    void remove_zeros<T, I>(exec_t exec, array<T>& values,                      \
                            array<I>& row_idxs, array<I>& col_idxs)             \
    {                                                                           \
        const auto nnz = static_cast<int64_t>(values.get_size());               \
        assembly_scratch w(exec, gkoc_compact_workspace_bytes(nnz));            \
        int64_t kept = 0;                                                       \
        GKOC_CALL(gkoc_remove_zeros_count_##TN(                                 \
            stream_of(exec), nnz, values.get_const_data(), w.buf.get_data(),    \
            w.buf.get_size(), &kept));                                          \
        if (kept < nnz) {                                                       \
            array<T> new_values{exec, static_cast<size_type>(kept)};            \
            array<I> new_row_idxs{exec, static_cast<size_type>(kept)};          \
            array<I> new_col_idxs{exec, static_cast<size_type>(kept)};          \
            GKOC_CALL(gkoc_remove_zeros_fill_##TN##_##IN(                       \
                stream_of(exec), nnz, row_idxs.get_const_data(),                \
                col_idxs.get_const_data(), values.get_const_data(),             \
                w.buf.get_const_data(), new_row_idxs.get_data(),                \
                new_col_idxs.get_data(), new_values.get_data()));               \
            exec->synchronize();                                                \
            values = std::move(new_values);                                     \
            row_idxs = std::move(new_row_idxs);                                 \
            col_idxs = std::move(new_col_idxs);                                 \
        }                                                                       \
    }                                                                           \
    template <>                                                                 \
    void sum_duplicates<T, I>(exec_t exec, size_type, array<T>& values,         \
                              array<I>& row_idxs, array<I>& col_idxs)           \
    {                                                                           \
        const auto nnz = static_cast<int64_t>(values.get_size());               \
        assembly_scratch w(exec, gkoc_compact_workspace_bytes(nnz));            \
        int64_t kept = 0;                                                       \
        GKOC_CALL(gkoc_sum_duplicates_count_##IN(                               \
            stream_of(exec), nnz, row_idxs.get_const_data(),                    \
            col_idxs.get_const_data(), w.buf.get_data(), w.buf.get_size(),      \
            &kept));                                                            \
        if (kept < nnz) {                                                       \
            array<T> new_values{exec, static_cast<size_type>(kept)};            \
            array<I> new_row_idxs{exec, static_cast<size_type>(kept)};          \
            array<I> new_col_idxs{exec, static_cast<size_type>(kept)};          \
            GKOC_CALL(gkoc_sum_duplicates_fill_##TN##_##IN(                     \
                stream_of(exec), nnz, row_idxs.get_const_data(),                \
                col_idxs.get_const_data(), values.get_const_data(),             \
                w.buf.get_const_data(), new_row_idxs.get_data(),                \
                new_col_idxs.get_data(), new_values.get_data()));               \
            exec->synchronize();                                                \
            values = std::move(new_values);                                     \
            row_idxs = std::move(new_row_idxs);                                 \
            col_idxs = std::move(new_col_idxs);                                 \
        }                                                                       \
    }
FOR_VT_IT(DEF)
#undef DEF

}  // namespace components
}  // namespace hip
}  // namespace kernels
}  // namespace gko
