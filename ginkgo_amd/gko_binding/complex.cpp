// gko::kernels::hip for std::complex<float / double>, as far as this backend goes: the kernels that
// move or measure complex data without multiplying it - device_matrix_data assembly, fill_array /
// fill_seq_array, and the column 2-norms of a complex Dense (what stop::ResidualNorm asks for).
// std::complex<R> is layout-compatible with the pair {R re, im} of the C ABI (gkoc_c128 / gkoc_c64).
// Everything else complex stays with Ginkgo's NotCompiled stubs.
#include <complex>

#include <ginkgo/core/base/device_matrix_data.hpp>
#include <ginkgo/core/base/matrix_data.hpp>
#include <ginkgo/core/matrix/coo.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/matrix/diagonal.hpp>

#include "core/base/device_matrix_data_kernels.hpp"
#include "core/components/absolute_array_kernels.hpp"
#include "core/components/precision_conversion_kernels.hpp"
#include "core/components/reduce_array_kernels.hpp"
#include "core/components/fill_array_kernels.hpp"
#include "core/matrix/coo_kernels.hpp"
#include "core/matrix/csr_kernels.hpp"
#include "core/matrix/dense_kernels.hpp"
#include "core/matrix/diagonal_kernels.hpp"
#include "core/preconditioner/jacobi_kernels.hpp"
#include "core/stop/residual_norm_kernels.hpp"
#include "shim_common.hpp"

namespace gko {
namespace kernels {
namespace hip {

using cdna4::cols;
using cdna4::ld;
using cdna4::rows;
using cdna4::stream_of;
using exec_t = std::shared_ptr<const HipExecutor>;

static_assert(sizeof(std::complex<double>) == sizeof(gkoc_c128) &&
                  sizeof(std::complex<float>) == sizeof(gkoc_c64),
              "complex layout");

inline gkoc_c128* pairs(std::complex<double>* p) { return reinterpret_cast<gkoc_c128*>(p); }
inline const gkoc_c128* pairs(const std::complex<double>* p)
{
    return reinterpret_cast<const gkoc_c128*>(p);
}
inline gkoc_c64* pairs(std::complex<float>* p) { return reinterpret_cast<gkoc_c64*>(p); }
inline const gkoc_c64* pairs(const std::complex<float>* p)
{
    return reinterpret_cast<const gkoc_c64*>(p);
}

#define FOR_CT(M) M(std::complex<double>, gkoc_c128, c128, double, f64) M(std::complex<float>, gkoc_c64, c64, float, f32)
#define FOR_CT_IT(M)                                                                             \
    M(std::complex<double>, c128, int32, i32) M(std::complex<double>, c128, int64, i64)          \
        M(std::complex<float>, c64, int32, i32) M(std::complex<float>, c64, int64, i64)


namespace components {

// the workspace lives until the stream has drained
struct complex_scratch {
    exec_t exec;
    array<char> buf;
    complex_scratch(exec_t e, size_t bytes) : exec{e}, buf{e, bytes} {}
    ~complex_scratch() { exec->synchronize(); }
};

#define DEF(C, P, TN, R, RN)                                                                     \
    template <>                                                                                  \
    void fill_array<C>(exec_t exec, C* data, size_type n, C val)                                 \
    {                                                                                            \
        GKOC_CALL(gkoc_fill_array_##TN(stream_of(exec), pairs(data), static_cast<int64_t>(n),    \
                                       P{val.real(), val.imag()}));                              \
    }                                                                                            \
    template <>                                                                                  \
    void fill_seq_array<C>(exec_t exec, C* data, size_type n)                                    \
    {                                                                                            \
        GKOC_CALL(gkoc_fill_seq_array_##TN(stream_of(exec), pairs(data), static_cast<int64_t>(n))); \
    }
FOR_CT(DEF)
#undef DEF

#define DEF(C, TN, I, IN)                                                                        \
    template <>                                                                                  \
    void aos_to_soa<C, I>(exec_t exec, const array<matrix_data_entry<C, I>>& in,                 \
                          device_matrix_data<C, I>& out)                                         \
    {                                                                                            \
        GKOC_CALL(gkoc_aos_to_soa_##TN##_##IN(stream_of(exec), static_cast<int64_t>(in.get_size()), \
                                              in.get_const_data(), out.get_row_idxs(),           \
                                              out.get_col_idxs(), pairs(out.get_values())));     \
    }                                                                                            \
    template <>                                                                                  \
    void soa_to_aos<C, I>(exec_t exec, const device_matrix_data<C, I>& in,                       \
                          array<matrix_data_entry<C, I>>& out)                                   \
    {                                                                                            \
        GKOC_CALL(gkoc_soa_to_aos_##TN##_##IN(                                                   \
            stream_of(exec), static_cast<int64_t>(in.get_num_stored_elements()),                 \
            in.get_const_row_idxs(), in.get_const_col_idxs(), pairs(in.get_const_values()),      \
            out.get_data()));                                                                    \
    }                                                                                            \
    template <>                                                                                  \
    void sort_row_major<C, I>(exec_t exec, size_type num_elems, I* row_idxs, I* col_idxs,        \
                              C* values)                                                         \
    {                                                                                            \
        const auto nnz = static_cast<int64_t>(num_elems);                                        \
        complex_scratch w(exec, gkoc_sort_row_major_workspace_bytes(nnz, sizeof(C), sizeof(I))); \
        GKOC_CALL(gkoc_sort_row_major_##TN##_##IN(stream_of(exec), nnz, row_idxs, col_idxs,      \
                                                  pairs(values), w.buf.get_data(),               \
                                                  w.buf.get_size()));                            \
    }                                                                                            \
    template <>                                                                                  \
    void remove_zeros<C, I>(exec_t exec, array<C>& values, array<I>& row_idxs,                   \
                            array<I>& col_idxs)                                                  \
    {                                                                                            \
        const auto nnz = static_cast<int64_t>(values.get_size());                                \
        complex_scratch w(exec, gkoc_compact_workspace_bytes(nnz));                              \
        int64_t kept = 0;                                                                        \
        GKOC_CALL(gkoc_remove_zeros_count_##TN(stream_of(exec), nnz,                             \
                                               pairs(values.get_const_data()),                   \
                                               w.buf.get_data(), w.buf.get_size(), &kept));      \
        if (kept < nnz) {                                                                        \
            array<C> new_values{exec, static_cast<size_type>(kept)};                             \
            array<I> new_row_idxs{exec, static_cast<size_type>(kept)};                           \
            array<I> new_col_idxs{exec, static_cast<size_type>(kept)};                           \
            GKOC_CALL(gkoc_remove_zeros_fill_##TN##_##IN(                                        \
                stream_of(exec), nnz, row_idxs.get_const_data(), col_idxs.get_const_data(),      \
                pairs(values.get_const_data()), w.buf.get_const_data(),                          \
                new_row_idxs.get_data(), new_col_idxs.get_data(), pairs(new_values.get_data())));\
            exec->synchronize();                                                                 \
            values = std::move(new_values);                                                      \
            row_idxs = std::move(new_row_idxs);                                                  \
            col_idxs = std::move(new_col_idxs);                                                  \
        }                                                                                        \
    }                                                                                            \
    template <>                                                                                  \
    void sum_duplicates<C, I>(exec_t exec, size_type, array<C>& values, array<I>& row_idxs,      \
                              array<I>& col_idxs)                                                \
    {                                                                                            \
        const auto nnz = static_cast<int64_t>(values.get_size());                                \
        complex_scratch w(exec, gkoc_compact_workspace_bytes(nnz));                              \
        int64_t kept = 0;                                                                        \
        GKOC_CALL(gkoc_sum_duplicates_count_##IN(stream_of(exec), nnz, row_idxs.get_const_data(),\
                                                 col_idxs.get_const_data(), w.buf.get_data(),    \
                                                 w.buf.get_size(), &kept));                      \
        if (kept < nnz) {                                                                        \
            array<C> new_values{exec, static_cast<size_type>(kept)};                             \
            array<I> new_row_idxs{exec, static_cast<size_type>(kept)};                           \
            array<I> new_col_idxs{exec, static_cast<size_type>(kept)};                           \
            GKOC_CALL(gkoc_sum_duplicates_fill_##TN##_##IN(                                      \
                stream_of(exec), nnz, row_idxs.get_const_data(), col_idxs.get_const_data(),      \
                pairs(values.get_const_data()), w.buf.get_const_data(),                          \
                new_row_idxs.get_data(), new_col_idxs.get_data(), pairs(new_values.get_data())));\
            exec->synchronize();                                                                 \
            values = std::move(new_values);                                                      \
            row_idxs = std::move(new_row_idxs);                                                  \
            col_idxs = std::move(new_col_idxs);                                                  \
        }                                                                                        \
    }
FOR_CT_IT(DEF)
#undef DEF

// |z| of a complex array (a column of n rows), sums and precision changes of complex arrays
#define DEF(C, P, TN, R, RN)                                                                     \
    template <>                                                                                  \
    void inplace_absolute_array<C>(exec_t exec, C* data, size_type n)                            \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_absolute_##TN(stream_of(exec), static_cast<int64_t>(n), 1,         \
                                            pairs(data), 1, nullptr, 0, 0));                     \
    }                                                                                            \
    template <>                                                                                  \
    void outplace_absolute_array<C>(exec_t exec, const C* in, size_type n, R* out)               \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_absolute_##TN(stream_of(exec), static_cast<int64_t>(n), 1,         \
                                            const_cast<P*>(pairs(in)), 1, out, 1, 1));           \
    }                                                                                            \
    template <>                                                                                  \
    void reduce_add_array<C>(exec_t exec, const array<C>& arr, array<C>& val)                    \
    {                                                                                            \
        const auto n = static_cast<int64_t>(arr.get_size());                                     \
        array<C> tmp(exec, 2);                                                                   \
        const auto s = stream_of(exec);                                                          \
        GKOC_CALL(gkoc_fill_array_##TN(s, pairs(tmp.get_data()) + 1, 1, P{1, 0}));               \
        GKOC_CALL(gkoc_cdense_compute_sum_##TN(s, n, 1, pairs(arr.get_const_data()), 1,          \
                                               pairs(tmp.get_data())));                          \
        /* val[0] += 1 * sum */                                                                  \
        GKOC_CALL(gkoc_cdense_add_scaled_##TN(s, 1, 1, tmp.get_const_data() + 1, 1, 0,           \
                                              pairs(tmp.get_const_data()), 1,                    \
                                              pairs(val.get_data()), 1));                        \
        exec->synchronize(); /* the temporaries are released on return */                        \
    }
FOR_CT(DEF)
#undef DEF

template <>
void convert_precision<std::complex<float>, std::complex<double>>(exec_t exec, size_type size,
                                                                  const std::complex<float>* in,
                                                                  std::complex<double>* out)
{
    GKOC_CALL(gkoc_convert_precision_f32_f64(stream_of(exec), 2 * static_cast<int64_t>(size),
                                             reinterpret_cast<const float*>(in),
                                             reinterpret_cast<double*>(out)));
}
template <>
void convert_precision<std::complex<double>, std::complex<float>>(exec_t exec, size_type size,
                                                                  const std::complex<double>* in,
                                                                  std::complex<float>* out)
{
    GKOC_CALL(gkoc_convert_precision_f64_f32(stream_of(exec), 2 * static_cast<int64_t>(size),
                                             reinterpret_cast<const double*>(in),
                                             reinterpret_cast<float*>(out)));
}

}  // namespace components


namespace dense {

#define DEF(C, P, TN, R, RN)                                                                     \
    template <>                                                                                  \
    void compute_norm2<C>(exec_t exec, const matrix::Dense<C>* x, matrix::Dense<R>* result,      \
                          array<char>& tmp)                                                      \
    {                                                                                            \
        const auto s = stream_of(exec);                                                          \
        const size_t bytes = gkoc_reduction_workspace_bytes(rows(x), cols(x), sizeof(R));        \
        if (tmp.get_size() < bytes) tmp.resize_and_reset(bytes);                                 \
        GKOC_CALL(gkoc_dense_compute_norm2_##TN(s, rows(x), cols(x), pairs(x->get_const_values()),\
                                                ld(x), result->get_values(), tmp.get_data(),     \
                                                bytes));                                         \
    }                                                                                            \
    template <>                                                                                  \
    void compute_norm2_dispatch<C>(exec_t exec, const matrix::Dense<C>* x,                       \
                                   matrix::Dense<R>* result, array<char>& tmp)                   \
    {                                                                                            \
        compute_norm2<C>(exec, x, result, tmp);                                                  \
    }                                                                                            \
    template <>                                                                                  \
    void fill<C>(exec_t exec, matrix::Dense<C>* mat, C value)                                    \
    {                                                                                            \
        GKOC_CALL(gkoc_dense_fill_##TN(stream_of(exec), rows(mat), cols(mat),                    \
                                       pairs(mat->get_values()), ld(mat),                        \
                                       P{value.real(), value.imag()}));                          \
    }
FOR_CT(DEF)
#undef DEF

// Dense -> Csr
#define DEF(C, P, TN, R, RN)                                                                     \
    template <>                                                                                  \
    void count_nonzeros_per_row<C, int32>(exec_t exec, const matrix::Dense<C>* source,           \
                                          int32* result)                                         \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_count_nonzeros_per_row_##TN(                                       \
            stream_of(exec), rows(source), cols(source), pairs(source->get_const_values()),      \
            ld(source), result, 4));                                                             \
    }                                                                                            \
    template <>                                                                                  \
    void count_nonzeros_per_row<C, int64>(exec_t exec, const matrix::Dense<C>* source,           \
                                          int64* result)                                         \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_count_nonzeros_per_row_##TN(                                       \
            stream_of(exec), rows(source), cols(source), pairs(source->get_const_values()),      \
            ld(source), result, 8));                                                             \
    }                                                                                            \
    template <>                                                                                  \
    void count_nonzeros_per_row<C, size_type>(exec_t exec, const matrix::Dense<C>* source,       \
                                              size_type* result)                                 \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_count_nonzeros_per_row_##TN(                                       \
            stream_of(exec), rows(source), cols(source), pairs(source->get_const_values()),      \
            ld(source), result, 8));                                                             \
    }
FOR_CT(DEF)
#undef DEF
#define DEF(C, TN, I, IN)                                                                        \
    template <>                                                                                  \
    void convert_to_csr<C, I>(exec_t exec, const matrix::Dense<C>* source,                       \
                              matrix::Csr<C, I>* result)                                         \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_to_csr_##TN##_##IN(                                                \
            stream_of(exec), rows(source), cols(source), pairs(source->get_const_values()),      \
            ld(source), result->get_const_row_ptrs(), result->get_col_idxs(),                    \
            pairs(result->get_values())));                                                       \
    }
FOR_CT_IT(DEF)
#undef DEF

// BLAS-1 on complex columns (csrc/complex_blas.hip); S = C (complex scalars) or R (real scalars)
#define DEF_AXPY(C, P, TN, R, S, IS_REAL)                                                        \
    template <>                                                                                  \
    void scale<C, S>(exec_t exec, const matrix::Dense<S>* alpha, matrix::Dense<C>* x)            \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_scale_##TN(stream_of(exec), rows(x), cols(x),                      \
                                         alpha->get_const_values(), cols(alpha), IS_REAL,        \
                                         pairs(x->get_values()), ld(x)));                        \
    }                                                                                            \
    template <>                                                                                  \
    void inv_scale<C, S>(exec_t exec, const matrix::Dense<S>* alpha, matrix::Dense<C>* x)        \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_inv_scale_##TN(stream_of(exec), rows(x), cols(x),                  \
                                             alpha->get_const_values(), cols(alpha), IS_REAL,    \
                                             pairs(x->get_values()), ld(x)));                    \
    }                                                                                            \
    template <>                                                                                  \
    void add_scaled<C, S>(exec_t exec, const matrix::Dense<S>* alpha, const matrix::Dense<C>* x, \
                          matrix::Dense<C>* y)                                                   \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_add_scaled_##TN(stream_of(exec), rows(y), cols(y),                 \
                                              alpha->get_const_values(), cols(alpha), IS_REAL,   \
                                              pairs(x->get_const_values()), ld(x),               \
                                              pairs(y->get_values()), ld(y)));                   \
    }                                                                                            \
    template <>                                                                                  \
    void sub_scaled<C, S>(exec_t exec, const matrix::Dense<S>* alpha, const matrix::Dense<C>* x, \
                          matrix::Dense<C>* y)                                                   \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_sub_scaled_##TN(stream_of(exec), rows(y), cols(y),                 \
                                              alpha->get_const_values(), cols(alpha), IS_REAL,   \
                                              pairs(x->get_const_values()), ld(x),               \
                                              pairs(y->get_values()), ld(y)));                   \
    }
#define DEF(C, P, TN, R, RN) DEF_AXPY(C, P, TN, R, C, 0) DEF_AXPY(C, P, TN, R, R, 1)
FOR_CT(DEF)
#undef DEF
#undef DEF_AXPY

#define DEF(C, P, TN, R, RN)                                                                     \
    template <>                                                                                  \
    void compute_dot<C>(exec_t exec, const matrix::Dense<C>* x, const matrix::Dense<C>* y,       \
                        matrix::Dense<C>* result, array<char>&)                                  \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_compute_dot_##TN(stream_of(exec), rows(x), cols(x),                \
                                               pairs(x->get_const_values()), ld(x),              \
                                               pairs(y->get_const_values()), ld(y),              \
                                               pairs(result->get_values()), 0));                 \
    }                                                                                            \
    template <>                                                                                  \
    void compute_dot_dispatch<C>(exec_t exec, const matrix::Dense<C>* x,                         \
                                 const matrix::Dense<C>* y, matrix::Dense<C>* result,            \
                                 array<char>& tmp)                                               \
    {                                                                                            \
        compute_dot<C>(exec, x, y, result, tmp);                                                 \
    }                                                                                            \
    template <>                                                                                  \
    void compute_conj_dot<C>(exec_t exec, const matrix::Dense<C>* x, const matrix::Dense<C>* y,  \
                             matrix::Dense<C>* result, array<char>&)                             \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_compute_dot_##TN(stream_of(exec), rows(x), cols(x),                \
                                               pairs(x->get_const_values()), ld(x),              \
                                               pairs(y->get_const_values()), ld(y),              \
                                               pairs(result->get_values()), 1));                 \
    }                                                                                            \
    template <>                                                                                  \
    void compute_conj_dot_dispatch<C>(exec_t exec, const matrix::Dense<C>* x,                    \
                                      const matrix::Dense<C>* y, matrix::Dense<C>* result,       \
                                      array<char>& tmp)                                          \
    {                                                                                            \
        compute_conj_dot<C>(exec, x, y, result, tmp);                                            \
    }                                                                                            \
    template <>                                                                                  \
    void compute_squared_norm2<C>(exec_t exec, const matrix::Dense<C>* x,                        \
                                  matrix::Dense<R>* result, array<char>&)                        \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_compute_squared_norm2_##TN(stream_of(exec), rows(x), cols(x),      \
                                                         pairs(x->get_const_values()), ld(x),    \
                                                         result->get_values()));                 \
    }                                                                                            \
    template <>                                                                                  \
    void compute_mean<C>(exec_t exec, const matrix::Dense<C>* x, matrix::Dense<C>* result,       \
                         array<char>&)                                                           \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_compute_mean_##TN(stream_of(exec), rows(x), cols(x),               \
                                                pairs(x->get_const_values()), ld(x),             \
                                                pairs(result->get_values())));                   \
    }                                                                                            \
    template <>                                                                                  \
    void compute_mean<R>(exec_t exec, const matrix::Dense<R>* x, matrix::Dense<R>* result,       \
                         array<char>&)                                                           \
    {                                                                                            \
        GKOC_CALL(gkoc_dense_compute_mean_##RN(stream_of(exec), rows(x), cols(x),                \
                                               x->get_const_values(), ld(x),                     \
                                               result->get_values()));                           \
    }                                                                                            \
    template <>                                                                                  \
    void make_complex<R>(exec_t exec, const matrix::Dense<R>* source, matrix::Dense<C>* result)  \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_convert_##TN(stream_of(exec), rows(source), cols(source),          \
                                           source->get_const_values(), ld(source),               \
                                           result->get_values(), ld(result), 0));                \
    }                                                                                            \
    template <>                                                                                  \
    void make_complex<C>(exec_t exec, const matrix::Dense<C>* source, matrix::Dense<C>* result)  \
    {                                                                                            \
        GKOC_CALL(gkoc_dense_copy_##RN(stream_of(exec), rows(source), 2 * cols(source),          \
                                       reinterpret_cast<const R*>(source->get_const_values()),   \
                                       2 * ld(source),                                           \
                                       reinterpret_cast<R*>(result->get_values()),               \
                                       2 * ld(result)));                                         \
    }                                                                                            \
    template <>                                                                                  \
    void get_real<C>(exec_t exec, const matrix::Dense<C>* source, matrix::Dense<R>* result)      \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_convert_##TN(stream_of(exec), rows(source), cols(source),          \
                                           source->get_const_values(), ld(source),               \
                                           result->get_values(), ld(result), 1));                \
    }                                                                                            \
    template <>                                                                                  \
    void get_imag<C>(exec_t exec, const matrix::Dense<C>* source, matrix::Dense<R>* result)      \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_convert_##TN(stream_of(exec), rows(source), cols(source),          \
                                           source->get_const_values(), ld(source),               \
                                           result->get_values(), ld(result), 2));                \
    }                                                                                            \
    template <>                                                                                  \
    void conj_transpose<C>(exec_t exec, const matrix::Dense<C>* orig, matrix::Dense<C>* trans)   \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_convert_##TN(stream_of(exec), rows(orig), cols(orig),              \
                                           orig->get_const_values(), ld(orig),                   \
                                           trans->get_values(), ld(trans), 3));                  \
    }
FOR_CT(DEF)
#undef DEF

#define DEF(C, P, TN, R, RN)                                                                     \
    template <>                                                                                  \
    void inplace_absolute_dense<C>(exec_t exec, matrix::Dense<C>* source)                        \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_absolute_##TN(stream_of(exec), rows(source), cols(source),         \
                                            pairs(source->get_values()), ld(source), nullptr, 0, \
                                            0));                                                 \
    }                                                                                            \
    template <>                                                                                  \
    void outplace_absolute_dense<C>(exec_t exec, const matrix::Dense<C>* source,                 \
                                    matrix::Dense<R>* result)                                    \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_absolute_##TN(stream_of(exec), rows(source), cols(source),         \
                                            const_cast<P*>(pairs(source->get_const_values())),   \
                                            ld(source), result->get_values(), ld(result), 1));   \
    }                                                                                            \
    template <>                                                                                  \
    void compute_norm1<C>(exec_t exec, const matrix::Dense<C>* x, matrix::Dense<R>* result,      \
                          array<char>&)                                                          \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_compute_norm1_##TN(stream_of(exec), rows(x), cols(x),              \
                                                 pairs(x->get_const_values()), ld(x),            \
                                                 result->get_values()));                         \
    }
FOR_CT(DEF)
#undef DEF

#define DEF(C, TN, I, IN)                                                                        \
    template <>                                                                                  \
    void row_gather<C, C, I>(exec_t exec, const I* gather_indices, const matrix::Dense<C>* orig, \
                             matrix::Dense<C>* row_collection)                                   \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_row_gather_##TN##_##IN(                                            \
            stream_of(exec), rows(row_collection), cols(orig), gather_indices,                   \
            pairs(orig->get_const_values()), ld(orig), pairs(row_collection->get_values()),      \
            ld(row_collection)));                                                                \
    }                                                                                            \
    template <>                                                                                  \
    void fill_in_matrix_data<C, I>(exec_t exec, const device_matrix_data<C, I>& data,            \
                                   matrix::Dense<C>* output)                                     \
    {                                                                                            \
        GKOC_CALL(gkoc_cdense_fill_in_matrix_data_##TN##_##IN(                                   \
            stream_of(exec), static_cast<int64_t>(data.get_num_stored_elements()),               \
            data.get_const_row_idxs(), data.get_const_col_idxs(), pairs(data.get_const_values()),\
            pairs(output->get_values()), ld(output)));                                           \
    }
FOR_CT_IT(DEF)
#undef DEF

}  // namespace dense


namespace csr {

// Csr with complex values: plain kernels (csrc/complex_blas.hip), so that Ginkgo's distributed
// Matrix / Schwarz work for every value type their tests instantiate
#define DEF(C, TN, I, IN)                                                                        \
    template <>                                                                                  \
    void spmv<C, C, C, I>(exec_t exec, const matrix::Csr<C, I>* a, const matrix::Dense<C>* b,    \
                          matrix::Dense<C>* c)                                                   \
    {                                                                                            \
        GKOC_CALL(gkoc_ccsr_spmv_##TN##_##IN(                                                    \
            stream_of(exec), rows(c), cols(c), a->get_const_row_ptrs(), a->get_const_col_idxs(), \
            pairs(a->get_const_values()), nullptr, pairs(b->get_const_values()), ld(b), nullptr, \
            pairs(c->get_values()), ld(c)));                                                     \
    }                                                                                            \
    template <>                                                                                  \
    void advanced_spmv<C, C, C, I>(exec_t exec, const matrix::Dense<C>* alpha,                   \
                                   const matrix::Csr<C, I>* a, const matrix::Dense<C>* b,        \
                                   const matrix::Dense<C>* beta, matrix::Dense<C>* c)            \
    {                                                                                            \
        GKOC_CALL(gkoc_ccsr_spmv_##TN##_##IN(                                                    \
            stream_of(exec), rows(c), cols(c), a->get_const_row_ptrs(), a->get_const_col_idxs(), \
            pairs(a->get_const_values()), pairs(alpha->get_const_values()),                      \
            pairs(b->get_const_values()), ld(b), pairs(beta->get_const_values()),                \
            pairs(c->get_values()), ld(c)));                                                     \
    }                                                                                            \
    template <>                                                                                  \
    void extract_diagonal<C, I>(exec_t exec, const matrix::Csr<C, I>* orig,                      \
                                matrix::Diagonal<C>* diag)                                       \
    {                                                                                            \
        GKOC_CALL(gkoc_ccsr_row_scan_##TN##_##IN(                                                \
            stream_of(exec), static_cast<int64_t>(diag->get_size()[0]),                          \
            orig->get_const_row_ptrs(), orig->get_const_col_idxs(),                              \
            pairs(orig->get_const_values()), pairs(diag->get_values()), 0));                     \
    }                                                                                            \
    template <>                                                                                  \
    void row_wise_absolute_sum<C, I>(exec_t exec, const matrix::Csr<C, I>* orig, array<C>& sum)  \
    {                                                                                            \
        GKOC_CALL(gkoc_ccsr_row_scan_##TN##_##IN(                                                \
            stream_of(exec), static_cast<int64_t>(orig->get_size()[0]),                          \
            orig->get_const_row_ptrs(), orig->get_const_col_idxs(),                              \
            pairs(orig->get_const_values()), pairs(sum.get_data()), 1));                         \
    }
FOR_CT_IT(DEF)
#undef DEF

#define DEF(C, TN, I, IN)                                                                        \
    template <>                                                                                  \
    void is_sorted_by_column_index<C, I>(exec_t exec, const matrix::Csr<C, I>* to_check,         \
                                         bool* is_sorted)                                        \
    {                                                                                            \
        int flag = 1; /* a question about the column indices only */                             \
        GKOC_CALL(gkoc_csr_is_sorted_by_column_index_f64_##IN(                                   \
            stream_of(exec), to_check->get_size()[0], to_check->get_const_row_ptrs(),            \
            to_check->get_const_col_idxs(), &flag));                                             \
        *is_sorted = flag != 0;                                                                  \
    }                                                                                            \
    template <>                                                                                  \
    void sort_by_column_index<C, I>(exec_t exec, matrix::Csr<C, I>* to_sort)                     \
    {                                                                                            \
        GKOC_CALL(gkoc_csr_sort_by_column_index_##TN##_##IN(                                     \
            stream_of(exec), to_sort->get_size()[0], to_sort->get_const_row_ptrs(),              \
            to_sort->get_col_idxs(), pairs(to_sort->get_values())));                             \
    }                                                                                            \
    static void ctranspose_impl_##TN##_##IN(exec_t exec, const matrix::Csr<C, I>* orig,          \
                                            matrix::Csr<C, I>* trans, bool conjugate)            \
    {                                                                                            \
        const int64_t nnz = orig->get_num_stored_elements();                                     \
        array<char> work(exec, gkoc_csr_transpose_workspace_bytes(nnz, orig->get_size()[1],      \
                                                                  sizeof(I)));                   \
        GKOC_CALL(gkoc_csr_transpose_##TN##_##IN(                                                \
            stream_of(exec), orig->get_size()[0], orig->get_size()[1],                           \
            orig->get_const_row_ptrs(), orig->get_const_col_idxs(),                              \
            pairs(orig->get_const_values()), nnz, trans->get_row_ptrs(), trans->get_col_idxs(),  \
            pairs(trans->get_values()), work.get_data(), work.get_size()));                      \
        if (conjugate && nnz > 0) {                                                              \
            /* conj of the nnz x 1 column, in place */                                           \
            GKOC_CALL(gkoc_cdense_convert_##TN(stream_of(exec), nnz, 1, trans->get_const_values(), \
                                               1, trans->get_values(), nnz, 3));                 \
        }                                                                                        \
        exec->synchronize(); /* work is released on return */                                    \
    }                                                                                            \
    template <>                                                                                  \
    void transpose<C, I>(exec_t exec, const matrix::Csr<C, I>* orig, matrix::Csr<C, I>* trans)   \
    {                                                                                            \
        ctranspose_impl_##TN##_##IN(exec, orig, trans, false);                                   \
    }                                                                                            \
    template <>                                                                                  \
    void conj_transpose<C, I>(exec_t exec, const matrix::Csr<C, I>* orig,                        \
                              matrix::Csr<C, I>* trans)                                          \
    {                                                                                            \
        ctranspose_impl_##TN##_##IN(exec, orig, trans, true);                                    \
    }
FOR_CT_IT(DEF)
#undef DEF

#define DEF(T, TN, I, IN)                                                                        \
    template <>                                                                                  \
    void row_wise_absolute_sum<T, I>(exec_t exec, const matrix::Csr<T, I>* orig, array<T>& sum)  \
    {                                                                                            \
        GKOC_CALL(gkoc_csr_row_wise_absolute_sum_##TN##_##IN(                                    \
            stream_of(exec), static_cast<int64_t>(orig->get_size()[0]),                          \
            orig->get_const_row_ptrs(), orig->get_const_values(), sum.get_data()));              \
    }
DEF(double, f64, int32, i32)
DEF(double, f64, int64, i64)
DEF(float, f32, int32, i32)
DEF(float, f32, int64, i64)
#undef DEF

}  // namespace csr


namespace diagonal {

#define DEF(C, P, TN, R, RN)                                                                     \
    template <>                                                                                  \
    void conj_transpose<C>(exec_t exec, const matrix::Diagonal<C>* orig,                         \
                           matrix::Diagonal<C>* trans)                                           \
    {                                                                                            \
        const auto n = static_cast<int64_t>(orig->get_size()[0]);                                \
        GKOC_CALL(gkoc_cdense_convert_##TN(stream_of(exec), n, 1, orig->get_const_values(), 1,   \
                                           trans->get_values(), n, 3));                          \
    }
FOR_CT(DEF)
#undef DEF

}  // namespace diagonal


namespace diagonal {

#define DEF(C, TN, I, IN)                                                                        \
    template <>                                                                                  \
    void apply_to_csr<C, I>(exec_t exec, const matrix::Diagonal<C>* a, const matrix::Csr<C, I>* b, \
                            matrix::Csr<C, I>* c, bool inverse)                                  \
    {                                                                                            \
        c->copy_from(b);                                                                         \
        GKOC_CALL(gkoc_ccsr_scale_by_diagonal_##TN##_##IN(                                       \
            stream_of(exec), static_cast<int64_t>(c->get_size()[0]), c->get_const_row_ptrs(),    \
            c->get_const_col_idxs(), pairs(a->get_const_values()), inverse ? 1 : 0,              \
            pairs(c->get_values())));                                                            \
    }                                                                                            \
    template <>                                                                                  \
    void convert_to_csr<C, I>(exec_t exec, const matrix::Diagonal<C>* source,                    \
                              matrix::Csr<C, I>* result)                                         \
    {                                                                                            \
        /* row i holds the one entry (i, i): pointers and columns count up, the values move */   \
        const auto n = static_cast<int64_t>(source->get_size()[0]);                              \
        const auto s = stream_of(exec);                                                          \
        GKOC_CALL(gkoc_fill_seq_array_##IN(s, result->get_row_ptrs(), n + 1));                   \
        GKOC_CALL(gkoc_fill_seq_array_##IN(s, result->get_col_idxs(), n));                       \
        GKOC_CALL(gkoc_memcpy_d2d(result->get_values(), source->get_const_values(),              \
                                  sizeof(C) * static_cast<size_t>(n), s));                       \
    }                                                                                            \
    template <>                                                                                  \
    void right_apply_to_csr<C, I>(exec_t exec, const matrix::Diagonal<C>* a,                     \
                                  const matrix::Csr<C, I>* b, matrix::Csr<C, I>* c)              \
    {                                                                                            \
        c->copy_from(b);                                                                         \
        GKOC_CALL(gkoc_ccsr_scale_by_diagonal_##TN##_##IN(                                       \
            stream_of(exec), static_cast<int64_t>(c->get_size()[0]), c->get_const_row_ptrs(),    \
            c->get_const_col_idxs(), pairs(a->get_const_values()), 2, pairs(c->get_values())));  \
    }
FOR_CT_IT(DEF)
#undef DEF

}  // namespace diagonal


namespace jacobi {

#define DEF(C, P, TN, R, RN)                                                                     \
    template <>                                                                                  \
    void invert_diagonal<C>(exec_t exec, const array<C>& diag, array<C>& inv_diag)               \
    {                                                                                            \
        GKOC_CALL(gkoc_cjacobi_invert_diagonal_##TN(stream_of(exec),                             \
                                                    static_cast<int64_t>(diag.get_size()),       \
                                                    pairs(diag.get_const_data()),                \
                                                    pairs(inv_diag.get_data())));                \
    }                                                                                            \
    template <>                                                                                  \
    void simple_scalar_apply<C>(exec_t exec, const array<C>& diag, const matrix::Dense<C>* b,    \
                                matrix::Dense<C>* x)                                             \
    {                                                                                            \
        GKOC_CALL(gkoc_cjacobi_scalar_apply_##TN(stream_of(exec), rows(x), cols(x),              \
                                                 pairs(diag.get_const_data()), nullptr,          \
                                                 pairs(b->get_const_values()), ld(b), nullptr,   \
                                                 pairs(x->get_values()), ld(x)));                \
    }                                                                                            \
    template <>                                                                                  \
    void scalar_apply<C>(exec_t exec, const array<C>& diag, const matrix::Dense<C>* alpha,       \
                         const matrix::Dense<C>* b, const matrix::Dense<C>* beta,                \
                         matrix::Dense<C>* x)                                                    \
    {                                                                                            \
        GKOC_CALL(gkoc_cjacobi_scalar_apply_##TN(                                                \
            stream_of(exec), rows(x), cols(x), pairs(diag.get_const_data()),                     \
            pairs(alpha->get_const_values()), pairs(b->get_const_values()), ld(b),               \
            pairs(beta->get_const_values()), pairs(x->get_values()), ld(x)));                    \
    }                                                                                            \
    template <>                                                                                  \
    void scalar_conj<C>(exec_t exec, const array<C>& diag, array<C>& conj_diag)                  \
    {                                                                                            \
        const auto n = static_cast<int64_t>(diag.get_size());                                    \
        GKOC_CALL(gkoc_cdense_convert_##TN(stream_of(exec), n, 1, diag.get_const_data(), 1,      \
                                           conj_diag.get_data(), n, 3));                         \
    }
FOR_CT(DEF)
#undef DEF

}  // namespace jacobi


namespace coo {

#define DEF(C, TN, I, IN)                                                                        \
    template <>                                                                                  \
    void spmv2<C, I>(exec_t exec, const matrix::Coo<C, I>* a, const matrix::Dense<C>* b,         \
                     matrix::Dense<C>* c)                                                        \
    {                                                                                            \
        GKOC_CALL(gkoc_ccoo_spmv2_##TN##_##IN(                                                   \
            stream_of(exec), static_cast<int64_t>(a->get_num_stored_elements()), cols(b),        \
            a->get_const_row_idxs(), a->get_const_col_idxs(), pairs(a->get_const_values()),      \
            nullptr, pairs(b->get_const_values()), ld(b), pairs(c->get_values()), ld(c)));       \
    }                                                                                            \
    template <>                                                                                  \
    void advanced_spmv2<C, I>(exec_t exec, const matrix::Dense<C>* alpha,                        \
                              const matrix::Coo<C, I>* a, const matrix::Dense<C>* b,             \
                              matrix::Dense<C>* c)                                               \
    {                                                                                            \
        GKOC_CALL(gkoc_ccoo_spmv2_##TN##_##IN(                                                   \
            stream_of(exec), static_cast<int64_t>(a->get_num_stored_elements()), cols(b),        \
            a->get_const_row_idxs(), a->get_const_col_idxs(), pairs(a->get_const_values()),      \
            pairs(alpha->get_const_values()), pairs(b->get_const_values()), ld(b),               \
            pairs(c->get_values()), ld(c)));                                                     \
    }                                                                                            \
    template <>                                                                                  \
    void spmv<C, I>(exec_t exec, const matrix::Coo<C, I>* a, const matrix::Dense<C>* b,          \
                    matrix::Dense<C>* c)                                                         \
    {                                                                                            \
        dense::fill<C>(exec, c, C{});                                                            \
        spmv2<C, I>(exec, a, b, c);                                                              \
    }                                                                                            \
    template <>                                                                                  \
    void advanced_spmv<C, I>(exec_t exec, const matrix::Dense<C>* alpha,                         \
                             const matrix::Coo<C, I>* a, const matrix::Dense<C>* b,              \
                             const matrix::Dense<C>* beta, matrix::Dense<C>* c)                  \
    {                                                                                            \
        dense::scale<C, C>(exec, beta, c);                                                       \
        advanced_spmv2<C, I>(exec, alpha, a, b, c);                                              \
    }
FOR_CT_IT(DEF)
#undef DEF

}  // namespace coo


namespace implicit_residual_norm {

// sqrt(|tau|) <= goal * orig_tau with a complex tau: |tau| first, then the real criterion
#define DEF(C, P, TN, R, RN)                                                                     \
    template <>                                                                                  \
    void implicit_residual_norm<C>(exec_t exec, const matrix::Dense<C>* tau,                     \
                                   const matrix::Dense<R>* orig_tau, R rel_residual_goal,        \
                                   uint8 stoppingId, bool setFinalized,                          \
                                   array<stopping_status>* stop_status,                          \
                                   array<bool>* device_storage, bool* all_converged,             \
                                   bool* one_changed)                                            \
    {                                                                                            \
        if (device_storage->get_size() < 2) device_storage->resize_and_reset(2);                 \
        array<R> mod(exec, tau->get_size()[1]);                                                  \
        GKOC_CALL(gkoc_cdense_absolute_##TN(stream_of(exec), 1, cols(tau),                       \
                                            const_cast<P*>(pairs(tau->get_const_values())),      \
                                            ld(tau), mod.get_data(), cols(tau), 1));             \
        int allc = 0, chg = 0;                                                                   \
        GKOC_CALL(gkoc_implicit_residual_norm_##RN(                                              \
            stream_of(exec), cols(tau), mod.get_const_data(), orig_tau->get_const_values(),      \
            rel_residual_goal, stoppingId, setFinalized ? 1 : 0, cdna4::raw(stop_status),        \
            reinterpret_cast<uint8_t*>(device_storage->get_data()), &allc, &chg));               \
        *all_converged = allc != 0;                                                              \
        *one_changed = chg != 0;                                                                 \
    }
FOR_CT(DEF)
#undef DEF

}  // namespace implicit_residual_norm


}  // namespace hip
}  // namespace kernels
}  // namespace gko
