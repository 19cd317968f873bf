// gko::kernels::hip for std::complex<float / double>, as far as this backend goes: the kernels that
// move or measure complex data without multiplying it - device_matrix_data assembly, fill_array /
// fill_seq_array, and the column 2-norms of a complex Dense (what stop::ResidualNorm asks for).
// std::complex<R> is layout-compatible with the pair {R re, im} of the C ABI (gkoc_c128 / gkoc_c64).
// Everything else complex stays with Ginkgo's NotCompiled stubs.
#include <complex>

#include <ginkgo/core/base/device_matrix_data.hpp>
#include <ginkgo/core/base/matrix_data.hpp>
#include <ginkgo/core/matrix/dense.hpp>

#include "core/base/device_matrix_data_kernels.hpp"
#include "core/components/fill_array_kernels.hpp"
#include "core/matrix/dense_kernels.hpp"
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

}  // namespace dense


}  // namespace hip
}  // namespace kernels
}  // namespace gko
