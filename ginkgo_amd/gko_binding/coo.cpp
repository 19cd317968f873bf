// gko::kernels::hip::coo::{spmv, advanced_spmv, spmv2, advanced_spmv2},
// hybrid::compute_coo_row_ptrs, csr::convert_to_hybrid and
// components::convert_ptrs_to_idxs forwarded to the C ABI (csrc/coo.hip).
// With these, Ginkgo's own Coo and Hybrid (Ell + Coo) matrices apply on this
// backend, and Csr -> Coo / Hybrid conversions run on the device.
#include <complex>

#include <ginkgo/core/base/device_matrix_data.hpp>
#include <ginkgo/core/matrix/coo.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/ell.hpp>
#include <ginkgo/core/matrix/hybrid.hpp>

#include "core/components/format_conversion_kernels.hpp"
#include "core/matrix/coo_kernels.hpp"
#include "core/matrix/csr_kernels.hpp"
#include "core/matrix/ell_kernels.hpp"
#include "core/matrix/hybrid_kernels.hpp"
#include "shim_common.hpp"

namespace gko {
namespace kernels {
namespace hip {

using cdna4::cols;
using cdna4::ld;
using cdna4::rows;
using cdna4::stream_of;
using exec_t = std::shared_ptr<const HipExecutor>;

#define FOR_VT_IT(M)                                                     \
    M(double, f64, int32, i32) M(double, f64, int64, i64) M(float, f32, int32, i32) \
        M(float, f32, int64, i64)


namespace coo {

// workspace of one call; released after the stream has drained (array<char> frees
// on destruction, the kernels are asynchronous)
struct scratch {
    exec_t exec;
    array<char> buf;
    scratch(exec_t e, size_t bytes) : exec{e}, buf{e, bytes} {}
    ~scratch() { exec->synchronize(); }
};

#define DEF(T, TN, I, IN)                                                                 \
    template <>                                                                           \
    void spmv<T, I>(exec_t exec, const matrix::Coo<T, I>* a, const matrix::Dense<T>* b,   \
                    matrix::Dense<T>* c)                                                  \
    {                                                                                     \
        const int64_t n = a->get_size()[0];                                               \
        scratch w(exec, gkoc_coo_workspace_bytes(n, sizeof(I), sizeof(T)));               \
        GKOC_CALL(gkoc_coo_spmv_##TN##_##IN(                                              \
            stream_of(exec), n, a->get_size()[1], a->get_num_stored_elements(),           \
            a->get_const_row_idxs(), a->get_const_col_idxs(), a->get_const_values(),      \
            b->get_const_values(), ld(b), c->get_values(), ld(c), cols(b),                \
            w.buf.get_data(), w.buf.get_size()));                                         \
    }                                                                                     \
    template <>                                                                           \
    void advanced_spmv<T, I>(exec_t exec, const matrix::Dense<T>* alpha,                  \
                             const matrix::Coo<T, I>* a, const matrix::Dense<T>* b,       \
                             const matrix::Dense<T>* beta, matrix::Dense<T>* c)           \
    {                                                                                     \
        const int64_t n = a->get_size()[0];                                               \
        scratch w(exec, gkoc_coo_workspace_bytes(n, sizeof(I), sizeof(T)));               \
        GKOC_CALL(gkoc_coo_advanced_spmv_##TN##_##IN(                                     \
            stream_of(exec), n, a->get_size()[1], a->get_num_stored_elements(),           \
            alpha->get_const_values(), a->get_const_row_idxs(), a->get_const_col_idxs(),  \
            a->get_const_values(), b->get_const_values(), ld(b),                          \
            beta->get_const_values(), c->get_values(), ld(c), cols(b), w.buf.get_data(),  \
            w.buf.get_size()));                                                           \
    }                                                                                     \
    template <>                                                                           \
    void spmv2<T, I>(exec_t exec, const matrix::Coo<T, I>* a, const matrix::Dense<T>* b,  \
                     matrix::Dense<T>* c)                                                 \
    {                                                                                     \
        const int64_t n = a->get_size()[0];                                               \
        scratch w(exec, gkoc_coo_workspace_bytes(n, sizeof(I), sizeof(T)));               \
        GKOC_CALL(gkoc_coo_spmv2_##TN##_##IN(                                             \
            stream_of(exec), n, a->get_size()[1], a->get_num_stored_elements(),           \
            a->get_const_row_idxs(), a->get_const_col_idxs(), a->get_const_values(),      \
            b->get_const_values(), ld(b), c->get_values(), ld(c), cols(b),                \
            w.buf.get_data(), w.buf.get_size()));                                         \
    }                                                                                     \
    template <>                                                                           \
    void advanced_spmv2<T, I>(exec_t exec, const matrix::Dense<T>* alpha,                 \
                              const matrix::Coo<T, I>* a, const matrix::Dense<T>* b,      \
                              matrix::Dense<T>* c)                                        \
    {                                                                                     \
        const int64_t n = a->get_size()[0];                                               \
        scratch w(exec, gkoc_coo_workspace_bytes(n, sizeof(I), sizeof(T)));               \
        GKOC_CALL(gkoc_coo_advanced_spmv2_##TN##_##IN(                                    \
            stream_of(exec), n, a->get_size()[1], a->get_num_stored_elements(),           \
            alpha->get_const_values(), a->get_const_row_idxs(), a->get_const_col_idxs(),  \
            a->get_const_values(), b->get_const_values(), ld(b), c->get_values(), ld(c),  \
            cols(b), w.buf.get_data(), w.buf.get_size()));                                \
    }
FOR_VT_IT(DEF)
#undef DEF

}  // namespace coo


namespace ell {

#define DEF(T, TN, I, IN)                                                                 \
    template <>                                                                           \
    void copy<T, I>(exec_t exec, const matrix::Ell<T, I>* source, matrix::Ell<T, I>* result) \
    {                                                                                     \
        GKOC_CALL(gkoc_ell_copy_##TN##_##IN(                                              \
            stream_of(exec), source->get_size()[0],                                       \
            source->get_num_stored_elements_per_row(), source->get_stride(),              \
            source->get_const_col_idxs(), source->get_const_values(), result->get_stride(), \
            result->get_col_idxs(), result->get_values()));                               \
    }
FOR_VT_IT(DEF)
DEF(std::complex<double>, c128, int32, i32)
DEF(std::complex<double>, c128, int64, i64)
DEF(std::complex<float>, c64, int32, i32)
DEF(std::complex<float>, c64, int64, i64)
#undef DEF

}  // namespace ell


namespace hybrid {

void compute_coo_row_ptrs(exec_t exec, const array<size_type>& row_nnz, size_type ell_lim,
                          int64* coo_row_ptrs)
{
    GKOC_CALL(gkoc_hybrid_compute_coo_row_ptrs(
        stream_of(exec), static_cast<int64_t>(row_nnz.get_size()),
        reinterpret_cast<const uint64_t*>(row_nnz.get_const_data()), ell_lim, coo_row_ptrs));
}

void compute_row_nnz(exec_t exec, const array<int64>& row_ptrs, size_type* row_nnzs)
{
    GKOC_CALL(gkoc_convert_ptrs_to_sizes_i64(
        stream_of(exec), static_cast<int64_t>(row_ptrs.get_size()) - 1, row_ptrs.get_const_data(),
        reinterpret_cast<uint64_t*>(row_nnzs)));
}

#define DEF(T, TN, I, IN)                                                                 \
    template <>                                                                           \
    void fill_in_matrix_data<T, I>(exec_t exec, const device_matrix_data<T, I>& data,     \
                                   const int64* row_ptrs, const int64* coo_row_ptrs,      \
                                   matrix::Hybrid<T, I>* result)                          \
    {                                                                                     \
        const auto n = result->get_size()[0];                                             \
        array<I> narrowed(exec);                                                          \
        const I* ptrs = nullptr;                                                          \
        if (sizeof(I) == sizeof(int64)) {                                                 \
            ptrs = reinterpret_cast<const I*>(row_ptrs);                                  \
        } else {                                                                          \
            narrowed.resize_and_reset(n + 1);                                             \
            GKOC_CALL(gkoc_narrow_i64_to_i32(stream_of(exec), static_cast<int64_t>(n + 1), \
                                             row_ptrs,                                    \
                                             reinterpret_cast<int32_t*>(narrowed.get_data()))); \
            ptrs = narrowed.get_const_data();                                             \
        }                                                                                 \
        GKOC_CALL(gkoc_csr_convert_to_hybrid_##TN##_##IN(                                 \
            stream_of(exec), n, ptrs, data.get_const_col_idxs(), data.get_const_values(), \
            result->get_ell_num_stored_elements_per_row(), result->get_ell_stride(),      \
            result->get_ell_col_idxs(), result->get_ell_values(), coo_row_ptrs,           \
            result->get_coo_row_idxs(), result->get_coo_col_idxs(),                       \
            result->get_coo_values()));                                                   \
        exec->synchronize();                                                              \
    }
FOR_VT_IT(DEF)
DEF(std::complex<double>, c128, int32, i32)
DEF(std::complex<double>, c128, int64, i64)
DEF(std::complex<float>, c64, int32, i32)
DEF(std::complex<float>, c64, int64, i64)
#undef DEF

}  // namespace hybrid


namespace csr {

#define DEF(T, TN, I, IN)                                                                 \
    template <>                                                                           \
    void convert_to_hybrid<T, I>(exec_t exec, const matrix::Csr<T, I>* source,            \
                                 const int64* coo_row_ptrs, matrix::Hybrid<T, I>* result) \
    {                                                                                     \
        GKOC_CALL(gkoc_csr_convert_to_hybrid_##TN##_##IN(                                 \
            stream_of(exec), source->get_size()[0], source->get_const_row_ptrs(),         \
            source->get_const_col_idxs(), source->get_const_values(),                     \
            result->get_ell_num_stored_elements_per_row(), result->get_ell_stride(),      \
            result->get_ell_col_idxs(), result->get_ell_values(), coo_row_ptrs,           \
            result->get_coo_row_idxs(), result->get_coo_col_idxs(),                       \
            result->get_coo_values()));                                                   \
    }
FOR_VT_IT(DEF)
DEF(std::complex<double>, c128, int32, i32)
DEF(std::complex<double>, c128, int64, i64)
DEF(std::complex<float>, c64, int32, i32)
DEF(std::complex<float>, c64, int64, i64)
#undef DEF

// transpose / conj_transpose (real value types: the same operation)
#define DEF(T, TN, I, IN)                                                                 \
    static void transpose_impl_##TN##_##IN(exec_t exec, const matrix::Csr<T, I>* orig,    \
                                           matrix::Csr<T, I>* trans)                      \
    {                                                                                     \
        const int64_t nnz = orig->get_num_stored_elements();                              \
        array<char> work(exec, gkoc_csr_transpose_workspace_bytes(                        \
                                   nnz, orig->get_size()[1], sizeof(I)));                 \
        GKOC_CALL(gkoc_csr_transpose_##TN##_##IN(                                         \
            stream_of(exec), orig->get_size()[0], orig->get_size()[1],                    \
            orig->get_const_row_ptrs(), orig->get_const_col_idxs(),                       \
            orig->get_const_values(), nnz, trans->get_row_ptrs(), trans->get_col_idxs(),  \
            trans->get_values(), work.get_data(), work.get_size()));                      \
        exec->synchronize(); /* work is released on return */                             \
    }                                                                                     \
    template <>                                                                           \
    void transpose<T, I>(exec_t exec, const matrix::Csr<T, I>* orig,                      \
                         matrix::Csr<T, I>* trans)                                        \
    {                                                                                     \
        transpose_impl_##TN##_##IN(exec, orig, trans);                                    \
    }                                                                                     \
    template <>                                                                           \
    void conj_transpose<T, I>(exec_t exec, const matrix::Csr<T, I>* orig,                 \
                              matrix::Csr<T, I>* trans)                                   \
    {                                                                                     \
        transpose_impl_##TN##_##IN(exec, orig, trans);                                    \
    }
FOR_VT_IT(DEF)
#undef DEF

}  // namespace csr


namespace components {

template <>
void convert_ptrs_to_idxs<int32, int32>(exec_t exec, const int32* ptrs, size_type num_blocks,
                                        int32* idxs)
{
    GKOC_CALL(gkoc_convert_ptrs_to_idxs_i32(stream_of(exec), ptrs, num_blocks, idxs));
}
template <>
void convert_ptrs_to_idxs<int64, int64>(exec_t exec, const int64* ptrs, size_type num_blocks,
                                        int64* idxs)
{
    GKOC_CALL(gkoc_convert_ptrs_to_idxs_i64(stream_of(exec), ptrs, num_blocks, idxs));
}

}  // namespace components

}  // namespace hip
}  // namespace kernels
}  // namespace gko
