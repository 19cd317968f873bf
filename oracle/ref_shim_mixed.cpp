// TEST INFRASTRUCTURE -- C-callable shim over the UNMODIFIED reference built with
// GINKGO_MIXED_PRECISION (oracle/_ref/mixed, oracle/build_ref_mixed.py): Csr / Ell apply on
// gko::ReferenceExecutor for a (matrix, input, output) triple of float / double.  Used to pin
// oracle/gko_oracle_mixed.inc and to generate tests/golden/mixed_spmv.npz
// (tests/golden/make_mixed_golden.py).  Nothing here is part of the product.
#include <cstdint>
#include <memory>

#include <ginkgo/core/base/array.hpp>
#include <ginkgo/core/base/executor.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/matrix/ell.hpp>

namespace {

template <typename T>
std::unique_ptr<gko::matrix::Dense<T>> dense_view(std::shared_ptr<const gko::Executor> exec, gko::size_type rows,
                                                  gko::size_type cols, void* data)
{
    return gko::matrix::Dense<T>::create(exec, gko::dim<2>{rows, cols},
                                         gko::make_array_view(exec, rows * cols, static_cast<T*>(data)), cols);
}

// fmt 0: Csr (ptrs = row_ptrs, k / stride unused); 1: Ell (ptrs unused)
template <typename MT, typename IT, typename OT, typename I>
int run(int fmt, int64_t n_rows, int64_t n_cols, int64_t k, int64_t stride, void* ptrs, void* cols, void* vals,
        const void* alpha, void* b, int64_t nrhs, const void* beta, void* c)
{
    auto exec = gko::ReferenceExecutor::create();
    const gko::dim<2> size{static_cast<gko::size_type>(n_rows), static_cast<gko::size_type>(n_cols)};
    std::unique_ptr<gko::LinOp> a;
    if (fmt == 0) {
        const auto nnz = static_cast<gko::size_type>(static_cast<I*>(ptrs)[n_rows]);
        a = gko::matrix::Csr<MT, I>::create(exec, size, gko::make_array_view(exec, nnz, static_cast<MT*>(vals)),
                                            gko::make_array_view(exec, nnz, static_cast<I*>(cols)),
                                            gko::make_array_view(exec, n_rows + 1, static_cast<I*>(ptrs)));
    } else {
        const auto total = static_cast<gko::size_type>(k * stride);
        a = gko::matrix::Ell<MT, I>::create(exec, size, gko::make_array_view(exec, total, static_cast<MT*>(vals)),
                                            gko::make_array_view(exec, total, static_cast<I*>(cols)), k, stride);
    }
    auto bv = dense_view<IT>(exec, n_cols, nrhs, b);
    auto cv = dense_view<OT>(exec, n_rows, nrhs, c);
    if (alpha) {
        MT av = *static_cast<const MT*>(alpha);
        OT bvl = *static_cast<const OT*>(beta);
        auto al = dense_view<MT>(exec, 1, 1, &av);
        auto be = dense_view<OT>(exec, 1, 1, &bvl);
        a->apply(al, bv, be, cv);
    } else {
        a->apply(bv, cv);
    }
    return 0;
}

template <typename MT, typename IT, typename I>
int pick_o(int ot, int fmt, int64_t n_rows, int64_t n_cols, int64_t k, int64_t stride, void* ptrs, void* cols,
           void* vals, const void* alpha, void* b, int64_t nrhs, const void* beta, void* c)
{
    return ot == 0 ? run<MT, IT, double, I>(fmt, n_rows, n_cols, k, stride, ptrs, cols, vals, alpha, b, nrhs, beta, c)
                   : run<MT, IT, float, I>(fmt, n_rows, n_cols, k, stride, ptrs, cols, vals, alpha, b, nrhs, beta, c);
}

template <typename MT, typename I>
int pick_i(int it, int ot, int fmt, int64_t n_rows, int64_t n_cols, int64_t k, int64_t stride, void* ptrs,
           void* cols, void* vals, const void* alpha, void* b, int64_t nrhs, const void* beta, void* c)
{
    return it == 0 ? pick_o<MT, double, I>(ot, fmt, n_rows, n_cols, k, stride, ptrs, cols, vals, alpha, b, nrhs, beta, c)
                   : pick_o<MT, float, I>(ot, fmt, n_rows, n_cols, k, stride, ptrs, cols, vals, alpha, b, nrhs, beta, c);
}

template <typename I>
int pick_m(int mt, int it, int ot, int fmt, int64_t n_rows, int64_t n_cols, int64_t k, int64_t stride, void* ptrs,
           void* cols, void* vals, const void* alpha, void* b, int64_t nrhs, const void* beta, void* c)
{
    return mt == 0 ? pick_i<double, I>(it, ot, fmt, n_rows, n_cols, k, stride, ptrs, cols, vals, alpha, b, nrhs, beta, c)
                   : pick_i<float, I>(it, ot, fmt, n_rows, n_cols, k, stride, ptrs, cols, vals, alpha, b, nrhs, beta, c);
}

}  // namespace

// type codes: 0 = double, 1 = float (GKOC_VT_F64 / GKOC_VT_F32); idx64 != 0: int64 indices.
// b is n_cols x nrhs, c is n_rows x nrhs, both row-major with stride nrhs; alpha (one MT) and beta
// (one OT) both NULL: c = A b.
extern "C" int ref_mixed_apply(int fmt, int mt, int it, int ot, int idx64, int64_t n_rows, int64_t n_cols,
                               int64_t k, int64_t stride, void* ptrs, void* cols, void* vals, const void* alpha,
                               void* b, int64_t nrhs, const void* beta, void* c)
{
    try {
        return idx64 ? pick_m<gko::int64>(mt, it, ot, fmt, n_rows, n_cols, k, stride, ptrs, cols, vals, alpha, b,
                                          nrhs, beta, c)
                     : pick_m<gko::int32>(mt, it, ot, fmt, n_rows, n_cols, k, stride, ptrs, cols, vals, alpha, b,
                                          nrhs, beta, c);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ref_mixed_apply: %s\n", e.what());
        return 1;
    }
}
