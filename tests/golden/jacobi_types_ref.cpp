// TEST INFRASTRUCTURE - a C entry point over the UNMODIFIED reference (oracle/_ref, ReferenceExecutor)
// for tests/golden/make_jacobi_types_golden.py: block-Jacobi with a fixed reduced, block-wise or
// autodetected storage precision for float, complex<float>, complex<double>
// (include/ginkgo/core/preconditioner/jacobi.hpp:389-484; reference/preconditioner/jacobi_kernels.cpp).
// Storage groups of stride 64 - what the device executor uses (jacobi.hpp:589-620) - so that the
// fixture's decisions (one precision per group) are the ones a 64-wide backend must reproduce.
#include <complex>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <memory>

#include <ginkgo/core/base/executor.hpp>
#include <ginkgo/core/base/matrix_data.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/preconditioner/jacobi.hpp>

namespace {

gko::precision_reduction from_byte(uint8_t p)
{
    return p == 0xff ? gko::precision_reduction::autodetect() : gko::precision_reduction(p >> 4, p & 15);
}

template <typename T>
int64_t run(int64_t n, const int32_t* rp, const int32_t* ci, const void* vals_v, uint32_t max_bs,
            const uint8_t* requests, int64_t n_requests, double accuracy, int64_t nrhs, const void* b_v,
            void* x_v, void* xt_v, void* xh_v, int32_t* block_ptrs, uint8_t* prec, double* cond)
{
    using R = gko::remove_complex<T>;
    using Csr = gko::matrix::Csr<T, gko::int32>;
    using Dense = gko::matrix::Dense<T>;
    using Jacobi = gko::preconditioner::Jacobi<T, gko::int32>;
    auto exec = gko::ReferenceExecutor::create();
    const T* vals = static_cast<const T*>(vals_v);
    gko::matrix_data<T, gko::int32> md{gko::dim<2>(n, n)};
    for (int64_t r = 0; r < n; ++r)
        for (int32_t k = rp[r]; k < rp[r + 1]; ++k) md.nonzeros.emplace_back(gko::int32(r), ci[k], vals[k]);
    auto a = gko::share(Csr::create(exec));
    a->read(md);
    auto f = Jacobi::build().with_max_block_size(max_bs).with_max_block_stride(64u).with_accuracy(R(accuracy));
    if (n_requests == 1) {
        f.with_storage_optimization(from_byte(requests[0]));
    } else if (n_requests > 1) {
        gko::array<gko::precision_reduction> req(exec, n_requests);
        for (int64_t i = 0; i < n_requests; ++i) req.get_data()[i] = from_byte(requests[i]);
        f.with_storage_optimization(req);
    }
    auto j = f.on(exec)->generate(a);
    const auto nb = static_cast<int64_t>(j->get_num_blocks());
    std::memcpy(block_ptrs, j->get_parameters().block_pointers.get_const_data(), sizeof(int32_t) * (nb + 1));
    const auto& pw = j->get_parameters().storage_optimization.block_wise;
    for (int64_t i = 0; i < nb; ++i) {
        prec[i] = pw.get_size() ? static_cast<uint8_t>(pw.get_const_data()[i]) : 0;
        cond[i] = j->get_conditioning() ? double(j->get_conditioning()[i]) : 0.0;
    }
    auto b = Dense::create(exec, gko::dim<2>(n, nrhs));
    std::memcpy(b->get_values(), b_v, sizeof(T) * n * nrhs);
    auto x = Dense::create(exec, gko::dim<2>(n, nrhs));
    j->apply(b, x);
    std::memcpy(x_v, x->get_const_values(), sizeof(T) * n * nrhs);
    j->transpose()->apply(b, x);
    std::memcpy(xt_v, x->get_const_values(), sizeof(T) * n * nrhs);
    j->conj_transpose()->apply(b, x);
    std::memcpy(xh_v, x->get_const_values(), sizeof(T) * n * nrhs);
    return nb;
}

}  // namespace

// vt: 0 float, 1 complex<float>, 2 complex<double>; b, x*: n x nrhs row-major; returns the number of blocks
extern "C" int64_t ref_jacobi_types(int vt, int64_t n, const int32_t* rp, const int32_t* ci, const void* vals,
                                    uint32_t max_bs, const uint8_t* requests, int64_t n_requests,
                                    double accuracy, int64_t nrhs, const void* b, void* x, void* xt, void* xh,
                                    int32_t* block_ptrs, uint8_t* prec, double* cond)
{
    try {
        switch (vt) {
        case 0: return run<float>(n, rp, ci, vals, max_bs, requests, n_requests, accuracy, nrhs, b, x, xt, xh,
                                  block_ptrs, prec, cond);
        case 1: return run<std::complex<float>>(n, rp, ci, vals, max_bs, requests, n_requests, accuracy, nrhs, b,
                                                x, xt, xh, block_ptrs, prec, cond);
        case 2: return run<std::complex<double>>(n, rp, ci, vals, max_bs, requests, n_requests, accuracy, nrhs, b,
                                                 x, xt, xh, block_ptrs, prec, cond);
        default: return -1;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ref_jacobi_types: %s\n", e.what());
        return -2;
    }
}
