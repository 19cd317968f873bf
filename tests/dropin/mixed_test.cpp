// Mixed-precision acceptance test: the UNMODIFIED Ginkgo core built with GINKGO_MIXED_PRECISION
// (oracle/_ref/mixed, oracle/build_ref_mixed.py) on gko::HipExecutor = the shim.  With that switch
// Csr / Ell / Dense::row_gather dispatch on the run-time types of their operands
// (include/ginkgo/core/base/precision_dispatch.hpp: mixed_precision_dispatch_real_complex) and call
// the (matrix, input, output) kernel instantiations directly instead of converting the vectors first.
// Every triple is compared with gko::ReferenceExecutor in the same process: real triples bit for
// bit, complex triples to r<value_type> of the narrowest type involved.
// Pattern of the checks: reference/test/matrix/csr_kernels.cpp:367-675 (MixedApplies*),
// reference/test/matrix/ell_kernels.cpp:90-300, reference/test/matrix/dense_kernels.cpp (row_gather).
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <random>
#include <string>
#include <typeinfo>

#include <ginkgo/core/base/executor.hpp>
#include <ginkgo/core/base/matrix_data.hpp>
#include <ginkgo/core/matrix/csr.hpp>
#include <ginkgo/core/matrix/dense.hpp>
#include <ginkgo/core/matrix/ell.hpp>

static int failures = 0;
static int checks = 0;

#define CHECK(cond, msg)                                                   \
    do {                                                                   \
        ++checks;                                                          \
        if (!(cond)) {                                                     \
            ++failures;                                                    \
            std::cout << "FAILED: " << msg << " (" #cond ")" << std::endl; \
        }                                                                  \
    } while (0)

template <typename T>
struct name_of;
template <>
struct name_of<float> {
    static const char* get() { return "f32"; }
};
template <>
struct name_of<double> {
    static const char* get() { return "f64"; }
};
template <>
struct name_of<std::complex<float>> {
    static const char* get() { return "c64"; }
};
template <>
struct name_of<std::complex<double>> {
    static const char* get() { return "c128"; }
};

template <typename T>
T value_from(double re, double im)
{
    if constexpr (gko::is_complex<T>()) {
        return T{static_cast<gko::remove_complex<T>>(re), static_cast<gko::remove_complex<T>>(im)};
    } else {
        return static_cast<T>(re);
    }
}

template <typename T>
double eps_of()
{
    return std::numeric_limits<gko::remove_complex<T>>::epsilon();
}

// max |a - b| / max(1, max |b|); 0 iff the bits agree for real types
template <typename T>
double distance(const gko::matrix::Dense<T>* a, const gko::matrix::Dense<T>* b, bool& same_bits)
{
    double d = 0, scale = 1;
    same_bits = true;
    for (gko::size_type i = 0; i < a->get_size()[0]; ++i) {
        for (gko::size_type j = 0; j < a->get_size()[1]; ++j) {
            const T x = a->at(i, j), y = b->at(i, j);
            if (std::memcmp(&x, &y, sizeof(T)) != 0) same_bits = false;
            d = std::max(d, static_cast<double>(std::abs(x - y)));
            scale = std::max(scale, static_cast<double>(std::abs(y)));
        }
    }
    return d / scale;
}

template <typename T>
std::unique_ptr<gko::matrix::Dense<T>> random_dense(std::shared_ptr<const gko::Executor> ref, gko::size_type r,
                                                    gko::size_type c, std::mt19937& gen)
{
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    auto d = gko::matrix::Dense<T>::create(ref, gko::dim<2>{r, c});
    for (gko::size_type i = 0; i < r; ++i)
        for (gko::size_type j = 0; j < c; ++j) d->at(i, j) = value_from<T>(dist(gen), dist(gen));
    return d;
}

template <typename Mtx, typename MT, typename IT, typename OT, typename I>
void one_triple(std::shared_ptr<gko::ReferenceExecutor> ref, std::shared_ptr<gko::HipExecutor> hip, const char* fmt,
                const gko::matrix_data<MT, I>& md, gko::size_type nrhs)
{
    std::mt19937 gen(7u + 13u * sizeof(MT) + 5u * sizeof(IT) + sizeof(OT) + nrhs);
    auto a_ref = Mtx::create(ref);
    a_ref->read(md);
    auto a_hip = gko::clone(hip, a_ref);
    const auto n = md.size[0], m = md.size[1];
    auto b_ref = random_dense<IT>(ref, m, nrhs, gen);
    auto c_ref = random_dense<OT>(ref, n, nrhs, gen);
    auto b_hip = gko::clone(hip, b_ref);
    auto c_hip = gko::clone(hip, c_ref);
    const bool real = !gko::is_complex<MT>();
    const double tol = 40 * std::max({eps_of<MT>(), eps_of<IT>(), eps_of<OT>()});
    const std::string tag = std::string(fmt) + "<" + name_of<MT>::get() + ", " + name_of<IT>::get() + ", " +
                            name_of<OT>::get() + ", i" + std::to_string(8 * sizeof(I)) + "> x " +
                            std::to_string(nrhs);
    bool same = false;
    a_ref->apply(b_ref, c_ref);
    a_hip->apply(b_hip, c_hip);
    double d = distance(gko::clone(ref, c_hip).get(), c_ref.get(), same);
    CHECK(real ? same : d <= tol, tag + " spmv" + (real ? " bit-identical" : " to rounding"));
    // c = alpha A b + beta c; alpha has the matrix' type, beta the output's
    auto alpha = gko::initialize<gko::matrix::Dense<MT>>({value_from<MT>(-1.5, 0.25)}, ref);
    auto beta = gko::initialize<gko::matrix::Dense<OT>>({value_from<OT>(0.75, -0.5)}, ref);
    a_ref->apply(alpha, b_ref, beta, c_ref);
    a_hip->apply(gko::clone(hip, alpha), b_hip, gko::clone(hip, beta), c_hip);
    d = distance(gko::clone(ref, c_hip).get(), c_ref.get(), same);
    CHECK(real ? same : d <= tol, tag + " advanced_spmv");
    // beta == 0 must not read c (here: NaN in c)
    auto zero = gko::initialize<gko::matrix::Dense<OT>>({value_from<OT>(0.0, 0.0)}, ref);
    c_ref->fill(value_from<OT>(std::nan(""), 0.0));
    c_hip->fill(value_from<OT>(std::nan(""), 0.0));
    a_ref->apply(alpha, b_ref, zero, c_ref);
    a_hip->apply(gko::clone(hip, alpha), b_hip, gko::clone(hip, zero), c_hip);
    d = distance(gko::clone(ref, c_hip).get(), c_ref.get(), same);
    CHECK(real ? same : d <= tol, tag + " advanced_spmv, beta = 0 over NaN");
}

template <typename MT, typename I>
gko::matrix_data<MT, I> test_matrix(gko::size_type n, gko::size_type m, std::mt19937& gen)
{
    // rows of 0 ... 40 entries (one empty row, one long row of 300), unsorted columns are not needed:
    // matrix_data -> read() sorts
    gko::matrix_data<MT, I> md{gko::dim<2>{n, m}};
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    for (gko::size_type r = 0; r < n; ++r) {
        gko::size_type len = r == 3 ? 0 : (r == 70 ? std::min<gko::size_type>(300, m) : 1 + (r * 7) % 40);
        const gko::size_type step = std::max<gko::size_type>(1, m / (len + 1));
        for (gko::size_type k = 0; k < len; ++k) {
            md.nonzeros.emplace_back(static_cast<I>(r), static_cast<I>((r + k * step) % m),
                                     value_from<MT>(dist(gen), dist(gen)));
        }
    }
    md.sort_row_major();
    md.sum_duplicates();
    return md;
}

template <typename W, typename N, typename I>
void all_triples(std::shared_ptr<gko::ReferenceExecutor> ref, std::shared_ptr<gko::HipExecutor> hip)
{
    std::mt19937 gen(99);
    const auto md_w = test_matrix<W, I>(333, 401, gen);
    gko::matrix_data<N, I> md_n{md_w.size};
    for (const auto& e : md_w.nonzeros) md_n.nonzeros.emplace_back(e.row, e.column, static_cast<N>(e.value));
    for (gko::size_type nrhs : {1, 3}) {
#define BOTH(MT, md, IT, OT)                                                                     \
    one_triple<gko::matrix::Csr<MT, I>, MT, IT, OT, I>(ref, hip, "csr", md, nrhs);               \
    one_triple<gko::matrix::Ell<MT, I>, MT, IT, OT, I>(ref, hip, "ell", md, nrhs)
        BOTH(W, md_w, W, N);
        BOTH(W, md_w, N, W);
        BOTH(W, md_w, N, N);
        BOTH(N, md_n, W, W);
        BOTH(N, md_n, W, N);
        BOTH(N, md_n, N, W);
        // the uniform ones through the same core flavor
        BOTH(W, md_w, W, W);
        BOTH(N, md_n, N, N);
#undef BOTH
    }
}

template <typename VT, typename OT, typename I>
void gather_pair(std::shared_ptr<gko::ReferenceExecutor> ref, std::shared_ptr<gko::HipExecutor> hip)
{
    std::mt19937 gen(5);
    auto orig = random_dense<VT>(ref, 57, 5, gen);
    auto out_ref = random_dense<OT>(ref, 23, 5, gen);
    auto out_hip = gko::clone(hip, out_ref);
    gko::array<I> idx{ref, 23};
    for (int i = 0; i < 23; ++i) idx.get_data()[i] = static_cast<I>((i * 11 + 3) % 57);
    gko::array<I> idx_hip{hip, idx};
    const std::string tag = std::string("row_gather<") + name_of<VT>::get() + ", " + name_of<OT>::get() + ", i" +
                            std::to_string(8 * sizeof(I)) + ">";
    bool same = false;
    orig->row_gather(&idx, out_ref.get());
    gko::clone(hip, orig)->row_gather(&idx_hip, out_hip.get());
    distance(gko::clone(ref, out_hip).get(), out_ref.get(), same);
    CHECK(same, tag + " bit-identical");
    auto alpha = gko::initialize<gko::matrix::Dense<VT>>({value_from<VT>(1.25, -0.5)}, ref);
    auto beta = gko::initialize<gko::matrix::Dense<VT>>({value_from<VT>(-0.375, 0.125)}, ref);
    orig->row_gather(alpha, &idx, beta, out_ref.get());
    gko::clone(hip, orig)->row_gather(gko::clone(hip, alpha), &idx_hip, gko::clone(hip, beta), out_hip.get());
    const double d = distance(gko::clone(ref, out_hip).get(), out_ref.get(), same);
    CHECK(gko::is_complex<VT>() ? d <= 8 * std::max(eps_of<VT>(), eps_of<OT>()) : same, tag + " advanced");
}

int main()
{
    auto ref = gko::ReferenceExecutor::create();
    auto hip = gko::HipExecutor::create(0, ref);
    all_triples<double, float, gko::int32>(ref, hip);
    all_triples<double, float, gko::int64>(ref, hip);
    all_triples<std::complex<double>, std::complex<float>, gko::int32>(ref, hip);
    all_triples<std::complex<double>, std::complex<float>, gko::int64>(ref, hip);
    gather_pair<double, float, gko::int32>(ref, hip);
    gather_pair<float, double, gko::int64>(ref, hip);
    gather_pair<std::complex<double>, std::complex<float>, gko::int64>(ref, hip);
    gather_pair<std::complex<float>, std::complex<double>, gko::int32>(ref, hip);
    std::cout << checks << " checks, " << failures << " failed\n"
              << (failures == 0 ? "MIXED OK" : "MIXED FAILED") << std::endl;
    return failures == 0 ? 0 : 1;
}
