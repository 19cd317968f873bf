// Ginkgo-side binding of the CB-GMRES kernels (core/solver/cb_gmres_kernels.hpp:101-142) to the
// C ABI.  The 3-d accessor range is unwrapped into storage pointer + strides (+ scalars).
#include <complex>
#include <type_traits>

#include <ginkgo/core/matrix/dense.hpp>

#include "core/solver/cb_gmres_accessor.hpp"
#include "core/solver/cb_gmres_kernels.hpp"
#include "shim_common.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace cb_gmres {

using namespace ::gko::cdna4;
using exec_t = std::shared_ptr<const HipExecutor>;

namespace {

template <typename S>
struct kind_of;
template <>
struct kind_of<double> {
    static constexpr int value = GKOC_CB_KEEP;  // corrected below for float arithmetic
};
template <>
struct kind_of<float> {
    static constexpr int value = GKOC_CB_F32;
};
template <>
struct kind_of<half> {
    static constexpr int value = GKOC_CB_F16;
};
template <>
struct kind_of<int64> {
    static constexpr int value = GKOC_CB_I64;
};
template <>
struct kind_of<int32> {
    static constexpr int value = GKOC_CB_I32;
};
template <>
struct kind_of<int16> {
    static constexpr int value = GKOC_CB_I16;
};

template <>
struct kind_of<std::complex<double>> {
    static constexpr int value = GKOC_CB_KEEP;
};
template <>
struct kind_of<std::complex<float>> {
    static constexpr int value = GKOC_CB_F32;    // complex<double> arithmetic, basis stored as complex<float>
};

// storage pointer, strides and scalars of either accessor kind
template <typename T>
struct unwrapped {
    int kind;
    void* bases;
    int64_t st0, st1;
    T* scal;
    int64_t sst;
};

template <typename T, typename Range>
unwrapped<T> unwrap(Range range)
{
    using accessor = typename Range::accessor;
    using storage = std::remove_const_t<typename accessor::storage_type>;
    const auto& acc = range.get_accessor();
    unwrapped<T> u{};
    u.kind = std::is_same<storage, T>::value ? GKOC_CB_KEEP : kind_of<storage>::value;
    u.bases = const_cast<storage*>(acc.get_const_storage());
    if constexpr (::gko::cb_gmres::detail::has_3d_scaled_accessor<Range>::value) {
        const auto st = acc.get_storage_stride();
        u.st0 = static_cast<int64_t>(st[0]);
        u.st1 = static_cast<int64_t>(st[1]);
        u.scal = const_cast<T*>(acc.get_const_scalar());
        u.sst = static_cast<int64_t>(acc.get_scalar_stride()[0]);
    } else {
        const auto st = acc.get_stride();
        u.st0 = static_cast<int64_t>(st[0]);
        u.st1 = static_cast<int64_t>(st[1]);
        u.scal = nullptr;
        u.sst = 0;
    }
    return u;
}

template <typename T>
struct abi;
#define GKOC_CB_ABI(T, TN)                                                      \
    template <>                                                                 \
    struct abi<T> {                                                             \
        static constexpr auto initialize = gkoc_common_gmres_initialize_##TN;   \
        static constexpr auto restart = gkoc_cb_gmres_restart_##TN;             \
        static constexpr auto arnoldi = gkoc_cb_gmres_arnoldi_##TN;             \
        static constexpr auto solve_krylov = gkoc_cb_gmres_solve_krylov_##TN;   \
    }
GKOC_CB_ABI(double, f64);
GKOC_CB_ABI(float, f32);
#undef GKOC_CB_ABI

template <typename T>
constexpr bool is_real_v = std::is_same<T, double>::value || std::is_same<T, float>::value;

// complex value types: csrc/cb_gmres_complex.hip (no scalars: the basis is a reduced_row_major accessor)
template <typename T>
struct cx_abi;
#define GKOC_CB_CX_ABI(T, TN)                                                   \
    template <>                                                                 \
    struct cx_abi<T> {                                                          \
        static constexpr auto initialize = gkoc_common_gmres_initialize_##TN;   \
        static constexpr auto restart = gkoc_cb_gmres_restart_##TN;             \
        static constexpr auto arnoldi = gkoc_cb_gmres_arnoldi_##TN;             \
        static constexpr auto solve_krylov = gkoc_cb_gmres_solve_krylov_##TN;   \
    }
GKOC_CB_CX_ABI(std::complex<double>, c128);
GKOC_CB_CX_ABI(std::complex<float>, c64);
#undef GKOC_CB_CX_ABI

}  // namespace


namespace {


template <typename ValueType>
void initialize_impl(exec_t exec, const matrix::Dense<ValueType>* b, matrix::Dense<ValueType>* residual,
                matrix::Dense<ValueType>* givens_sin, matrix::Dense<ValueType>* givens_cos,
                array<stopping_status>* stop_status, size_type krylov_dim)
{
    if constexpr (is_real_v<ValueType>) {
        GKOC_CALL(abi<ValueType>::initialize(
            stream_of(exec), rows(b), cols(b), b->get_const_values(), ld(b), residual->get_values(),
            ld(residual), givens_sin->get_values(), ld(givens_sin), givens_cos->get_values(),
            ld(givens_cos), static_cast<int64_t>(krylov_dim), raw(stop_status)));
    } else {
        GKOC_CALL(cx_abi<ValueType>::initialize(
            stream_of(exec), rows(b), cols(b), b->get_const_values(), ld(b), residual->get_values(),
            ld(residual), givens_sin->get_values(), ld(givens_sin), givens_cos->get_values(),
            ld(givens_cos), static_cast<int64_t>(krylov_dim), raw(stop_status)));
    }
}



template <typename ValueType, typename Accessor3d>
void restart_impl(exec_t exec, const matrix::Dense<ValueType>* residual,
             matrix::Dense<remove_complex<ValueType>>* residual_norm,
             matrix::Dense<ValueType>* residual_norm_collection,
             matrix::Dense<remove_complex<ValueType>>* arnoldi_norm, Accessor3d krylov_bases,
             matrix::Dense<ValueType>* next_krylov_basis, array<size_type>* final_iter_nums,
             array<char>&, size_type krylov_dim)
{
    if constexpr (is_real_v<ValueType>) {
        const auto u = unwrap<ValueType>(krylov_bases);
        GKOC_CALL(abi<ValueType>::restart(
            stream_of(exec), rows(residual), cols(residual), static_cast<int64_t>(krylov_dim),
            residual->get_const_values(), ld(residual), residual_norm->get_values(),
            residual_norm_collection->get_values(), ld(residual_norm_collection),
            arnoldi_norm->get_values(), ld(arnoldi_norm), u.kind, u.bases, u.st0, u.st1, u.scal, u.sst,
            next_krylov_basis->get_values(), ld(next_krylov_basis),
            reinterpret_cast<uint64_t*>(final_iter_nums->get_data())));
    } else {
        const auto u = unwrap<ValueType>(krylov_bases);
        GKOC_CALL(cx_abi<ValueType>::restart(
            stream_of(exec), rows(residual), cols(residual), static_cast<int64_t>(krylov_dim),
            residual->get_const_values(), ld(residual), residual_norm->get_values(),
            residual_norm_collection->get_values(), ld(residual_norm_collection), u.kind, u.bases, u.st0,
            u.st1, next_krylov_basis->get_values(), ld(next_krylov_basis),
            reinterpret_cast<uint64_t*>(final_iter_nums->get_data())));
    }
}



template <typename ValueType, typename Accessor3d>
void arnoldi_impl(exec_t exec, matrix::Dense<ValueType>* next_krylov_basis,
             matrix::Dense<ValueType>* givens_sin, matrix::Dense<ValueType>* givens_cos,
             matrix::Dense<remove_complex<ValueType>>* residual_norm,
             matrix::Dense<ValueType>* residual_norm_collection, Accessor3d krylov_bases,
             matrix::Dense<ValueType>* hessenberg_iter, matrix::Dense<ValueType>* buffer_iter,
             matrix::Dense<remove_complex<ValueType>>* arnoldi_norm, size_type iter,
             array<size_type>* final_iter_nums, const array<stopping_status>* stop_status,
             array<stopping_status>*, array<size_type>*)
{
    if constexpr (is_real_v<ValueType>) {
        const auto u = unwrap<ValueType>(krylov_bases);
        // the re-orthogonalisation buffer is optional (Ginkgo's own test hands over an empty one)
        const bool has_buffer = buffer_iter && rows(buffer_iter) >= static_cast<int64_t>(iter) + 1 &&
                                cols(buffer_iter) >= cols(next_krylov_basis);
        GKOC_CALL(abi<ValueType>::arnoldi(
            stream_of(exec), rows(next_krylov_basis), cols(next_krylov_basis),
            static_cast<int64_t>(iter), next_krylov_basis->get_values(), ld(next_krylov_basis),
            givens_sin->get_values(), ld(givens_sin), givens_cos->get_values(), ld(givens_cos),
            residual_norm->get_values(), residual_norm_collection->get_values(),
            ld(residual_norm_collection), u.kind, u.bases, u.st0, u.st1, u.scal, u.sst,
            hessenberg_iter->get_values(), ld(hessenberg_iter),
            has_buffer ? buffer_iter->get_values() : nullptr, has_buffer ? ld(buffer_iter) : 0,
            arnoldi_norm->get_values(), ld(arnoldi_norm),
            reinterpret_cast<uint64_t*>(final_iter_nums->get_data()), raw(stop_status)));
    } else {
        const auto u = unwrap<ValueType>(krylov_bases);
        const bool has_buffer = buffer_iter && rows(buffer_iter) >= static_cast<int64_t>(iter) + 1 &&
                                cols(buffer_iter) >= cols(next_krylov_basis);
        GKOC_CALL(cx_abi<ValueType>::arnoldi(
            stream_of(exec), rows(next_krylov_basis), cols(next_krylov_basis),
            static_cast<int64_t>(iter), next_krylov_basis->get_values(), ld(next_krylov_basis),
            givens_sin->get_values(), ld(givens_sin), givens_cos->get_values(), ld(givens_cos),
            residual_norm->get_values(), residual_norm_collection->get_values(),
            ld(residual_norm_collection), u.kind, u.bases, u.st0, u.st1, hessenberg_iter->get_values(),
            ld(hessenberg_iter), has_buffer ? buffer_iter->get_values() : nullptr,
            has_buffer ? ld(buffer_iter) : 0, arnoldi_norm->get_values(), ld(arnoldi_norm),
            reinterpret_cast<uint64_t*>(final_iter_nums->get_data()), raw(stop_status)));
    }
}



template <typename ValueType, typename ConstAccessor3d>
void solve_krylov_impl(exec_t exec, const matrix::Dense<ValueType>* residual_norm_collection,
                  ConstAccessor3d krylov_bases, const matrix::Dense<ValueType>* hessenberg,
                  matrix::Dense<ValueType>* y, matrix::Dense<ValueType>* before_preconditioner,
                  const array<size_type>* final_iter_nums)
{
    if constexpr (is_real_v<ValueType>) {
        const auto u = unwrap<ValueType>(krylov_bases);
        GKOC_CALL(abi<ValueType>::solve_krylov(
            stream_of(exec), rows(before_preconditioner), cols(before_preconditioner),
            residual_norm_collection->get_const_values(), ld(residual_norm_collection), u.kind,
            u.bases, u.st0, u.st1, u.scal, u.sst, hessenberg->get_const_values(), ld(hessenberg),
            y->get_values(), ld(y), before_preconditioner->get_values(), ld(before_preconditioner),
            reinterpret_cast<const uint64_t*>(final_iter_nums->get_const_data())));
    } else {
        const auto u = unwrap<ValueType>(krylov_bases);
        GKOC_CALL(cx_abi<ValueType>::solve_krylov(
            stream_of(exec), rows(before_preconditioner), cols(before_preconditioner),
            residual_norm_collection->get_const_values(), ld(residual_norm_collection), u.kind,
            u.bases, u.st0, u.st1, hessenberg->get_const_values(), ld(hessenberg), y->get_values(), ld(y),
            before_preconditioner->get_values(), ld(before_preconditioner),
            reinterpret_cast<const uint64_t*>(final_iter_nums->get_const_data())));
    }
}



}  // namespace


// Explicit SPECIALISATIONS (strong symbols) for the type list of
// GKO_INSTANTIATE_FOR_EACH_CB_GMRES_TYPE (core/solver/cb_gmres_kernels.hpp:37-94): an explicit
// instantiation would only be a weak definition next to Ginkgo's weakened stub.
#define CB_INIT(V)                                                                              \
    template <>                                                                                 \
    void initialize<V>(exec_t exec, const matrix::Dense<V>* b, matrix::Dense<V>* residual,      \
                       matrix::Dense<V>* givens_sin, matrix::Dense<V>* givens_cos,              \
                       array<stopping_status>* stop_status, size_type krylov_dim)               \
    {                                                                                           \
        initialize_impl<V>(exec, b, residual, givens_sin, givens_cos, stop_status, krylov_dim); \
    }
CB_INIT(double)
CB_INIT(float)
CB_INIT(std::complex<double>)
CB_INIT(std::complex<float>)
#undef CB_INIT

#define CB_UNPACK(...) __VA_ARGS__
#define CB_MUTABLE(V, R)                                                                          \
    template <>                                                                                   \
    void restart<V, CB_UNPACK R>(                                                                 \
        exec_t exec, const matrix::Dense<V>* residual,                                            \
        matrix::Dense<remove_complex<V>>* residual_norm, matrix::Dense<V>* rnc,                   \
        matrix::Dense<remove_complex<V>>* arnoldi_norm, CB_UNPACK R krylov_bases,                 \
        matrix::Dense<V>* next_krylov_basis, array<size_type>* final_iter_nums,                   \
        array<char>& tmp, size_type krylov_dim)                                                   \
    {                                                                                             \
        restart_impl<V>(exec, residual, residual_norm, rnc, arnoldi_norm, krylov_bases,           \
                        next_krylov_basis, final_iter_nums, tmp, krylov_dim);                     \
    }                                                                                             \
    template <>                                                                                   \
    void arnoldi<V, CB_UNPACK R>(                                                                 \
        exec_t exec, matrix::Dense<V>* next_krylov_basis, matrix::Dense<V>* givens_sin,           \
        matrix::Dense<V>* givens_cos, matrix::Dense<remove_complex<V>>* residual_norm,            \
        matrix::Dense<V>* rnc, CB_UNPACK R krylov_bases, matrix::Dense<V>* hessenberg_iter,       \
        matrix::Dense<V>* buffer_iter, matrix::Dense<remove_complex<V>>* arnoldi_norm,            \
        size_type iter, array<size_type>* final_iter_nums,                                        \
        const array<stopping_status>* stop_status, array<stopping_status>* reorth_status,         \
        array<size_type>* num_reorth)                                                             \
    {                                                                                             \
        arnoldi_impl<V>(exec, next_krylov_basis, givens_sin, givens_cos, residual_norm, rnc,      \
                        krylov_bases, hessenberg_iter, buffer_iter, arnoldi_norm, iter,           \
                        final_iter_nums, stop_status, reorth_status, num_reorth);                 \
    }
#define CB_CONST(V, R)                                                                            \
    template <>                                                                                   \
    void solve_krylov<V, CB_UNPACK R>(                                                            \
        exec_t exec, const matrix::Dense<V>* rnc, CB_UNPACK R krylov_bases,                       \
        const matrix::Dense<V>* hessenberg, matrix::Dense<V>* y,                                  \
        matrix::Dense<V>* before_preconditioner, const array<size_type>* final_iter_nums)         \
    {                                                                                             \
        solve_krylov_impl<V>(exec, rnc, krylov_bases, hessenberg, y, before_preconditioner,       \
                             final_iter_nums);                                                    \
    }
#define CB_EACH(M, C)                                                                  \
    M(double, (acc::range<acc::reduced_row_major<3, double, C double>>))               \
    M(double, (acc::range<acc::reduced_row_major<3, double, C float>>))                \
    M(double, (acc::range<acc::reduced_row_major<3, double, C half>>))                 \
    M(double, (acc::range<acc::scaled_reduced_row_major<3, double, C int64, 0b101>>))  \
    M(double, (acc::range<acc::scaled_reduced_row_major<3, double, C int32, 0b101>>))  \
    M(double, (acc::range<acc::scaled_reduced_row_major<3, double, C int16, 0b101>>))  \
    M(float, (acc::range<acc::reduced_row_major<3, float, C float>>))                  \
    M(float, (acc::range<acc::reduced_row_major<3, float, C half>>))                   \
    M(float, (acc::range<acc::scaled_reduced_row_major<3, float, C int32, 0b101>>))    \
    M(float, (acc::range<acc::scaled_reduced_row_major<3, float, C int16, 0b101>>))    \
    M(std::complex<double>,                                                            \
      (acc::range<acc::reduced_row_major<3, std::complex<double>, C std::complex<double>>>)) \
    M(std::complex<double>,                                                            \
      (acc::range<acc::reduced_row_major<3, std::complex<double>, C std::complex<float>>>))  \
    M(std::complex<float>,                                                             \
      (acc::range<acc::reduced_row_major<3, std::complex<float>, C std::complex<float>>>))
CB_EACH(CB_MUTABLE, )
CB_EACH(CB_CONST, const)
#undef CB_EACH
#undef CB_CONST
#undef CB_MUTABLE
#undef CB_UNPACK


}  // namespace cb_gmres
}  // namespace hip
}  // namespace kernels
}  // namespace gko
