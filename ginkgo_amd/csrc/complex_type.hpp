// The complex value types of the C ABI as C++ classes (library side; users of include/gko_cdna4.h
// see two plain structs of the same layout: { re, im }).  With the arithmetic below the element-wise
// kernel templates of the Krylov solvers instantiate for complex<float> / complex<double> as they
// stand (include/ginkgo/core/base/types.hpp:471, 689: the value types every kernel is declared for).
// Products and sums are the textbook expressions in a fixed order (what std::complex's operators
// compute for finite operands); the quotient is Smith's scaled quotient, not bit for bit libstdc++'s
// __divdc3 - complex kernels agree with the reference to rounding (the reference's own tolerance
// r<value_type>), not bit for bit, and are documented as such.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

template <typename R>
struct gkoc_cplx {
    R re, im;
    gkoc_cplx() = default;
    __host__ __device__ constexpr gkoc_cplx(R r, R i = R(0)) : re(r), im(i) {}
    __host__ __device__ gkoc_cplx& operator+=(gkoc_cplx b)
    {
        re += b.re;
        im += b.im;
        return *this;
    }
    __host__ __device__ gkoc_cplx& operator-=(gkoc_cplx b)
    {
        re -= b.re;
        im -= b.im;
        return *this;
    }
    __host__ __device__ gkoc_cplx& operator*=(gkoc_cplx b)
    {
        const R r = re * b.re - im * b.im;
        im = re * b.im + im * b.re;
        re = r;
        return *this;
    }
};
typedef gkoc_cplx<double> gkoc_c128;
typedef gkoc_cplx<float> gkoc_c64;
#define GKOC_COMPLEX_TYPES_DEFINED 1

#define GKOC_CX __host__ __device__ __forceinline__
template <typename R>
GKOC_CX gkoc_cplx<R> operator+(gkoc_cplx<R> a, gkoc_cplx<R> b) { return {a.re + b.re, a.im + b.im}; }
template <typename R>
GKOC_CX gkoc_cplx<R> operator-(gkoc_cplx<R> a, gkoc_cplx<R> b) { return {a.re - b.re, a.im - b.im}; }
template <typename R>
GKOC_CX gkoc_cplx<R> operator-(gkoc_cplx<R> a) { return {-a.re, -a.im}; }
template <typename R>
GKOC_CX gkoc_cplx<R> operator*(gkoc_cplx<R> a, gkoc_cplx<R> b)
{
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename R>
GKOC_CX gkoc_cplx<R> operator*(gkoc_cplx<R> a, R b) { return {a.re * b, a.im * b}; }
template <typename R>
GKOC_CX gkoc_cplx<R> operator*(R a, gkoc_cplx<R> b) { return {a * b.re, a * b.im}; }
template <typename R>
GKOC_CX gkoc_cplx<R> operator/(gkoc_cplx<R> a, R b) { return {a.re / b, a.im / b}; }
// Quotient by Smith's method (the divisor is scaled by its larger part): |b|^2 is never formed, so
// a divisor of magnitude 1e-20 in float - a Krylov scalar near convergence - does not underflow to
// zero as it does in the textbook formula (complex<float> Bicgstab returned NaN with that one;
// libstdc++'s operator/ scales as well)
template <typename R>
GKOC_CX gkoc_cplx<R> operator/(gkoc_cplx<R> a, gkoc_cplx<R> b)
{
    if ((b.re < R(0) ? -b.re : b.re) >= (b.im < R(0) ? -b.im : b.im)) {
        const R r = b.im / b.re, den = b.re + b.im * r;
        return {(a.re + a.im * r) / den, (a.im - a.re * r) / den};
    }
    const R r = b.re / b.im, den = b.re * r + b.im;
    return {(a.re * r + a.im) / den, (a.im * r - a.re) / den};
}
// principal square root (std::sqrt(std::complex) up to rounding)
template <typename R>
GKOC_CX gkoc_cplx<R> sqrt(gkoc_cplx<R> z)
{
    const R m = ::hypot(z.re, z.im);
    const R a = ::sqrt((m + z.re) / R(2)), b = ::sqrt((m - z.re) / R(2));
    return {a, z.im < R(0) ? -b : b};
}
template <typename R>
GKOC_CX bool operator==(gkoc_cplx<R> a, gkoc_cplx<R> b) { return a.re == b.re && a.im == b.im; }
template <typename R>
GKOC_CX bool operator!=(gkoc_cplx<R> a, gkoc_cplx<R> b) { return !(a == b); }

#ifdef __HIPCC__
// cross-lane moves of a complex value: its two parts (the reductions of common.hpp call these names)
template <typename R>
__device__ __forceinline__ gkoc_cplx<R> __shfl_xor(gkoc_cplx<R> v, int mask, int width = 64)
{
    return {__shfl_xor(v.re, mask, width), __shfl_xor(v.im, mask, width)};
}
template <typename R>
__device__ __forceinline__ gkoc_cplx<R> __shfl(gkoc_cplx<R> v, int lane, int width = 64)
{
    return {__shfl(v.re, lane, width), __shfl(v.im, lane, width)};
}
template <typename R>
__device__ __forceinline__ gkoc_cplx<R> __shfl_down(gkoc_cplx<R> v, unsigned delta, int width = 64)
{
    return {__shfl_down(v.re, delta, width), __shfl_down(v.im, delta, width)};
}
#endif

namespace gkoc {

// remove_complex<T> (include/ginkgo/core/base/math.hpp) and the value functions generic kernels use
template <typename T>
struct real_type {
    using type = T;
};
template <typename R>
struct real_type<gkoc_cplx<R>> {
    using type = R;
};
template <typename T>
using real_t = typename real_type<T>::type;

GKOC_CX float conj_v(float a) { return a; }
GKOC_CX double conj_v(double a) { return a; }
template <typename R>
GKOC_CX gkoc_cplx<R> conj_v(gkoc_cplx<R> a) { return {a.re, -a.im}; }
GKOC_CX float real_v(float a) { return a; }
GKOC_CX double real_v(double a) { return a; }
template <typename R>
GKOC_CX R real_v(gkoc_cplx<R> a) { return a.re; }
GKOC_CX float imag_v(float) { return 0.0f; }
GKOC_CX double imag_v(double) { return 0.0; }
template <typename R>
GKOC_CX R imag_v(gkoc_cplx<R> a) { return a.im; }
// |a|^2 and |a| (math.hpp squared_norm / abs)
GKOC_CX float squared_norm_v(float a) { return a * a; }
GKOC_CX double squared_norm_v(double a) { return a * a; }
template <typename R>
GKOC_CX R squared_norm_v(gkoc_cplx<R> a) { return a.re * a.re + a.im * a.im; }
GKOC_CX float abs_v(float a) { return fabsf(a); }
GKOC_CX double abs_v(double a) { return fabs(a); }
GKOC_CX float abs_v(gkoc_cplx<float> a) { return hypotf(a.re, a.im); }
GKOC_CX double abs_v(gkoc_cplx<double> a) { return hypot(a.re, a.im); }

}  // namespace gkoc
#undef GKOC_CX
