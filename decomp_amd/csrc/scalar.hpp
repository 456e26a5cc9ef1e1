// Scalar helpers shared by all kernels: a minimal complex type and overloaded
// arithmetic so that one epilogue / elementwise template serves
// float, double, complex64 and complex128.
#pragma once
#include <hip/hip_runtime.h>

namespace dcp {

#define DCP_HD __host__ __device__ __forceinline__

template <class R>
struct cx {
    R re, im;
};
typedef cx<float> c64;
typedef cx<double> c128;

template <class T> struct scalar_traits;
template <> struct scalar_traits<float>  { typedef float  real; static constexpr bool is_complex = false; };
template <> struct scalar_traits<double> { typedef double real; static constexpr bool is_complex = false; };
template <> struct scalar_traits<c64>    { typedef float  real; static constexpr bool is_complex = true; };
template <> struct scalar_traits<c128>   { typedef double real; static constexpr bool is_complex = true; };
template <class T> using real_t = typename scalar_traits<T>::real;

// ---- construction --------------------------------------------------------
template <class T> DCP_HD T zero_of();
template <> DCP_HD float  zero_of<float>()  { return 0.0f; }
template <> DCP_HD double zero_of<double>() { return 0.0; }
template <> DCP_HD c64    zero_of<c64>()    { return c64{0.0f, 0.0f}; }
template <> DCP_HD c128   zero_of<c128>()   { return c128{0.0, 0.0}; }

template <class T> DCP_HD T from_real(real_t<T> r);
template <> DCP_HD float  from_real<float>(float r)   { return r; }
template <> DCP_HD double from_real<double>(double r) { return r; }
template <> DCP_HD c64    from_real<c64>(float r)     { return c64{r, 0.0f}; }
template <> DCP_HD c128   from_real<c128>(double r)   { return c128{r, 0.0}; }

// ---- arithmetic ----------------------------------------------------------
DCP_HD float  add(float a, float b)   { return a + b; }
DCP_HD double add(double a, double b) { return a + b; }
template <class R> DCP_HD cx<R> add(cx<R> a, cx<R> b) { return cx<R>{a.re + b.re, a.im + b.im}; }

DCP_HD float  sub(float a, float b)   { return a - b; }
DCP_HD double sub(double a, double b) { return a - b; }
template <class R> DCP_HD cx<R> sub(cx<R> a, cx<R> b) { return cx<R>{a.re - b.re, a.im - b.im}; }

DCP_HD float  mul(float a, float b)   { return a * b; }
DCP_HD double mul(double a, double b) { return a * b; }
template <class R> DCP_HD cx<R> mul(cx<R> a, cx<R> b) {
    return cx<R>{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

// scale by a real
DCP_HD float  scale(float a, float s)   { return a * s; }
DCP_HD double scale(double a, double s) { return a * s; }
template <class R> DCP_HD cx<R> scale(cx<R> a, R s) { return cx<R>{a.re * s, a.im * s}; }

// acc + a*b
DCP_HD float  madd(float acc, float a, float b)    { return fmaf(a, b, acc); }
DCP_HD double madd(double acc, double a, double b) { return fma(a, b, acc); }
template <class R> DCP_HD cx<R> madd(cx<R> acc, cx<R> a, cx<R> b) {
    acc.re += a.re * b.re - a.im * b.im;
    acc.im += a.re * b.im + a.im * b.re;
    return acc;
}

// acc - a*b
DCP_HD float  msub(float acc, float a, float b)    { return fmaf(-a, b, acc); }
DCP_HD double msub(double acc, double a, double b) { return fma(-a, b, acc); }
template <class R> DCP_HD cx<R> msub(cx<R> acc, cx<R> a, cx<R> b) {
    acc.re -= a.re * b.re - a.im * b.im;
    acc.im -= a.re * b.im + a.im * b.re;
    return acc;
}

// acc + a*b and acc - a*b as EXPLICIT fused multiply-adds in a fixed order (complex: re = fma(-a.im, b.im,
// fma(a.re, b.re, acc.re)), im = fma(a.im, b.re, fma(a.re, b.im, acc.im))): the compiler's own contraction of
// mul + add depends on the surrounding code, these do not -- two kernels that must agree bit for bit (the register
// and the memory-resident coordinate-descent sweeps) use them.
DCP_HD float  fmadd(float acc, float a, float b)    { return fmaf(a, b, acc); }
DCP_HD double fmadd(double acc, double a, double b) { return fma(a, b, acc); }
DCP_HD c64 fmadd(c64 acc, c64 a, c64 b) {
    return c64{fmaf(-a.im, b.im, fmaf(a.re, b.re, acc.re)), fmaf(a.im, b.re, fmaf(a.re, b.im, acc.im))};
}
DCP_HD c128 fmadd(c128 acc, c128 a, c128 b) {
    return c128{fma(-a.im, b.im, fma(a.re, b.re, acc.re)), fma(a.im, b.re, fma(a.re, b.im, acc.im))};
}
DCP_HD float  fmsub(float acc, float a, float b)    { return fmaf(-a, b, acc); }
DCP_HD double fmsub(double acc, double a, double b) { return fma(-a, b, acc); }
DCP_HD c64 fmsub(c64 acc, c64 a, c64 b) {
    return c64{fmaf(a.im, b.im, fmaf(-a.re, b.re, acc.re)), fmaf(-a.im, b.re, fmaf(-a.re, b.im, acc.im))};
}
DCP_HD c128 fmsub(c128 acc, c128 a, c128 b) {
    return c128{fma(a.im, b.im, fma(-a.re, b.re, acc.re)), fma(-a.im, b.re, fma(-a.re, b.im, acc.im))};
}

DCP_HD float  conj_of(float a)  { return a; }
DCP_HD double conj_of(double a) { return a; }
template <class R> DCP_HD cx<R> conj_of(cx<R> a) { return cx<R>{a.re, -a.im}; }

// |a|^2 and |a|
DCP_HD float  abs2(float a)  { return a * a; }
DCP_HD double abs2(double a) { return a * a; }
template <class R> DCP_HD R abs2(cx<R> a) { return a.re * a.re + a.im * a.im; }

DCP_HD float  absval(float a)  { return fabsf(a); }
DCP_HD double absval(double a) { return fabs(a); }
DCP_HD float  absval(c64 a)  { return hypotf(a.re, a.im); }
DCP_HD double absval(c128 a) { return hypot(a.re, a.im); }

// np.maximum(a, floor): the larger value, and NaN when a is NaN (floor is a finite constant)
DCP_HD float  max_np(float a, float floor_)   { return (a > floor_ || a != a) ? a : floor_; }
DCP_HD double max_np(double a, double floor_) { return (a > floor_ || a != a) ? a : floor_; }

DCP_HD float  real_part(float a)  { return a; }
DCP_HD double real_part(double a) { return a; }
template <class R> DCP_HD R real_part(cx<R> a) { return a.re; }

}  // namespace dcp
