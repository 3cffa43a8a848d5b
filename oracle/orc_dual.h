/*
 * orc_dual.h -- forward-mode dual numbers for the oracle's restatement of the geometry-attached part of
 * PRBIntegrator.sample (src/python/python/ad/integrators/prb.py:124-141, 176-216, 261-297).
 *
 * TEST INFRASTRUCTURE (see mi_oracle.h): the reference obtains these derivatives from Dr.Jit's reverse-mode AD; the
 * oracle writes the attached computation down literally (replace_grad, relative_grad, detach as in the Python source) over a
 * dual type that carries the partial derivatives w.r.t. the 9 coordinates of the triangle a path vertex lies on.  No derivative formula is derived by hand here, which is
 * what makes it an independent check of the product's hand-derived adjoint kernels.  Values and derivatives are double.
 */
#pragma once
#include <cmath>

namespace orc {

template <int N> struct Dual {
    double v; double d[N];
    Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
    Dual(double c) : v(c) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
    static Dual param(double value, int slot) { Dual r(value); r.d[slot] = 1.0; return r; }
};
template <int N> static inline Dual<N> operator+(const Dual<N> &a, const Dual<N> &b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> static inline Dual<N> operator-(const Dual<N> &a, const Dual<N> &b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> static inline Dual<N> operator-(const Dual<N> &a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> static inline Dual<N> operator*(const Dual<N> &a, const Dual<N> &b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> static inline Dual<N> operator/(const Dual<N> &a, const Dual<N> &b) {
    Dual<N> r; double ib = 1.0 / b.v; r.v = a.v * ib;
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
template <int N> static inline Dual<N> operator*(const Dual<N> &a, double s) { Dual<N> r; r.v = a.v * s; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> static inline Dual<N> dsqrt(const Dual<N> &a) { Dual<N> r; r.v = std::sqrt(a.v); double k = 0.5 / r.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
template <int N> static inline Dual<N> dabs(const Dual<N> &a) { return a.v < 0.0 ? -a : a; }
/* dr.replace_grad(a, b): the value of a, the derivative of b */
template <int N> static inline Dual<N> replace_grad(double value, const Dual<N> &g) { Dual<N> r = g; r.v = value; return r; }

template <int N> struct Dual3 {
    Dual<N> x, y, z;
    Dual3() {}
    Dual3(double a, double b, double c) : x(a), y(b), z(c) {}
    Dual3(const Dual<N> &a, const Dual<N> &b, const Dual<N> &c) : x(a), y(b), z(c) {}
};
template <int N> static inline Dual3<N> operator+(const Dual3<N> &a, const Dual3<N> &b) { return Dual3<N>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <int N> static inline Dual3<N> operator-(const Dual3<N> &a, const Dual3<N> &b) { return Dual3<N>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <int N> static inline Dual3<N> operator*(const Dual3<N> &a, const Dual<N> &s) { return Dual3<N>(a.x * s, a.y * s, a.z * s); }
template <int N> static inline Dual3<N> operator*(const Dual3<N> &a, double s) { return Dual3<N>(a.x * s, a.y * s, a.z * s); }
template <int N> static inline Dual<N> ddot(const Dual3<N> &a, const Dual3<N> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <int N> static inline Dual3<N> dcross(const Dual3<N> &a, const Dual3<N> &b) {
    return Dual3<N>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <int N> static inline Dual3<N> dnormalize(const Dual3<N> &a) { Dual<N> il = Dual<N>(1.0) / dsqrt(ddot(a, a)); return a * il; }
template <int N> static inline Dual3<N> dvalue(const Dual3<N> &a) { return Dual3<N>(a.x.v, a.y.v, a.z.v); }       /* dr.detach */
template <int N> static inline Dual3<N> replace_grad3(double vx, double vy, double vz, const Dual3<N> &g) {
    return Dual3<N>(replace_grad(vx, g.x), replace_grad(vy, g.y), replace_grad(vz, g.z));
}

} // namespace orc
