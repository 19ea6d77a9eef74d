// ============================================================================
// oracle/ref_shim/include/eigen_shim.hpp  --  TEST INFRASTRUCTURE ONLY.
//
// A small stand-in for the part of Eigen 3.3 that the reference's registration
// path uses, so that the reference's OWN sources (include/registration/*.h,
// src/ivox_map/*.cpp, src/loam/*.cpp, include/common/*.h) can be compiled
// VERBATIM from /root/reference into oracle/_ref/libref.so (oracle/ref_shim/Makefile)
// and run as the pin of the CPU oracle.  Eigen itself is not in this container.
//
// What this pins and what it does not:
//   * pinned: every line of the reference's own control flow and data handling
//     (loops, gates, stale flags, deques, LRU lists, update rules, std::nth_element /
//     std::sort calls on the reference's own structs -- compiled, not restated);
//   * NOT pinned: Eigen's arithmetic.  Dense factorisations (ColPiv / FullPiv QR, LU,
//     3x3 inverse, 3x3 Jacobi SVD) forward to oracle/flo_linalg.h -- the same code the
//     oracle uses -- and the association order of sums / products below follows this
//     author's reading of Eigen 3.3.7 for SSE2 (redux unrollers, coefficient-based
//     product evaluator).  Oracle-vs-_ref comparisons are therefore exact on integer
//     outputs and to a tight tolerance on FP64 outputs (tests/test_ref_pin.py).
//
// Model of Eigen's evaluation order implemented here (all sizes on this path are
// small and fixed, i.e. completely unrolled in Eigen):
//   redux (sum, dot, norm, squaredNorm, mean) over n terms
//       packet access (all leaves contiguous, n >= P, P = 2 doubles / 4 floats):
//           packets p_k = terms [kP, kP+P), combined by the halving tree of
//           redux_vec_unroller, then predux, then + tree(the n % P tail terms)
//       otherwise: the halving tree of redux_novec_unroller: f(0,n) = f(0,n/2) + f(n/2,n-n/2)
//   product C = A * B, K inner terms
//       1x1 result: redux of a(0,k) b(k,0)
//       A plain column-major with Rows % P == 0: packet path, sequential k = 0..K-1
//       otherwise coefficient path: redux over k; it has packet access only when the rows
//           of A are contiguous (A is a transposed column-major matrix) and B is column-major
//   a value read through a strided view (a row of a column-major matrix) has no packet access,
//   and neither has any un-evaluated expression built from it ("NoPacket" taint below).
// ============================================================================
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include "flo_linalg.h"

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_ALIGN16 __attribute__((aligned(16)))
#define eigen_assert(x) assert(x)
#define EIGEN_SHIM 1

namespace Eigen {

constexpr int Dynamic = -1;
enum { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20 };
enum TransformTraits { Isometry = 0x1, Affine = 0x2, AffineCompact = 0x10 | Affine, Projective = 0x20 };
enum { RowMajorExpr = 1, NoPacket = 2 };  // shim-only expression flags carried in Matrix's 4th parameter

template <class T>
struct aligned_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = aligned_allocator<U>; };
    aligned_allocator() = default;
    template <class U> aligned_allocator(const aligned_allocator<U>&) {}
};

template <class T, int R, int C, int Opt = 0> class Matrix;
template <class T, int R, int C> class View;
template <class T, int R, int C> class ArrayW;
template <class D> class MatrixBase;

namespace internal {
constexpr int pick(int a, int b) { return a == Dynamic ? b : a; }
template <class T> struct packet { static constexpr int size = 1; };
template <> struct packet<float> { static constexpr int size = 4; };
template <> struct packet<double> { static constexpr int size = 2; };
template <class D> struct traits;
template <class T, int R, int C, int O> struct traits<Matrix<T, R, C, O>> { using Scalar = T; enum { Rows = R, Cols = C, Opt = O }; };
// a View is a Block / Map of a column-major matrix: a strided row has no packet access
template <class T, int R, int C> struct traits<View<T, R, C>> { using Scalar = T; enum { Rows = R, Cols = C, Opt = (C == 1 ? 0 : NoPacket) }; };

template <class S, class F>
inline S tree(int start, int len, F&& term) {  // redux_novec_unroller
    if (len == 1) return term(start);
    const int h = len / 2;
    const S a = tree<S>(start, h, term);
    const S b = tree<S>(start + h, len - h, term);
    return a + b;
}
template <class S, class F>
inline S redux(int n, bool packet_ok, F&& term) {
    constexpr int P = packet<S>::size;
    if (n <= 0) return S(0);
    if (!packet_ok || P == 1 || n < P) return tree<S>(0, n, term);
    const int np = n / P;
    std::array<S, 4> acc{};
    // halving tree over the packets, lane-wise
    struct Rec {
        static std::array<S, 4> run(int s, int l, F& t) {
            std::array<S, 4> r{};
            if (l == 1) { for (int q = 0; q < P; ++q) r[q] = t(s * P + q); return r; }
            const int h = l / 2;
            const auto a = run(s, h, t), b = run(s + h, l - h, t);
            for (int q = 0; q < P; ++q) r[q] = a[q] + b[q];
            return r;
        }
    };
    acc = Rec::run(0, np, term);
    S res = (P == 2) ? S(acc[0] + acc[1]) : S((acc[0] + acc[2]) + (acc[1] + acc[3]));  // SSE2 predux (no hadd)
    if (np * P != n) res = res + tree<S>(np * P, n - np * P, term);
    return res;
}
}  // namespace internal

// ---------------------------------------------------------------------------------------------
template <class T, int R, int C> class CommaInit;

template <class D>
class MatrixBase {
public:
    using Scalar = typename internal::traits<D>::Scalar;
    enum { RowsAtCompileTime = internal::traits<D>::Rows, ColsAtCompileTime = internal::traits<D>::Cols, ShimOpt = internal::traits<D>::Opt,
           SizeAtCompileTime = (RowsAtCompileTime == Dynamic || ColsAtCompileTime == Dynamic) ? Dynamic : RowsAtCompileTime * ColsAtCompileTime };
    using Plain = Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime>;
    const D& derived() const { return *static_cast<const D*>(this); }
    D& derived() { return *static_cast<D*>(this); }
    int rows() const { return derived().rows_(); }
    int cols() const { return derived().cols_(); }
    int size() const { return rows() * cols(); }
    Scalar coeff(int i, int j) const { return derived().at_(i, j); }
    Scalar vcoeff(int i) const { return cols() == 1 ? coeff(i, 0) : coeff(0, i); }
    Scalar operator()(int i, int j) const { return coeff(i, j); }
    Scalar operator()(int i) const { return vcoeff(i); }
    Scalar operator[](int i) const { return vcoeff(i); }
    Scalar x() const { return vcoeff(0); }
    Scalar y() const { return vcoeff(1); }
    Scalar z() const { return vcoeff(2); }
    Scalar w() const { return vcoeff(3); }
    static constexpr bool packet_access() { return (ShimOpt & NoPacket) == 0; }

    Plain eval() const { return Plain(*this); }
    Plain matrix() const { return Plain(*this); }
    template <class U> Matrix<U, RowsAtCompileTime, ColsAtCompileTime, ShimOpt> cast() const {
        Matrix<U, RowsAtCompileTime, ColsAtCompileTime, ShimOpt> r(rows(), cols());
        for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) r.ref(i, j) = static_cast<U>(coeff(i, j));
        return r;
    }
    // Transpose<colmajor matrix> is a row-major expression; Transpose<vector> stays linearly accessible
    Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime, (ShimOpt & NoPacket) | ((RowsAtCompileTime != 1 && ColsAtCompileTime != 1 && !(ShimOpt & RowMajorExpr)) ? RowMajorExpr : 0)>
    transpose() const {
        Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime, (ShimOpt & NoPacket) | ((RowsAtCompileTime != 1 && ColsAtCompileTime != 1 && !(ShimOpt & RowMajorExpr)) ? RowMajorExpr : 0)> r(cols(), rows());
        for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) r.ref(j, i) = coeff(i, j);
        return r;
    }
    Scalar sum() const {
        const int n = size(), r = rows();
        return internal::redux<Scalar>(n, packet_access(), [&](int k) { return coeff(k % r, k / r); });
    }
    Scalar mean() const { return sum() / Scalar(size()); }
    Scalar squaredNorm() const {
        const int n = size(), r = rows();
        return internal::redux<Scalar>(n, packet_access(), [&](int k) { const Scalar v = coeff(k % r, k / r); return v * v; });
    }
    Scalar norm() const { return std::sqrt(squaredNorm()); }
    Plain normalized() const {  // Eigen 3.3: z = squaredNorm(); z > 0 ? derived() / sqrt(z) : derived()
        const Scalar z = squaredNorm();
        Plain r(*this);
        if (z > Scalar(0)) { const Scalar s = std::sqrt(z); for (int k = 0; k < r.size(); ++k) r.data()[k] = r.data()[k] / s; }
        return r;
    }
    template <class O> Scalar dot(const MatrixBase<O>& o) const {
        const int n = size();
        return internal::redux<Scalar>(n, packet_access() && MatrixBase<O>::packet_access(), [&](int k) { return vcoeff(k) * o.vcoeff(k); });
    }
    template <class O> Matrix<Scalar, 3, 1> cross(const MatrixBase<O>& o) const {  // Eigen/src/Geometry/OrthoMethods.h
        Matrix<Scalar, 3, 1> r;
        r.ref(0, 0) = vcoeff(1) * o.vcoeff(2) - vcoeff(2) * o.vcoeff(1);
        r.ref(1, 0) = vcoeff(2) * o.vcoeff(0) - vcoeff(0) * o.vcoeff(2);
        r.ref(2, 0) = vcoeff(0) * o.vcoeff(1) - vcoeff(1) * o.vcoeff(0);
        return r;
    }
    template <class O> bool operator==(const MatrixBase<O>& o) const {
        if (rows() != o.rows() || cols() != o.cols()) return false;
        for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) if (!(coeff(i, j) == o.coeff(i, j))) return false;
        return true;
    }
    template <class O> bool operator!=(const MatrixBase<O>& o) const { return !(*this == o); }
    ArrayW<Scalar, RowsAtCompileTime, ColsAtCompileTime> array() const;
    Matrix<Scalar, internal::pick(SizeAtCompileTime, Dynamic), internal::pick(SizeAtCompileTime, Dynamic)> asDiagonal() const {
        const int n = size();
        Matrix<Scalar, internal::pick(SizeAtCompileTime, Dynamic), internal::pick(SizeAtCompileTime, Dynamic)> r(n, n);
        for (int k = 0; k < n; ++k) r.ref(k, k) = vcoeff(k);
        return r;
    }
    // ---- dense decompositions: forwarded to oracle/flo_linalg.h (third-party arithmetic shared with the oracle)
    Plain inverse() const {
        static_assert(std::is_same<Scalar, double>::value, "shim: inverse() only for double");
        Plain r(rows(), cols());
        const Plain a(*this);
        if (rows() == 3 && cols() == 3) flo::inverse3(a.data(), r.data());
        else if (rows() == 6 && cols() == 6) { flo::PartialPivLU<6> lu; lu.compute(a.data()); lu.inverse(r.data()); }
        else throw std::logic_error("eigen_shim: inverse() only for 3x3 and 6x6");
        return r;
    }
    Scalar determinant() const {
        static_assert(std::is_same<Scalar, double>::value, "shim: determinant() only for double");
        const Plain a(*this);
        if (rows() == 6 && cols() == 6) { flo::PartialPivLU<6> lu; lu.compute(a.data()); return lu.determinant(); }
        throw std::logic_error("eigen_shim: determinant() only for 6x6");
    }
    struct QrSolver {
        Plain m; bool full;
        template <class B> Matrix<Scalar, ColsAtCompileTime, 1> solve(const MatrixBase<B>& b) const {
            const Matrix<Scalar, RowsAtCompileTime, 1> bb(b);
            Matrix<Scalar, ColsAtCompileTime, 1> x(m.cols(), 1);
            if (full && m.rows() == 6 && m.cols() == 6) flo::fullpiv_qr_solve<6>(m.data(), bb.data(), x.data());
            else if (!full && m.rows() == 5 && m.cols() == 3) flo::colpiv_qr_solve<5, 3>(m.data(), bb.data(), x.data());
            else throw std::logic_error("eigen_shim: QR solve only for fullPiv 6x6 / colPiv 5x3");
            return x;
        }
    };
    QrSolver fullPivHouseholderQr() const { static_assert(std::is_same<Scalar, double>::value, "double only"); return QrSolver{Plain(*this), true}; }
    QrSolver colPivHouseholderQr() const { static_assert(std::is_same<Scalar, double>::value, "double only"); return QrSolver{Plain(*this), false}; }
    // ---- partial reductions (loam_full_kdtree.h:236-237)
    struct RowwiseOp {
        const D& m;
        Matrix<Scalar, RowsAtCompileTime, 1> mean() const {
            Matrix<Scalar, RowsAtCompileTime, 1> r(m.rows(), 1);
            for (int i = 0; i < m.rows(); ++i)  // a row of a column-major matrix: strided, no packet access
                r.ref(i, 0) = internal::redux<Scalar>(m.cols(), false, [&](int k) { return m.coeff(i, k); }) / Scalar(m.cols());
            return r;
        }
    };
    struct ColwiseOp {
        const D& m;
        template <class V> Plain operator-(const MatrixBase<V>& v) const {
            Plain r(m.rows(), m.cols());
            for (int j = 0; j < m.cols(); ++j) for (int i = 0; i < m.rows(); ++i) r.ref(i, j) = m.coeff(i, j) - v.vcoeff(i);
            return r;
        }
    };
    RowwiseOp rowwise() const { return RowwiseOp{derived()}; }
    ColwiseOp colwise() const { return ColwiseOp{derived()}; }
};

template <class D>
std::ostream& operator<<(std::ostream& os, const MatrixBase<D>& m) {
    for (int i = 0; i < m.rows(); ++i) { for (int j = 0; j < m.cols(); ++j) os << (j ? " " : "") << m.coeff(i, j); if (i + 1 < m.rows()) os << "\n"; }
    return os;
}

// ---------------------------------------------------------------------------------------------
// lvalue mix-in shared by Matrix and View
template <class D, class T, int R, int C>
struct Lvalue {
    D& self() { return *static_cast<D*>(this); }
    template <class O> void assign_(const MatrixBase<O>& o) {
        D& s = self();
        s.resize_like_(o.rows(), o.cols());
        assert(s.rows_() == o.rows() && s.cols_() == o.cols());
        // the right-hand side may alias this object (T.block = R * T.block): expressions are evaluated eagerly into
        // owning temporaries before they get here, except when o is itself a view -> copy through a temporary
        if (std::is_same<O, Matrix<typename internal::traits<O>::Scalar, internal::traits<O>::Rows, internal::traits<O>::Cols, internal::traits<O>::Opt>>::value) {
            for (int j = 0; j < s.cols_(); ++j) for (int i = 0; i < s.rows_(); ++i) s.ref(i, j) = T(o.coeff(i, j));
        } else {
            std::vector<T> tmp(size_t(o.rows()) * size_t(o.cols()));
            for (int j = 0; j < o.cols(); ++j) for (int i = 0; i < o.rows(); ++i) tmp[size_t(i) + size_t(j) * size_t(o.rows())] = T(o.coeff(i, j));
            for (int j = 0; j < s.cols_(); ++j) for (int i = 0; i < s.rows_(); ++i) s.ref(i, j) = tmp[size_t(i) + size_t(j) * size_t(o.rows())];
        }
    }
    T& operator()(int i, int j) { return self().ref(i, j); }
    T& operator()(int i) { return self().cols_() == 1 ? self().ref(i, 0) : self().ref(0, i); }
    T& operator[](int i) { return (*this)(i); }
    T& x() { return (*this)(0); }
    T& y() { return (*this)(1); }
    T& z() { return (*this)(2); }
    T& w() { return (*this)(3); }
    D& setZero() { return setConstant(T(0)); }
    D& setConstant(T v) { D& s = self(); for (int j = 0; j < s.cols_(); ++j) for (int i = 0; i < s.rows_(); ++i) s.ref(i, j) = v; return s; }
    D& setIdentity() { D& s = self(); for (int j = 0; j < s.cols_(); ++j) for (int i = 0; i < s.rows_(); ++i) s.ref(i, j) = (i == j) ? T(1) : T(0); return s; }
    template <class O> D& operator+=(const MatrixBase<O>& o) { D& s = self(); const auto t = o.eval(); for (int j = 0; j < s.cols_(); ++j) for (int i = 0; i < s.rows_(); ++i) s.ref(i, j) = s.ref(i, j) + t.coeff(i, j); return s; }
    template <class O> D& operator-=(const MatrixBase<O>& o) { D& s = self(); const auto t = o.eval(); for (int j = 0; j < s.cols_(); ++j) for (int i = 0; i < s.rows_(); ++i) s.ref(i, j) = s.ref(i, j) - t.coeff(i, j); return s; }
    template <class O> D& operator*=(const MatrixBase<O>& o) { D& s = self(); const auto t = (static_cast<const MatrixBase<D>&>(s) * o).eval(); s.assign_(t); return s; }
    template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
    D& operator*=(S v) { D& s = self(); for (int j = 0; j < s.cols_(); ++j) for (int i = 0; i < s.rows_(); ++i) s.ref(i, j) = s.ref(i, j) * T(v); return s; }
    template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
    D& operator/=(S v) { D& s = self(); for (int j = 0; j < s.cols_(); ++j) for (int i = 0; i < s.rows_(); ++i) s.ref(i, j) = s.ref(i, j) / T(v); return s; }
    // blocks (lvalue views)
    template <int BR, int BC> View<T, BR, BC> block(int i, int j) { D& s = self(); return View<T, BR, BC>(&s.ref(i, j), BR, BC, s.outer_()); }
    View<T, Dynamic, Dynamic> block(int i, int j, int r, int c) { D& s = self(); return View<T, Dynamic, Dynamic>(&s.ref(i, j), r, c, s.outer_()); }
    View<T, 1, C> row(int i) { D& s = self(); return View<T, 1, C>(&s.ref(i, 0), 1, s.cols_(), s.outer_()); }
    View<T, R, 1> col(int j) { D& s = self(); return View<T, R, 1>(&s.ref(0, j), s.rows_(), 1, s.outer_()); }
    View<T, R, Dynamic> leftCols(int n) { D& s = self(); return View<T, R, Dynamic>(&s.ref(0, 0), s.rows_(), n, s.outer_()); }
    View<T, R, Dynamic> rightCols(int n) { D& s = self(); return View<T, R, Dynamic>(&s.ref(0, s.cols_() - n), s.rows_(), n, s.outer_()); }
    View<T, Dynamic, 1> head(int n) { D& s = self(); assert(s.cols_() == 1); return View<T, Dynamic, 1>(&s.ref(0, 0), n, 1, s.outer_()); }
    View<T, Dynamic, 1> tail(int n) { D& s = self(); assert(s.cols_() == 1); return View<T, Dynamic, 1>(&s.ref(s.rows_() - n, 0), n, 1, s.outer_()); }
    template <int N> View<T, N, 1> head() { D& s = self(); return View<T, N, 1>(&s.ref(0, 0), N, 1, s.outer_()); }
    template <int N> View<T, N, 1> tail() { D& s = self(); return View<T, N, 1>(&s.ref(s.rows_() - N, 0), N, 1, s.outer_()); }
    template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
    CommaInit<T, R, C> operator<<(S v);
};

// read-only blocks of const objects: evaluated copies (the reference only reads through them)
#define EIGEN_SHIM_CONST_BLOCKS(T, R, C)                                                                                                        \
    template <int BR, int BC> Matrix<T, BR, BC, (BC == 1 ? 0 : NoPacket)> block(int i, int j) const {                                           \
        Matrix<T, BR, BC, (BC == 1 ? 0 : NoPacket)> r; for (int b = 0; b < BC; ++b) for (int a = 0; a < BR; ++a) r.ref(a, b) = this->at_(i + a, j + b); return r; } \
    Matrix<T, Dynamic, Dynamic, NoPacket> block(int i, int j, int rr, int cc) const {                                                           \
        Matrix<T, Dynamic, Dynamic, NoPacket> r(rr, cc); for (int b = 0; b < cc; ++b) for (int a = 0; a < rr; ++a) r.ref(a, b) = this->at_(i + a, j + b); return r; } \
    Matrix<T, 1, C, NoPacket> row(int i) const { Matrix<T, 1, C, NoPacket> r(1, this->cols_()); for (int b = 0; b < this->cols_(); ++b) r.ref(0, b) = this->at_(i, b); return r; } \
    Matrix<T, R, 1> col(int j) const { Matrix<T, R, 1> r(this->rows_(), 1); for (int a = 0; a < this->rows_(); ++a) r.ref(a, 0) = this->at_(a, j); return r; } \
    Matrix<T, Dynamic, 1> head(int n) const { Matrix<T, Dynamic, 1> r(n, 1); for (int a = 0; a < n; ++a) r.ref(a, 0) = this->at_(a, 0); return r; } \
    Matrix<T, Dynamic, 1> tail(int n) const { Matrix<T, Dynamic, 1> r(n, 1); for (int a = 0; a < n; ++a) r.ref(a, 0) = this->at_(this->rows_() - n + a, 0); return r; } \
    template <int N> Matrix<T, N, 1> head() const { Matrix<T, N, 1> r; for (int a = 0; a < N; ++a) r.ref(a, 0) = this->at_(a, 0); return r; }   \
    template <int N> Matrix<T, N, 1> tail() const { Matrix<T, N, 1> r; for (int a = 0; a < N; ++a) r.ref(a, 0) = this->at_(this->rows_() - N + a, 0); return r; } \
    Matrix<T, R, Dynamic> leftCols(int n) const { Matrix<T, R, Dynamic> r(this->rows_(), n); for (int b = 0; b < n; ++b) for (int a = 0; a < this->rows_(); ++a) r.ref(a, b) = this->at_(a, b); return r; } \
    Matrix<T, R, Dynamic> rightCols(int n) const { Matrix<T, R, Dynamic> r(this->rows_(), n); for (int b = 0; b < n; ++b) for (int a = 0; a < this->rows_(); ++a) r.ref(a, b) = this->at_(a, this->cols_() - n + b); return r; }

// ---------------------------------------------------------------------------------------------
// 1x1 products read as scalars (Eigen: inner products convert implicitly)
template <class D, class T, bool OneByOne> struct ScalarConv {};
template <class D, class T> struct ScalarConv<D, T, true> { operator T() const { return static_cast<const D*>(this)->at_(0, 0); } };

template <class T, int R, int C, int Opt>
class Matrix : public MatrixBase<Matrix<T, R, C, Opt>>, public Lvalue<Matrix<T, R, C, Opt>, T, R, C>, public ScalarConv<Matrix<T, R, C, Opt>, T, R == 1 && C == 1> {
    static constexpr bool kFixed = (R != Dynamic && C != Dynamic);
    typename std::conditional<kFixed, std::array<T, size_t(kFixed ? R * C : 1)>, std::vector<T>>::type d_;
    int r_ = (R == Dynamic ? 0 : R), c_ = (C == Dynamic ? 0 : C);
    void alloc_(std::true_type) { d_.fill(T(0)); }
    void alloc_(std::false_type) { d_.assign(size_t(r_) * size_t(c_), T(0)); }
    using LV = Lvalue<Matrix<T, R, C, Opt>, T, R, C>;
    using MB = MatrixBase<Matrix<T, R, C, Opt>>;

public:
    using Scalar = T;
    using LV::operator();
    using MB::operator();
    using LV::operator[];
    using MB::operator[];
    using LV::x; using MB::x; using LV::y; using MB::y; using LV::z; using MB::z; using LV::w; using MB::w;
    using LV::block; using LV::row; using LV::col; using LV::head; using LV::tail; using LV::leftCols; using LV::rightCols;
    EIGEN_SHIM_CONST_BLOCKS(T, R, C)

    Matrix() { alloc_(std::integral_constant<bool, kFixed>()); }
    // (rows, cols) for dynamic types, (x, y) for fixed 2-vectors
    template <class A, class B, class = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>::type>
    Matrix(A a, B b) {
        if (kFixed && R * C == 2) { alloc_(std::integral_constant<bool, kFixed>()); d_[0] = T(a); d_[1] = T(b); }  // Vector2(x, y)
        else {  // (rows, cols): sizes of a dynamic type, or the (redundant) sizes of a fixed one
            r_ = (R == Dynamic ? int(a) : R); c_ = (C == Dynamic ? int(b) : C);
            assert(int(a) == r_ && int(b) == c_);
            alloc_(std::integral_constant<bool, kFixed>());
        }
    }
    template <class A, class = typename std::enable_if<std::is_integral<A>::value && !kFixed && sizeof(A) != 0>::type>
    explicit Matrix(A n) { if (R == Dynamic) r_ = int(n); if (C == Dynamic) c_ = int(n); alloc_(std::integral_constant<bool, kFixed>()); }
    template <class A, class B, class E, class = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value && std::is_arithmetic<E>::value>::type>
    Matrix(A a, B b, E c) { alloc_(std::integral_constant<bool, kFixed>()); static_assert(R * C == 3, "3-vector ctor"); d_[0] = T(a); d_[1] = T(b); d_[2] = T(c); }
    template <class A, class B, class E, class F, class = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<F>::value>::type>
    Matrix(A a, B b, E c, F e) { alloc_(std::integral_constant<bool, kFixed>()); static_assert(R * C == 4, "4-vector ctor"); d_[0] = T(a); d_[1] = T(b); d_[2] = T(c); d_[3] = T(e); }
    Matrix(const Matrix&) = default;
    Matrix(Matrix&&) = default;
    Matrix& operator=(const Matrix&) = default;
    Matrix& operator=(Matrix&&) = default;
    template <class O> Matrix(const MatrixBase<O>& o) { r_ = o.rows(); c_ = o.cols(); check_(); alloc_(std::integral_constant<bool, kFixed>()); this->assign_(o); }
    template <class O> Matrix& operator=(const MatrixBase<O>& o) { this->assign_(o); return *this; }
    template <class U, int AR, int AC> Matrix(const ArrayW<U, AR, AC>& a);
    template <class U, int AR, int AC> Matrix& operator=(const ArrayW<U, AR, AC>& a) { this->assign_(a.m); return *this; }

    void check_() const {
        if ((R != Dynamic && r_ != R) || (C != Dynamic && c_ != C)) {
            // a dynamic vector assigned to a fixed one of the same length in the other orientation is not supported
            throw std::logic_error("eigen_shim: size mismatch in fixed-size construction");
        }
    }
    int rows_() const { return r_; }
    int cols_() const { return c_; }
    int outer_() const { return r_; }
    T at_(int i, int j) const { return d_[size_t(i) + size_t(j) * size_t(r_)]; }
    T& ref(int i, int j) { return d_[size_t(i) + size_t(j) * size_t(r_)]; }
    T* data() { return d_.data(); }
    const T* data() const { return d_.data(); }
    void resize_like_(int r, int c) { if (!kFixed && (r != r_ || c != c_)) resize(r, c); }
    void resize(int r, int c) { assert((R == Dynamic || r == R) && (C == Dynamic || c == C)); r_ = r; c_ = c; alloc_(std::integral_constant<bool, kFixed>()); }
    void resize(int n) { if (R == Dynamic) r_ = n; else c_ = n; alloc_(std::integral_constant<bool, kFixed>()); }

    static Matrix Zero() { return Matrix(); }
    static Matrix Zero(int r, int c) { return Matrix(r, c); }
    static Matrix Ones() { Matrix m; m.setConstant(T(1)); return m; }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix Identity(int r, int c) { Matrix m(r, c); m.setIdentity(); return m; }
    static Matrix Constant(T v) { Matrix m; m.setConstant(v); return m; }
};

// a strided window onto a Matrix (Block / VectorBlock / Map): always an lvalue here
template <class T, int R, int C>
class View : public MatrixBase<View<T, R, C>>, public Lvalue<View<T, R, C>, T, R, C> {
    T* p_; int r_, c_, os_;
    using LV = Lvalue<View<T, R, C>, T, R, C>;
    using MB = MatrixBase<View<T, R, C>>;

public:
    using Scalar = T;
    using LV::operator(); using MB::operator(); using LV::operator[]; using MB::operator[];
    using LV::x; using MB::x; using LV::y; using MB::y; using LV::z; using MB::z; using LV::w; using MB::w;
    using LV::block; using LV::row; using LV::col; using LV::head; using LV::tail; using LV::leftCols; using LV::rightCols;
    View(T* p, int r, int c, int os) : p_(p), r_(r), c_(c), os_(os) {}
    View(const View&) = default;
    View& operator=(const View& o) { this->assign_(o); return *this; }
    template <class O> View& operator=(const MatrixBase<O>& o) { this->assign_(o); return *this; }
    template <class U, int AR, int AC> View& operator=(const ArrayW<U, AR, AC>& a) { this->assign_(a.m); return *this; }
    int rows_() const { return r_; }
    int cols_() const { return c_; }
    int outer_() const { return os_; }
    T at_(int i, int j) const { return p_[size_t(i) + size_t(j) * size_t(os_)]; }
    T& ref(int i, int j) { return p_[size_t(i) + size_t(j) * size_t(os_)]; }
    void resize_like_(int, int) {}
};

template <class PlainVector> using Map = PlainVector;  // getVector3fMap() returns by value in the pcl shim (read-only uses)

// A.row(j) << a, b, c;
template <class T, int R, int C>
class CommaInit {
    View<T, Dynamic, Dynamic> v_; int k_;
public:
    CommaInit(View<T, Dynamic, Dynamic> v, T first) : v_(v), k_(0) { put(first); }
    void put(T x) { const int r = v_.rows(), c = v_.cols(); const int i = k_ / c, j = k_ % c; assert(i < r); (void)r; v_.ref(i, j) = x; ++k_; }  // row by row
    template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
    CommaInit& operator,(S x) { put(T(x)); return *this; }
};
template <class D, class T, int R, int C>
template <class S, class>
CommaInit<T, R, C> Lvalue<D, T, R, C>::operator<<(S v) {
    D& s = self();
    return CommaInit<T, R, C>(View<T, Dynamic, Dynamic>(&s.ref(0, 0), s.rows_(), s.cols_(), s.outer_()), T(v));
}

// ---------------------------------------------------------------------------------------------
// element-wise expressions (evaluated eagerly; the NoPacket taint and the row-major flag are inherited)
template <class A, class B>
using SumType = Matrix<typename internal::traits<A>::Scalar, internal::pick(internal::traits<A>::Rows, internal::traits<B>::Rows),
                       internal::pick(internal::traits<A>::Cols, internal::traits<B>::Cols),
                       ((internal::traits<A>::Opt | internal::traits<B>::Opt) & NoPacket) | (internal::traits<A>::Opt & internal::traits<B>::Opt & RowMajorExpr)>;
template <class A, class B> SumType<A, B> operator+(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    assert(a.rows() == b.rows() && a.cols() == b.cols());
    SumType<A, B> r(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref(i, j) = a.coeff(i, j) + b.coeff(i, j);
    return r;
}
template <class A, class B> SumType<A, B> operator-(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    assert(a.rows() == b.rows() && a.cols() == b.cols());
    SumType<A, B> r(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref(i, j) = a.coeff(i, j) - b.coeff(i, j);
    return r;
}
template <class A> using SameType = Matrix<typename internal::traits<A>::Scalar, internal::traits<A>::Rows, internal::traits<A>::Cols, internal::traits<A>::Opt>;
template <class A> SameType<A> operator-(const MatrixBase<A>& a) {
    SameType<A> r(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref(i, j) = -a.coeff(i, j);
    return r;
}
template <class A, class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
SameType<A> operator*(const MatrixBase<A>& a, S s) {
    using T = typename internal::traits<A>::Scalar;
    SameType<A> r(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref(i, j) = a.coeff(i, j) * T(s);
    return r;
}
template <class A, class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
SameType<A> operator*(S s, const MatrixBase<A>& a) {
    using T = typename internal::traits<A>::Scalar;
    SameType<A> r(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref(i, j) = T(s) * a.coeff(i, j);
    return r;
}
template <class A, class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
SameType<A> operator/(const MatrixBase<A>& a, S s) {
    using T = typename internal::traits<A>::Scalar;
    SameType<A> r(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref(i, j) = a.coeff(i, j) / T(s);
    return r;
}

// matrix product (see the header comment for the evaluation-order model)
template <class A, class B>
Matrix<typename internal::traits<A>::Scalar, internal::traits<A>::Rows, internal::traits<B>::Cols>
operator*(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    using T = typename internal::traits<A>::Scalar;
    static_assert(std::is_same<T, typename internal::traits<B>::Scalar>::value, "mixed scalar product");
    constexpr int AR = internal::traits<A>::Rows, AC = internal::traits<A>::Cols, BC = internal::traits<B>::Cols, BR = internal::traits<B>::Rows;
    constexpr int AO = internal::traits<A>::Opt, BO = internal::traits<B>::Opt;
    constexpr int P = internal::packet<T>::size;
    assert(a.cols() == b.rows());
    const int M = a.rows(), N = b.cols(), K = a.cols();
    Matrix<T, AR, BC> r(M, N);
    const bool taint = ((AO | BO) & NoPacket) != 0;
    // contiguous rows of A: a transposed column-major matrix, or a row vector that is a transposed column vector
    const bool a_rows_contig = (AO & RowMajorExpr) || (AR == 1);
    const bool b_cols_contig = !(BO & RowMajorExpr) || (BC == 1 && BR != 1);
    if (AR == 1 && BC == 1) {  // inner product
        r.ref(0, 0) = internal::redux<T>(K, !taint, [&](int k) { return a.coeff(0, k) * b.coeff(k, 0); });
        return r;
    }
    if (K == 1 || AC == 1) {  // outer product: one multiplication per coefficient
        for (int j = 0; j < N; ++j) for (int i = 0; i < M; ++i) r.ref(i, j) = a.coeff(i, 0) * b.coeff(0, j);
        return r;
    }
    const bool lhs_packets = !(AO & RowMajorExpr) && !(AO & NoPacket) && AR != Dynamic && AR != 1 && (AR % P == 0) && P > 1;
    if (lhs_packets) {  // etor_product_packet_impl<ColMajor>: res = a(:,0) b(0,j); res = a(:,k) b(k,j) + res
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < M; ++i) {
                T acc = a.coeff(i, 0) * b.coeff(0, j);
                for (int k = 1; k < K; ++k) acc = a.coeff(i, k) * b.coeff(k, j) + acc;
                r.ref(i, j) = acc;
            }
        return r;
    }
    const bool ok = !taint && a_rows_contig && b_cols_contig;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) r.ref(i, j) = internal::redux<T>(K, ok, [&](int k) { return a.coeff(i, k) * b.coeff(k, j); });
    return r;
}

// ---------------------------------------------------------------------------------------------
// .array(): floor / round / + scalar / * scalar / cast, assignable to a matrix
template <class T, int R, int C>
class ArrayW {
public:
    Matrix<T, R, C> m;
    explicit ArrayW(const Matrix<T, R, C>& mm) : m(mm) {}
    ArrayW floor() const { ArrayW r(m); for (int k = 0; k < m.size(); ++k) r.m.data()[k] = std::floor(m.data()[k]); return r; }
    ArrayW round() const { ArrayW r(m); for (int k = 0; k < m.size(); ++k) r.m.data()[k] = std::round(m.data()[k]); return r; }
    ArrayW abs() const { ArrayW r(m); for (int k = 0; k < m.size(); ++k) r.m.data()[k] = std::abs(m.data()[k]); return r; }
    template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
    ArrayW operator+(S s) const { ArrayW r(m); for (int k = 0; k < m.size(); ++k) r.m.data()[k] = m.data()[k] + T(s); return r; }
    template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
    ArrayW operator-(S s) const { ArrayW r(m); for (int k = 0; k < m.size(); ++k) r.m.data()[k] = m.data()[k] - T(s); return r; }
    template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
    ArrayW operator*(S s) const { ArrayW r(m); for (int k = 0; k < m.size(); ++k) r.m.data()[k] = m.data()[k] * T(s); return r; }
    template <class U> ArrayW<U, R, C> cast() const { return ArrayW<U, R, C>(m.template cast<U>()); }
    Matrix<T, R, C> matrix() const { return m; }
};
template <class D>
ArrayW<typename MatrixBase<D>::Scalar, MatrixBase<D>::RowsAtCompileTime, MatrixBase<D>::ColsAtCompileTime> MatrixBase<D>::array() const {
    return ArrayW<Scalar, RowsAtCompileTime, ColsAtCompileTime>(Plain(*this));
}
template <class T, int R, int C, int Opt>
template <class U, int AR, int AC>
Matrix<T, R, C, Opt>::Matrix(const ArrayW<U, AR, AC>& a) {
    r_ = a.m.rows(); c_ = a.m.cols(); check_(); alloc_(std::integral_constant<bool, kFixed>()); this->assign_(a.m);
}

// ---------------------------------------------------------------------------------------------
template <class T, int Dim, int Mode>
class Transform {
    Matrix<T, Dim + 1, Dim + 1> m_;
public:
    Transform() { m_.setIdentity(); }
    template <class O> explicit Transform(const MatrixBase<O>& m) : m_(m) {}
    const Matrix<T, Dim + 1, Dim + 1>& matrix() const { return m_; }
    Matrix<T, Dim + 1, Dim + 1>& matrix() { return m_; }
};
using Affine3d = Transform<double, 3, Affine>;
using Affine3f = Transform<float, 3, Affine>;

template <class T>
class Quaternion {  // storage only: nothing on the compiled path does quaternion algebra
    T c_[4] = {0, 0, 0, 1};
public:
    Quaternion() = default;
    Quaternion(T w, T x, T y, T z) { c_[0] = x; c_[1] = y; c_[2] = z; c_[3] = w; }
    static Quaternion Identity() { return Quaternion(1, 0, 0, 0); }
    T w() const { return c_[3]; }
    T x() const { return c_[0]; }
    T y() const { return c_[1]; }
    T z() const { return c_[2]; }
};
using Quaterniond = Quaternion<double>;
using Quaternionf = Quaternion<float>;

template <class MatrixType>
class JacobiSVD {  // 3x3 double only (incremental_ndt.h:166, loam_full_kdtree.h:244): forwards to flo::jacobi_svd3
    using T = typename MatrixType::Scalar;
    enum { Rr = internal::traits<MatrixType>::Rows, Cc = internal::traits<MatrixType>::Cols };
    Matrix<T, Rr, Rr> u_;
    Matrix<T, Cc, Cc> v_;
    Matrix<T, internal::pick(Rr, Cc), 1> s_;
public:
    using SingularValuesType = Matrix<T, internal::pick(Rr, Cc), 1>;
    template <class O> JacobiSVD(const MatrixBase<O>& m, unsigned int = 0) {
        static_assert(std::is_same<T, double>::value, "shim: JacobiSVD only for double");
        if (m.rows() != 3 || m.cols() != 3) throw std::logic_error("eigen_shim: JacobiSVD only for 3x3");
        const Matrix<double, 3, 3> a(m);
        Matrix<double, 3, 3> U, V;
        Matrix<double, 3, 1> S;
        flo::jacobi_svd3(a.data(), U.data(), S.data(), V.data());
        u_ = U; v_ = V; s_ = S;
    }
    const SingularValuesType& singularValues() const { return s_; }
    const Matrix<T, Rr, Rr>& matrixU() const { return u_; }
    const Matrix<T, Cc, Cc>& matrixV() const { return v_; }
};
template <class O> JacobiSVD(const MatrixBase<O>&, unsigned int) -> JacobiSVD<typename MatrixBase<O>::Plain>;
template <class O> JacobiSVD(const MatrixBase<O>&) -> JacobiSVD<typename MatrixBase<O>::Plain>;

#define EIGEN_SHIM_TYPEDEFS(T, S)                                   \
    using Vector2##S = Matrix<T, 2, 1>; using Vector3##S = Matrix<T, 3, 1>; using Vector4##S = Matrix<T, 4, 1>; \
    using VectorX##S = Matrix<T, Dynamic, 1>;                      \
    using Matrix2##S = Matrix<T, 2, 2>; using Matrix3##S = Matrix<T, 3, 3>; using Matrix4##S = Matrix<T, 4, 4>; \
    using MatrixX##S = Matrix<T, Dynamic, Dynamic>;
EIGEN_SHIM_TYPEDEFS(float, f)
EIGEN_SHIM_TYPEDEFS(double, d)
EIGEN_SHIM_TYPEDEFS(int, i)
#undef EIGEN_SHIM_TYPEDEFS

}  // namespace Eigen
