// oracle/refshim/mini_eigen.hpp — TEST INFRASTRUCTURE (builds oracle/_ref), not product code.
//
// A minimal header-only stand-in for exactly the Eigen surface the reference's
// own sources use (Numerics.h, PreIntegrator.cc, Ransac.cc, Updater.cc,
// Tracker.cc, System.cc), so that those files can be compiled UNMODIFIED from
// /root/reference into oracle/_ref/libref.so and run beside the restatement in
// oracle/filter.cpp.  Eigen itself is not installed in this image
//  and is not copied here: this file is written from Eigen 3.3's
// documented semantics:
//   * column-major dense storage, coefficient-wise lazy expressions for
//     + - scalar* unary- transpose block diagonal, evaluated column-major straight
//     into the destination WITHOUT a temporary — so `A = .5*(A + A.transpose())`
//     aliases exactly as Eigen's evaluator does in a Release build (C.4);
//   * products / solve / inverse evaluate into a temporary first (C.5);
//   * JacobiRotation::makeGivens + applyOnTheLeft (C.1);
//   * ColPivHouseholderQR::solve (C.2), PartialPivLU::inverse (C.3).
// Dot products are summed in index order.  Eigen's SSE2 kernels pair some sums
// differently (redux over packets, 4-column GEMV groups); that is last-bit
// noise this file does not claim to reproduce.
#ifndef RVIO_REFSHIM_MINI_EIGEN_HPP
#define RVIO_REFSHIM_MINI_EIGEN_HPP
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <limits>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

const int Dynamic = -1;
typedef int Index;

template <class S, int R, int C> class Matrix;
template <class X> class Block;
template <class X> class Transpose;
template <class X> class Diagonal;
template <class A, class B, int Op> class CwiseBinary;
template <class A, int Op> class CwiseUnary;
template <class S> class JacobiRotation;
template <class M> class ColPivHouseholderQR;
template <class D> struct traits;

// operands: plain matrices are held by reference, expression nodes by value
template <class T> struct nested { typedef T type; };
template <class S, int R, int C> struct nested<Matrix<S, R, C> > { typedef const Matrix<S, R, C>& type; };
template <class S, int R, int C> struct nested<const Matrix<S, R, C> > { typedef const Matrix<S, R, C>& type; };
template <class T> struct lv_nested { typedef T type; };
template <class S, int R, int C> struct lv_nested<Matrix<S, R, C> > { typedef Matrix<S, R, C>& type; };

template <class D> class CommaInitializer;
template <class D> class NoAlias;

template <class D>
class MatrixBase {
public:
    typedef typename traits<D>::Scalar Scalar;
    enum { RowsAtCompileTime = traits<D>::Rows, ColsAtCompileTime = traits<D>::Cols };
    typedef Matrix<Scalar, traits<D>::Rows, traits<D>::Cols> PlainObject;

    const D& derived() const { return *static_cast<const D*>(this); }
    D& derived() { return *static_cast<D*>(this); }
    int rows() const { return derived().rows(); }
    int cols() const { return derived().cols(); }
    int size() const { return rows() * cols(); }
    Scalar coeff(int i, int j) const { return derived().coeff(i, j); }
    Scalar coeff(int i) const { return cols() == 1 ? coeff(i, 0) : coeff(0, i); }

    Scalar operator()(int i, int j) const { return coeff(i, j); }
    Scalar operator()(int i) const { return coeff(i); }
    decltype(auto) operator()(int i, int j) { return derived().coeffRef(i, j); }
    decltype(auto) operator()(int i) { return cols() == 1 ? derived().coeffRef(i, 0) : derived().coeffRef(0, i); }

    PlainObject eval() const { return PlainObject(derived()); }

    // ---- views
    Transpose<const D> transpose() const { return Transpose<const D>(derived()); }
    Block<const D> block(int i, int j, int r, int c) const { return Block<const D>(derived(), i, j, r, c); }
    Block<D> block(int i, int j, int r, int c) { return Block<D>(derived(), i, j, r, c); }
    template <int R, int C> Block<const D> block(int i, int j) const { return Block<const D>(derived(), i, j, R, C); }
    template <int R, int C> Block<D> block(int i, int j) { return Block<D>(derived(), i, j, R, C); }
    Block<const D> col(int j) const { return block(0, j, rows(), 1); }
    Block<D> col(int j) { return block(0, j, rows(), 1); }
    Block<const D> row(int i) const { return block(i, 0, 1, cols()); }
    Block<D> row(int i) { return block(i, 0, 1, cols()); }
    Block<const D> head(int n) const { return cols() == 1 ? block(0, 0, n, 1) : block(0, 0, 1, n); }
    Block<D> head(int n) { return cols() == 1 ? block(0, 0, n, 1) : block(0, 0, 1, n); }
    Block<const D> tail(int n) const { return cols() == 1 ? block(rows() - n, 0, n, 1) : block(0, cols() - n, 1, n); }
    Block<D> tail(int n) { return cols() == 1 ? block(rows() - n, 0, n, 1) : block(0, cols() - n, 1, n); }
    Diagonal<const D> diagonal() const { return Diagonal<const D>(derived()); }
    Diagonal<D> diagonal() { return Diagonal<D>(derived()); }
    NoAlias<D> noalias() { return NoAlias<D>(derived()); }

    // ---- reductions (index order)
    Scalar squaredNorm() const {
        Scalar s = 0;
        bool first = true;
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) {
                Scalar v = coeff(i, j);
                if (first) { s = v * v; first = false; } else s += v * v;
            }
        return s;
    }
    Scalar norm() const { return std::sqrt(squaredNorm()); }
    Scalar trace() const {
        Scalar s = coeff(0, 0);
        for (int i = 1; i < std::min(rows(), cols()); ++i) s += coeff(i, i);
        return s;
    }
    void normalize() {
        Scalar z = squaredNorm();
        if (z > 0) derived() /= std::sqrt(z);
    }

    // ---- assignment: coefficient-wise, column-major, no temporary
    template <class O> D& assign(const MatrixBase<O>& o) {
        derived().resizeLike(o.rows(), o.cols());
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) derived().coeffRef(i, j) = o.coeff(i, j);
        return derived();
    }
    template <class O> D& operator+=(const MatrixBase<O>& o) {
        assert(rows() == o.rows() && cols() == o.cols());
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) derived().coeffRef(i, j) += o.coeff(i, j);
        return derived();
    }
    template <class O> D& operator-=(const MatrixBase<O>& o) {
        assert(rows() == o.rows() && cols() == o.cols());
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) derived().coeffRef(i, j) -= o.coeff(i, j);
        return derived();
    }
    D& operator*=(const Scalar& s) {
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) derived().coeffRef(i, j) *= s;
        return derived();
    }
    D& operator/=(const Scalar& s) {
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) derived().coeffRef(i, j) /= s;
        return derived();
    }
    D& setZero() { return setConstant(Scalar(0)); }
    D& setOnes() { return setConstant(Scalar(1)); }
    D& setConstant(const Scalar& v) {
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) derived().coeffRef(i, j) = v;
        return derived();
    }
    D& setIdentity() {
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) derived().coeffRef(i, j) = (i == j) ? Scalar(1) : Scalar(0);
        return derived();
    }

    CommaInitializer<D> operator<<(const Scalar& s) { return CommaInitializer<D>(derived(), s); }
    template <class O> CommaInitializer<D> operator<<(const MatrixBase<O>& o) { return CommaInitializer<D>(derived(), o); }

    // rows p,q <- [c s; -s c] applied as Eigen's apply_rotation_in_the_plane (Jacobi.h): x' = c x + s y, y' = -s x + c y
    void applyOnTheLeft(int p, int q, const JacobiRotation<Scalar>& j) {
        const Scalar c = j.c(), s = j.s();
        if (c == Scalar(1) && s == Scalar(0)) return;
        for (int k = 0; k < cols(); ++k) {
            Scalar xi = coeff(p, k), yi = coeff(q, k);
            derived().coeffRef(p, k) = c * xi + s * yi;
            derived().coeffRef(q, k) = -s * xi + c * yi;
        }
    }

    ColPivHouseholderQR<PlainObject> colPivHouseholderQr() const { return ColPivHouseholderQR<PlainObject>(eval()); }
    PlainObject inverse() const;
};

template <class D>
class NoAlias {
    typename lv_nested<D>::type m_;
public:
    explicit NoAlias(D& m) : m_(m) {}
    template <class O> D& operator=(const MatrixBase<O>& o) { return m_.assign(o); }
    template <class O> D& operator+=(const MatrixBase<O>& o) { return m_ += o; }
    template <class O> D& operator-=(const MatrixBase<O>& o) { return m_ -= o; }
};

// ------------------------------------------------------------------ Matrix
template <class S, int R, int C> struct traits<Matrix<S, R, C> > {
    typedef S Scalar;
    enum { Rows = R, Cols = C };
};
template <class T> struct traits<const T> : traits<T> {};

template <class S, int R, int C>
class Matrix : public MatrixBase<Matrix<S, R, C> > {
    std::vector<S> d_;
    int r_, c_;
    typedef MatrixBase<Matrix<S, R, C> > Base;
public:
    Matrix() : d_((R == Dynamic ? 0 : R) * (C == Dynamic ? 0 : C)), r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C) {}
    explicit Matrix(int n) : r_(R), c_(C) {
        if (R == Dynamic && C == Dynamic) { r_ = n; c_ = 1; }
        else if (R == Dynamic) r_ = n;
        else if (C == Dynamic) c_ = n;
        d_.assign(r_ * c_, S(0));
    }
    // (rows, cols) for anything but a fixed 2-vector, whose two arguments are its coefficients (as in Eigen)
    template <class A, class B, class = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>::type>
    Matrix(A a, B b) : r_(R), c_(C) {
        if (R != Dynamic && C != Dynamic && R * C == 2) {
            d_.resize(2);
            d_[0] = S(a);
            d_[1] = S(b);
        } else {
            r_ = int(a);
            c_ = int(b);
            assert((R == Dynamic || R == r_) && (C == Dynamic || C == c_));
            d_.assign(r_ * c_, S(0));
        }
    }
    Matrix(const S& x, const S& y, const S& z) : d_(3), r_(R == Dynamic ? 3 : R), c_(C == Dynamic ? 1 : C) {
        assert(r_ * c_ == 3);
        d_[0] = x; d_[1] = y; d_[2] = z;
    }
    Matrix(const S& x, const S& y, const S& z, const S& w) : d_(4), r_(R == Dynamic ? 4 : R), c_(C == Dynamic ? 1 : C) {
        assert(r_ * c_ == 4);
        d_[0] = x; d_[1] = y; d_[2] = z; d_[3] = w;
    }
    Matrix(const Matrix&) = default;
    Matrix(Matrix&&) = default;
    template <class O> Matrix(const MatrixBase<O>& o) : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C) {
        d_.resize(r_ * c_);
        // evaluate into fresh storage, then adopt: a source may reference *this only in operator=
        Base::assign(o);
    }
    Matrix& operator=(const Matrix& o) { return Base::assign(o); }
    template <class O> Matrix& operator=(const MatrixBase<O>& o) { return Base::assign(o); }

    int rows() const { return r_; }
    int cols() const { return c_; }
    S coeff(int i, int j) const { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return d_[i + (size_t)j * r_]; }
    S& coeffRef(int i, int j) { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return d_[i + (size_t)j * r_]; }
    const S& coeffRef(int i, int j) const { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return d_[i + (size_t)j * r_]; }
    S* data() { return d_.data(); }
    const S* data() const { return d_.data(); }

    // Eigen: resizing assignment (no-op when the sizes agree; contents are NOT preserved otherwise)
    void resizeLike(int r, int c) {
        // a row vector assigned to a column-vector type (and vice versa) transposes implicitly, as in Eigen
        if (R != Dynamic || C != Dynamic) {
            if (C == 1 && c != 1 && r == 1) std::swap(r, c);
            if (R == 1 && r != 1 && c == 1) std::swap(r, c);
        }
        if (r == r_ && c == c_) return;
        assert((R == Dynamic || R == r) && (C == Dynamic || C == c));
        r_ = r; c_ = c;
        d_.assign((size_t)r * c, S(0));
    }
    void resize(int r, int c) { resizeLike(r, c); }
    void resize(int n) { if (C == 1 || (R == Dynamic && C == Dynamic)) resizeLike(n, C == 1 ? 1 : 1); else resizeLike(1, n); }
    // Eigen's conservativeResize is not used by the reference.

    using Base::setZero;
    using Base::setOnes;
    using Base::setIdentity;
    Matrix& setZero(int r, int c) { resize(r, c); return Base::setZero(); }
    Matrix& setZero(int n) { resize(n); return Base::setZero(); }
    Matrix& setOnes(int r, int c) { resize(r, c); return Base::setOnes(); }
    Matrix& setOnes(int n) { resize(n); return Base::setOnes(); }
    Matrix& setIdentity(int r, int c) { resize(r, c); return Base::setIdentity(); }

    static Matrix Zero() { Matrix m; m.Base::setZero(); return m; }
    static Matrix Zero(int r, int c) { Matrix m(r, c); m.Base::setZero(); return m; }
    static Matrix Zero(int n) { Matrix m(n); m.Base::setZero(); return m; }
    static Matrix Ones() { Matrix m; m.Base::setOnes(); return m; }
    static Matrix Ones(int r, int c) { Matrix m(r, c); m.Base::setOnes(); return m; }
    static Matrix Identity() { Matrix m; m.Base::setIdentity(); return m; }
    static Matrix Identity(int r, int c) { Matrix m(r, c); m.Base::setIdentity(); return m; }
};

// 1x1 fixed results convert to their scalar (Eigen: inner products)
template <class S>
class Matrix<S, 1, 1> : public MatrixBase<Matrix<S, 1, 1> > {
    S v_;
    typedef MatrixBase<Matrix<S, 1, 1> > Base;
public:
    Matrix() : v_(0) {}
    template <class O> Matrix(const MatrixBase<O>& o) : v_(0) { assert(o.rows() == 1 && o.cols() == 1); v_ = o.coeff(0, 0); }
    template <class O> Matrix& operator=(const MatrixBase<O>& o) { return Base::assign(o); }
    int rows() const { return 1; }
    int cols() const { return 1; }
    S coeff(int, int) const { return v_; }
    S& coeffRef(int, int) { return v_; }
    const S& coeffRef(int, int) const { return v_; }
    void resizeLike(int r, int c) { assert(r == 1 && c == 1); (void)r; (void)c; }
    operator S() const { return v_; }
};

// ------------------------------------------------------------------ views
template <class X> struct traits<Block<X> > {
    typedef typename traits<X>::Scalar Scalar;
    enum { Rows = Dynamic, Cols = Dynamic };
};
// a block of a column vector is still a column vector at compile time, so that x.block(..).transpose()*M*y ends 1x1
template <class S, int R> struct traits<Block<Matrix<S, R, 1> > > { typedef S Scalar; enum { Rows = Dynamic, Cols = 1 }; };
template <class S, int R> struct traits<Block<const Matrix<S, R, 1> > > { typedef S Scalar; enum { Rows = Dynamic, Cols = 1 }; };

template <class X>
class Block : public MatrixBase<Block<X> > {
    X& x_;
    int i0_, j0_, r_, c_;
    typedef MatrixBase<Block<X> > Base;
public:
    typedef typename traits<X>::Scalar Scalar;
    Block(X& x, int i, int j, int r, int c) : x_(x), i0_(i), j0_(j), r_(r), c_(c) {
        assert(i >= 0 && j >= 0 && r >= 0 && c >= 0 && i + r <= x.rows() && j + c <= x.cols());
    }
    Block(const Block&) = default;
    int rows() const { return r_; }
    int cols() const { return c_; }
    Scalar coeff(int i, int j) const { return x_.coeff(i0_ + i, j0_ + j); }
    decltype(auto) coeffRef(int i, int j) { return x_.coeffRef(i0_ + i, j0_ + j); }
    void resizeLike(int r, int c) {
        if (r == c_ && c == r_ && (r == 1 || c == 1)) return;  // vector transposition on assignment
        assert(r == r_ && c == c_);
        (void)r; (void)c;
    }
    Block& operator=(const Block& o) { return Base::assign(o); }
    template <class O> Block& operator=(const MatrixBase<O>& o) { return Base::assign(o); }
};

template <class X> struct traits<Transpose<X> > {
    typedef typename traits<X>::Scalar Scalar;
    enum { Rows = traits<X>::Cols, Cols = traits<X>::Rows };
};
template <class X>
class Transpose : public MatrixBase<Transpose<X> > {
    typename nested<X>::type x_;
public:
    typedef typename traits<X>::Scalar Scalar;
    explicit Transpose(const X& x) : x_(x) {}
    int rows() const { return x_.cols(); }
    int cols() const { return x_.rows(); }
    Scalar coeff(int i, int j) const { return x_.coeff(j, i); }
};

template <class X> struct traits<Diagonal<X> > {
    typedef typename traits<X>::Scalar Scalar;
    enum { Rows = Dynamic, Cols = 1 };
};
template <class X>
class Diagonal : public MatrixBase<Diagonal<X> > {
    X& x_;
    typedef MatrixBase<Diagonal<X> > Base;
public:
    typedef typename traits<X>::Scalar Scalar;
    explicit Diagonal(X& x) : x_(x) {}
    Diagonal(const Diagonal&) = default;
    int rows() const { return std::min(x_.rows(), x_.cols()); }
    int cols() const { return 1; }
    Scalar coeff(int i, int) const { return x_.coeff(i, i); }
    decltype(auto) coeffRef(int i, int) { return x_.coeffRef(i, i); }
    void resizeLike(int r, int c) { assert(r == rows() && c == 1); (void)r; (void)c; }
    template <class O> Diagonal& operator=(const MatrixBase<O>& o) { return Base::assign(o); }
};

// ------------------------------------------------------------------ coefficient-wise expressions
enum { OpSum = 0, OpDiff = 1, OpScaleL = 2, OpScaleR = 3, OpDivR = 4, OpNeg = 5 };
template <int A, int B> struct pick_dim { enum { value = (A != Dynamic ? A : B) }; };

template <class A, class B, int Op> struct traits<CwiseBinary<A, B, Op> > {
    typedef typename traits<A>::Scalar Scalar;
    enum { Rows = pick_dim<traits<A>::Rows, traits<B>::Rows>::value, Cols = pick_dim<traits<A>::Cols, traits<B>::Cols>::value };
};
template <class A, class B, int Op>
class CwiseBinary : public MatrixBase<CwiseBinary<A, B, Op> > {
    typename nested<A>::type a_;
    typename nested<B>::type b_;
public:
    typedef typename traits<A>::Scalar Scalar;
    CwiseBinary(const A& a, const B& b) : a_(a), b_(b) { assert(a.rows() == b.rows() && a.cols() == b.cols()); }
    int rows() const { return a_.rows(); }
    int cols() const { return a_.cols(); }
    Scalar coeff(int i, int j) const { return Op == OpSum ? a_.coeff(i, j) + b_.coeff(i, j) : a_.coeff(i, j) - b_.coeff(i, j); }
};
template <class A, int Op> struct traits<CwiseUnary<A, Op> > : traits<A> {};
template <class A, int Op>
class CwiseUnary : public MatrixBase<CwiseUnary<A, Op> > {
    typename nested<A>::type a_;
    typename traits<A>::Scalar s_;
public:
    typedef typename traits<A>::Scalar Scalar;
    CwiseUnary(const A& a, Scalar s) : a_(a), s_(s) {}
    int rows() const { return a_.rows(); }
    int cols() const { return a_.cols(); }
    Scalar coeff(int i, int j) const {
        return Op == OpScaleL ? s_ * a_.coeff(i, j) : Op == OpScaleR ? a_.coeff(i, j) * s_ : Op == OpDivR ? a_.coeff(i, j) / s_ : -a_.coeff(i, j);
    }
};

template <class A, class B> CwiseBinary<A, B, OpSum> operator+(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    return CwiseBinary<A, B, OpSum>(a.derived(), b.derived());
}
template <class A, class B> CwiseBinary<A, B, OpDiff> operator-(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    return CwiseBinary<A, B, OpDiff>(a.derived(), b.derived());
}
template <class A> CwiseUnary<A, OpNeg> operator-(const MatrixBase<A>& a) { return CwiseUnary<A, OpNeg>(a.derived(), 0); }
template <class T, class A, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
CwiseUnary<A, OpScaleL> operator*(T s, const MatrixBase<A>& a) { return CwiseUnary<A, OpScaleL>(a.derived(), typename traits<A>::Scalar(s)); }
template <class T, class A, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
CwiseUnary<A, OpScaleR> operator*(const MatrixBase<A>& a, T s) { return CwiseUnary<A, OpScaleR>(a.derived(), typename traits<A>::Scalar(s)); }
template <class T, class A, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
CwiseUnary<A, OpDivR> operator/(const MatrixBase<A>& a, T s) { return CwiseUnary<A, OpDivR>(a.derived(), typename traits<A>::Scalar(s)); }

// ------------------------------------------------------------------ products (evaluated into a temporary, C.5)
template <class A, class B>
Matrix<typename traits<A>::Scalar, traits<A>::Rows, traits<B>::Cols> operator*(const MatrixBase<A>& a_, const MatrixBase<B>& b_) {
    typedef typename traits<A>::Scalar S;
    // operands that are themselves expressions are evaluated once (Eigen nests by value through a temporary too)
    const Matrix<S, Dynamic, Dynamic> a(a_), b(b_);
    assert(a.cols() == b.rows());
    Matrix<S, traits<A>::Rows, traits<B>::Cols> r;
    r.resizeLike(a.rows(), b.cols());
    const int K = a.cols();
    for (int j = 0; j < b.cols(); ++j)
        for (int i = 0; i < a.rows(); ++i) {
            S s = 0;
            if (K > 0) {
                s = a.coeff(i, 0) * b.coeff(0, j);
                for (int k = 1; k < K; ++k) s += a.coeff(i, k) * b.coeff(k, j);
            }
            r.coeffRef(i, j) = s;
        }
    return r;
}

// ------------------------------------------------------------------ comma initializer (Eigen/src/Core/CommaInitializer.h semantics)
template <class D>
class CommaInitializer {
    typename lv_nested<D>::type m_;
    int row_, col_, blockRows_;
public:
    typedef typename traits<D>::Scalar Scalar;
    CommaInitializer(D& m, const Scalar& s) : m_(m), row_(0), col_(1), blockRows_(1) { m_.coeffRef(0, 0) = s; }
    template <class O> CommaInitializer(D& m, const MatrixBase<O>& o) : m_(m), row_(0), col_(o.cols()), blockRows_(o.rows()) { put(0, 0, o); }
    CommaInitializer& operator,(const Scalar& s) {
        if (col_ == m_.cols()) { row_ += blockRows_; col_ = 0; blockRows_ = 1; }
        assert(row_ < m_.rows() && col_ < m_.cols());
        m_.coeffRef(row_, col_++) = s;
        return *this;
    }
    template <class O> CommaInitializer& operator,(const MatrixBase<O>& o) {
        if (o.rows() == 0 || o.cols() == 0) return *this;
        if (col_ == m_.cols()) { row_ += blockRows_; col_ = 0; blockRows_ = o.rows(); }
        put(row_, col_, o);
        col_ += o.cols();
        return *this;
    }
private:
    template <class O> void put(int r0, int c0, const MatrixBase<O>& o) {
        assert(r0 + o.rows() <= m_.rows() && c0 + o.cols() <= m_.cols());
        for (int j = 0; j < o.cols(); ++j)
            for (int i = 0; i < o.rows(); ++i) m_.coeffRef(r0 + i, c0 + j) = o.coeff(i, j);
    }
};

// ------------------------------------------------------------------ Jacobi / Givens (Eigen/src/Jacobi/Jacobi.h, real case; C.1)
template <class S>
class JacobiRotation {
    S c_, s_;
public:
    JacobiRotation() : c_(1), s_(0) {}
    JacobiRotation(const S& c, const S& s) : c_(c), s_(s) {}
    S c() const { return c_; }
    S s() const { return s_; }
    JacobiRotation transpose() const { return JacobiRotation(c_, -s_); }
    JacobiRotation adjoint() const { return JacobiRotation(c_, -s_); }
    void makeGivens(const S& p, const S& q, S* r = 0) {
        using std::abs;
        using std::sqrt;
        if (q == S(0)) {
            c_ = p < S(0) ? S(-1) : S(1);
            s_ = S(0);
            if (r) *r = abs(p);
        } else if (p == S(0)) {
            c_ = S(0);
            s_ = q < S(0) ? S(1) : S(-1);
            if (r) *r = abs(q);
        } else if (abs(p) > abs(q)) {
            S t = q / p;
            S u = sqrt(S(1) + t * t);
            if (p < S(0)) u = -u;
            c_ = S(1) / u;
            s_ = -t * c_;
            if (r) *r = p * u;
        } else {
            S t = p / q;
            S u = sqrt(S(1) + t * t);
            if (q < S(0)) u = -u;
            s_ = -S(1) / u;
            c_ = -t * s_;
            if (r) *r = q * u;
        }
    }
};

// ------------------------------------------------------------------ ColPivHouseholderQR (Eigen 3.3 QR/ColPivHouseholderQR.h; C.2)
template <class M>
class ColPivHouseholderQR {
    typedef typename traits<M>::Scalar S;
    Matrix<S, Dynamic, Dynamic> qr_;
    std::vector<S> h_;
    std::vector<int> perm_;  // column now at position k came from perm_[k]
    int nonzero_pivots_;
    S maxpivot_;
public:
    explicit ColPivHouseholderQR(const M& m) : qr_(m) { compute(); }
    int nonzeroPivots() const { return nonzero_pivots_; }

    template <class B>
    Matrix<S, traits<M>::Cols, traits<B>::Cols> solve(const MatrixBase<B>& b_) const {
        const int rows = qr_.rows(), cols = qr_.cols(), nz = nonzero_pivots_;
        Matrix<S, Dynamic, Dynamic> c(b_);
        assert(c.rows() == rows);
        Matrix<S, traits<M>::Cols, traits<B>::Cols> x;
        x.resizeLike(cols, c.cols());
        x.setZero();
        if (nz == 0) return x;
        // c <- H_{nz-1} ... H_0 c   (Q^T = (H_0 H_1 ...)^T, applied in order 0..nz-1)
        for (int k = 0; k < nz; ++k) apply_reflector(c, k, 0, c.cols());
        // back substitution on the leading nz x nz triangle
        for (int j = 0; j < c.cols(); ++j)
            for (int i = nz - 1; i >= 0; --i) {
                S s = c.coeff(i, j);
                for (int k = i + 1; k < nz; ++k) s -= qr_.coeff(i, k) * c.coeff(k, j);
                c.coeffRef(i, j) = s / qr_.coeff(i, i);
            }
        for (int i = 0; i < nz; ++i)
            for (int j = 0; j < c.cols(); ++j) x.coeffRef(perm_[i], j) = c.coeff(i, j);
        return x;
    }

private:
    // rows k..rows-1 of columns [c0,c1) of a:  a -= tau v (v^T a),  v = [1; essential stored under the diagonal of column k]
    void apply_reflector(Matrix<S, Dynamic, Dynamic>& a, int k, int c0, int c1) const {
        const int rows = qr_.rows();
        const S tau = h_[k];
        if (rows - k == 1) {
            for (int j = c0; j < c1; ++j) a.coeffRef(k, j) *= S(1) - tau;
            return;
        }
        if (tau == S(0)) return;
        for (int j = c0; j < c1; ++j) {
            S tmp = 0;
            for (int i = k + 1; i < rows; ++i) {
                S t = qr_.coeff(i, k) * a.coeff(i, j);
                if (i == k + 1) tmp = t; else tmp += t;
            }
            tmp += a.coeff(k, j);
            a.coeffRef(k, j) -= tau * tmp;
            for (int i = k + 1; i < rows; ++i) a.coeffRef(i, j) -= tau * qr_.coeff(i, k) * tmp;
        }
    }
    S tail_norm(int col, int from) const {
        S s = 0;
        for (int i = from; i < qr_.rows(); ++i) s += qr_.coeff(i, col) * qr_.coeff(i, col);
        return std::sqrt(s);
    }
    void compute() {
        using std::abs;
        using std::sqrt;
        const int rows = qr_.rows(), cols = qr_.cols(), size = std::min(rows, cols);
        h_.assign(size, S(0));
        perm_.resize(cols);
        std::vector<S> upd(cols), dir(cols);
        S maxnorm = 0;
        for (int k = 0; k < cols; ++k) {
            perm_[k] = k;
            dir[k] = upd[k] = tail_norm(k, 0);
            maxnorm = std::max(maxnorm, upd[k]);
        }
        const S eps = std::numeric_limits<S>::epsilon();
        S th = maxnorm * eps / S(rows);
        const S threshold_helper = th * th;
        const S norm_downdate_threshold = sqrt(eps);
        nonzero_pivots_ = size;
        maxpivot_ = S(0);
        for (int k = 0; k < size; ++k) {
            int big = k;
            for (int j = k + 1; j < cols; ++j)
                if (upd[j] > upd[big]) big = j;
            S biggest_sq = upd[big] * upd[big];
            if (nonzero_pivots_ == size && biggest_sq < threshold_helper * S(rows - k)) nonzero_pivots_ = k;
            if (big != k) {
                for (int i = 0; i < rows; ++i) std::swap(qr_.coeffRef(i, k), qr_.coeffRef(i, big));
                std::swap(upd[k], upd[big]);
                std::swap(dir[k], dir[big]);
                std::swap(perm_[k], perm_[big]);
            }
            // makeHouseholderInPlace on rows k.. of column k
            S tailSq = 0;
            for (int i = k + 1; i < rows; ++i) tailSq += qr_.coeff(i, k) * qr_.coeff(i, k);
            S c0 = qr_.coeff(k, k), beta, tau;
            if (tailSq <= std::numeric_limits<S>::min()) {
                tau = S(0);
                beta = c0;
                for (int i = k + 1; i < rows; ++i) qr_.coeffRef(i, k) = S(0);
            } else {
                beta = sqrt(c0 * c0 + tailSq);
                if (c0 >= S(0)) beta = -beta;
                for (int i = k + 1; i < rows; ++i) qr_.coeffRef(i, k) /= (c0 - beta);
                tau = (beta - c0) / beta;
            }
            h_[k] = tau;
            qr_.coeffRef(k, k) = beta;
            if (abs(beta) > maxpivot_) maxpivot_ = abs(beta);
            apply_reflector(qr_, k, k + 1, cols);
            for (int j = k + 1; j < cols; ++j) {
                if (upd[j] != S(0)) {
                    S temp = abs(qr_.coeff(k, j)) / upd[j];
                    temp = (S(1) + temp) * (S(1) - temp);
                    temp = temp < S(0) ? S(0) : temp;
                    S r = upd[j] / dir[j];
                    S temp2 = temp * r * r;
                    if (temp2 <= norm_downdate_threshold) {
                        dir[j] = tail_norm(j, k + 1);
                        upd[j] = dir[j];
                    } else {
                        upd[j] *= sqrt(temp);
                    }
                }
            }
        }
    }
};

// ------------------------------------------------------------------ inverse: PartialPivLU (Eigen/src/LU; C.3), solve against I
template <class D>
typename MatrixBase<D>::PlainObject MatrixBase<D>::inverse() const {
    typedef Scalar S;
    const int n = rows();
    assert(n == cols());
    Matrix<S, Dynamic, Dynamic> lu(derived());
    std::vector<int> piv(n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        S best = std::abs(lu.coeff(k, k));
        for (int i = k + 1; i < n; ++i)
            if (std::abs(lu.coeff(i, k)) > best) { best = std::abs(lu.coeff(i, k)); p = i; }
        piv[k] = p;
        if (best != S(0)) {
            if (p != k)
                for (int j = 0; j < n; ++j) std::swap(lu.coeffRef(k, j), lu.coeffRef(p, j));
            for (int i = k + 1; i < n; ++i) lu.coeffRef(i, k) /= lu.coeff(k, k);
        }
        for (int j = k + 1; j < n; ++j) {
            const S u = lu.coeff(k, j);
            for (int i = k + 1; i < n; ++i) lu.coeffRef(i, j) -= lu.coeff(i, k) * u;
        }
    }
    // inv = U^-1 L^-1 P
    Matrix<S, Dynamic, Dynamic> x = Matrix<S, Dynamic, Dynamic>::Identity(n, n);
    for (int k = 0; k < n; ++k)
        if (piv[k] != k)
            for (int j = 0; j < n; ++j) std::swap(x.coeffRef(k, j), x.coeffRef(piv[k], j));
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) {  // unit lower
            S s = x.coeff(i, j);
            for (int k = 0; k < i; ++k) s -= lu.coeff(i, k) * x.coeff(k, j);
            x.coeffRef(i, j) = s;
        }
        for (int i = n - 1; i >= 0; --i) {  // upper
            S s = x.coeff(i, j);
            for (int k = i + 1; k < n; ++k) s -= lu.coeff(i, k) * x.coeff(k, j);
            x.coeffRef(i, j) = s / lu.coeff(i, i);
        }
    }
    PlainObject out;
    out.resizeLike(n, n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) out.coeffRef(i, j) = x.coeff(i, j);
    return out;
}

// ------------------------------------------------------------------ typedefs the reference spells
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<int, Dynamic, Dynamic> MatrixXi;
typedef Matrix<float, Dynamic, Dynamic> MatrixXf;

}  // namespace Eigen
#endif
