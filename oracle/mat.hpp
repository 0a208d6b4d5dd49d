// oracle/mat.hpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Minimal dependency-free dense linear algebra used by the CPU restatement of
// the R-VIO hot path.  Stands in for the slice of Eigen the reference uses
// (SURVEY.md appendix C): column-major double matrices, plain k-ascending
// products, Eigen's real JacobiRotation::makeGivens, column-pivoted
// Householder QR solve (colPivHouseholderQr().solve, Updater.cc:239,420) and
// partial-pivot LU inverse (MatrixXd::inverse(), Updater.cc:543).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>
#include <algorithm>
#include <limits>

namespace orc {

struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    inline double& operator()(int i, int j) { return a[(size_t)j * r + i]; }
    inline double operator()(int i, int j) const { return a[(size_t)j * r + i]; }
    void zero() { std::fill(a.begin(), a.end(), 0.0); }
    static Mat identity(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1; return m; }
    Mat block(int i0, int j0, int nr, int nc) const {
        Mat b(nr, nc);
        for (int j = 0; j < nc; ++j) for (int i = 0; i < nr; ++i) b(i, j) = (*this)(i0 + i, j0 + j);
        return b;
    }
    void set_block(int i0, int j0, const Mat& b) {
        for (int j = 0; j < b.c; ++j) for (int i = 0; i < b.r; ++i) (*this)(i0 + i, j0 + j) = b(i, j);
    }
    Mat t() const { Mat m(c, r); for (int j = 0; j < c; ++j) for (int i = 0; i < r; ++i) m(j, i) = (*this)(i, j); return m; }
};

inline Mat mul(const Mat& A, const Mat& B) {
    Mat C(A.r, B.c);
    for (int j = 0; j < B.c; ++j)
        for (int k = 0; k < A.c; ++k) {
            double b = B(k, j);
            if (b == 0.0) continue;  // exact: adding 0*x never changes a finite sum
            const double* ac = &A.a[(size_t)k * A.r];
            double* cc = &C.a[(size_t)j * C.r];
            for (int i = 0; i < A.r; ++i) cc[i] += ac[i] * b;
        }
    return C;
}
// A * B^T
inline Mat mul_nt(const Mat& A, const Mat& B) {
    Mat C(A.r, B.r);
    for (int k = 0; k < A.c; ++k)
        for (int j = 0; j < B.r; ++j) {
            double b = B(j, k);
            if (b == 0.0) continue;
            const double* ac = &A.a[(size_t)k * A.r];
            double* cc = &C.a[(size_t)j * C.r];
            for (int i = 0; i < A.r; ++i) cc[i] += ac[i] * b;
        }
    return C;
}
inline Mat add(const Mat& A, const Mat& B) { Mat C = A; for (size_t i = 0; i < C.a.size(); ++i) C.a[i] += B.a[i]; return C; }
inline Mat sub(const Mat& A, const Mat& B) { Mat C = A; for (size_t i = 0; i < C.a.size(); ++i) C.a[i] -= B.a[i]; return C; }
inline Mat scale(const Mat& A, double s) { Mat C = A; for (auto& v : C.a) v *= s; return C; }

// A = .5*(A+A^T).  The reference's in-place Eigen expression aliases
//  and leaves an O(asymmetry) non-symmetric result;
// the oracle symmetrises exactly — a deliberate, tolerance-safe deviation.
inline void symmetrize(Mat& A) {
    for (int j = 0; j < A.c; ++j)
        for (int i = j + 1; i < A.r; ++i) {
            double v = .5 * (A(i, j) + A(j, i));
            A(i, j) = v; A(j, i) = v;
        }
}

// ---- fixed 3-vector / 3x3 helpers (row-major storage, m[i][j]) ----
struct V3 { double v[3]; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
struct M3 { double m[3][3]; };
struct Q4 { double v[4]; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };

inline V3 v3(double a, double b, double c) { V3 r; r.v[0] = a; r.v[1] = b; r.v[2] = c; return r; }
inline V3 operator+(const V3& a, const V3& b) { return v3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline V3 operator-(const V3& a, const V3& b) { return v3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline V3 operator*(double s, const V3& a) { return v3(s * a[0], s * a[1], s * a[2]); }
inline double norm(const V3& a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
inline V3 normalized(const V3& a) { double n = norm(a); return v3(a[0] / n, a[1] / n, a[2] / n); }
inline M3 m3_zero() { M3 r; std::memset(&r, 0, sizeof r); return r; }
inline M3 m3_eye() { M3 r = m3_zero(); r.m[0][0] = r.m[1][1] = r.m[2][2] = 1; return r; }
inline M3 operator*(const M3& A, const M3& B) {
    M3 C;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}
inline V3 operator*(const M3& A, const V3& x) {
    return v3(A.m[0][0] * x[0] + A.m[0][1] * x[1] + A.m[0][2] * x[2],
              A.m[1][0] * x[0] + A.m[1][1] * x[1] + A.m[1][2] * x[2],
              A.m[2][0] * x[0] + A.m[2][1] * x[1] + A.m[2][2] * x[2]);
}
inline M3 operator+(const M3& A, const M3& B) { M3 C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][j] + B.m[i][j]; return C; }
inline M3 operator-(const M3& A, const M3& B) { M3 C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][j] - B.m[i][j]; return C; }
inline M3 operator*(double s, const M3& A) { M3 C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[i][j] = s * A.m[i][j]; return C; }
inline M3 transpose(const M3& A) { M3 C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[j][i]; return C; }
inline double trace(const M3& A) { return A.m[0][0] + A.m[1][1] + A.m[2][2]; }

inline void put(Mat& M, int i0, int j0, const M3& B) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M(i0 + i, j0 + j) = B.m[i][j]; }

// Eigen JacobiRotation<double>::makeGivens(p,q).
struct Givens { double c, s; };
inline Givens make_givens(double p, double q) {
    Givens g;
    if (q == 0.0) { g.c = p < 0 ? -1.0 : 1.0; g.s = 0.0; }
    else if (p == 0.0) { g.c = 0.0; g.s = q < 0 ? 1.0 : -1.0; }
    else if (std::fabs(p) > std::fabs(q)) {
        double t = q / p; double u = std::sqrt(1.0 + t * t); if (p < 0) u = -u;
        g.c = 1.0 / u; g.s = -t * g.c;
    } else {
        double t = p / q; double u = std::sqrt(1.0 + t * t); if (q < 0) u = -u;
        g.s = -1.0 / u; g.c = -t * g.s;
    }
    return g;
}
// rows (x,y) -> (c x - s y, s x + c y): block.applyOnTheLeft(0,1,G.adjoint())
inline void apply_givens_rows(Mat& A, int rx, int ry, int c0, int nc, Givens g) {
    for (int j = c0; j < c0 + nc; ++j) {
        double x = A(rx, j), y = A(ry, j);
        A(rx, j) = g.c * x - g.s * y;
        A(ry, j) = g.s * x + g.c * y;
    }
}

// x = A.colPivHouseholderQr().solve(b)   (square A; appendix C.2)
inline std::vector<double> colpiv_qr_solve(Mat A, std::vector<double> b) {
    const int n = A.r;
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    std::vector<double> cn(n);
    double maxn = 0;
    for (int j = 0; j < n; ++j) { double s = 0; for (int i = 0; i < n; ++i) s += A(i, j) * A(i, j); cn[j] = s; maxn = std::max(maxn, s); }
    const double eps = std::numeric_limits<double>::epsilon();
    const double thr = maxn * eps * eps / (double)n;
    int rank = n;
    for (int k = 0; k < n; ++k) {
        // pivot: column of largest remaining norm (recomputed exactly each step)
        int piv = k; double best = -1;
        for (int j = k; j < n; ++j) { double s = 0; for (int i = k; i < n; ++i) s += A(i, j) * A(i, j); cn[j] = s; if (s > best) { best = s; piv = j; } }
        if (rank == n && best < thr * (double)(n - k)) rank = k;
        if (piv != k) { for (int i = 0; i < n; ++i) std::swap(A(i, k), A(i, piv)); std::swap(perm[k], perm[piv]); }
        // Householder on column k
        double tail = 0; for (int i = k + 1; i < n; ++i) tail += A(i, k) * A(i, k);
        double c0 = A(k, k);
        if (tail <= std::numeric_limits<double>::min()) continue;  // tau = 0, H = I
        double beta = std::sqrt(c0 * c0 + tail); if (c0 >= 0) beta = -beta;
        double tau = (beta - c0) / beta;
        double inv = 1.0 / (c0 - beta);
        for (int i = k + 1; i < n; ++i) A(i, k) *= inv;  // essential part v (v_k = 1)
        A(k, k) = beta;
        for (int j = k + 1; j < n; ++j) {
            double w = A(k, j); for (int i = k + 1; i < n; ++i) w += A(i, k) * A(i, j);
            w *= tau;
            A(k, j) -= w; for (int i = k + 1; i < n; ++i) A(i, j) -= w * A(i, k);
        }
        double w = b[k]; for (int i = k + 1; i < n; ++i) w += A(i, k) * b[i];
        w *= tau;
        b[k] -= w; for (int i = k + 1; i < n; ++i) b[i] -= w * A(i, k);
    }
    // back substitution on the leading rank x rank triangle
    std::vector<double> y(n, 0.0);
    for (int i = rank - 1; i >= 0; --i) {
        double s = b[i]; for (int j = i + 1; j < rank; ++j) s -= A(i, j) * y[j];
        y[i] = s / A(i, i);
    }
    std::vector<double> x(n, 0.0);
    for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
    return x;
}

// MatrixXd::inverse(): PartialPivLU (appendix C.3)
inline Mat lu_inverse(Mat A) {
    const int n = A.r;
    std::vector<int> p(n);
    for (int i = 0; i < n; ++i) p[i] = i;
    for (int k = 0; k < n; ++k) {
        int piv = k; double best = std::fabs(A(k, k));
        for (int i = k + 1; i < n; ++i) if (std::fabs(A(i, k)) > best) { best = std::fabs(A(i, k)); piv = i; }
        if (piv != k) { for (int j = 0; j < n; ++j) std::swap(A(k, j), A(piv, j)); std::swap(p[k], p[piv]); }
        double d = A(k, k);
        for (int i = k + 1; i < n; ++i) A(i, k) /= d;
        for (int j = k + 1; j < n; ++j) { double u = A(k, j); if (u == 0) continue; for (int i = k + 1; i < n; ++i) A(i, j) -= A(i, k) * u; }
    }
    Mat X(n, n);
    for (int c = 0; c < n; ++c) {
        std::vector<double> y(n);
        for (int i = 0; i < n; ++i) y[i] = (p[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < n; ++i) { double s = y[i]; for (int j = 0; j < i; ++j) s -= A(i, j) * y[j]; y[i] = s; }
        for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int j = i + 1; j < n; ++j) s -= A(i, j) * y[j]; y[i] = s / A(i, i); }
        for (int i = 0; i < n; ++i) X(i, c) = y[i];
    }
    return X;
}

}  // namespace orc
