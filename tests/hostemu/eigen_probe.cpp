// Probes of oracle/refshim/mini_eigen.hpp — the Eigen surface the reference's sources are compiled against in oracle/_ref — called from
// tests/test_mini_eigen.py and held there against NumPy / LAPACK and against the behaviour Eigen 3.3 documents (Jacobi.h, the aliasing
// page, ColPivHouseholderQR, PartialPivLU).  TEST INFRASTRUCTURE; written against the Eigen API exactly as the reference spells it.
//   g++ -O1 -std=c++17 -shared -fPIC -Ioracle/refshim tests/hostemu/eigen_probe.cpp -o tests/hostemu/libeigen_probe.so
#include <Eigen/Dense>

using Eigen::MatrixXd;
using Eigen::VectorXd;

static MatrixXd load(const double* a, int r, int c) {   // column-major, like Eigen's default storage
    MatrixXd m(r, c);
    for (int j = 0; j < c; ++j)
        for (int i = 0; i < r; ++i) m(i, j) = a[i + (size_t)j * r];
    return m;
}
static void store(const MatrixXd& m, double* o) {
    for (int j = 0; j < m.cols(); ++j)
        for (int i = 0; i < m.rows(); ++i) o[i + (size_t)j * m.rows()] = m(i, j);
}

extern "C" {

// Updater.cc:543 `S.inverse()`
void probe_inverse(const double* a, int n, double* o) { MatrixXd A = load(a, n, n); MatrixXd X = A.inverse(); store(X, o); }

// Updater.cc:239,420 `A.colPivHouseholderQr().solve(b)`
void probe_qr_solve(const double* a, int r, int c, const double* b, int nb, double* o) {
    MatrixXd A = load(a, r, c), B = load(b, r, nb);
    MatrixXd X = A.colPivHouseholderQr().solve(B);
    store(X, o);
}
void probe_qr_solve3(const double* a, const double* b, double* o) {   // the fixed-size form of Updater.cc:239
    Eigen::Matrix3d A;
    Eigen::Vector3d B;
    for (int j = 0; j < 3; ++j) { B(j) = b[j]; for (int i = 0; i < 3; ++i) A(i, j) = a[i + 3 * j]; }
    Eigen::Vector3d x = A.colPivHouseholderQr().solve(B);
    for (int i = 0; i < 3; ++i) o[i] = x(i);
}

// Updater.cc:388-400 / 501-510: makeGivens on two entries, the adjoint applied to the two rows
void probe_givens(double p, double q, const double* rows2xk, int k, double* out_rows, double* csr) {
    Eigen::JacobiRotation<double> G;
    double r = 0;
    G.makeGivens(p, q, &r);
    MatrixXd M = load(rows2xk, 2, k);
    (M.block(0, 0, 2, k)).applyOnTheLeft(0, 1, G.adjoint());
    store(M, out_rows);
    csr[0] = G.c(); csr[1] = G.s(); csr[2] = r;
}

// System.cc:297,321,358 / PreIntegrator.cc:193 / Updater.cc:541,619: `P = .5*(P+P.transpose())` evaluated in place
void probe_symmetrise(double* a, int n) { MatrixXd A = load(a, n, n); A = .5 * (A + A.transpose()); store(A, a); }
// ... and the form with a temporary (`.eval()`), which IS symmetric: the difference between the two is the aliasing
void probe_symmetrise_eval(double* a, int n) { MatrixXd A = load(a, n, n); A = (.5 * (A + A.transpose())).eval(); store(A, a); }

// products with transposes, and a product assigned into a block of its own operand's matrix (evaluated through a temporary)
void probe_products(const double* a, int m, int k, const double* b, int n, double* ab, double* abt_in, double* atb_in) {
    MatrixXd A = load(a, m, k), B = load(b, k, n);
    MatrixXd AB = A * B;
    store(AB, ab);
    MatrixXd C = load(abt_in, m, m);          // C (m x m) <- A * A^T + C
    C = A * A.transpose() + C;
    store(C, abt_in);
    MatrixXd D = load(atb_in, k, k);          // D.block <- (A^T A) scaled, written into D itself
    D.block(0, 0, k, k) = 2. * (A.transpose() * A);
    store(D, atb_in);
}

// the comma initialiser fills row by row; head/tail/segment-like blocks; squaredNorm / norm / normalize; Identity / Zero; diagonal()
void probe_misc(double* o) {
    Eigen::Matrix3d M;
    M << 1, 2, 3,
         4, 5, 6,
         7, 8, 10;
    Eigen::Vector3d v(3, 4, 12);
    int t = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[t++] = M(i, j);       // row-major read-out
    o[t++] = v.squaredNorm(); o[t++] = v.norm();
    v.normalize();
    for (int i = 0; i < 3; ++i) o[t++] = v(i);
    VectorXd x(6);
    x << 1, 2, 3, 4, 5, 6;
    o[t++] = x.head(2)(1); o[t++] = x.tail(2)(0);
    MatrixXd I = MatrixXd::Identity(3, 3), Z = MatrixXd::Zero(2, 2);
    o[t++] = I(1, 1) + I(0, 1) + Z(1, 1);
    o[t++] = M.diagonal()(2); o[t++] = M.trace();
    MatrixXd T = M.transpose();
    o[t++] = T(0, 2);                                                               // = M(2, 0) = 7
    M.block(0, 0, 2, 2) = M.block(1, 1, 2, 2).eval();                               // overlapping blocks through a temporary
    o[t++] = M(0, 0); o[t++] = M(1, 1);
}
}
