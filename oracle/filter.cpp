// oracle/filter.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
// CPU restatement of the filter half of the R-VIO hot path:
//   N1  util/Numerics.h:30-167          quaternion helpers
//   --  System.cc:115-170               initialize
//   P1  PreIntegrator.cc:51-194         propagate
//   U1..U10 Updater.cc:72-628           update
//   S1/S2 System.cc:279-365             augmentation, window slide, composition
// Pinned against the reference's own sources through oracle/_ref (see rvio_oracle.h, tests/test_ref_pins.py).
#include "rvio_oracle.h"
#include "mat.hpp"
#include <cstdio>
#include <cstdlib>

using namespace orc;

namespace {

const double kChi2[500] = {
#include "chi2_table.inc"
};

// ---------------------------------------------------------------- N1
// QuatMul, Numerics.h:30-63 (JPL; normalises and forces w>=0)
Q4 quat_mul(const Q4& q1, const Q4& q2) {
    double m[4][4] = {{q1[3], q1[2], -q1[1], q1[0]},
                      {-q1[2], q1[3], q1[0], q1[1]},
                      {q1[1], -q1[0], q1[3], q1[2]},
                      {-q1[0], -q1[1], -q1[2], q1[3]}};
    Q4 q;
    for (int i = 0; i < 4; ++i) q[i] = m[i][0] * q2[0] + m[i][1] * q2[1] + m[i][2] * q2[2] + m[i][3] * q2[3];
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
    if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
    return q;
}
// SkewSymm, Numerics.h:97-105
M3 skew(const V3& w) {
    M3 S = m3_zero();
    S.m[0][1] = -w[2]; S.m[0][2] = w[1];
    S.m[1][0] = w[2];  S.m[1][2] = -w[0];
    S.m[2][0] = -w[1]; S.m[2][1] = w[0];
    return S;
}
// QuatToRot, Numerics.h:111-120:  I - 2 w [q]x + 2 [q]x^2
M3 quat_to_rot(const Q4& q) {
    M3 qx = skew(v3(q[0], q[1], q[2]));
    return (m3_eye() - (2 * q[3]) * qx) + 2 * (qx * qx);
}
// RotToQuat, Numerics.h:126-167 (Breckenridge 4-branch)
Q4 rot_to_quat(const M3& R) {
    Q4 q;
    double T = trace(R);
    const double r00 = R.m[0][0], r11 = R.m[1][1], r22 = R.m[2][2];
    if (r00 > T && r00 > r11 && r00 > r22) {
        q[0] = std::sqrt((1 + 2 * r00 - T) / 4);
        double k = 1 / (4 * q[0]);
        q[1] = k * (R.m[0][1] + R.m[1][0]); q[2] = k * (R.m[0][2] + R.m[2][0]); q[3] = k * (R.m[1][2] - R.m[2][1]);
    } else if (r11 > T && r11 > r00 && r11 > r22) {
        q[1] = std::sqrt((1 + 2 * r11 - T) / 4);
        double k = 1 / (4 * q[1]);
        q[0] = k * (R.m[0][1] + R.m[1][0]); q[2] = k * (R.m[1][2] + R.m[2][1]); q[3] = k * (R.m[2][0] - R.m[0][2]);
    } else if (r22 > T && r22 > r00 && r22 > r11) {
        q[2] = std::sqrt((1 + 2 * r22 - T) / 4);
        double k = 1 / (4 * q[2]);
        q[0] = k * (R.m[0][2] + R.m[2][0]); q[1] = k * (R.m[1][2] + R.m[2][1]); q[3] = k * (R.m[0][1] - R.m[1][0]);
    } else {
        q[3] = std::sqrt((1 + T) / 4);
        double k = 1 / (4 * q[3]);
        q[0] = k * (R.m[1][2] - R.m[2][1]); q[1] = k * (R.m[2][0] - R.m[0][2]); q[2] = k * (R.m[0][1] - R.m[1][0]);
    }
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
    if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
    return q;
}

Q4 q4_at(const double* x) { Q4 q; for (int i = 0; i < 4; ++i) q[i] = x[i]; return q; }
V3 v3_at(const double* x) { return v3(x[0], x[1], x[2]); }

struct Extr { M3 Ric, Rci; V3 tic, tci; };
// Updater ctor, Updater.cc:46-53
Extr extrinsics(const rvio_config* cfg) {
    Extr e;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) e.Ric.m[i][j] = cfg->T_bc[4 * i + j]; e.tic[i] = cfg->T_bc[4 * i + 3]; }
    e.Rci = transpose(e.Ric);
    e.tci = -1.0 * (e.Rci * e.tic);
    return e;
}
// Updater.cc:42-44: float max, then widened to double
double sigma_im(const rvio_config* cfg) { float s = std::max(cfg->sigma_px, cfg->sigma_py); return (double)s; }

// dq from a half-angle error vector (Updater.cc:549-563 and the two copies below it)
Q4 small_quat(double ex, double ey, double ez) {
    Q4 dq; dq[0] = .5 * ex; dq[1] = .5 * ey; dq[2] = .5 * ez;
    double n = std::sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]);
    if (n < 1) dq[3] = std::sqrt(1 - n * n);
    else { double k = 1 / std::sqrt(1 + n * n); dq[0] *= k; dq[1] *= k; dq[2] *= k; dq[3] = k; }
    return dq;
}

// ------------------------------------------------------------ per-feature U1..U5
struct FeatOut {
    bool accepted = false;
    int ndof = 0;
    double gamma = 0, phi = 0, psi = 0, rho = 0;
    Mat Hx_;                 // ndof x 6n   (nullspace-projected)
    std::vector<double> r_;  // ndof
};

// Updater.cc:109-455 for one feature.  Pcc = Pk1k.block(24,24,6n,6n).
// probe (tests only): evaluate the measurement model at a GIVEN inverse-depth triple instead of the LM result, and hand out the
// residual / Jacobians as they are before the nullspace projection (finite-difference pin of U3, tests/test_oracle_pins.py)
struct FeatProbe { const double* pf = nullptr; Mat* Hx_raw = nullptr; Mat* Hf_raw = nullptr; std::vector<double>* r_raw = nullptr; };

FeatOut feature_rows(const rvio_config* cfg, const Extr& ex, double sig, const double* x, int n_clones,
                     const Mat& Pcc, unsigned char type, const float* meas, int L, const FeatProbe* probe = nullptr) {
    FeatOut out;
    const int nc6 = 6 * n_clones;
    int nTrackLength = L, nTrackPhases = L - 1;
    const double* rel = (type == '1') ? (x + 26 + 7 * n_clones - 7 * nTrackPhases) : (x + 26);  // :118-122

    // [qIi1,tIi1] :125-132
    std::vector<Q4> qI(nTrackPhases); std::vector<V3> tI(nTrackPhases);
    qI[0] = q4_at(rel);
    tI[0] = -1.0 * (quat_to_rot(qI[0]) * v3_at(rel + 4));
    for (int i = 1; i < nTrackPhases; ++i) {
        Q4 qi = q4_at(rel + 7 * i);
        qI[i] = quat_mul(qi, qI[i - 1]);
        tI[i] = quat_to_rot(qi) * (tI[i - 1] - v3_at(rel + 7 * i + 4));
    }
    // [qCi1,tCi1] :135-141
    std::vector<Q4> qC(nTrackPhases); std::vector<V3> tC(nTrackPhases);
    for (int i = 0; i < nTrackPhases; ++i) {
        M3 RI = quat_to_rot(qI[i]);
        qC[i] = rot_to_quat((ex.Rci * RI) * ex.Ric);
        tC[i] = ((ex.Rci * RI) * ex.tic + ex.Rci * tI[i]) + ex.tci;
    }

    // inverse-depth init :146-158
    const float fx0 = meas[0], fy0 = meas[1];
    double phi = std::atan2((double)fy0, std::sqrt(std::pow((double)fx0, 2) + 1));
    double psi = std::atan2((double)fx0, 1.0);
    double rho = 0.;
    out.phi = phi; out.psi = psi; out.rho = rho;
    if (std::fabs(phi) > .5 * 3.14 || std::fabs(psi) > .5 * 3.14) return out;

    V3 ep = v3(std::cos(phi) * std::sin(psi), std::sin(phi), std::cos(phi) * std::cos(psi));
    double Jang[3][2] = {{-std::sin(phi) * std::sin(psi), std::cos(phi) * std::cos(psi)},
                         {std::cos(phi), 0},
                         {-std::sin(phi) * std::cos(psi), -std::cos(phi) * std::sin(psi)}};
    const double ri = 1. / std::pow(sig, 2);  // Rinv diagonal :172-174

    // LM :176-263
    const int maxIter = 10;
    double lambda = 0.01;
    double lastCost = std::numeric_limits<double>::infinity();
    if (probe && probe->pf) {
        phi = probe->pf[0]; psi = probe->pf[1]; rho = probe->pf[2];
        ep = v3(std::cos(phi) * std::sin(psi), std::sin(phi), std::cos(phi) * std::cos(psi));
        Jang[0][0] = -std::sin(phi) * std::sin(psi); Jang[0][1] = std::cos(phi) * std::cos(psi);
        Jang[1][0] = std::cos(phi);                  Jang[1][1] = 0;
        Jang[2][0] = -std::sin(phi) * std::cos(psi); Jang[2][1] = -std::cos(phi) * std::sin(psi);
    }
    for (int it = 0; it < ((probe && probe->pf) ? 0 : maxIter); ++it) {
        double HTH[3][3] = {{0}}; double HTe[3] = {0}; double cost = 0;
        auto accumulate = [&](const double H[2][3], float exf, float eyf) {
            double e[2] = {(double)exf, (double)eyf};
            cost += (e[0] * ri) * e[0] + (e[1] * ri) * e[1];
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) HTH[a][b] += (H[0][a] * ri) * H[0][b] + (H[1][a] * ri) * H[1][b];
                HTe[a] += (H[0][a] * ri) * e[0] + (H[1][a] * ri) * e[1];
            }
        };
        {   // first measurement :186-206
            V3 h = ep;
            double Hp[2][3] = {{1 / h[2], 0, -h[0] / std::pow(h[2], 2)}, {0, 1 / h[2], -h[1] / std::pow(h[2], 2)}};
            double H[2][3];
            for (int a = 0; a < 2; ++a) { for (int b = 0; b < 2; ++b) H[a][b] = Hp[a][0] * Jang[0][b] + Hp[a][1] * Jang[1][b] + Hp[a][2] * Jang[2][b]; H[a][2] = 0; }
            float px = (float)(h[0] / h[2]), py = (float)(h[1] / h[2]);  // cv::Point2f :194-196
            accumulate(H, fx0 - px, fy0 - py);
        }
        for (int i = 0; i < nTrackPhases; ++i) {  // :208-233
            M3 Rc = quat_to_rot(qC[i]);
            V3 h = Rc * ep + rho * tC[i];
            double Hp[2][3] = {{1 / h[2], 0, -h[0] / std::pow(h[2], 2)}, {0, 1 / h[2], -h[1] / std::pow(h[2], 2)}};
            double HpRc[2][3];
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) HpRc[a][b] = Hp[a][0] * Rc.m[0][b] + Hp[a][1] * Rc.m[1][b] + Hp[a][2] * Rc.m[2][b];
            double H[2][3];
            for (int a = 0; a < 2; ++a) {
                for (int b = 0; b < 2; ++b) H[a][b] = HpRc[a][0] * Jang[0][b] + HpRc[a][1] * Jang[1][b] + HpRc[a][2] * Jang[2][b];
                H[a][2] = Hp[a][0] * tC[i][0] + Hp[a][1] * tC[i][1] + Hp[a][2] * tC[i][2];
            }
            float px = (float)(h[0] / h[2]), py = (float)(h[1] / h[2]);
            accumulate(H, meas[2 * (i + 1)] - px, meas[2 * (i + 1) + 1] - py);
        }
        if (cost <= lastCost) {  // :235-256
            Mat A(3, 3);
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A(a, b) = HTH[a][b];
            for (int a = 0; a < 3; ++a) A(a, a) += lambda * A(a, a);
            std::vector<double> dp = colpiv_qr_solve(A, {HTe[0], HTe[1], HTe[2]});
            phi += dp[0]; psi += dp[1]; rho += dp[2];
            ep = v3(std::cos(phi) * std::sin(psi), std::sin(phi), std::cos(phi) * std::cos(psi));
            Jang[0][0] = -std::sin(phi) * std::sin(psi); Jang[0][1] = std::cos(phi) * std::cos(psi);
            Jang[1][0] = std::cos(phi);                  Jang[1][1] = 0;
            Jang[2][0] = -std::sin(phi) * std::cos(psi); Jang[2][1] = -std::cos(phi) * std::sin(psi);
            if (std::fabs(lastCost - cost) < 1e-6 && dp[2] < 1e-6) break;
            lambda *= .1; lastCost = cost;
        } else {  // :257-262 (quirk D.5: lastCost is overwritten on reject too)
            lambda *= 10; lastCost = cost;
        }
    }
    out.phi = phi; out.psi = psi; out.rho = rho;
    if (std::fabs(phi) > .5 * 3.14 || std::fabs(psi) > .5 * 3.14 || std::isinf(rho) || rho < 0) return out;  // :265-269

    if (type == '2') { nTrackLength = (int)std::ceil(.5 * nTrackLength); nTrackPhases = nTrackLength - 1; }  // :271-275

    // residual + Jacobians :280-368
    const int M = 2 * nTrackLength;
    std::vector<double> r(M, 0.0);
    Mat Hx(M, nc6), Hf(M, 3);
    const int nStartCol = (type == '1') ? 6 * (n_clones - nTrackPhases) : 0;  // :288-293
    {
        V3 h = ep;
        float px = (float)(h[0] / h[2]), py = (float)(h[1] / h[2]);
        double Hp[2][3] = {{1 / h[2], 0, -h[0] / std::pow(h[2], 2)}, {0, 1 / h[2], -h[1] / std::pow(h[2], 2)}};
        float e1x = fx0 - px, e1y = fy0 - py;  // :307-308
        r[0] = e1x; r[1] = e1y;
        for (int a = 0; a < 2; ++a) { for (int b = 0; b < 2; ++b) Hf(a, b) = Hp[a][0] * Jang[0][b] + Hp[a][1] * Jang[1][b] + Hp[a][2] * Jang[2][b]; Hf(a, 2) = 0; }
    }
    for (int i = 1; i < nTrackLength; ++i) {
        const int row = 2 * i;
        M3 R = quat_to_rot(qI[i - 1]);
        M3 Rc = quat_to_rot(qC[i - 1]);
        V3 tc = tC[i - 1];
        V3 h = Rc * ep + rho * tc;
        float px = (float)(h[0] / h[2]), py = (float)(h[1] / h[2]);
        double Hp[2][3] = {{1 / h[2], 0, -h[0] / std::pow(h[2], 2)}, {0, 1 / h[2], -h[1] / std::pow(h[2], 2)}};
        float ex_ = meas[2 * i] - px, ey_ = meas[2 * i + 1] - py;  // :338-339
        r[row] = ex_; r[row + 1] = ey_;
        // Hproj*mRci*R  (left-to-right as Eigen evaluates the chain)
        double HpRci[2][3], HRR[2][3];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) HpRci[a][b] = Hp[a][0] * ex.Rci.m[0][b] + Hp[a][1] * ex.Rci.m[1][b] + Hp[a][2] * ex.Rci.m[2][b];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) HRR[a][b] = HpRci[a][0] * R.m[0][b] + HpRci[a][1] * R.m[1][b] + HpRci[a][2] * R.m[2][b];
        for (int j = 0; j < i; ++j) {
            M3 RjT = transpose(quat_to_rot(qI[j]));
            V3 tj = tI[j];
            M3 dpx = skew((ex.Ric * ep + rho * ex.tic) + rho * (RjT * tj));  // :345,357
            M3 left = dpx * RjT;
            M3 right = (j == 0) ? (-rho) * m3_eye() : (-rho) * transpose(quat_to_rot(qI[j - 1]));  // :347,358
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 3; ++b) {
                    Hx(row + a, nStartCol + 6 * j + b) = HRR[a][0] * left.m[0][b] + HRR[a][1] * left.m[1][b] + HRR[a][2] * left.m[2][b];
                    Hx(row + a, nStartCol + 6 * j + 3 + b) = HRR[a][0] * right.m[0][b] + HRR[a][1] * right.m[1][b] + HRR[a][2] * right.m[2][b];
                }
        }
        double HpRc[2][3];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) HpRc[a][b] = Hp[a][0] * Rc.m[0][b] + Hp[a][1] * Rc.m[1][b] + Hp[a][2] * Rc.m[2][b];
        for (int a = 0; a < 2; ++a) {
            for (int b = 0; b < 2; ++b) Hf(row + a, b) = HpRc[a][0] * Jang[0][b] + HpRc[a][1] * Jang[1][b] + HpRc[a][2] * Jang[2][b];
            Hf(row + a, 2) = Hp[a][0] * tc[0] + Hp[a][1] * tc[1] + Hp[a][2] * tc[2];
        }
    }

    if (probe) {
        if (probe->Hx_raw) *probe->Hx_raw = Hx;
        if (probe->Hf_raw) *probe->Hf_raw = Hf;
        if (probe->r_raw) *probe->r_raw = r;
    }
    // Givens left-nullspace marginalisation :370-402
    int N = 3;
    { double s = 0; for (int i = 0; i < M; ++i) s += Hf(i, 2) * Hf(i, 2); if (std::sqrt(s) < 1e-4) N--; }
    Mat rM(M, 1); for (int i = 0; i < M; ++i) rM(i, 0) = r[i];
    for (int n = 0; n < N; ++n)
        for (int m = M - 1; m > n; --m) {
            Givens g = make_givens(Hf(m - 1, n), Hf(m, n));
            apply_givens_rows(Hf, m - 1, m, n, N - n, g);
            apply_givens_rows(Hx, m - 1, m, 0, nc6, g);
            apply_givens_rows(rM, m - 1, m, 0, 1, g);
        }

    // Mahalanobis gate :404-422
    const int nDOF = M - N;
    out.ndof = nDOF;
    out.Hx_ = Hx.block(N, 0, nDOF, nc6);
    out.r_.resize(nDOF); for (int i = 0; i < nDOF; ++i) out.r_[i] = rM(N + i, 0);
    Mat S = mul_nt(mul(out.Hx_, Pcc), out.Hx_);
    for (int i = 0; i < nDOF; ++i) S(i, i) += std::pow(sig, 2);
    symmetrize(S);
    std::vector<double> y = colpiv_qr_solve(S, out.r_);
    double g = 0; for (int i = 0; i < nDOF; ++i) g += out.r_[i] * y[i];
    out.gamma = std::fabs(g);
    out.accepted = out.gamma < kChi2[nDOF - 1];
    return out;
}

// U8..U10 given the compressed pair (Hn = [0 | Hw], rn): Updater.cc:540-619
void ekf_apply(const double* x, int xdim, const Mat& P, const Mat& Hw, const std::vector<double>& rn, double sig,
               double* x_out, Mat& P_out) {
    const int d = P.r, r = Hw.r, n = (xdim - 26) / 7;
    Mat Hn(r, d);
    Hn.set_block(0, 24, Hw);
    Mat PHt = mul_nt(P, Hn);           // d x r
    Mat S = mul(Hn, PHt);              // r x r
    for (int i = 0; i < r; ++i) S(i, i) += std::pow(sig, 2);
    symmetrize(S);
    Mat K = mul(PHt, lu_inverse(S));   // :543
    std::vector<double> dx(d, 0.0);
    for (int k = 0; k < r; ++k) for (int i = 0; i < d; ++i) dx[i] += K(i, k) * rn[k];

    // state injection :546-613
    Q4 q = quat_mul(small_quat(dx[0], dx[1], dx[2]), q4_at(x));
    for (int i = 0; i < 4; ++i) x_out[i] = q[i];
    for (int i = 0; i < 6; ++i) x_out[4 + i] = dx[3 + i] + x[4 + i];
    V3 g = normalized(v3_at(x_out + 7));
    for (int i = 0; i < 3; ++i) x_out[7 + i] = g[i];
    q = quat_mul(small_quat(dx[9], dx[10], dx[11]), q4_at(x + 10));
    for (int i = 0; i < 4; ++i) x_out[10 + i] = q[i];
    for (int i = 0; i < 12; ++i) x_out[14 + i] = dx[12 + i] + x[14 + i];
    for (int p = 0; p < n; ++p) {
        q = quat_mul(small_quat(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), q4_at(x + 26 + 7 * p));
        for (int i = 0; i < 4; ++i) x_out[26 + 7 * p + i] = q[i];
        for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
    }
    // Joseph form :615-619
    Mat IKH = sub(Mat::identity(d), mul(K, Hn));
    P_out = mul_nt(mul(IKH, P), IKH);
    Mat KKt = mul_nt(K, K);
    const double Rn0 = std::pow(sig, 2);
    for (size_t i = 0; i < P_out.a.size(); ++i) P_out.a[i] += Rn0 * KKt.a[i];
    symmetrize(P_out);
}

Mat mat_from(const double* P, int d) { Mat M(d, d); std::memcpy(M.a.data(), P, sizeof(double) * d * d); return M; }

}  // namespace

// =================================================================== C API
extern "C" {

void orc_quat_mul(const double q1[4], const double q2[4], double out[4]) { Q4 q = quat_mul(q4_at(q1), q4_at(q2)); for (int i = 0; i < 4; ++i) out[i] = q[i]; }
void orc_quat_to_rot(const double q[4], double R[9]) { M3 m = quat_to_rot(q4_at(q)); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = m.m[i][j]; }
void orc_rot_to_quat(const double R[9], double q[4]) { M3 m; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.m[i][j] = R[3 * i + j]; Q4 o = rot_to_quat(m); for (int i = 0; i < 4; ++i) q[i] = o[i]; }
double orc_chi2_95(int dof) { return kChi2[dof - 1]; }

// System::initialize, System.cc:115-170
void orc_initialize(const rvio_config* cfg, const double w[3], const double a[3], int n_imu, double x[26], double P[576]) {
    V3 g = normalized(v3_at(a));
    M3 R = m3_eye();
    if (cfg->ini_enable_alignment) {
        V3 zv = g;
        // xv = ex - zv*zv^T*ex  evaluated left to right: (zv zv^T) ex
        V3 exv = v3(1, 0, 0);
        V3 xv;
        for (int i = 0; i < 3; ++i) xv[i] = exv[i] - (zv[i] * zv[0] * exv[0] + zv[i] * zv[1] * exv[1] + zv[i] * zv[2] * exv[2]);
        xv = normalized(xv);
        V3 yv = normalized(skew(zv) * xv);
        for (int i = 0; i < 3; ++i) { R.m[i][0] = xv[i]; R.m[i][1] = yv[i]; R.m[i][2] = zv[i]; }
    }
    for (int i = 0; i < 26; ++i) x[i] = 0;
    Q4 q = rot_to_quat(R);
    for (int i = 0; i < 4; ++i) x[i] = q[i];
    for (int i = 0; i < 3; ++i) x[7 + i] = g[i];
    if (n_imu > 1) for (int i = 0; i < 3; ++i) { x[20 + i] = w[i]; x[23 + i] = a[i] - cfg->gravity * g[i]; }
    double dt = 1. / cfg->imu_rate;
    for (int i = 0; i < 576; ++i) P[i] = 0;
    auto D = [&](int i, double v) { P[i * 24 + i] = v; };
    for (int i = 0; i < 6; ++i) D(i, std::pow(1e-3, 2));
    for (int i = 6; i < 9; ++i) D(i, n_imu * dt * std::pow(cfg->sigma_a, 2));
    for (int i = 18; i < 21; ++i) D(i, n_imu * dt * std::pow(cfg->sigma_wg, 2));
    for (int i = 21; i < 24; ++i) D(i, n_imu * dt * std::pow(cfg->sigma_wa, 2));
}

// PreIntegrator::propagate, PreIntegrator.cc:51-194
void orc_propagate(const rvio_config* cfg, const double* x, int xdim, double* Pio, int d,
                   const rvio_imu* imu, int m, double* x_out) {
    V3 gk = v3_at(x + 7);
    Q4 qk = q4_at(x + 10);
    V3 pk = v3_at(x + 14), vk = v3_at(x + 17), bg = v3_at(x + 20), ba = v3_at(x + 23);
    const V3 gR = gk, vR = vk;
    M3 Rk = quat_to_rot(qk), RkT = transpose(Rk);
    V3 dp = v3(0, 0, 0), dv = v3(0, 0, 0);
    Mat F(24, 24), Psi = Mat::identity(24), G(24, 12), Sig(12, 12);
    const double sg[4] = {cfg->sigma_g, cfg->sigma_wg, cfg->sigma_a, cfg->sigma_wa};  // :40-44
    for (int b = 0; b < 4; ++b) for (int i = 0; i < 3; ++i) Sig(3 * b + i, 3 * b + i) = std::pow(sg[b], 2);
    Mat P = mat_from(Pio, d);
    const M3 I = m3_eye();
    const double nG = cfg->gravity;
    double Dt = 0;
    for (int s = 0; s < m; ++s) {
        V3 wm = v3_at(imu[s].w), am = v3_at(imu[s].a);
        double dt = imu[s].dt;
        Dt += dt;
        V3 w = wm - bg, a = am - ba;
        bool small = norm(w) < cfg->small_angle;
        double w1 = norm(w), wdt = w1 * dt, wdt2 = wdt * wdt;
        double cw = std::cos(wdt), sw = std::sin(wdt);
        M3 wx = skew(w), wx2 = wx * wx, vx = skew(vk);
        // covariance :123-142
        put(F, 9, 9, -1.0 * wx);
        put(F, 9, 18, -1.0 * I);
        put(F, 12, 9, (-1.0 * RkT) * vx);
        put(F, 12, 15, RkT);
        put(F, 15, 6, (-nG) * Rk);
        put(F, 15, 9, (-nG) * skew(gk));
        put(F, 15, 15, -1.0 * wx);
        put(F, 15, 18, -1.0 * vx);
        put(F, 15, 21, -1.0 * I);
        Mat Phi = add(Mat::identity(24), scale(F, dt));
        Psi = mul(Phi, Psi);
        put(G, 9, 0, -1.0 * I);
        put(G, 15, 0, -1.0 * vx);
        put(G, 15, 6, -1.0 * I);
        put(G, 18, 3, I);
        put(G, 21, 9, I);
        Mat Q = mul_nt(mul(scale(G, dt), Sig), G);
        Mat P11 = add(mul_nt(mul(Phi, P.block(0, 0, 24, 24)), Phi), Q);
        P.set_block(0, 0, P11);
        // state :145-178
        M3 dR; double f1, f2, f3, f4;
        if (small) {
            dR = (I - dt * wx) + (std::pow(dt, 2) / 2) * wx2;
            f1 = -std::pow(dt, 3) / 3; f2 = std::pow(dt, 4) / 8; f3 = -std::pow(dt, 2) / 2; f4 = std::pow(dt, 3) / 6;
        } else {
            dR = (I - (sw / w1) * wx) + ((1 - cw) / std::pow(w1, 2)) * wx2;
            f1 = (wdt * cw - sw) / std::pow(w1, 3);
            f2 = .5 * (wdt2 - 2 * cw - 2 * wdt * sw + 2) / std::pow(w1, 4);
            f3 = (cw - 1) / std::pow(w1, 2);
            f4 = (wdt - sw) / std::pow(w1, 3);
        }
        Rk = dR * Rk; RkT = transpose(Rk);
        dp = dp + dt * dv;
        dp = dp + (RkT * (((.5 * std::pow(dt, 2)) * I + f1 * wx) + f2 * wx2)) * a;
        dv = dv + (RkT * ((dt * I + f3 * wx) + f4 * wx2)) * a;
        pk = (Dt * vR - (.5 * nG * std::pow(Dt, 2)) * gR) + dp;
        vk = Rk * ((vR - (nG * Dt) * gR) + dv);
        gk = normalized(Rk * gR);
    }
    for (int i = 0; i < xdim; ++i) x_out[i] = x[i];
    Q4 qn = rot_to_quat(Rk);
    for (int i = 0; i < 4; ++i) x_out[10 + i] = qn[i];
    for (int i = 0; i < 3; ++i) { x_out[14 + i] = pk[i]; x_out[17 + i] = vk[i]; }
    int n = (xdim - 26) / 7;
    if (n > 0) {  // :186-191
        Mat P12 = mul(Psi, P.block(0, 24, 24, 6 * n));
        P.set_block(0, 24, P12);
        P.set_block(24, 0, P12.t());
    }
    symmetrize(P);
    std::memcpy(Pio, P.a.data(), sizeof(double) * d * d);
}

// U1..U6 (Updater.cc:89-463): the stacked pair (Hw = Hx[:,24:], r) of the accepted features, in feature order.
// Hw_out: nRows x 6n column-major with leading dimension ld_rows (>= sum of 2*len); returns nRowCount.
static int stack_rows(const rvio_config* cfg, const double* x, int xdim, const Mat& P, const rvio_tracks* tr,
                      Mat& Hw, std::vector<double>& r, int32_t* accepted, double* gamma, int32_t* ndof, double* pfinv, int* n_good) {
    const int n = (xdim - 26) / 7, nc6 = 6 * n;
    const Extr ex = extrinsics(cfg);
    const double sig = sigma_im(cfg);
    Mat Pcc = P.block(24, 24, nc6, nc6);
    int nRows = 0;
    for (int f = 0; f < tr->n_feat; ++f) nRows += 2 * tr->len[f];
    r.assign(nRows, 0.0);
    Hw = Mat(nRows, nc6);  // Hx[:,24:]; columns 0..23 of Hx are identically zero (:102-104,425)
    int nRowCount = 0, nGood = 0;
    for (int f = 0; f < tr->n_feat; ++f) {
        FeatOut fo = feature_rows(cfg, ex, sig, x, n, Pcc, tr->types[f], tr->meas + (size_t)f * tr->max_len * 2, tr->len[f]);
        if (accepted) accepted[f] = fo.accepted;
        if (gamma) gamma[f] = fo.gamma;
        if (ndof) ndof[f] = fo.ndof;
        if (pfinv) { pfinv[3 * f] = fo.phi; pfinv[3 * f + 1] = fo.psi; pfinv[3 * f + 2] = fo.rho; }
        if (!fo.accepted) continue;
        for (int i = 0; i < fo.ndof; ++i) { r[nRowCount + i] = fo.r_[i]; for (int j = 0; j < nc6; ++j) Hw(nRowCount + i, j) = fo.Hx_(i, j); }
        nRowCount += fo.ndof; nGood++;
    }
    *n_good = nGood;
    return nRowCount;
}

// The tall branch of the measurement compression, Updater.cc:474-523, in place: trailing all-zero columns dropped (:482-491), the
// sequential Givens sweep column by column, rows bottom-up (:496-512), and the leading-row scan (:516-523).  Returns nRank.
static int givens_compress(Mat& Ho, Mat& roM, int M, int nc6, int* n_cols = nullptr) {
    int N = nc6;
    for (int i = N; i > 0; --i) {  // drop trailing all-zero columns :482-491
        double s = 0; for (int k = 0; k < M; ++k) s += Ho(k, i - 1) * Ho(k, i - 1);
        if (std::sqrt(s) == 0) N--; else break;
    }
    for (int c = 0; c < N; ++c)
        for (int m = M - 1; m > c; --m) {
            Givens g = make_givens(Ho(m - 1, c), Ho(m, c));
            apply_givens_rows(Ho, m - 1, m, c, N - c, g);
            apply_givens_rows(roM, m - 1, m, 0, 1, g);
        }
    int nRank = 0;  // :516-523
    for (int i = 0; i < M; ++i) {
        double s = 0; for (int j = 0; j < nc6; ++j) s += Ho(i, j) * Ho(i, j);
        if (std::sqrt(s) < 1e-4) break; else nRank++;
    }
    if (n_cols) *n_cols = N;
    return nRank;
}

// U7..U10 (Updater.cc:460-627) on a given stacked pair: Ho = first nRowCount rows of Hw.
// row_norms (may be NULL): norms of the rows of Ho after the Givens sweep, min(M, 2*nc6) entries (diagnostic).
static void compress_and_apply(const rvio_config* cfg, const double* x, int xdim, const double* Pin, int d, const Mat& P,
                               const Mat& Hw, const std::vector<double>& r, int nRowCount, int nGood,
                               double* x_out, double* P_out, int32_t info[4], double* row_norms) {
    const int n = (xdim - 26) / 7, nc6 = 6 * n;
    const double sig = sigma_im(cfg);
    if (info) { info[0] = nGood; info[1] = nRowCount; info[2] = -1; info[3] = 0; }
    if (nGood > 2) {  // :460
        Mat Ho = Hw.block(0, 0, nRowCount, nc6);
        std::vector<double> ro(r.begin(), r.begin() + nRowCount);
        Mat Hn; std::vector<double> rn;
        if (nRowCount > nc6) {  // tall :474-529
            const int M = nRowCount;
            Mat roM(M, 1); for (int i = 0; i < M; ++i) roM(i, 0) = ro[i];
            const int nRank = givens_compress(Ho, roM, M, nc6);
            if (row_norms)
                for (int i = 0; i < std::min(M, 2 * nc6); ++i) {
                    double s = 0; for (int j = 0; j < nc6; ++j) s += Ho(i, j) * Ho(i, j);
                    row_norms[i] = std::sqrt(s);
                }
            Hn = Ho.block(0, 0, nRank, nc6);
            rn.resize(nRank); for (int i = 0; i < nRank; ++i) rn[i] = roM(i, 0);
            if (info) info[2] = nRank;
        } else { Hn = Ho; rn = ro; }
        Mat Pn;
        ekf_apply(x, xdim, P, Hn, rn, sig, x_out, Pn);
        std::memcpy(P_out, Pn.a.data(), sizeof(double) * d * d);
        if (info) info[3] = 1;
    } else {  // :621-627
        std::memcpy(x_out, x, sizeof(double) * xdim);
        std::memcpy(P_out, Pin, sizeof(double) * d * d);
    }
}

// Updater::update, Updater.cc:72-628
void orc_update(const rvio_config* cfg, const double* x, int xdim, const double* Pin, int d,
                const rvio_tracks* tr, double* x_out, double* P_out,
                int32_t* accepted, double* gamma, int32_t* ndof, double* pfinv, int32_t info[4]) {
    Mat P = mat_from(Pin, d);
    Mat Hw; std::vector<double> r; int nGood = 0;
    int nRowCount = stack_rows(cfg, x, xdim, P, tr, Hw, r, accepted, gamma, ndof, pfinv, &nGood);
    compress_and_apply(cfg, x, xdim, Pin, d, P, Hw, r, nRowCount, nGood, x_out, P_out, info, nullptr);
}

// U1..U3 of one feature at a given (phi, psi, rho) (pf == NULL: the LM estimate): residual (2 Lu), Hx (2 Lu x 6n, row-major) and
// Hf (2 Lu x 3) before the nullspace projection; returns 2 Lu (Lu = L, or ceil(L/2) for type '2'), 0 if the feature is rejected early
int orc_feature_model(const rvio_config* cfg, const double* x, int xdim, unsigned char type, const float* meas, int L,
                      const double* pf, double* r_out, double* Hx_rowmajor, double* Hf_rowmajor, double* pf_out) {
    const int n = (xdim - 26) / 7, nc6 = 6 * n;
    const Extr ex = extrinsics(cfg);
    Mat Pcc = Mat::identity(nc6);
    Mat Hx, Hf; std::vector<double> r;
    FeatProbe pr; pr.pf = pf; pr.Hx_raw = &Hx; pr.Hf_raw = &Hf; pr.r_raw = &r;
    FeatOut fo = feature_rows(cfg, ex, sigma_im(cfg), x, n, Pcc, type, meas, L, &pr);
    if (pf_out) { pf_out[0] = fo.phi; pf_out[1] = fo.psi; pf_out[2] = fo.rho; }
    if (r.empty()) return 0;
    const int M = (int)r.size();
    for (int i = 0; i < M; ++i) {
        r_out[i] = r[i];
        for (int j = 0; j < nc6; ++j) Hx_rowmajor[(size_t)i * nc6 + j] = Hx(i, j);
        for (int j = 0; j < 3; ++j) Hf_rowmajor[i * 3 + j] = Hf(i, j);
    }
    return M;
}

// The two halves of orc_update, separately (analysis of the rank truncation, tests/test_truncation.py):
// orc_update_stack returns the stacked pair (row-major M x 6n, M = return value; buffers sized sum(2*len) rows).
int orc_update_stack(const rvio_config* cfg, const double* x, int xdim, const double* Pin, int d,
                     const rvio_tracks* tr, double* Hw_rowmajor, double* r_out, int32_t* n_good) {
    const int nc6 = 6 * ((xdim - 26) / 7);
    Mat P = mat_from(Pin, d);
    Mat Hw; std::vector<double> r; int nGood = 0;
    int M = stack_rows(cfg, x, xdim, P, tr, Hw, r, nullptr, nullptr, nullptr, nullptr, &nGood);
    for (int i = 0; i < M; ++i) { r_out[i] = r[i]; for (int j = 0; j < nc6; ++j) Hw_rowmajor[(size_t)i * nc6 + j] = Hw(i, j); }
    *n_good = nGood;
    return M;
}
void orc_update_from_stack(const rvio_config* cfg, const double* x, int xdim, const double* Pin, int d,
                           const double* Hw_rowmajor, const double* r_in, int M, int n_good,
                           double* x_out, double* P_out, int32_t info[4], double* row_norms) {
    const int nc6 = 6 * ((xdim - 26) / 7);
    Mat P = mat_from(Pin, d);
    Mat Hw(M, nc6); std::vector<double> r(r_in, r_in + M);
    for (int i = 0; i < M; ++i) for (int j = 0; j < nc6; ++j) Hw(i, j) = Hw_rowmajor[(size_t)i * nc6 + j];
    compress_and_apply(cfg, x, xdim, Pin, d, P, Hw, r, M, n_good, x_out, P_out, info, row_norms);
}

// ---------------------------------------------------------------------------------------------------------------
// Information-form compression (CPU mirror of the device design).
//
// [A|b] = sum_f Hx_f^T [Hx_f | r_f] replaces the Givens QR of Updater.cc:469-512, and the reference's leading-row rank scan
// (Updater.cc:516-529) is reproduced by its structural equivalent.  What the scan does to the result (derived from the
// sweep's handling of exact zeros — makeGivens(0,q) swaps, makeGivens(p,0) is the identity — and checked against the
// literal path above in tests/test_truncation.py): rows of type-'2' features span columns [0, e2], e2 = 6(ceil(L/2)-1)-1;
// rows of type-'1' features span [6(n-L+1), 6n-1].  If every accepted type-'1' feature starts behind e2, the sweep of
// columns 0..e2 never mixes the two families.  The type-'2' block has the scale gauge of a monocular window as null
// direction, so its column e2 is dependent: when that block has more than e2 rows, a left-over row of rounding residue
// (norm ~1e-15) is carried to position e2, the scan stops there (nRank = e2) and every type-'1' row is discarded.  In all
// other constellations the scan only drops rows that are zero to rounding.  Hence: keep the type-'2' sum apart from the
// type-'1' sum, and drop the latter iff  (a) min start column of the accepted type-'1' features > e2,  (b) the accepted
// type-'2' rows number >= e2+1,  (c) column e2 of the type-'2' block is dependent on columns 0..e2-1: the Schur
// complement of A2[e2][e2] is < (1e-4)^2 (the scan's threshold on the row norm),  (d) the stack is tall (rows > 6n).
//
// block (per shard; doubles): part0 = type-'2' sum [6n x (6n+1)], part1 = type-'1' sum, then 8 doubles
//   {n_good, n_rows, rows of type '2', e2 (-1: none), min start column of type '1' (1e9: none), literal mode, literal nRank, columns swept (6n less the trailing all-zero columns)}.
//
// Round 6 — the LITERAL sweep for small stacks (the device: csrc/literal.h).  The structural rule above covers every simulated
// sequence but not every stack: with a handful of accepted features a column gap or a weak row stops the reference's scan where
// the rule predicts nothing (tests/test_truncation.py, round 5: 0.2-0.4 % of such updates, up to 2e-3 of state).  So an update
// that is handed at most ORC_LIT_FEATS features and whose stack is barely tall (rows - 6n <= 8) or has a column gap behind an
// over-determined group (gap_trigger below) runs the reference's own sequence — Givens nullspace per feature
// (Updater.cc:370-402: feature_rows above), Givens QR of the stack in the reference's row order + leading-row scan
// (Updater.cc:493-523: givens_compress) — and hands [A|b] = Rn^T [Rn | zn] of the nRank leading rows to the same solve.
// Such an update is not sharded: every rank builds every feature, block 0 alone is used (literal mode 1: part0 = the literal
// [A|b]; 2: the stack was not tall or had <= 2 accepted features — part0/part1 are the plain sums of ALL features).
static int e2_of(int L) { return 6 * ((int)std::ceil(.5 * L) - 1) - 1; }
// = LIT_FEATS of csrc/literal.h; the environment variable ORC_LIT_FEATS overrides it (0: the structural rule alone, as up to round 5 — tests / studies)
static int orc_lit_feats() { const char* e = std::getenv("ORC_LIT_FEATS"); return e ? std::atoi(e) : 24; }

// = LIT_SPARE of csrc/literal.h: a gap only stops the scan when the stack has few rows to spare (rows - 6n small) — with many, informative rows
// move up into the gap and the residue ends at the bottom (every exception found: rows - 6n <= 21; the stock sequence's one trigger: 390)
static int orc_lit_spare() { const char* e = std::getenv("ORC_LIT_SPARE"); return e ? std::atoi(e) : 48; }
static int orc_lit_slack() { const char* e = std::getenv("ORC_LIT_SLACK"); return e ? std::atoi(e) : 0; }   // = LIT_SLACK of csrc/literal.h (0: the barely-tall trigger is off)
// A column gap behind an over-determined group (csrc/literal.h:lit_gap_trigger): the accepted features' rows are counted against their
// column spans in the order of the start columns (type '2': columns 0..e2, rank <= e2 — the scale gauge of a monocular window; type
// '1': columns 6 (n - L + 1) .. 6n - 1).  True when a column no feature can fill comes up while later features still wait AND a group
// before it had more rows than its span can hold: the left-over rows of that group are rounding residue, the sweep compacts them
// into the gap, the reference's scan stops there.
static bool gap_trigger(int n, const std::vector<int>& types, const std::vector<int>& lens, const std::vector<int>& nrows) {
    std::vector<int> rows_k(n + 1, 0), end_k(n + 1, -1);
    bool gauge0 = false, any1_0 = false;
    for (size_t f = 0; f < types.size(); ++f) {
        const int L = lens[f];
        int k, e;
        if (types[f] == '2') { k = 0; e = e2_of(L); gauge0 = true; }
        else { k = n - (L - 1); e = 6 * n - 1; if (k == 0) any1_0 = true; }
        if (k < 0 || k > n || nrows[f] <= 0) continue;
        rows_k[k] += nrows[f]; end_k[k] = std::max(end_k[k], e);
    }
    const bool only2 = gauge0 && !any1_0 && rows_k[0] > 0;     // a type-'2' block alone at column 0
    int p = 0, done = 0; bool over = false;
    for (int k = 0; k <= n; ++k) {
        if (rows_k[k] == 0) continue;
        if (p < 6 * k) {
            if (!(only2 && done == 1)) return over;   // (the gap right behind a lone type-'2' block is the structural rule's own constellation, conditions (a)-(d) above: the count goes on behind it)
        }
        const int cap = (k == 0 && only2 && end_k[0] < 6 * n - 1) ? end_k[0] : end_k[k] + 1;
        if (p + rows_k[k] > cap) over = true;
        p = std::min(p + rows_k[k], cap);
        ++done;
    }
    return false;
}

void orc_update_local(const rvio_config* cfg, const double* x, int xdim, const double* Pin, int d,
                      const rvio_tracks* tr, int rank, int world, double* block) {
    const int n = (xdim - 26) / 7, nc6 = 6 * n, ld = nc6 + 1, part = nc6 * ld;
    const Extr ex = extrinsics(cfg);
    const double sig = sigma_im(cfg);
    Mat P = mat_from(Pin, d);
    Mat Pcc = P.block(24, 24, nc6, nc6);
    for (int i = 0; i < 2 * part + 8; ++i) block[i] = 0;
    int good = 0, rows = 0, rows2 = 0, e2 = -1, smin = 1000000000;
    const bool lit = tr->n_feat <= orc_lit_feats();
    if (lit) { rank = 0; world = 1; }
    std::vector<FeatOut> kept; std::vector<int> kept_types, kept_lens, kept_rows;
    for (int f = rank; f < tr->n_feat; f += world) {
        FeatOut fo = feature_rows(cfg, ex, sig, x, n, Pcc, tr->types[f], tr->meas + (size_t)f * tr->max_len * 2, tr->len[f]);
        if (!fo.accepted) continue;
        good++; rows += fo.ndof;
        double* B = block;
        if (tr->types[f] == '2') { rows2 += fo.ndof; e2 = std::max(e2, e2_of(tr->len[f])); }
        else { B = block + part; smin = std::min(smin, 6 * (n - (tr->len[f] - 1))); }
        for (int k = 0; k < fo.ndof; ++k)
            for (int i = 0; i < nc6; ++i) {
                double hi = fo.Hx_(k, i);
                if (hi == 0) continue;
                for (int j = 0; j < nc6; ++j) B[i * ld + j] += hi * fo.Hx_(k, j);
                B[i * ld + nc6] += hi * fo.r_[k];
            }
        if (lit) { kept.push_back(fo); kept_types.push_back(tr->types[f]); kept_lens.push_back(tr->len[f]); kept_rows.push_back(fo.ndof); }
    }
    double* m = block + 2 * part;
    m[0] = good; m[1] = rows; m[2] = rows2; m[3] = e2; m[4] = smin;
    if (!lit) return;
    m[5] = 2;
    if (!(good > 2 && rows > nc6)) return;
    if (!(rows - nc6 <= orc_lit_slack() || (rows - nc6 <= orc_lit_spare() && gap_trigger(n, kept_types, kept_lens, kept_rows)))) return;
    // the literal sweep + scan on the stack in feature order
    Mat Ho(rows, nc6), roM(rows, 1);
    int at = 0;
    for (const FeatOut& fo : kept) {
        for (int k = 0; k < fo.ndof; ++k) { roM(at + k, 0) = fo.r_[k]; for (int j = 0; j < nc6; ++j) Ho(at + k, j) = fo.Hx_(k, j); }
        at += fo.ndof;
    }
    int nCols = nc6;
    const int nRank = givens_compress(Ho, roM, rows, nc6, &nCols);
    for (int i = 0; i < 2 * part; ++i) block[i] = 0;
    for (int k = 0; k < nRank; ++k)
        for (int i = k; i < nc6; ++i) {      // (the residue below the diagonal, 1e-17, is not carried: the device's cells do not keep it either)
            const double hi = Ho(k, i);
            if (hi == 0) continue;
            for (int j = k; j < nc6; ++j) block[i * ld + j] += hi * Ho(k, j);
            block[i * ld + nc6] += hi * roM(k, 0);
        }
    m[5] = 1; m[6] = nRank; m[7] = nCols;
}

// (c) above: eliminate columns 0..e-1 of the leading (e+1)x(e+1) block of A2 (square-root-free, unpivoted; a column whose
// pivot has cancelled to rounding level is skipped, as the sweep's mixture row leaves the later columns' span alone)
// and return what is left of A2[e][e].
static double schur_last(const Mat& A2, int e) {
    const int m = e + 1;
    Mat M = A2.block(0, 0, m, m);
    std::vector<double> d0(m);
    for (int i = 0; i < m; ++i) d0[i] = M(i, i);
    for (int k = 0; k < e; ++k) {
        const double dk = M(k, k);
        if (!(dk > 1e-12 * d0[k])) continue;
        const double rd = 1.0 / dk;
        for (int i = k + 1; i < m; ++i) {
            const double f = M(i, k) * rd;
            for (int j = k + 1; j <= i; ++j) M(i, j) -= f * M(j, k);
        }
    }
    return M(e, e);
}

// dx = Pc (s2 I + A Pcc)^-1 b ;  Joseph form written through A ("information-form update")
void orc_update_global(const rvio_config* cfg, const double* x, int xdim, const double* Pin, int d,
                       const double* blocks, int world, double* x_out, double* P_out, int32_t info[4]) {
    const int n = (xdim - 26) / 7, nc6 = 6 * n, ld = nc6 + 1, part = nc6 * ld, blen = 2 * part + 8;
    const double sig = sigma_im(cfg), s2 = std::pow(sig, 2);
    Mat A2(nc6, nc6), A1(nc6, nc6); std::vector<double> b2(nc6, 0.0), b1(nc6, 0.0);
    int good = 0, rows = 0, rows2 = 0, e2 = -1, smin = 1000000000;
    const int lit_mode = (int)blocks[2 * part + 5];   // 0: sharded sums; 1 / 2: the update was small — block 0 holds the whole of it (1: literal [A|b])
    if (lit_mode) world = 1;
    for (int w = 0; w < world; ++w) {
        const double* B = blocks + (size_t)w * blen;
        for (int i = 0; i < nc6; ++i) {
            for (int j = 0; j < nc6; ++j) { A2(i, j) += B[i * ld + j]; A1(i, j) += B[part + i * ld + j]; }
            b2[i] += B[i * ld + nc6]; b1[i] += B[part + i * ld + nc6];
        }
        const double* m = B + 2 * part;
        good += (int)m[0]; rows += (int)m[1]; rows2 += (int)m[2]; e2 = std::max(e2, (int)m[3]); smin = std::min(smin, (int)m[4]);
    }
    // the reference's rank truncation (Updater.cc:516-529), structural form (see above) — unless the literal scan has taken the decision
    bool truncate = false;
    if (lit_mode != 1 && good > 2 && rows > nc6 && e2 >= 0 && e2 < nc6 && smin < 1000000000 && smin > e2 && rows2 >= e2 + 1)
        truncate = !(schur_last(A2, e2) >= 1e-8);
    Mat A = A2; std::vector<double> b = b2;
    if (!truncate) { A = add(A2, A1); for (int i = 0; i < nc6; ++i) b[i] = b2[i] + b1[i]; }
    if (lit_mode == 1) { e2 = (int)blocks[2 * part + 6]; truncate = e2 < (int)blocks[2 * part + 7]; }   // reported: the literal nRank when the scan stopped early
    if (info) { info[0] = good; info[1] = rows; info[2] = truncate ? e2 : -1; info[3] = 0; }
    if (good <= 2) { std::memcpy(x_out, x, sizeof(double) * xdim); std::memcpy(P_out, Pin, sizeof(double) * d * d); return; }
    // Reuse ekf_apply with the Cholesky-free equivalent pair: eigen-free route —
    // form Hn := A^(1/2) is not needed; solve directly.
    Mat P = mat_from(Pin, d);
    Mat Pc = P.block(0, 24, d, nc6), Pcc = P.block(24, 24, nc6, nc6);
    Mat T = mul(A, Pcc); for (int i = 0; i < nc6; ++i) T(i, i) += s2;
    Mat W = lu_inverse(T);
    Mat WA = mul(W, A);                       // = Hw^T S^-1 Hw (symmetric)
    std::vector<double> Wb(nc6, 0.0);
    for (int k = 0; k < nc6; ++k) for (int i = 0; i < nc6; ++i) Wb[i] += W(i, k) * b[k];
    std::vector<double> dx(d, 0.0);
    for (int k = 0; k < nc6; ++k) for (int i = 0; i < d; ++i) dx[i] += Pc(i, k) * Wb[k];
    // state injection identical to ekf_apply
    {
        Q4 q = quat_mul(small_quat(dx[0], dx[1], dx[2]), q4_at(x));
        for (int i = 0; i < 4; ++i) x_out[i] = q[i];
        for (int i = 0; i < 6; ++i) x_out[4 + i] = dx[3 + i] + x[4 + i];
        V3 g = normalized(v3_at(x_out + 7));
        for (int i = 0; i < 3; ++i) x_out[7 + i] = g[i];
        q = quat_mul(small_quat(dx[9], dx[10], dx[11]), q4_at(x + 10));
        for (int i = 0; i < 4; ++i) x_out[10 + i] = q[i];
        for (int i = 0; i < 12; ++i) x_out[14 + i] = dx[12 + i] + x[14 + i];
        for (int p = 0; p < n; ++p) {
            q = quat_mul(small_quat(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), q4_at(x + 26 + 7 * p));
            for (int i = 0; i < 4; ++i) x_out[26 + 7 * p + i] = q[i];
            for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
        }
    }
    // Joseph: P+ = P - G Pc^T - Pc G^T + G Pcc G^T + s2 * (Pc W) A (Pc W)^T,  G = Pc W A
    Mat G = mul(Pc, WA), PcW = mul(Pc, W);
    Mat GPct = mul_nt(G, Pc);
    Mat Pn = sub(sub(P, GPct), GPct.t());
    Pn = add(Pn, mul_nt(mul(G, Pcc), G));
    Pn = add(Pn, scale(mul_nt(mul(PcW, A), PcW), s2));
    symmetrize(Pn);
    std::memcpy(P_out, Pn.a.data(), sizeof(double) * d * d);
    if (info) info[3] = 1;
}

// System.cc:279-365
void orc_augment_compose(const rvio_config* cfg, double* x, int* xdim_io, double* Pio, int* d_io,
                         int do_augment, double pose_p[3], double pose_q[4]) {
    int xdim = *xdim_io, d = *d_io, n = (xdim - 26) / 7;
    const int win = cfg->max_track_len - 1;  // mnSlidingWindowSize, System.cc:71-72
    Mat P = mat_from(Pio, d);
    if (do_augment) {
        // J = [I; rows 9..14]; tempP = J P J^T then .5(tempP+tempP^T)  (:288-298,:308-317)
        Mat J(d + 6, d);
        for (int i = 0; i < d; ++i) J(i, i) = 1;
        for (int i = 0; i < 3; ++i) { J(d + i, 9 + i) = 1; J(d + 3 + i, 12 + i) = 1; }
        Mat tP = mul_nt(mul(J, P), J);
        symmetrize(tP);
        if (n < win) {
            for (int i = 0; i < 7; ++i) x[xdim + i] = x[10 + i];
            xdim += 7; d += 6; n += 1;
            P = tP;
        } else {
            // drop oldest clone :303-321
            std::vector<double> nx(x, x + 26);
            nx.insert(nx.end(), x + 26 + 7, x + 26 + 7 * win);
            nx.insert(nx.end(), x + 10, x + 17);
            for (int i = 0; i < xdim; ++i) x[i] = nx[i];
            const int w6 = 6 * win;
            P.set_block(0, 0, tP.block(0, 0, 24, 24));
            P.set_block(0, 24, tP.block(0, 30, 24, w6));
            P.set_block(24, 0, tP.block(30, 0, w6, 24));
            P.set_block(24, 24, tP.block(30, 30, w6, w6));
        }
    }
    // composition :325-365
    Q4 qG = q4_at(x), qk = q4_at(x + 10);
    V3 pG = v3_at(x + 4), gk = v3_at(x + 7), pk = v3_at(x + 14);
    M3 RG = quat_to_rot(qG), Rk = quat_to_rot(qk);
    gk = normalized(Rk * gk);
    Q4 qkG = quat_mul(qk, qG);
    V3 pkG = Rk * (pG - pk);
    V3 pGk = transpose(RG) * (pk - pG);
    Mat Vk(24, 24);
    put(Vk, 0, 0, Rk); put(Vk, 0, 9, m3_eye());
    put(Vk, 3, 3, Rk); put(Vk, 3, 9, skew(pkG)); put(Vk, 3, 12, -1.0 * Rk);
    put(Vk, 6, 6, Rk); put(Vk, 6, 9, skew(gk));
    for (int i = 15; i < 24; ++i) Vk(i, i) = 1;
    P.set_block(0, 0, mul_nt(mul(Vk, P.block(0, 0, 24, 24)), Vk));
    if (n > 0) {
        Mat P12 = mul(Vk, P.block(0, 24, 24, 6 * n));
        P.set_block(0, 24, P12);
        P.set_block(24, 0, P12.t());
    }
    symmetrize(P);
    for (int i = 0; i < 4; ++i) x[i] = qkG[i];
    for (int i = 0; i < 3; ++i) { x[4 + i] = pkG[i]; x[7 + i] = gk[i]; x[14 + i] = 0; }
    x[10] = 0; x[11] = 0; x[12] = 0; x[13] = 1;
    if (pose_p) for (int i = 0; i < 3; ++i) pose_p[i] = pGk[i];
    if (pose_q) for (int i = 0; i < 4; ++i) pose_q[i] = qkG[i];
    *xdim_io = xdim; *d_io = d;
    std::memcpy(Pio, P.a.data(), sizeof(double) * d * d);
}

}  // extern "C"
