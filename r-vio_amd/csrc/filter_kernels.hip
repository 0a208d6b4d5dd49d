// filter_kernels.hip — hand-written gfx950 kernels for the filter half of the
// R-VIO hot path.
//
//   propagate_kernel     PreIntegrator::propagate          PreIntegrator.cc:51-194
//   feat_build_kernel    Updater::update per-feature loop  Updater.cc:109-455
//   gram_kernel/gram_reduce_kernel   measurement compression (Updater.cc:469-536)
//                        in information form [A|b] = Hw^T [Hw | r]   
//   gemm_f64_kernel      FP64-MFMA (v_mfma_f64_16x16x4_f64) tiled GEMM used for
//                        every dense covariance product of Updater.cc:540-619
//   gj_kernel            (sigma^2 I + A Pcc)^-1 by Gauss-Jordan with partial pivoting
//   inject_kernel        state injection                   Updater.cc:546-613
//   symm_out_kernel      P = .5 (P + P^T)                  Updater.cc:619
//   augment_kernel / compose_kernel   System.cc:279-365
#include "rvio_dev.h"
#include "../../include/rvio_hip.h"

__device__ const double kChi2Dev[500] = {
#include "chi2_table.inc"
};

// =============================================================== P1 propagate
// One workgroup, 256 threads.  Only rows 9..17 of Phi = I + dt F differ from the
// identity (PreIntegrator.cc:123-132), so Phi P Phi^T touches 9 rows then 9
// columns of the 24x24 IMU block; the scalar state integration is evaluated
// redundantly by every lane (uniform control flow, no broadcasts).
__device__ __forceinline__ double phi9_entry(int r, int c, double dt, const m33& wx, const m33& RkTvx, const m33& RkT,
                                             const m33& Rk, const m33& gx, const m33& vx, double nG) {
    const int br = r / 3, i = r % 3, bc = c / 3, j = c % 3;
    const double id = (i == j) ? 1.0 : 0.0;
    if (br == 0) {          // theta_k rows (F rows 9..11)
        if (bc == 3) return id - dt * wx.m[3 * i + j];
        if (bc == 6) return -dt * id;
        return 0.0;
    } else if (br == 1) {   // p_k rows (12..14)
        if (bc == 3) return -dt * RkTvx.m[3 * i + j];
        if (bc == 4) return id;
        if (bc == 5) return dt * RkT.m[3 * i + j];
        return 0.0;
    } else {                // v rows (15..17)
        if (bc == 2) return -dt * nG * Rk.m[3 * i + j];
        if (bc == 3) return -dt * nG * gx.m[3 * i + j];
        if (bc == 5) return id - dt * wx.m[3 * i + j];
        if (bc == 6) return -dt * vx.m[3 * i + j];
        if (bc == 7) return -dt * id;
        return 0.0;
    }
}

__global__ __launch_bounds__(256) void propagate_kernel(DevCfg cfg, FilterMeta* meta, double* x, double* P,
                                                        const rvio_imu* imu, int m) {
    __shared__ double Pl[24][25];
    __shared__ double Psi[24][25];
    __shared__ double Phi9[9][25];
    const int tid = threadIdx.x;
    const int n = meta->n_clones;
    const int ld = cfg.dmax;
    if (tid == 0) { meta->n_good = 0; meta->n_rows = 0; meta->updated = 0; }   // per-frame update statistics
    for (int e = tid; e < 576; e += 256) {
        int i = e % 24, j = e / 24;
        Pl[i][j] = P[i + (size_t)j * ld];
        Psi[i][j] = (i == j) ? 1.0 : 0.0;
    }
    d3 gk = ld3(x + 7);
    const q4 qk0 = ldq(x + 10);
    d3 pk = ld3(x + 14), vk = ld3(x + 17);
    const d3 bg = ld3(x + 20), ba = ld3(x + 23);
    const d3 gR = gk, vR = vk;
    m33 Rk = q2r(qk0), RkT = tr33(Rk);
    d3 dp = mk3(0, 0, 0), dv = mk3(0, 0, 0);
    const m33 I = eye33();
    const double nG = cfg.gravity;
    double Dt = 0;
    const int r9 = tid / 24, c9 = tid % 24;  // valid for tid < 216
    __syncthreads();
    for (int s = 0; s < m; ++s) {
        const d3 wm = mk3(imu[s].w[0], imu[s].w[1], imu[s].w[2]);
        const d3 am = mk3(imu[s].a[0], imu[s].a[1], imu[s].a[2]);
        const double dt = imu[s].dt;
        Dt += dt;
        const d3 w = sub3(wm, bg), a = sub3(am, ba);
        const double w1 = nrm3(w);
        const bool small = w1 < cfg.small_angle;
        const double wdt = w1 * dt, wdt2 = wdt * wdt;
        const double cw = cos(wdt), sw = sin(wdt);
        const m33 wx = skew33(w), wx2 = mul33(wx, wx), vx = skew33(vk);
        const m33 RkTvx = mul33(RkT, vx), gx = skew33(gk);
        if (tid < 216) Phi9[r9][c9] = phi9_entry(r9, c9, dt, wx, RkTvx, RkT, Rk, gx, vx, nG);
        __syncthreads();
        double accP = 0, accS = 0;
        if (tid < 216) {
#pragma unroll 8
            for (int k = 0; k < 24; ++k) { double f = Phi9[r9][k]; accP += f * Pl[k][c9]; accS += f * Psi[k][c9]; }
        }
        __syncthreads();
        if (tid < 216) { Pl[9 + r9][c9] = accP; Psi[9 + r9][c9] = accS; }
        __syncthreads();
        // P' = (Phi P) Phi^T: only columns 9..17 change; thread (i = c9, r = r9) -> P'[i][9+r]
        double accC = 0;
        if (tid < 216) {
#pragma unroll 8
            for (int k = 0; k < 24; ++k) accC += Pl[c9][k] * Phi9[r9][k];
            // Q = dt G Sigma G^T  (PreIntegrator.cc:135-140), nonzero blocks only
            const int i = c9, j = 9 + r9;
            const int bi = i / 3, ii = i % 3, bj = j / 3, jj = j % 3;
            if (bi == 3 && bj == 3) accC += (ii == jj) ? dt * cfg.sg2 : 0.0;
            else if (bi == 3 && bj == 5) accC += dt * cfg.sg2 * vx.m[3 * jj + ii];
            else if (bi == 5 && bj == 3) accC += dt * cfg.sg2 * vx.m[3 * ii + jj];
            else if (bi == 5 && bj == 5) {
                double q = ((dt * vx.m[3 * ii]) * cfg.sg2) * vx.m[3 * jj] + ((dt * vx.m[3 * ii + 1]) * cfg.sg2) * vx.m[3 * jj + 1] +
                           ((dt * vx.m[3 * ii + 2]) * cfg.sg2) * vx.m[3 * jj + 2];
                if (ii == jj) q += dt * cfg.sa2;
                accC += q;
            }
        }
        __syncthreads();
        if (tid < 216) Pl[c9][9 + r9] = accC;
        if (tid >= 216 && tid < 219) Pl[18 + tid - 216][18 + tid - 216] += dt * cfg.swg2;
        if (tid >= 219 && tid < 222) Pl[21 + tid - 219][21 + tid - 219] += dt * cfg.swa2;
        // state (PreIntegrator.cc:145-178)
        m33 dR; double f1, f2, f3, f4;
        if (small) {
            dR = add33(sub33(I, scl33(dt, wx)), scl33(dt * dt / 2, wx2));
            f1 = -(dt * dt * dt) / 3; f2 = (dt * dt * dt * dt) / 8; f3 = -(dt * dt) / 2; f4 = (dt * dt * dt) / 6;
        } else {
            const double w2 = w1 * w1, w3 = w2 * w1, w4 = w2 * w2;
            dR = add33(sub33(I, scl33(sw / w1, wx)), scl33((1 - cw) / w2, wx2));
            f1 = (wdt * cw - sw) / w3;
            f2 = .5 * (wdt2 - 2 * cw - 2 * wdt * sw + 2) / w4;
            f3 = (cw - 1) / w2;
            f4 = (wdt - sw) / w3;
        }
        Rk = mul33(dR, Rk); RkT = tr33(Rk);
        dp = add3(dp, scl3(dt, dv));
        dp = add3(dp, mv33(mul33(RkT, add33(add33(scl33(.5 * dt * dt, I), scl33(f1, wx)), scl33(f2, wx2))), a));
        dv = add3(dv, mv33(mul33(RkT, add33(add33(scl33(dt, I), scl33(f3, wx)), scl33(f4, wx2))), a));
        pk = add3(sub3(scl3(Dt, vR), scl3(.5 * nG * Dt * Dt, gR)), dp);
        vk = mv33(Rk, add3(sub3(vR, scl3(nG * Dt, gR)), dv));
        gk = unit3(mv33(Rk, gR));
        __syncthreads();
    }
    if (tid == 0) {
        stq(x + 10, r2q(Rk));
        st3(x + 14, pk);
        st3(x + 17, vk);
    }
    // P11 back (symmetrised, PreIntegrator.cc:192); P22 is untouched and already symmetric
    for (int e = tid; e < 576; e += 256) {
        int i = e % 24, j = e / 24;
        P[i + (size_t)j * ld] = .5 * (Pl[i][j] + Pl[j][i]);
    }
    // P12 = Psi P12, P21 = P12^T (PreIntegrator.cc:186-191): one thread per clone column
    for (int c = tid; c < 6 * n; c += 256) {
        double col[24];
        double* pc = P + (size_t)(24 + c) * ld;
#pragma unroll
        for (int k = 0; k < 24; ++k) col[k] = pc[k];
        for (int i = 0; i < 24; ++i) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 24; ++k) acc += Psi[i][k] * col[k];
            pc[i] = acc;
            P[(24 + c) + (size_t)i * ld] = acc;
        }
    }
}

// =============================================================== U1..U5 per feature
// One workgroup per feature slot.  Dynamic LDS (doubles):
//   pose[(L-1)*24]  RI(9) tI(3) Rc(9) tc(3) per track phase
//   hrr[L*6] hf[2L*3] lr[(L-1)*18] vh[3*2L] misc[16]
//   Hx[2L][ldh]   ([Hx | r], row-major)   Tm[rho][ldh]   S[(rho+1)][rho+1]
struct FeatLds {
    double *pose, *hrr, *hf, *lr, *vh, *misc, *Hx, *Tm, *S;
};
__host__ __device__ inline size_t feat_lds_doubles(int max_len, int ldh, bool tm_in_lds) {
    const int L = max_len, M2 = 2 * L, rho = 2 * L - 2;
    size_t n = (size_t)(L - 1) * 24 + L * 6 + M2 * 3 + (L - 1) * 18 + 3 * M2 + 16;
    n += (size_t)M2 * ldh;
    if (tm_in_lds) n += (size_t)rho * ldh;
    n += (size_t)(rho + 1) * (rho + 1);
    return n;
}

__global__ void feat_build_kernel(DevCfg cfg, const FilterMeta* meta, const double* x, const double* P,
                                  const int* n_feat_ptr, const unsigned char* types, const int* lens, const float* meas,
                                  int shard_rank, int shard_world,
                                  double* Hstack, int* nrows_out, int* acc_out, int* ndof_out, double* gamma_out,
                                  double* pfinv_out, double* tm_global) {
    extern __shared__ __align__(16) double lds[];
    const int tid = threadIdx.x, T = blockDim.x, f = blockIdx.x;
    const int n = meta->n_clones, c6 = 6 * n, ldh = cfg.ldh, ld = cfg.dmax;
    const int n_feat = *n_feat_ptr;
    if (f >= n_feat || (f % shard_world) != shard_rank) {
        if (tid == 0) { nrows_out[f] = 0; acc_out[f] = 0; ndof_out[f] = 0; gamma_out[f] = 0; }
        return;
    }
    const unsigned char type = types[f];
    const int L = lens[f];
    const float* mz = meas + (size_t)f * cfg.max_len * 2;
    // carve LDS
    const int ML = cfg.max_len, M2max = 2 * ML, rhomax = 2 * ML - 2;
    double* p = lds;
    double* pose = p; p += (size_t)(ML - 1) * 24;
    double* hrr = p;  p += ML * 6;
    double* hf = p;   p += M2max * 3;
    double* lr = p;   p += (ML - 1) * 18;
    double* vh = p;   p += 3 * M2max;
    double* misc = p; p += 16;
    double* Hx = p;   p += (size_t)M2max * ldh;
    double* Tm;
    if (tm_global) Tm = tm_global + (size_t)f * rhomax * ldh; else { Tm = p; p += (size_t)rhomax * ldh; }
    double* S = p;
    const int lane = tid & 63;
    const bool wave0 = tid < 64;
    const int nPh = L - 1;
    const double sig = cfg.sigma_im, sig2 = sig * sig;
    const m33 Ric = ldm33(cfg.Ric), Rci = ldm33(cfg.Rci);
    const d3 tic = ld3(cfg.tic), tci = ld3(cfg.tci);

    // ---- U1 relative-pose chain (Updater.cc:114-141): sequential, lane 0 writes
    if (wave0) {
        const double* rel = (type == '1') ? (x + 26 + 7 * n - 7 * nPh) : (x + 26);
        q4 qI = ldq(rel);
        d3 tI = scl3(-1.0, mv33(q2r(qI), ld3(rel + 4)));
        for (int i = 0; i < nPh; ++i) {
            if (i > 0) {
                q4 qi = ldq(rel + 7 * i);
                tI = mv33(q2r(qi), sub3(tI, ld3(rel + 7 * i + 4)));
                qI = qmul(qi, qI);
            }
            m33 RI = q2r(qI);
            m33 RcRaw = mul33(mul33(Rci, RI), Ric);
            q4 qC = r2q(RcRaw);
            m33 Rc = q2r(qC);
            d3 tC = add3(add3(mv33(mul33(Rci, RI), tic), mv33(Rci, tI)), tci);
            if (lane == 0) {
                double* o = pose + i * 24;
                for (int k = 0; k < 9; ++k) { o[k] = RI.m[k]; o[12 + k] = Rc.m[k]; }
                st3(o + 9, tI); st3(o + 21, tC);
            }
        }
    }
    __syncthreads();

    // ---- U2 inverse-depth LM triangulation (Updater.cc:143-269): lane i <-> observation i
    double phi = 0, psi = 0, rho = 0;
    bool valid = true;
    if (wave0) {
        const float fx0 = mz[0], fy0 = mz[1];
        phi = atan2((double)fy0, sqrt((double)fx0 * (double)fx0 + 1));
        psi = atan2((double)fx0, 1.0);
        if (fabs(phi) > .5 * 3.14 || fabs(psi) > .5 * 3.14) valid = false;
        const bool act = lane < L;
        float mx = 0, my = 0;
        m33 Rc = eye33(); d3 tc = mk3(0, 0, 0);
        if (act) { mx = mz[2 * lane]; my = mz[2 * lane + 1]; }
        if (act && lane > 0) { Rc = ldm33(pose + (lane - 1) * 24 + 12); tc = ld3(pose + (lane - 1) * 24 + 21); }
        const double ri = 1. / sig2;
        double lambda = 0.01, lastCost = INFINITY;
        if (valid) {
            for (int it = 0; it < 10; ++it) {
                const double sph = sin(phi), cph = cos(phi), sps = sin(psi), cps = cos(psi);
                const d3 ep = mk3(cph * sps, sph, cph * cps);
                const double J00 = -sph * sps, J01 = cph * cps, J10 = cph, J20 = -sph * cps, J21 = -cph * sps;
                double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0, g0 = 0, g1 = 0, g2 = 0, cost = 0;
                if (act) {
                    d3 h = (lane == 0) ? ep : add3(mv33(Rc, ep), scl3(rho, tc));
                    const double iz = 1 / h.z, iz2 = h.z * h.z;
                    const double Hp0[3] = {iz, 0, -h.x / iz2}, Hp1[3] = {0, iz, -h.y / iz2};
                    double HR0[3], HR1[3];
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        HR0[b] = Hp0[0] * Rc.m[b] + Hp0[1] * Rc.m[3 + b] + Hp0[2] * Rc.m[6 + b];
                        HR1[b] = Hp1[0] * Rc.m[b] + Hp1[1] * Rc.m[3 + b] + Hp1[2] * Rc.m[6 + b];
                    }
                    double H0[3], H1[3];
                    H0[0] = HR0[0] * J00 + HR0[1] * J10 + HR0[2] * J20;
                    H0[1] = HR0[0] * J01 + HR0[2] * J21;
                    H1[0] = HR1[0] * J00 + HR1[1] * J10 + HR1[2] * J20;
                    H1[1] = HR1[0] * J01 + HR1[2] * J21;
                    if (lane == 0) { H0[2] = 0; H1[2] = 0; }
                    else { H0[2] = Hp0[0] * tc.x + Hp0[2] * tc.z; H1[2] = Hp1[1] * tc.y + Hp1[2] * tc.z; }
                    const float px = (float)(h.x / h.z), py = (float)(h.y / h.z);  // cv::Point2f rounding (Updater.cc:197-202)
                    const double e0 = (double)(mx - px), e1 = (double)(my - py);
                    cost = (e0 * ri) * e0 + (e1 * ri) * e1;
                    c00 = (H0[0] * ri) * H0[0] + (H1[0] * ri) * H1[0];
                    c01 = (H0[0] * ri) * H0[1] + (H1[0] * ri) * H1[1];
                    c02 = (H0[0] * ri) * H0[2] + (H1[0] * ri) * H1[2];
                    c11 = (H0[1] * ri) * H0[1] + (H1[1] * ri) * H1[1];
                    c12 = (H0[1] * ri) * H0[2] + (H1[1] * ri) * H1[2];
                    c22 = (H0[2] * ri) * H0[2] + (H1[2] * ri) * H1[2];
                    g0 = (H0[0] * ri) * e0 + (H1[0] * ri) * e1;
                    g1 = (H0[1] * ri) * e0 + (H1[1] * ri) * e1;
                    g2 = (H0[2] * ri) * e0 + (H1[2] * ri) * e1;
                }
                cost = wave_sum(cost);
                c00 = wave_sum(c00); c01 = wave_sum(c01); c02 = wave_sum(c02);
                c11 = wave_sum(c11); c12 = wave_sum(c12); c22 = wave_sum(c22);
                g0 = wave_sum(g0); g1 = wave_sum(g1); g2 = wave_sum(g2);
                if (cost <= lastCost) {
                    // damped normal equations, SPD 3x3: Cholesky solve (reference: colPivHouseholderQr, Updater.cc:239)
                    const double a00 = c00 + lambda * c00, a11 = c11 + lambda * c11, a22 = c22 + lambda * c22;
                    const double l00 = sqrt(a00), l10 = c01 / l00, l20 = c02 / l00;
                    const double l11 = sqrt(a11 - l10 * l10), l21 = (c12 - l20 * l10) / l11;
                    const double l22 = sqrt(a22 - l20 * l20 - l21 * l21);
                    const double y0 = g0 / l00, y1 = (g1 - l10 * y0) / l11, y2 = (g2 - l20 * y0 - l21 * y1) / l22;
                    const double d2 = y2 / l22, d1 = (y1 - l21 * d2) / l11, d0 = (y0 - l10 * d1 - l20 * d2) / l00;
                    phi += d0; psi += d1; rho += d2;
                    if (fabs(lastCost - cost) < 1e-6 && d2 < 1e-6) break;
                    lambda *= .1; lastCost = cost;
                } else { lambda *= 10; lastCost = cost; }  // Updater.cc:257-262
            }
            if (fabs(phi) > .5 * 3.14 || fabs(psi) > .5 * 3.14 || isinf(rho) || rho < 0 || isnan(rho) || isnan(phi) || isnan(psi)) valid = false;
        }
        if (lane == 0) { misc[0] = phi; misc[1] = psi; misc[2] = rho; misc[3] = valid ? 1.0 : 0.0; }
    }
    __syncthreads();
    phi = misc[0]; psi = misc[1]; rho = misc[2]; valid = misc[3] != 0.0;
    if (tid == 0) { pfinv_out[3 * f] = phi; pfinv_out[3 * f + 1] = psi; pfinv_out[3 * f + 2] = rho; }
    if (!valid) {
        if (tid == 0) { nrows_out[f] = 0; acc_out[f] = 0; ndof_out[f] = 0; gamma_out[f] = 0; }
        return;
    }

    // ---- U3 residual + Jacobians (Updater.cc:271-368)
    const int Lu = (type == '2') ? (L + 1) / 2 : L;   // ceil(.5 L)
    const int M2 = 2 * Lu;
    const int nStartCol = (type == '1') ? 6 * (n - (Lu - 1)) : 0;
    const int cLo = nStartCol, cHi = nStartCol + 6 * (Lu - 1);   // non-zero column range of this feature
    const double sph = sin(phi), cph = cos(phi), sps = sin(psi), cps = cos(psi);
    const d3 ep = mk3(cph * sps, sph, cph * cps);
    const double J00 = -sph * sps, J01 = cph * cps, J10 = cph, J20 = -sph * cps, J21 = -cph * sps;
    for (int e = tid; e < M2 * ldh; e += T) Hx[e] = 0.0;
    __syncthreads();
    for (int i = tid; i < Lu; i += T) {
        m33 Rc = eye33(); d3 tc = mk3(0, 0, 0);
        if (i > 0) { Rc = ldm33(pose + (i - 1) * 24 + 12); tc = ld3(pose + (i - 1) * 24 + 21); }
        d3 h = (i == 0) ? ep : add3(mv33(Rc, ep), scl3(rho, tc));
        const double iz = 1 / h.z, iz2 = h.z * h.z;
        const double Hp0[3] = {iz, 0, -h.x / iz2}, Hp1[3] = {0, iz, -h.y / iz2};
        const float px = (float)(h.x / h.z), py = (float)(h.y / h.z);
        const float ex = mz[2 * i] - px, ey = mz[2 * i + 1] - py;  // float32 residual (Updater.cc:307-308,338-339)
        Hx[(size_t)(2 * i) * ldh + c6] = (double)ex;
        Hx[(size_t)(2 * i + 1) * ldh + c6] = (double)ey;
        double HR0[3], HR1[3];
        for (int b = 0; b < 3; ++b) {
            HR0[b] = Hp0[0] * Rc.m[b] + Hp0[1] * Rc.m[3 + b] + Hp0[2] * Rc.m[6 + b];
            HR1[b] = Hp1[0] * Rc.m[b] + Hp1[1] * Rc.m[3 + b] + Hp1[2] * Rc.m[6 + b];
        }
        double* h0 = hf + (2 * i) * 3; double* h1 = h0 + 3;
        h0[0] = HR0[0] * J00 + HR0[1] * J10 + HR0[2] * J20; h0[1] = HR0[0] * J01 + HR0[2] * J21;
        h1[0] = HR1[0] * J00 + HR1[1] * J10 + HR1[2] * J20; h1[1] = HR1[0] * J01 + HR1[2] * J21;
        if (i == 0) { h0[2] = 0; h1[2] = 0; }
        else { h0[2] = Hp0[0] * tc.x + Hp0[2] * tc.z; h1[2] = Hp1[1] * tc.y + Hp1[2] * tc.z; }
        if (i > 0) {  // Hproj * Rci * R  (Updater.cc:349)
            m33 RR = mul33(Rci, ldm33(pose + (i - 1) * 24));
            double* o = hrr + i * 6;
            for (int b = 0; b < 3; ++b) {
                o[b] = Hp0[0] * RR.m[b] + Hp0[1] * RR.m[3 + b] + Hp0[2] * RR.m[6 + b];
                o[3 + b] = Hp1[0] * RR.m[b] + Hp1[1] * RR.m[3 + b] + Hp1[2] * RR.m[6 + b];
            }
        }
    }
    // per-clone 3x6 right factors [ dpx*Rj^T | -rho*R(j-1)^T ]  (Updater.cc:341-362)
    for (int j = tid; j < Lu - 1; j += T) {
        m33 RjT = tr33(ldm33(pose + j * 24));
        d3 tj = ld3(pose + j * 24 + 9);
        m33 dpx = skew33(add3(add3(mv33(Ric, ep), scl3(rho, tic)), scl3(rho, mv33(RjT, tj))));
        m33 left = mul33(dpx, RjT);
        m33 right = (j == 0) ? scl33(-rho, eye33()) : scl33(-rho, tr33(ldm33(pose + (j - 1) * 24)));
        double* o = lr + j * 18;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { o[a * 6 + b] = left.m[3 * a + b]; o[a * 6 + 3 + b] = right.m[3 * a + b]; }
    }
    __syncthreads();
    for (int i = 1; i < Lu; ++i) {
        const double* hr = hrr + i * 6;
        for (int e = tid; e < i * 12; e += T) {
            const int j = e / 12, a = (e % 12) / 6, b = e % 6;
            const double* l = lr + j * 18;
            Hx[(size_t)(2 * i + a) * ldh + nStartCol + 6 * j + b] = hr[3 * a] * l[b] + hr[3 * a + 1] * l[6 + b] + hr[3 * a + 2] * l[12 + b];
        }
    }
    // ---- U4 left-nullspace projection (Updater.cc:370-402).  The reference sweeps Givens
    // rotations; any orthonormal basis of null(Hf^T) gives the same gate statistic and the
    // same information [A|b], so 3 Householder reflections are used:
    // lane <-> row of Hf for the reflectors, then thread <-> column of [Hx|r] to apply them.
    int N = 3;
    if (wave0) {
        double h0 = 0, h1 = 0, h2 = 0;
        if (lane < M2) { h0 = hf[lane * 3]; h1 = hf[lane * 3 + 1]; h2 = hf[lane * 3 + 2]; }
        if (sqrt(wave_sum(h2 * h2)) < 1e-4) N = 2;   // rank-deficient Hf (Updater.cc:374-378)
        double hc[3] = {h0, h1, h2};
        for (int k = 0; k < 3; ++k) {
            double v = 0, beta = 0;
            if (k < N) {
                const double xk = (lane >= k && lane < M2) ? hc[k] : 0.0;
                const double s = wave_sum(xk * xk);
                const double akk = __shfl(hc[k], k, 64);
                if (s > 0) {
                    const double alpha = (akk >= 0) ? -sqrt(s) : sqrt(s);
                    v = (lane == k) ? (akk - alpha) : xk;
                    const double vtv = s - akk * akk + (akk - alpha) * (akk - alpha);
                    beta = 2.0 / vtv;
                    for (int c = k + 1; c < 3; ++c) {
                        const double wdot = wave_sum(v * hc[c]);
                        hc[c] -= beta * wdot * v;
                    }
                }
            }
            if (lane < M2max) vh[k * M2max + lane] = v;
            if (lane == 0) misc[4 + k] = beta;
        }
        if (lane == 0) misc[7] = (double)N;
    }
    __syncthreads();
    N = (int)misc[7];
    {
        const int nact = (cHi - cLo) + 1;   // active columns + the residual column
        for (int e = tid; e < nact; e += T) {
            const int c = (e < cHi - cLo) ? (cLo + e) : c6;
            for (int k = 0; k < N; ++k) {
                const double beta = misc[4 + k];
                const double* v = vh + k * M2max;
                double wdot = 0;
                for (int i = k; i < M2; ++i) wdot += v[i] * Hx[(size_t)i * ldh + c];
                wdot *= beta;
                for (int i = k; i < M2; ++i) Hx[(size_t)i * ldh + c] -= wdot * v[i];
            }
        }
    }
    __syncthreads();
    // ---- U5 Mahalanobis gate (Updater.cc:404-422) on rows N..M2-1
    const int rr = M2 - N;             // nDOF
    const double* Hn = Hx + (size_t)N * ldh;
    const int wa = cHi - cLo;          // active width
    // Tm = Hn * Pcc restricted to the active clone range; Pcc[k][c] read as its mirror (coalesced)
    for (int e = tid; e < wa; e += T) {
        const int c = cLo + e;
        const double* pcol = P + (size_t)(24 + c) + (size_t)(24 + cLo) * ld;   // P[(24+c),(24+cLo+k)]
        for (int i0 = 0; i0 < rr; i0 += 20) {
            double acc[20];
#pragma unroll
            for (int ii = 0; ii < 20; ++ii) acc[ii] = 0;
            for (int k = 0; k < wa; ++k) {
                const double pv = pcol[(size_t)k * ld];
                const double* hk = Hn + (size_t)i0 * ldh + cLo + k;
#pragma unroll
                for (int ii = 0; ii < 20; ++ii) if (i0 + ii < rr) acc[ii] += hk[(size_t)ii * ldh] * pv;
            }
#pragma unroll
            for (int ii = 0; ii < 20; ++ii) if (i0 + ii < rr) Tm[(size_t)(i0 + ii) * ldh + c] = acc[ii];
        }
    }
    __syncthreads();
    const int lds_s = rhomax + 1;
    for (int e = tid; e < (rr + 1) * rr; e += T) {
        const int i = e / rr, j = e % rr;   // row i (i == rr: the residual row), col j <= i
        if (i < rr) {
            if (j > i) continue;
            double acc = 0;
            const double* ti = Tm + (size_t)i * ldh + cLo;
            const double* hj = Hn + (size_t)j * ldh + cLo;
            for (int c = 0; c < wa; ++c) acc += ti[c] * hj[c];
            // the lower triangle stands for the symmetrised matrix .5 (S + S^T) (Updater.cc:418)
            if (i == j) acc += sig2;
            S[i * lds_s + j] = acc;
        } else {
            S[rr * lds_s + j] = Hn[(size_t)j * ldh + c6];
        }
    }
    __syncthreads();
    // Cholesky of S with the residual appended as an extra row: its entries become y = L^-1 r,
    // gamma = |r^T S^-1 r| = y^T y.  (reference: colPivHouseholderQr().solve, Updater.cc:420)
    for (int k = 0; k < rr; ++k) {
        if (tid == 0) { double dkk = S[k * lds_s + k]; S[k * lds_s + k] = sqrt(dkk > 0 ? dkk : 1e-300); }
        __syncthreads();
        for (int i = k + 1 + tid; i <= rr; i += T) S[i * lds_s + k] /= S[k * lds_s + k];
        __syncthreads();
        for (int i = k + 1 + tid; i <= rr; i += T) {
            const double lik = S[i * lds_s + k];
            const int jmax = (i < rr) ? i : rr - 1;
            for (int j = k + 1; j <= jmax; ++j) S[i * lds_s + j] -= lik * S[j * lds_s + k];
        }
        __syncthreads();
    }
    double gam = 0;
    if (wave0) {
        double part = 0;
        for (int k = lane; k < rr; k += 64) { double y = S[rr * lds_s + k]; part += y * y; }
        gam = fabs(wave_sum(part));
        if (lane == 0) misc[8] = gam;
    }
    __syncthreads();
    gam = misc[8];
    const bool accept = gam < kChi2Dev[rr - 1];
    if (tid == 0) { acc_out[f] = accept ? 1 : 0; ndof_out[f] = rr; gamma_out[f] = gam; nrows_out[f] = accept ? rr : 0; }
    if (accept) {
        double* out = Hstack + (size_t)f * rhomax * ldh;
        for (int e = tid; e < rr * ldh; e += T) out[e] = Hn[e];
    }
}

// =============================================================== U7 compression, information form
// partial[g][p][q] = sum over the rows of feature group g of H[row][p] * H[row][q],
// q = 0..c6 (column c6 is the residual -> b).  grid = (groups, ceil(c6/16)), 256 threads.
#define GRAM_FG 4
__global__ __launch_bounds__(256) void gram_kernel(DevCfg cfg, const FilterMeta* meta, const double* Hstack, const int* nrows,
                                                   double* partial) {
    const int n = meta->n_clones, c6 = 6 * n, ldh = cfg.ldh, rhomax = cfg.rho_max;
    const int g = blockIdx.x, p0 = blockIdx.y * 16;
    if (p0 >= c6) return;
    const int f0 = g * GRAM_FG;
    const int ncol = c6 + 1;
    double* out = partial + (size_t)g * cfg.ldh * cfg.ldh;
    for (int e = threadIdx.x; e < 16 * ncol; e += 256) {
        const int p = p0 + e / ncol, q = e % ncol;
        if (p >= c6) continue;
        double acc = 0;
        for (int ff = 0; ff < GRAM_FG; ++ff) {
            const int f = f0 + ff;
            if (f >= cfg.Fu) break;
            const int nr = nrows[f];
            const double* H = Hstack + (size_t)f * rhomax * ldh;
            for (int r = 0; r < nr; ++r) acc += H[(size_t)r * ldh + p] * H[(size_t)r * ldh + q];
        }
        out[(size_t)p * ldh + q] = acc;
    }
}

// block = [A|b] (c6 x ldh row-major) + {n_good, n_rows}: the all-gather payload of the sharded updater
__global__ __launch_bounds__(256) void gram_reduce_kernel(DevCfg cfg, const FilterMeta* meta, const double* partial, int n_groups,
                                                          const int* nrows, double* block) {
    const int n = meta->n_clones, c6 = 6 * n, ldh = cfg.ldh;
    const int total = c6 * ldh;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int q = e % ldh;
        double acc = 0;
        if (q <= c6) for (int g = 0; g < n_groups; ++g) acc += partial[(size_t)g * ldh * ldh + e];
        block[e] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int good = 0, rows = 0;
        for (int f = 0; f < cfg.Fu; ++f) { if (nrows[f] > 0) { good++; rows += nrows[f]; } }
        block[(size_t)cfg.ldh * (cfg.ldh - 1)] = (double)good;
        block[(size_t)cfg.ldh * (cfg.ldh - 1) + 1] = (double)rows;
    }
}

// Sum `world` gathered blocks in rank order -> Ab (c6 x ldh), set meta counters, build
// the Gauss-Jordan tableau Aug (c6 x 2*ldh, row-major) at FIXED column offsets so that no
// pointer depends on n_clones:  cols [0,c6) = s2 I + A Pcc,  col ldh-1 = b,  cols [ldh, ldh+c6) = I.
// The product A*Pcc is added by gemm_f64_kernel afterwards.
__global__ __launch_bounds__(256) void block_sum_kernel(DevCfg cfg, FilterMeta* meta, const double* blocks, int world, size_t block_stride,
                                                        double* Ab, double* Aug) {
    const int n = meta->n_clones, c6 = 6 * n, ldh = cfg.ldh, lda = 2 * ldh;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    const int total = c6 * ldh;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        double acc = 0;
        for (int w = 0; w < world; ++w) acc += blocks[(size_t)w * block_stride + e];
        Ab[e] = acc;
        const int p = e / ldh, q = e % ldh;
        // every tableau entry of row p is (re)written here: stale values never survive a frame
        // (slot ldh-1 holds b and is written only by the q == c6 thread: no write race while n < nmax)
        if (q != ldh - 1) Aug[(size_t)p * lda + q] = (q < c6) ? ((p == q) ? s2 : 0.0) : 0.0;
        Aug[(size_t)p * lda + ldh + q] = (q < c6 && p == q) ? 1.0 : 0.0;
        if (q == c6) Aug[(size_t)p * lda + ldh - 1] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double good = 0, rows = 0;
        for (int w = 0; w < world; ++w) { good += blocks[(size_t)w * block_stride + (size_t)ldh * (ldh - 1)]; rows += blocks[(size_t)w * block_stride + (size_t)ldh * (ldh - 1) + 1]; }
        meta->n_good = (int)good; meta->n_rows = (int)rows;
        meta->updated = ((int)good > 2) ? 1 : 0;    // Updater.cc:460
    }
}

// =============================================================== FP64 MFMA GEMM
// Cout = alpha * A(MxK) * B(KxN) + beta * Cin, arbitrary element strides (handles
// transposes and the column-major P without copies).  One workgroup = 4 waves = a
// 32x32 output tile, each wave one 16x16 tile with v_mfma_f64_16x16x4_f64:
//   A operand lane l : A[i = l&15][k = l>>4]      B operand lane l : B[k = l>>4][j = l&15]
//   C/D      lane l : 4 values, row = (l>>4) + 4*r, col = l&15.
// dims[] (device): M,N,K are read from meta at run time so shapes follow n_clones.
typedef double d4 __attribute__((ext_vector_type(4)));
struct GemmArgs {
    const double* A; long sar, sac;
    const double* B; long sbr, sbc;
    const double* Cin; long scr, scc;
    double* Cout; long sor, soc;
    double alpha, beta;
    int mode;   // 0: M=d,N=c6,K=c6   1: M=d,N=d,K=c6   2: M=c6,N=c6,K=c6
    int need_update;  // 1: skip when meta->updated == 0
};
__global__ __launch_bounds__(256) void gemm_f64_kernel(const FilterMeta* meta, GemmArgs g) {
    if (g.need_update && !meta->updated) return;
    const int n = meta->n_clones, c6 = 6 * n, d = 24 + c6;
    const int M = (g.mode == 2) ? c6 : d, N = (g.mode == 1) ? d : c6, K = c6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = blockIdx.y * 32 + (wave >> 1) * 16, j0 = blockIdx.x * 32 + (wave & 1) * 16;
    if (i0 >= M || j0 >= N) return;
    const int li = lane & 15, lk = lane >> 4;
    const int ai = i0 + li, bj = j0 + li;
    const bool aok = ai < M, bok = bj < N;
    const double* ap = g.A + (long)ai * g.sar;
    const double* bp = g.B + (long)bj * g.sbc;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 4 * u + lk;
            a[u] = (aok && k < K) ? ap[(long)k * g.sac] : 0.0;
            b[u] = (bok && k < K) ? bp[(long)k * g.sbr] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    const int col = j0 + li;
    if (col < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + lk + 4 * r;
            if (row < M) {
                double v = g.alpha * acc[r];
                if (g.beta != 0.0) v += g.beta * g.Cin[(long)row * g.scr + (long)col * g.scc];
                g.Cout[(long)row * g.sor + (long)col * g.soc] = v;
            }
        }
    }
}

// =============================================================== Gauss-Jordan
// In-place reduction of the tableau [T | b | I] (c6 x 2*ldh, fixed offsets, see block_sum_kernel)
// to [I | T^-1 b | T^-1] with partial pivoting.  One workgroup (1024 threads); staged in LDS when it fits.
__global__ __launch_bounds__(1024) void gj_kernel(DevCfg cfg, FilterMeta* meta, double* AugG, int use_lds) {
    extern __shared__ __align__(16) double gl[];
    __shared__ double red_v[16];
    __shared__ int red_i[16];
    __shared__ double s_piv;
    __shared__ int s_prow;
    if (!meta->updated) return;
    const int n = meta->n_clones, c6 = 6 * n, lda = 2 * cfg.ldh, ncol = 2 * cfg.ldh;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wv = tid >> 6;
    double* M = AugG;
    int ldm = lda;
    if (use_lds) {
        ldm = ncol | 1;   // odd stride: conflict-free column walks
        for (int e = tid; e < c6 * ncol; e += T) { int r = e / ncol, c = e % ncol; gl[(size_t)r * ldm + c] = AugG[(size_t)r * lda + c]; }
        M = gl;
        __syncthreads();
    }
    for (int k = 0; k < c6; ++k) {
        // pivot search over rows k..c6-1 of column k
        double best = -1.0; int bi = k;
        for (int i = k + tid; i < c6; i += T) { double v = fabs(M[(size_t)i * ldm + k]); if (v > best) { best = v; bi = i; } }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            double ov = __shfl_xor(best, off, 64); int oi = __shfl_xor(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { red_v[wv] = best; red_i[wv] = bi; }
        __syncthreads();
        if (tid == 0) {
            double b = red_v[0]; int ib = red_i[0];
            for (int w = 1; w < (T >> 6); ++w) if (red_v[w] > b || (red_v[w] == b && red_i[w] < ib)) { b = red_v[w]; ib = red_i[w]; }
            s_prow = ib;
            if (!(b > 0)) meta->err |= 1;
        }
        __syncthreads();
        const int pr = s_prow;
        if (pr != k) for (int c = tid; c < ncol; c += T) { double t0 = M[(size_t)k * ldm + c]; M[(size_t)k * ldm + c] = M[(size_t)pr * ldm + c]; M[(size_t)pr * ldm + c] = t0; }
        __syncthreads();
        const double ipiv = 1.0 / M[(size_t)k * ldm + k];
        // eliminate column k from every other row; columns <= k of the left block are already final
        const int nc = ncol - (k + 1);
        for (int e = tid; e < c6 * nc; e += T) {
            const int i = e / nc, c = k + 1 + e % nc;
            if (i == k) continue;
            const double fct = M[(size_t)i * ldm + k] * ipiv;
            M[(size_t)i * ldm + c] -= fct * M[(size_t)k * ldm + c];
        }
        __syncthreads();
        // scale the pivot row, clear column k
        for (int c = k + 1 + tid; c < ncol; c += T) M[(size_t)k * ldm + c] *= ipiv;
        for (int i = tid; i < c6; i += T) M[(size_t)i * ldm + k] = (i == k) ? 1.0 : 0.0;
        __syncthreads();
    }
    if (use_lds) for (int e = tid; e < c6 * ncol; e += T) { int r = e / ncol, c = e % ncol; AugG[(size_t)r * lda + c] = gl[(size_t)r * ldm + c]; }
}

// =============================================================== U9 state injection
// dx = Pc * y  (y = W b = last tableau column), then Updater.cc:546-613.  Writes x_out.
__global__ __launch_bounds__(256) void inject_kernel(DevCfg cfg, const FilterMeta* meta, const double* x, const double* P,
                                                     const double* Aug, double* x_out) {
    __shared__ double dx[24 + 6 * RVIO_MAX_LEN];
    const int n = meta->n_clones, c6 = 6 * n, d = 24 + c6, ld = cfg.dmax, lda = 2 * cfg.ldh, xd = 26 + 7 * n;
    const int tid = threadIdx.x;
    if (!meta->updated) { for (int i = tid; i < xd; i += 256) x_out[i] = x[i]; return; }
    for (int i = tid; i < d; i += 256) {
        double acc = 0;
        for (int k = 0; k < c6; ++k) acc += P[(size_t)i + (size_t)(24 + k) * ld] * Aug[(size_t)k * lda + cfg.ldh - 1];
        dx[i] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        stq(x_out, qmul(small_q(dx[0], dx[1], dx[2]), ldq(x)));
        for (int i = 0; i < 6; ++i) x_out[4 + i] = dx[3 + i] + x[4 + i];
        st3(x_out + 7, unit3(ld3(x_out + 7)));
        stq(x_out + 10, qmul(small_q(dx[9], dx[10], dx[11]), ldq(x + 10)));
        for (int i = 0; i < 12; ++i) x_out[14 + i] = dx[12 + i] + x[14 + i];
    }
    for (int p = tid - 1; p >= 0 && p < n; p += 255) {
        stq(x_out + 26 + 7 * p, qmul(small_q(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), ldq(x + 26 + 7 * p)));
        for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
    }
}

// P_out = .5 (Pt + Pt^T) if updated else P_in   (Updater.cc:619 / :621-627)
__global__ __launch_bounds__(256) void symm_out_kernel(DevCfg cfg, const FilterMeta* meta, const double* Pt, const double* Pin, double* Pout) {
    const int n = meta->n_clones, d = 24 + 6 * n, ld = cfg.dmax;
    const bool upd = meta->updated;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < d * d; e += gridDim.x * 256) {
        const int i = e % d, j = e / d;
        Pout[(size_t)i + (size_t)j * ld] = upd ? .5 * (Pt[(size_t)i + (size_t)j * ld] + Pt[(size_t)j + (size_t)i * ld]) : Pin[(size_t)i + (size_t)j * ld];
    }
}

// =============================================================== S1 augmentation / slide
// System.cc:279-323.  J P J^T with J = [I; rows 9..14] is a pure gather: out[a][b] = P[src(a)][src(b)].
__device__ __forceinline__ int aug_src(int a, int n, int nmax, int do_aug) {
    if (a < 24 || !do_aug) return a;
    const int cb = (a - 24) / 6, off = (a - 24) % 6;
    if (n < nmax) return (cb < n) ? a : 9 + off;
    return (cb < nmax - 1) ? a + 6 : 9 + off;
}
__global__ __launch_bounds__(256) void augment_kernel(DevCfg cfg, const FilterMeta* meta, const double* x, const double* P,
                                                      double* x_out, double* P_out, int do_aug) {
    const int n = meta->n_clones, nmax = cfg.nmax, ld = cfg.dmax;
    const int n2 = do_aug ? ((n < nmax) ? n + 1 : nmax) : n;
    const int d2 = 24 + 6 * n2;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < d2 * d2; e += gridDim.x * 256) {
        const int i = e % d2, j = e / d2;
        P_out[(size_t)i + (size_t)j * ld] = P[(size_t)aug_src(i, n, nmax, do_aug) + (size_t)aug_src(j, n, nmax, do_aug) * ld];
    }
    if (blockIdx.x == 0) {
        const int xd2 = 26 + 7 * n2;
        for (int i = threadIdx.x; i < xd2; i += 256) {
            int src = i;
            if (do_aug && i >= 26) {
                const int cb = (i - 26) / 7, off = (i - 26) % 7;
                if (n < nmax) src = (cb < n) ? i : 10 + off;
                else src = (cb < nmax - 1) ? i + 7 : 10 + off;
            }
            x_out[i] = x[src];
        }
    }
}
// meta update after augmentation (separate tiny kernel keeps augment_kernel race-free)
__global__ void meta_after_augment_kernel(DevCfg cfg, FilterMeta* meta, int do_aug) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (do_aug && meta->n_clones < cfg.nmax) meta->n_clones += 1;
    }
}

// =============================================================== S2 composition
// System.cc:325-365.  Reads (x_in,P_in) written by augment_kernel, writes (x_out,P_out).
// block 0: the 24x24 corner Vk P11 Vk^T (symmetrised); other blocks: rows<24 x clone columns.
__global__ __launch_bounds__(256) void compose_kernel(DevCfg cfg, const FilterMeta* meta, const double* x, const double* P,
                                                      double* x_out, double* P_out, double* pose_out) {
    __shared__ double Vk[24][25];
    __shared__ double P11[24][25];
    __shared__ double Tm[24][25];
    const int n = meta->n_clones, d = 24 + 6 * n, ld = cfg.dmax, xd = 26 + 7 * n;
    const int tid = threadIdx.x;
    const q4 qG = ldq(x), qk = ldq(x + 10);
    const d3 pG = ld3(x + 4), pk = ld3(x + 14);
    const m33 RG = q2r(qG), Rk = q2r(qk);
    const d3 gk = unit3(mv33(Rk, ld3(x + 7)));
    const q4 qkG = qmul(qk, qG);
    const d3 pkG = mv33(Rk, sub3(pG, pk));
    for (int e = tid; e < 576; e += 256) Vk[e / 24][e % 24] = 0.0;
    __syncthreads();
    if (tid < 9) {
        const int i = tid / 3, j = tid % 3;
        const m33 spx = skew33(pkG), sgx = skew33(gk);
        Vk[i][j] = Rk.m[3 * i + j];           Vk[i][9 + j] = (i == j) ? 1.0 : 0.0;
        Vk[3 + i][3 + j] = Rk.m[3 * i + j];   Vk[3 + i][9 + j] = spx.m[3 * i + j];   Vk[3 + i][12 + j] = -Rk.m[3 * i + j];
        Vk[6 + i][6 + j] = Rk.m[3 * i + j];   Vk[6 + i][9 + j] = sgx.m[3 * i + j];
        Vk[15 + tid][15 + tid] = 1.0;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int e = tid; e < 576; e += 256) P11[e % 24][e / 24] = P[(size_t)(e % 24) + (size_t)(e / 24) * ld];
        __syncthreads();
        for (int e = tid; e < 576; e += 256) { int i = e / 24, j = e % 24; double a = 0; for (int k = 0; k < 24; ++k) a += Vk[i][k] * P11[k][j]; Tm[i][j] = a; }
        __syncthreads();
        for (int e = tid; e < 576; e += 256) { int i = e / 24, j = e % 24; double a = 0; for (int k = 0; k < 24; ++k) a += Tm[i][k] * Vk[j][k]; P11[i][j] = a; }
        __syncthreads();
        for (int e = tid; e < 576; e += 256) { int i = e / 24, j = e % 24; P_out[(size_t)i + (size_t)j * ld] = .5 * (P11[i][j] + P11[j][i]); }
        // state (System.cc:360-365) + pose line (System.cc:371-374)
        for (int i = tid; i < xd; i += 256) {
            double v = x[i];
            if (i < 4) v = (&qkG.x)[i];
            else if (i < 7) v = (&pkG.x)[i - 4];
            else if (i < 10) v = (&gk.x)[i - 7];
            else if (i < 13) v = 0.0;
            else if (i == 13) v = 1.0;
            else if (i < 17) v = 0.0;
            x_out[i] = v;
        }
        if (tid == 0) {
            const d3 pGk = mv33(tr33(RG), sub3(pk, pG));
            st3(pose_out, pGk); stq(pose_out + 3, qkG);
        }
    } else {
        // columns c >= 24: out[0:24, c] = Vk * P[0:24, c]; mirror; lower-right block copied
        const int c6 = 6 * n;
        for (int c = (blockIdx.x - 1) * 256 + tid; c < c6; c += (gridDim.x - 1) * 256) {
            double col[24];
            const double* pc = P + (size_t)(24 + c) * ld;
            for (int k = 0; k < 24; ++k) col[k] = pc[k];
            for (int i = 0; i < 24; ++i) {
                double a = 0;
                for (int k = 0; k < 24; ++k) a += Vk[i][k] * col[k];
                P_out[(size_t)i + (size_t)(24 + c) * ld] = a;
                P_out[(size_t)(24 + c) + (size_t)i * ld] = a;
            }
            for (int r = 24; r < d; ++r) P_out[(size_t)r + (size_t)(24 + c) * ld] = pc[r];
        }
    }
}
