// filter_kernels.hip — hand-written gfx950 kernels for the filter half of the
// R-VIO hot path.
//
//   propagate_kernel     PreIntegrator::propagate          PreIntegrator.cc:51-194
//   feat_build_kernel    Updater::update per-feature loop  Updater.cc:109-455
//   gram_kernel / gram_reduce_kernel / block_sum_kernel
//                        measurement compression (Updater.cc:469-536) in information
//                        form [A|b] = Hw^T [Hw | r]   
//   gemm_f64_kernel      FP64-MFMA (v_mfma_f64_16x16x4_f64) tiled GEMM:  T = s2 I + A Pcc
//   solve_kernel         W = T^-1, y = W b by Gauss-Jordan (partial pivoting, one barrier
//                        per column), then dx = Pc y and state injection  (Updater.cc:540-613)
//   ug_kernel            U = Pc W, G = U A, P1 = P - G Pc^T   (FP64 MFMA, one 16-row strip / WG)
//   final_kernel         P+ = sym( P1 - P1c G^T + s2 G U^T )   (Joseph form, Updater.cc:615-619)
//   augcomp_kernel       augmentation/slide + composition, fused  (System.cc:279-365)
//
// Design rules learnt from the first profile (profiles/r01_a): every kernel front-loads its
// global reads in one batch (a dependent global load after a kernel boundary costs 1-2 us),
// n_clones is a kernel argument (data-independent, mirrored on the host), wave reductions use
// DPP, serial chains are hoisted out of barrier-separated loops.
#include "rvio_dev.h"
#include "../../include/rvio_hip.h"

__device__ const double kChi2Dev[500] = {
#include "chi2_table.inc"
};

typedef double d4 __attribute__((ext_vector_type(4)));

// =============================================================== P1 propagate
// One workgroup, 256 threads.
//  phase A  lane s <-> IMU sample s: everything that depends on the sample alone (trig, dR, f1..f4)
//  phase B  the short serial chain Rk <- dR Rk, dp, dv, pk, vk, gk (reference order)
//  phase C  Phi rows 9..17 (the only non-identity rows of Phi = I + dt F, PreIntegrator.cc:123-132)
//           for all samples at once
//  phase D  per sample: rows 9..17 of (Phi P), Psi <- Phi Psi, columns 9..17 of (Phi P) Phi^T + Q
#define PROP_CH 16
struct PropSample {      // per-sample scratch in LDS
    double dR[9], up[3], uv[3], w[3], dt, Dt, small;
    double Rk[9], vk[3], gk[3];   // PRE-step values used by F (PreIntegrator.cc:123-131)
};

__global__ __launch_bounds__(256) void propagate_kernel(DevCfg cfg, FilterMeta* meta, int n, double* x, double* P,
                                                        const rvio_imu* imu, int m) {
    __shared__ double Pl[24][25];
    __shared__ double Psi[24][25];
    __shared__ double Phi9[PROP_CH][9][25];
    __shared__ double vxs[PROP_CH][9];
    __shared__ PropSample sm[PROP_CH];
    __shared__ double xs[26];
    const int tid = threadIdx.x;
    const int ld = cfg.dmax;
    if (tid == 0) { meta->n_good = 0; meta->n_rows = 0; meta->updated = 0; }
    for (int e = tid; e < 576; e += 256) {
        int i = e % 24, j = e / 24;
        Pl[i][j] = P[i + (size_t)j * ld];
        Psi[i][j] = (i == j) ? 1.0 : 0.0;
    }
    if (tid < 26) xs[tid] = x[tid];
    __syncthreads();
    const d3 bg = ld3(xs + 20), ba = ld3(xs + 23);
    const d3 gR = ld3(xs + 7), vR = ld3(xs + 17);
    m33 Rk = q2r(ldq(xs + 10)), RkT = tr33(Rk);
    d3 pk = ld3(xs + 14), vk = vR, gk = gR;
    d3 dp = mk3(0, 0, 0), dv = mk3(0, 0, 0);
    const m33 I = eye33();
    const double nG = cfg.gravity;
    double Dt = 0;
    const int r9 = tid / 24, c9 = tid % 24;  // valid for tid < 216
    DBG_T(0);
    for (int s0 = 0; s0 < m; s0 += PROP_CH) {
        const int mc = (m - s0 < PROP_CH) ? (m - s0) : PROP_CH;
        // ---- phase A
        DBG_T(1);
        if (tid < mc) {
            const rvio_imu u = imu[s0 + tid];
            const d3 w = sub3(mk3(u.w[0], u.w[1], u.w[2]), bg), a = sub3(mk3(u.a[0], u.a[1], u.a[2]), ba);
            const double dt = u.dt, w1 = nrm3(w);
            const bool small = w1 < cfg.small_angle;
            const double wdt = w1 * dt, wdt2 = wdt * wdt;
            const double cw = cos(wdt), sw = sin(wdt);
            const m33 wx = skew33(w), wx2 = mul33(wx, wx);
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
            PropSample& q = sm[tid];
            for (int k = 0; k < 9; ++k) q.dR[k] = dR.m[k];
            st3(q.up, mv33(add33(add33(scl33(.5 * dt * dt, I), scl33(f1, wx)), scl33(f2, wx2)), a));
            st3(q.uv, mv33(add33(add33(scl33(dt, I), scl33(f3, wx)), scl33(f4, wx2)), a));
            st3(q.w, w); q.dt = dt;
        }
        __syncthreads();
        DBG_T(2);
        // ---- phase B (every thread runs the same short chain: no broadcast needed)
        for (int s = 0; s < mc; ++s) {
            PropSample& q = sm[s];
            if (tid == 0) { for (int k = 0; k < 9; ++k) q.Rk[k] = Rk.m[k]; st3(q.vk, vk); st3(q.gk, gk); }
            const double dt = q.dt;
            Dt += dt;
            Rk = mul33(ldm33(q.dR), Rk); RkT = tr33(Rk);
            dp = add3(dp, scl3(dt, dv));
            dp = add3(dp, mv33(RkT, ld3(q.up)));
            dv = add3(dv, mv33(RkT, ld3(q.uv)));
            pk = add3(sub3(scl3(Dt, vR), scl3(.5 * nG * Dt * Dt, gR)), dp);
            vk = mv33(Rk, add3(sub3(vR, scl3(nG * Dt, gR)), dv));
            gk = unit3(mv33(Rk, gR));
        }
        __syncthreads();
        DBG_T(3);
        // ---- phase C: Phi9[s][r][c] for all samples of the chunk
        for (int e = tid; e < mc * 216; e += 256) {
            const int s = e / 216, rc = e % 216, r = rc / 24, c = rc % 24;
            const PropSample& q = sm[s];
            const int br = r / 3, i = r % 3, bc = c / 3, j = c % 3;
            const double id = (i == j) ? 1.0 : 0.0, dt = q.dt;
            double v = 0.0;
            if (br == 0) {            // theta_k rows
                if (bc == 3) v = id - dt * skew33(ld3(q.w)).m[3 * i + j];
                else if (bc == 6) v = -dt * id;
            } else if (br == 1) {     // p_k rows:  -Rk^T [v]x | I | Rk^T
                if (bc == 3) {
                    const m33 vx = skew33(ld3(q.vk));
                    v = -dt * (q.Rk[i] * vx.m[j] + q.Rk[3 + i] * vx.m[3 + j] + q.Rk[6 + i] * vx.m[6 + j]);
                } else if (bc == 4) v = id;
                else if (bc == 5) v = dt * q.Rk[3 * j + i];
            } else {                  // v rows
                if (bc == 2) v = -dt * nG * q.Rk[3 * i + j];
                else if (bc == 3) v = -dt * nG * skew33(ld3(q.gk)).m[3 * i + j];
                else if (bc == 5) v = id - dt * skew33(ld3(q.w)).m[3 * i + j];
                else if (bc == 6) v = -dt * skew33(ld3(q.vk)).m[3 * i + j];
                else if (bc == 7) v = -dt * id;
            }
            Phi9[s][r][c] = v;
            if (rc < 9) vxs[s][rc] = skew33(ld3(q.vk)).m[rc];
        }
        __syncthreads();
        DBG_T(4);
        // ---- phase D
        for (int s = 0; s < mc; ++s) {
            const double dt = sm[s].dt;
            double accP = 0, accS = 0;
            if (tid < 216) {
#pragma unroll 8
                for (int k = 0; k < 24; ++k) { const double f = Phi9[s][r9][k]; accP += f * Pl[k][c9]; accS += f * Psi[k][c9]; }
            }
            __syncthreads();
            if (tid < 216) { Pl[9 + r9][c9] = accP; Psi[9 + r9][c9] = accS; }
            __syncthreads();
            double accC = 0;
            if (tid < 216) {
#pragma unroll 8
                for (int k = 0; k < 24; ++k) accC += Pl[c9][k] * Phi9[s][r9][k];
                // Q = dt G Sigma G^T (PreIntegrator.cc:135-140), non-zero blocks only
                const int i = c9, j = 9 + r9;
                const int bi = i / 3, ii = i % 3, bj = j / 3, jj = j % 3;
                const double* vx = vxs[s];
                if (bi == 3 && bj == 3) accC += (ii == jj) ? dt * cfg.sg2 : 0.0;
                else if (bi == 3 && bj == 5) accC += dt * cfg.sg2 * vx[3 * jj + ii];
                else if (bi == 5 && bj == 3) accC += dt * cfg.sg2 * vx[3 * ii + jj];
                else if (bi == 5 && bj == 5) {
                    double q = ((dt * vx[3 * ii]) * cfg.sg2) * vx[3 * jj] + ((dt * vx[3 * ii + 1]) * cfg.sg2) * vx[3 * jj + 1] +
                               ((dt * vx[3 * ii + 2]) * cfg.sg2) * vx[3 * jj + 2];
                    if (ii == jj) q += dt * cfg.sa2;
                    accC += q;
                }
            }
            __syncthreads();
            if (tid < 216) Pl[c9][9 + r9] = accC;
            if (tid >= 216 && tid < 219) Pl[18 + tid - 216][18 + tid - 216] += dt * cfg.swg2;
            if (tid >= 219 && tid < 222) Pl[21 + tid - 219][21 + tid - 219] += dt * cfg.swa2;
            __syncthreads();
        }
    }
    DBG_T(5);
    if (tid == 0) {
        stq(x + 10, r2q(Rk));
        st3(x + 14, pk);
        st3(x + 17, vk);
    }
    DBG_T(6);
    // P11 back (symmetrised, PreIntegrator.cc:192); P22 is untouched and already symmetric
    for (int e = tid; e < 576; e += 256) {
        int i = e % 24, j = e / 24;
        P[i + (size_t)j * ld] = .5 * (Pl[i][j] + Pl[j][i]);
    }
    DBG_T(7);
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
    DBG_T(8);
}

// =============================================================== U1..U5 per feature
// One workgroup per feature slot.  Dynamic LDS (doubles):
//   xcl[7*nmax] pose[(L-1)*24] hrr[L*6] hf[2L*3] lr[(L-1)*18] vh[3*2L] misc[16]
//   Hx[2L][ldh]  ([Hx | r], row-major)   Tm[rho][ldh]   S[(rho+1)][rho+1]
__host__ __device__ inline size_t feat_lds_doubles(int max_len, int ldh, bool tm_in_lds) {
    const int L = max_len, M2 = 2 * L, rho = 2 * L - 2;
    size_t n = (size_t)7 * (L - 1) + (size_t)(L - 1) * 24 + L * 6 + M2 * 3 + (L - 1) * 18 + 3 * M2 + 16;
    n += (size_t)M2 * ldh;
    if (tm_in_lds) n += (size_t)rho * ldh;
    n += (size_t)(rho + 1) * (rho + 1);
    return n;
}

__global__ void feat_build_kernel(DevCfg cfg, int n, const double* x, const double* P,
                                  const int* n_feat_ptr, const unsigned char* types, const int* lens, const float* meas,
                                  int shard_rank, int shard_world,
                                  double* Hstack, int* nrows_out, int* acc_out, int* ndof_out, double* gamma_out,
                                  double* pfinv_out, double* tm_global) {
    extern __shared__ __align__(16) double lds[];
    const int tid = threadIdx.x, T = blockDim.x, f = blockIdx.x;
    const int c6 = 6 * n, ldh = cfg.ldh, ld = cfg.dmax;
    // carve LDS
    const int ML = cfg.max_len, M2max = 2 * ML, rhomax = 2 * ML - 2;
    double* p = lds;
    double* xcl = p;  p += 7 * (ML - 1);
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
    // ---- one batch of global reads: feature header, its observations, the clone poses
    const int n_feat = *n_feat_ptr;
    const unsigned char type = types[f];
    const int L = lens[f];
    const float* mz = meas + (size_t)f * cfg.max_len * 2;
    float mxv = 0, myv = 0;
    const int lane = tid & 63;
    if (lane < ML) { mxv = mz[2 * lane]; myv = mz[2 * lane + 1]; }
    for (int e = tid; e < 7 * n; e += T) xcl[e] = x[26 + e];
    if (f >= n_feat || (f % shard_world) != shard_rank) {
        if (tid == 0) { nrows_out[f] = 0; acc_out[f] = 0; ndof_out[f] = 0; gamma_out[f] = 0; }
        return;
    }
    const bool wave0 = tid < 64;
    const int nPh = L - 1;
    const double sig = cfg.sigma_im, sig2 = sig * sig;
    const m33 Ric = ldm33(cfg.Ric), Rci = ldm33(cfg.Rci);
    const d3 tic = ld3(cfg.tic), tci = ld3(cfg.tci);
    DBG_T(30);
    __syncthreads();
    DBG_T(31);

    // ---- U1 relative-pose chain (Updater.cc:114-141).  R(q_i) for every clone in parallel (lane <-> clone), the chain
    // R_I(i) = R(q_i) R_I(i-1), t_I(i) = R(q_i) (t_I(i-1) - p_i) as a short serial product of 3x3 matrices, then the
    // camera-frame poses in parallel again.  The reference carries the chain as normalised quaternions and passes
    // R_c through RotToQuat/QuatToRot; both are the same rotations up to O(1e-16).
    if (wave0) {
        const double* rel = (type == '1') ? (xcl + 7 * n - 7 * nPh) : xcl;
        if (lane < nPh) {
            const m33 Rl = q2r(ldq(rel + 7 * lane));
            double* o = pose + lane * 24 + 12;          // park R(q_i) in the Rc slot until the chain has consumed it
#pragma unroll
            for (int k = 0; k < 9; ++k) o[k] = Rl.m[k];
        }
        __builtin_amdgcn_wave_barrier();
        m33 RI = ldm33(pose + 12);
        d3 tI = scl3(-1.0, mv33(RI, ld3(rel + 4)));
        for (int i = 0; i < nPh; ++i) {
            if (i > 0) {
                const m33 Ri = ldm33(pose + i * 24 + 12);
                tI = mv33(Ri, sub3(tI, ld3(rel + 7 * i + 4)));
                RI = mul33(Ri, RI);
            }
            if (lane == 0) {
                double* o = pose + i * 24;
#pragma unroll
                for (int k = 0; k < 9; ++k) o[k] = RI.m[k];
                st3(o + 9, tI);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < nPh) {
            double* o = pose + lane * 24;
            const m33 RIl = ldm33(o);
            const d3 tIl = ld3(o + 9);
            const m33 RciRI = mul33(Rci, RIl);
            const m33 Rc = mul33(RciRI, Ric);
            const d3 tC = add3(add3(mv33(RciRI, tic), mv33(Rci, tIl)), tci);
#pragma unroll
            for (int k = 0; k < 9; ++k) o[12 + k] = Rc.m[k];
            st3(o + 21, tC);
        }
    }
    __syncthreads();
    DBG_T(32);

    // ---- U2 inverse-depth LM triangulation (Updater.cc:143-269): lane i <-> observation i
    double phi = 0, psi = 0, rho = 0;
    bool valid = true;
    const float fx0 = __shfl(mxv, 0, 64), fy0 = __shfl(myv, 0, 64);
    if (wave0) {
        phi = atan2((double)fy0, sqrt((double)fx0 * (double)fx0 + 1));
        psi = atan2((double)fx0, 1.0);
        if (fabs(phi) > .5 * 3.14 || fabs(psi) > .5 * 3.14) valid = false;
        const bool act = lane < L;
        const float mx = mxv, my = myv;
        m33 Rc = eye33(); d3 tc = mk3(0, 0, 0);
        if (act && lane > 0) { Rc = ldm33(pose + (lane - 1) * 24 + 12); tc = ld3(pose + (lane - 1) * 24 + 21); }
        const double ri = 1. / sig2;
        double lambda = 0.01, lastCost = INFINITY;
        if (valid) {
            for (int it = 0; it < 10; ++it) {
                double sph, cph, sps, cps;
                sincos(phi, &sph, &cph); sincos(psi, &sps, &cps);
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
                    // L D L^T (square-root free): 3 reciprocals instead of 3 sqrt + 9 divisions
                    const double a00 = c00 + lambda * c00, a11 = c11 + lambda * c11, a22 = c22 + lambda * c22;
                    const double i0 = 1.0 / a00, l10 = c01 * i0, l20 = c02 * i0;
                    const double dd1 = a11 - l10 * c01, i1 = 1.0 / dd1, l21 = (c12 - l20 * c01) * i1;
                    const double dd2 = a22 - l20 * c02 - l21 * (c12 - l20 * c01), i2 = 1.0 / dd2;
                    const double z0 = g0, z1 = g1 - l10 * z0, z2 = g2 - l20 * z0 - l21 * z1;
                    const double d2 = z2 * i2, d1 = z1 * i1 - l21 * d2, d0 = z0 * i0 - l10 * d1 - l20 * d2;
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
    DBG_T(33);
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
    double sph, cph, sps, cps;
    sincos(phi, &sph, &cph); sincos(psi, &sps, &cps);
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
        // T >= 64 > Lu: the loop body runs once with i == tid == lane, so (mxv, myv) is observation i
        const float ex = mxv - px, ey = myv - py;  // float32 residual (Updater.cc:307-308,338-339)
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
    DBG_T(34);
    {   // all (i, j<i) 2x6 blocks, flattened over the workgroup
        const int nitems = (Lu * (Lu - 1) / 2) * 12;
        for (int e = tid; e < nitems; e += T) {
            const int pr = e / 12, ab = e % 12, a = ab / 6, b = ab % 6;
            int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)pr)) * 0.5f);
            while (i * (i - 1) / 2 > pr) --i;
            while ((i + 1) * i / 2 <= pr) ++i;
            const int j = pr - i * (i - 1) / 2;
            const double* hr = hrr + i * 6;
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
        double vv[3] = {0, 0, 0}, bb[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double v = 0, beta = 0;
            if (k < N) {
                const double xk = (lane >= k && lane < M2) ? hc[k] : 0.0;
                const double s = wave_sum(xk * xk);
                const double akk = readlane_f64(hc[k], k);
                if (s > 0) {
                    const double alpha = (akk >= 0) ? -sqrt(s) : sqrt(s);
                    v = (lane == k) ? (akk - alpha) : xk;
                    const double vtv = s - akk * akk + (akk - alpha) * (akk - alpha);
                    beta = 2.0 / vtv;
#pragma unroll
                    for (int c = k + 1; c < 3; ++c) {
                        const double wdot = wave_sum(v * hc[c]);
                        hc[c] -= beta * wdot * v;
                    }
                }
            }
            if (lane < M2max) vh[k * M2max + lane] = v;
            vv[k] = v; bb[k] = beta;
        }
        // compact WY:  H0 H1 H2 = I - V T V^T  (T upper triangular), so the sweep H2 H1 H0 X = X - V T^T (V^T X)
        const double d01 = wave_sum(vv[0] * vv[1]), d02 = wave_sum(vv[0] * vv[2]), d12 = wave_sum(vv[1] * vv[2]);
        if (lane == 0) {
            const double T00 = bb[0], T11 = bb[1], T22 = bb[2];
            const double T01 = -bb[1] * T00 * d01;
            const double T02 = -bb[2] * (T00 * d02 + T01 * d12), T12 = -bb[2] * T11 * d12;
            misc[7] = (double)N;
            misc[9] = T00; misc[10] = T01; misc[11] = T02; misc[12] = T11; misc[13] = T12; misc[14] = T22;
        }
    }
    __syncthreads();
    DBG_T(35);
    N = (int)misc[7];
    {
        const int nact = (cHi - cLo) + 1;   // active columns + the residual column
        const double T00 = misc[9], T01 = misc[10], T02 = misc[11], T11 = misc[12], T12 = misc[13], T22 = misc[14];
        const double *v0 = vh, *v1 = vh + M2max, *v2 = vh + 2 * M2max;
        for (int e = tid; e < nact; e += T) {
            const int c = (e < cHi - cLo) ? (cLo + e) : c6;
            double w0 = 0, w1 = 0, w2 = 0;
            for (int i = 0; i < M2; ++i) { const double hv = Hx[(size_t)i * ldh + c]; w0 += v0[i] * hv; w1 += v1[i] * hv; w2 += v2[i] * hv; }
            const double u0 = T00 * w0, u1 = T01 * w0 + T11 * w1, u2 = T02 * w0 + T12 * w1 + T22 * w2;
            for (int i = 0; i < M2; ++i) Hx[(size_t)i * ldh + c] -= v0[i] * u0 + v1[i] * u1 + v2[i] * u2;
        }
    }
    __syncthreads();
    DBG_T(36);
    // ---- U5 Mahalanobis gate (Updater.cc:404-422) on rows N..M2-1
    const int rr = M2 - N;             // nDOF
    const double* Hn = Hx + (size_t)N * ldh;
    const int wa = cHi - cLo;          // active width
    // Tm = Hn * Pcc restricted to the active clone range, on the FP64 matrix cores (v_mfma_f64_16x16x4_f64):
    //   A[i][k] = Hn[i][cLo+k] (LDS),  B[k][j] = Pcc[cLo+k][cLo+j] read as its mirror P[24+cLo+j, 24+cLo+k] (coalesced)
    const int nwv = T >> 6, wvid = tid >> 6, li = lane & 15, lk = lane >> 4;
    {
        const int nit = (rr + 15) / 16, njt = (wa + 15) / 16;
        for (int t = wvid; t < nit * njt; t += nwv) {
            const int it = t / njt, jt = t % njt;
            const int ai = it * 16 + li, bj = jt * 16 + li;
            const bool aok = ai < rr, bok = bj < wa;
            const double* ap = Hn + (size_t)ai * ldh + cLo;
            const double* bp = P + (size_t)(24 + cLo + bj) + (size_t)(24 + cLo) * ld;
            d4 acc = {0, 0, 0, 0};
            for (int k0 = 0; k0 < wa; k0 += 16) {
                double a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 4 * u + lk;
                    a[u] = (aok && k < wa) ? ap[k] : 0.0;
                    b[u] = (bok && k < wa) ? bp[(size_t)k * ld] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = it * 16 + lk + 4 * r;
                if (row < rr && bj < wa) Tm[(size_t)row * ldh + cLo + bj] = acc[r];
            }
        }
    }
    __syncthreads();
    DBG_T(37);
    // S = Tm Hn^T + sig2 I  (lower triangle stands for the symmetrised matrix .5 (S + S^T), Updater.cc:418)
    const int lds_s = rhomax + 1;
    {
        const int nit = (rr + 15) / 16;
        for (int t = wvid; t < nit * nit; t += nwv) {
            const int it = t / nit, jt = t % nit;
            if (jt > it) continue;
            const int ai = it * 16 + li, bj = jt * 16 + li;
            const bool aok = ai < rr, bok = bj < rr;
            const double* ap = Tm + (size_t)ai * ldh + cLo;
            const double* bp = Hn + (size_t)bj * ldh + cLo;
            d4 acc = {0, 0, 0, 0};
            for (int k0 = 0; k0 < wa; k0 += 16) {
                double a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 4 * u + lk;
                    a[u] = (aok && k < wa) ? ap[k] : 0.0;
                    b[u] = (bok && k < wa) ? bp[k] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = it * 16 + lk + 4 * r;
                if (row < rr && bj < rr && bj <= row) S[row * lds_s + bj] = acc[r] + ((row == bj) ? sig2 : 0.0);
            }
        }
        for (int j = tid; j < rr; j += T) S[rr * lds_s + j] = Hn[(size_t)j * ldh + c6];   // residual as an extra row
    }
    __syncthreads();
    DBG_T(38);
    // gamma = |r^T S^-1 r| by a square-root-free L D L^T of S with the residual row appended (reference:
    // colPivHouseholderQr().solve, Updater.cc:420).  Columns stay unscaled (S[i][k] = l_ik d_k), so the residual row
    // carries w = L^-1 r and gamma = sum_k w_k^2 / d_k.  Wave 0 only, lane <-> row: no block barriers.
    double gam = 0;
    if (wave0) {
        double gsum = 0;
        const int i = lane;
        for (int k = 0; k < rr; ++k) {
            const double dk = S[k * lds_s + k];
            const double rd = 1.0 / (dk > 0 ? dk : 1e-300);
            double fik = 0;
            if (i > k && i <= rr) fik = S[i * lds_s + k] * rd;
            if (i == rr) gsum += S[rr * lds_s + k] * fik;
            const int jmax = (i < rr) ? i : rr - 1;
            if (i > k && i <= rr)
                for (int j = k + 1; j <= jmax; ++j) S[i * lds_s + j] -= fik * S[j * lds_s + k];
            __builtin_amdgcn_wave_barrier();
        }
        gam = fabs(__shfl(gsum, rr, 64));
        if (lane == 0) misc[8] = gam;
    }
    DBG_T(39);
    __syncthreads();
    gam = misc[8];
    const bool accept = gam < kChi2Dev[rr - 1];
    if (tid == 0) { acc_out[f] = accept ? 1 : 0; ndof_out[f] = rr; gamma_out[f] = gam; nrows_out[f] = accept ? rr : 0; }
    if (accept) {
        double* out = Hstack + (size_t)f * rhomax * ldh;
        for (int e = tid; e < rr * ldh; e += T) out[e] = Hn[e];
    }
    DBG_T(40);
}

// =============================================================== U7 compression, information form
// partial[g][p][q] = sum over the rows of feature group g of H[row][p] * H[row][q],
// q = 0..c6 (column c6 is the residual -> b).  grid = (groups, ceil(c6/16)), 256 threads.
#define GRAM_FG 4
__global__ __launch_bounds__(256) void gram_kernel(DevCfg cfg, int n, const double* Hstack, const int* nrows, double* partial) {
    const int c6 = 6 * n, ldh = cfg.ldh, rhomax = cfg.rho_max;
    const int g = blockIdx.x, p0 = blockIdx.y * 16;
    if (p0 >= c6) return;
    const int f0 = g * GRAM_FG;
    const int ncol = c6 + 1;
    int nr[GRAM_FG];
#pragma unroll
    for (int ff = 0; ff < GRAM_FG; ++ff) nr[ff] = (f0 + ff < cfg.Fu) ? nrows[f0 + ff] : 0;
    double* out = partial + (size_t)g * cfg.ldh * cfg.ldh;
    for (int e = threadIdx.x; e < 16 * ncol; e += 256) {
        const int p = p0 + e / ncol, q = e % ncol;
        if (p >= c6) continue;
        double acc = 0;
#pragma unroll
        for (int ff = 0; ff < GRAM_FG; ++ff) {
            const double* H = Hstack + (size_t)(f0 + ff) * rhomax * ldh;
            for (int r = 0; r < nr[ff]; ++r) acc += H[(size_t)r * ldh + p] * H[(size_t)r * ldh + q];
        }
        out[(size_t)p * ldh + q] = acc;
    }
}

// block = [A|b] (c6 x ldh row-major) + {n_good, n_rows}: the all-gather payload of the sharded updater
__global__ __launch_bounds__(256) void gram_reduce_kernel(DevCfg cfg, int n, const double* partial, int n_groups,
                                                          const int* nrows, double* block) {
    const int c6 = 6 * n, ldh = cfg.ldh;
    const int total = c6 * ldh;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int q = e % ldh;
        double acc = 0;
        if (q <= c6) for (int g = 0; g < n_groups; ++g) acc += partial[(size_t)g * ldh * ldh + e];
        block[e] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        int good = 0, rows = 0;
        for (int f = threadIdx.x; f < cfg.Fu; f += 64) { const int r = nrows[f]; if (r > 0) { good++; rows += r; } }
        good = (int)wave_sum_i64(good); rows = (int)wave_sum_i64(rows);
        if (threadIdx.x == 0) {
            block[(size_t)cfg.ldh * (cfg.ldh - 1)] = (double)good;
            block[(size_t)cfg.ldh * (cfg.ldh - 1) + 1] = (double)rows;
        }
    }
}

// world > 1: sum the gathered blocks in rank order -> Ab (same layout as a block, counts included)
__global__ __launch_bounds__(256) void block_sum_kernel(DevCfg cfg, int n, const double* blocks, int world, size_t block_stride, double* Ab) {
    const int c6 = 6 * n, ldh = cfg.ldh;
    const int total = c6 * ldh;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        double acc = 0;
        for (int w = 0; w < world; ++w) acc += blocks[(size_t)w * block_stride + e];
        Ab[e] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x < 2) {
        const size_t t = (size_t)ldh * (ldh - 1) + threadIdx.x;
        double acc = 0;
        for (int w = 0; w < world; ++w) acc += blocks[(size_t)w * block_stride + t];
        Ab[t] = acc;
    }
}

// =============================================================== FP64 MFMA GEMM:  T = s2 I + A Pcc
// One workgroup = 4 waves = a 32x32 output tile, each wave one 16x16 tile with v_mfma_f64_16x16x4_f64:
//   A operand lane l : A[i = l&15][k = l>>4]      B operand lane l : B[k = l>>4][j = l&15]
//   C/D      lane l : 4 values, row = (l>>4) + 4*r, col = l&15.
// A = Ab (row-major, ld = ldh), B = Pcc = P[24:,24:] (column-major, ld = dmax), T row-major ld = ldh.
__global__ __launch_bounds__(256) void gemm_T_kernel(DevCfg cfg, int n, const double* Ab, const double* P, double* Tm) {
    const int c6 = 6 * n, ldh = cfg.ldh, ld = cfg.dmax;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = blockIdx.y * 32 + (wave >> 1) * 16, j0 = blockIdx.x * 32 + (wave & 1) * 16;
    if (i0 >= c6 || j0 >= c6) return;
    const int li = lane & 15, lk = lane >> 4;
    const int ai = i0 + li, bj = j0 + li;
    const bool aok = ai < c6, bok = bj < c6;
    const double* ap = Ab + (size_t)ai * ldh;
    const double* bp = P + 24 + (size_t)(24 + bj) * ld;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < c6; k0 += 16) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 4 * u + lk;
            a[u] = (aok && k < c6) ? ap[k] : 0.0;
            b[u] = (bok && k < c6) ? bp[k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    const int col = j0 + li;
    if (col < c6) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + lk + 4 * r;
            if (row < c6) Tm[(size_t)row * ldh + col] = acc[r] + ((row == col) ? s2 : 0.0);
        }
    }
}

// =============================================================== solve: W = T^-1, y = W b, dx, x+
// One workgroup of 1024 threads.  Tableau M = [T | b | I]  (c6 x NC, NC = 2 c6 + 1) in LDS (or in
// global scratch when it does not fit).  Gauss-Jordan with partial pivoting, NO row swaps and
// deferred pivot scaling: step k picks the unused row p with max |M[i][k]|, every other row i does
// M[i][j] -= (M[i][k]/piv) M[p][j] for the still-active columns.  Every wave re-derives p itself from
// LDS, so ONE barrier per column suffices.  Solution row k is row p_k of the tableau times 1/piv_k.
template <bool USE_LDS>
__device__ __forceinline__ void solve_body(const DevCfg& cfg, FilterMeta* meta, int n, const double* Tg, const double* Ab,
                                           const double* x, const double* P, double* Wout, double* x_out, double* Mg, double* sh) {
    __shared__ int s_prow[6 * RVIO_MAX_LEN];
    __shared__ double s_ipiv[6 * RVIO_MAX_LEN];
    __shared__ double s_y[6 * RVIO_MAX_LEN];
    __shared__ double s_dx[24 + 6 * RVIO_MAX_LEN];
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax, xd = 26 + 7 * n;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wv = tid >> 6, NW = T >> 6;
    const int NC = 2 * c6 + 1;
    const int ldm = USE_LDS ? (NC | 1) : (2 * ldh);
    double* M = USE_LDS ? sh : Mg;
    const int n_good = (int)Ab[(size_t)ldh * (ldh - 1)], n_rows = (int)Ab[(size_t)ldh * (ldh - 1) + 1];
    const bool upd = n_good > 2;                       // Updater.cc:460
    if (tid == 0) { meta->n_good = n_good; meta->n_rows = n_rows; meta->updated = upd ? 1 : 0; }
    if (!upd) {                                        // pass-through (Updater.cc:621-627): W = 0 => U = G = 0 => P+ = P exactly
        for (int e = tid; e < c6 * c6; e += T) Wout[(size_t)(e / c6) * ldh + (e % c6)] = 0.0;
        for (int i = tid; i < xd; i += T) x_out[i] = x[i];
        return;
    }
    for (int e = tid; e < c6 * NC; e += T) {
        const int r = e / NC, c = e % NC;
        double v;
        if (c < c6) v = Tg[(size_t)r * ldh + c];
        else if (c == c6) v = Ab[(size_t)r * ldh + c6];
        else v = (c - c6 - 1 == r) ? 1.0 : 0.0;
        M[(size_t)r * ldm + c] = v;
    }
    __syncthreads();
    unsigned long long used0 = 0, used1 = 0, used2 = 0;   // rows already chosen as pivots (uniform per wave)
    for (int k = 0; k < c6; ++k) {
        // pivot search: lane <-> row (rows lane, lane+64, lane+128)
        double best = -1.0; int bi = 0;
        for (int i = lane, q = 0; i < c6; i += 64, ++q) {
            const unsigned long long um = (q == 0) ? used0 : (q == 1 ? used1 : used2);
            if (!((um >> lane) & 1ull)) { const double v = fabs(M[(size_t)i * ldm + k]); if (v > best) { best = v; bi = i; } }
        }
        // wave arg-max: 4 DPP row-rotate steps (16-lane rows), then the 4 row winners through readlane
#define ARGMAX_STEP(CTRL)                                                                             \
        {                                                                                             \
            const double ov = dpp_f64<CTRL>(best);                                                    \
            const int oi = __builtin_amdgcn_update_dpp(bi, bi, CTRL, 0xf, 0xf, false);                \
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }                         \
        }
        ARGMAX_STEP(0x128) ARGMAX_STEP(0x124) ARGMAX_STEP(0x122) ARGMAX_STEP(0x121)
#undef ARGMAX_STEP
        {
            double b0 = readlane_f64(best, 0); int i0 = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
            for (int rw = 16; rw < 64; rw += 16) {
                const double ov = readlane_f64(best, rw); const int oi = __builtin_amdgcn_readlane(bi, rw);
                if (ov > b0 || (ov == b0 && oi < i0)) { b0 = ov; i0 = oi; }
            }
            best = b0; bi = i0;
        }
        const int pr = bi;
        if (pr < 64) used0 |= 1ull << pr; else if (pr < 128) used1 |= 1ull << (pr - 64); else used2 |= 1ull << (pr - 128);
        const double piv = M[(size_t)pr * ldm + k];
        const double ipiv = 1.0 / piv;
        if (tid == 0) { s_prow[k] = pr; s_ipiv[k] = ipiv; if (!(best > 0)) meta->err |= 1; }
        // eliminate: wave w owns rows w, w+NW, ...; lanes own the active columns j > k
        const int j0 = k + 1 + lane;
        double prv[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) { const int j = j0 + 64 * u; prv[u] = (j < NC) ? M[(size_t)pr * ldm + j] : 0.0; }
        for (int i = wv; i < c6; i += NW) {
            if (i == pr) continue;
            const double f = M[(size_t)i * ldm + k] * ipiv;
#pragma unroll
            for (int u = 0; u < 6; ++u) { const int j = j0 + 64 * u; if (j < NC) M[(size_t)i * ldm + j] -= f * prv[u]; }
        }
        __syncthreads();
    }
    // unscramble: solution row k = tableau row prow[k] * ipiv[k]
    for (int e = tid; e < c6 * c6; e += T) {
        const int k = e / c6, j = e % c6;
        Wout[(size_t)k * ldh + j] = M[(size_t)s_prow[k] * ldm + c6 + 1 + j] * s_ipiv[k];
    }
    for (int k = tid; k < c6; k += T) s_y[k] = M[(size_t)s_prow[k] * ldm + c6] * s_ipiv[k];
    __syncthreads();
    // dx = K r = Pc y   (Updater.cc:544)
    for (int i = tid; i < d; i += T) {
        double acc = 0;
        for (int k = 0; k < c6; ++k) acc += P[(size_t)i + (size_t)(24 + k) * ld] * s_y[k];
        s_dx[i] = acc;
    }
    __syncthreads();
    // state injection (Updater.cc:546-613)
    const double* dx = s_dx;
    if (tid == 0) {
        stq(x_out, qmul(small_q(dx[0], dx[1], dx[2]), ldq(x)));
        for (int i = 0; i < 6; ++i) x_out[4 + i] = dx[3 + i] + x[4 + i];
        st3(x_out + 7, unit3(ld3(x_out + 7)));
        stq(x_out + 10, qmul(small_q(dx[9], dx[10], dx[11]), ldq(x + 10)));
        for (int i = 0; i < 12; ++i) x_out[14 + i] = dx[12 + i] + x[14 + i];
    }
    for (int p = tid - 64; p >= 0 && p < n; p += T - 64) {
        stq(x_out + 26 + 7 * p, qmul(small_q(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), ldq(x + 26 + 7 * p)));
        for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
    }
}
__global__ __launch_bounds__(1024) void solve_kernel_lds(DevCfg cfg, FilterMeta* meta, int n, const double* Tg, const double* Ab,
                                                         const double* x, const double* P, double* Wout, double* x_out) {
    extern __shared__ __align__(16) double sh[];
    solve_body<true>(cfg, meta, n, Tg, Ab, x, P, Wout, x_out, nullptr, sh);
}
__global__ __launch_bounds__(1024) void solve_kernel_glb(DevCfg cfg, FilterMeta* meta, int n, const double* Tg, const double* Ab,
                                                         const double* x, const double* P, double* Wout, double* x_out, double* Mg) {
    solve_body<false>(cfg, meta, n, Tg, Ab, x, P, Wout, x_out, Mg, nullptr);
}

// =============================================================== U = Pc W, G = U A, P1 = P - G Pc^T
// One workgroup (4 waves) per 16-row strip of the d rows.  K H = [0 | G];  (I - K H) P = P1.
// LDS: Us[16][c6p], Gs[16][c6p] (row-major, c6p = c6 rounded up to 16, +1 pad).
__global__ __launch_bounds__(256) void ug_kernel(DevCfg cfg, int n, const double* P, const double* W, const double* Ab,
                                                 double* U, double* G, double* P1) {
    extern __shared__ __align__(16) double sh[];
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax;
    const int c6t = (c6 + 15) / 16, dt = (d + 15) / 16;
    const int lds = c6t * 16 + 1;
    double* Us = sh; double* Gs = sh + 16 * lds;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int i0 = blockIdx.x * 16;
    if (i0 >= d) return;
    const int ai = i0 + li;
    const bool aok = ai < d;
    // U strip: A = Pc[i][k] = P[i + (24+k) ld], B = W[k][j] (row-major ldh)
    for (int jt = wave; jt < c6t; jt += 4) {
        const int bj = jt * 16 + li; const bool bok = bj < c6;
        d4 acc = {0, 0, 0, 0};
        for (int k0 = 0; k0 < c6; k0 += 16) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 4 * u + lk;
                a[u] = (aok && k < c6) ? P[(size_t)ai + (size_t)(24 + k) * ld] : 0.0;
                b[u] = (bok && k < c6) ? W[(size_t)k * ldh + bj] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lk + 4 * r;
            Us[row * lds + jt * 16 + li] = acc[r];
            if (i0 + row < d && bj < c6) U[(size_t)(i0 + row) * ldh + bj] = acc[r];
        }
    }
    __syncthreads();
    // G strip: A = Us[i][k], B = A[k][j] (Ab row-major; A is symmetric)
    for (int jt = wave; jt < c6t; jt += 4) {
        const int bj = jt * 16 + li; const bool bok = bj < c6;
        d4 acc = {0, 0, 0, 0};
        for (int k0 = 0; k0 < c6; k0 += 16) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 4 * u + lk;
                a[u] = (k < c6) ? Us[li * lds + k] : 0.0;
                b[u] = (bok && k < c6) ? Ab[(size_t)k * ldh + bj] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lk + 4 * r;
            Gs[row * lds + jt * 16 + li] = acc[r];
            if (i0 + row < d && bj < c6) G[(size_t)(i0 + row) * ldh + bj] = acc[r];
        }
    }
    __syncthreads();
    // P1 strip = P - G Pc^T: A = Gs[i][k], B[k][j] = Pc[j][k] = P[j + (24+k) ld]
    for (int jt = wave; jt < dt; jt += 4) {
        const int bj = jt * 16 + li; const bool bok = bj < d;
        d4 acc = {0, 0, 0, 0};
        for (int k0 = 0; k0 < c6; k0 += 16) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 4 * u + lk;
                a[u] = (k < c6) ? Gs[li * lds + k] : 0.0;
                b[u] = (bok && k < c6) ? P[(size_t)bj + (size_t)(24 + k) * ld] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + lk + 4 * r;
            if (row < d && bj < d) P1[(size_t)row + (size_t)bj * ld] = P[(size_t)row + (size_t)bj * ld] - acc[r];
        }
    }
}

// =============================================================== Joseph form, final: P+ = sym(X),
//   X = P1 - P1c G^T + s2 G U^T   =  (I-KH) P (I-KH)^T + s2 K K^T      (Updater.cc:615-619)
// One wave per unordered 16x16 tile pair (I <= J): computes X_IJ and X_JI, writes .5 (X_IJ + X_JI^T) to both.
__device__ __forceinline__ d4 final_tile(const double* P1, const double* G, const double* U, int d, int c6, int ld, int ldh, double s2,
                                         int i0, int j0, int li, int lk) {
    const int ai = i0 + li, bj = j0 + li;
    const bool aok = ai < d, bok = bj < d;
    d4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    for (int k0 = 0; k0 < c6; k0 += 16) {
        double a1[4], b1[4], a2[4], b2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 4 * u + lk;
            const bool kok = k < c6;
            a1[u] = (aok && kok) ? P1[(size_t)ai + (size_t)(24 + k) * ld] : 0.0;   // P1c[i][k]
            b1[u] = (bok && kok) ? G[(size_t)bj * ldh + k] : 0.0;                  // G[j][k]
            a2[u] = (aok && kok) ? G[(size_t)ai * ldh + k] : 0.0;                  // G[i][k]
            b2[u] = (bok && kok) ? U[(size_t)bj * ldh + k] : 0.0;                  // U[j][k]
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[u], b2[u], acc2, 0, 0, 0);
        }
    }
    d4 out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r;
        const double p1 = (row < d && bj < d) ? P1[(size_t)row + (size_t)bj * ld] : 0.0;
        out[r] = p1 - acc[r] + s2 * acc2[r];
    }
    return out;
}
__global__ __launch_bounds__(256) void final_kernel(DevCfg cfg, int n, const double* P1, const double* G, const double* U, double* Pout) {
    __shared__ double tl[4][16][17];
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    const int nt = (d + 15) / 16, npair = nt * (nt + 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int pr = blockIdx.x * 4 + wave;
    if (pr >= npair) return;
    int I = 0, rem = pr;
    while (rem >= nt - I) { rem -= nt - I; ++I; }
    const int J = I + rem;
    const d4 xij = final_tile(P1, G, U, d, c6, ld, ldh, s2, I * 16, J * 16, li, lk);
    d4 xji = xij;
    if (I != J) xji = final_tile(P1, G, U, d, c6, ld, ldh, s2, J * 16, I * 16, li, lk);
    // transpose X_JI through LDS (wave-private 16x16 tile)
#pragma unroll
    for (int r = 0; r < 4; ++r) tl[wave][lk + 4 * r][li] = xji[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = lk + 4 * r, row = I * 16 + rr, col = J * 16 + li;
        const double v = .5 * (xij[r] + tl[wave][li][rr]);     // X_JI[col_local][row_local]
        if (row < d && col < d) {
            Pout[(size_t)row + (size_t)col * ld] = v;
            if (I != J) Pout[(size_t)col + (size_t)row * ld] = v;
        }
    }
}

// =============================================================== S1 + S2 fused: augmentation/slide + composition
// System.cc:279-365.  J P J^T with J = [I; rows 9..14] is a gather (out[a][b] = P[src(a)][src(b)]); composition
// multiplies the first 24 rows/columns by Vk.  Out-of-place (reads cur, writes cur^1).
// block 0: the 24x24 corner Vk P11 Vk^T (symmetrised) + the state vector; other blocks: one clone column per thread.
__device__ __forceinline__ int aug_src(int a, int n, int nmax, int do_aug) {
    if (a < 24 || !do_aug) return a;
    const int cb = (a - 24) / 6, off = (a - 24) % 6;
    if (n < nmax) return (cb < n) ? a : 9 + off;
    return (cb < nmax - 1) ? a + 6 : 9 + off;
}
__global__ __launch_bounds__(256) void augcomp_kernel(DevCfg cfg, int n, int do_aug, const double* x, const double* P,
                                                      double* x_out, double* P_out, double* pose_out) {
    __shared__ double Vk[24][25];
    __shared__ double P11[24][25];
    __shared__ double Tm[24][25];
    __shared__ double xs[26];
    const int nmax = cfg.nmax, ld = cfg.dmax;
    const int n2 = do_aug ? ((n < nmax) ? n + 1 : nmax) : n;
    const int d2 = 24 + 6 * n2, xd2 = 26 + 7 * n2;
    const int tid = threadIdx.x;
    DBG_T(20);
    if (tid < 26) xs[tid] = x[tid];
    if (blockIdx.x == 0) for (int e = tid; e < 576; e += 256) P11[e % 24][e / 24] = P[(size_t)(e % 24) + (size_t)(e / 24) * ld];
    for (int e = tid; e < 576; e += 256) Vk[e / 24][e % 24] = 0.0;
    __syncthreads();
    DBG_T(21);
    const q4 qG = ldq(xs), qk = ldq(xs + 10);
    const d3 pG = ld3(xs + 4), pk = ld3(xs + 14);
    const m33 RG = q2r(qG), Rk = q2r(qk);
    const d3 gk = unit3(mv33(Rk, ld3(xs + 7)));
    const q4 qkG = qmul(qk, qG);
    const d3 pkG = mv33(Rk, sub3(pG, pk));
    if (tid < 9) {
        const int i = tid / 3, j = tid % 3;
        const m33 spx = skew33(pkG), sgx = skew33(gk);
        Vk[i][j] = Rk.m[3 * i + j];           Vk[i][9 + j] = (i == j) ? 1.0 : 0.0;
        Vk[3 + i][3 + j] = Rk.m[3 * i + j];   Vk[3 + i][9 + j] = spx.m[3 * i + j];   Vk[3 + i][12 + j] = -Rk.m[3 * i + j];
        Vk[6 + i][6 + j] = Rk.m[3 * i + j];   Vk[6 + i][9 + j] = sgx.m[3 * i + j];
        Vk[15 + tid][15 + tid] = 1.0;
    }
    __syncthreads();
    DBG_T(22);
    if (blockIdx.x == 0) {
        for (int e = tid; e < 576; e += 256) { int i = e / 24, j = e % 24; double a = 0; for (int k = 0; k < 24; ++k) a += Vk[i][k] * P11[k][j]; Tm[i][j] = a; }
        __syncthreads();
        DBG_T(23);
        for (int e = tid; e < 576; e += 256) { int i = e / 24, j = e % 24; double a = 0; for (int k = 0; k < 24; ++k) a += Tm[i][k] * Vk[j][k]; P11[i][j] = a; }
        __syncthreads();
        DBG_T(24);
        for (int e = tid; e < 576; e += 256) { int i = e / 24, j = e % 24; P_out[(size_t)i + (size_t)j * ld] = .5 * (P11[i][j] + P11[j][i]); }
        // state: augmentation (System.cc:282-287,303-306) then composition (System.cc:360-365)
        for (int i = tid; i < xd2; i += 256) {
            double v;
            if (i < 4) v = (&qkG.x)[i];
            else if (i < 7) v = (&pkG.x)[i - 4];
            else if (i < 10) v = (&gk.x)[i - 7];
            else if (i < 13) v = 0.0;
            else if (i == 13) v = 1.0;
            else if (i < 17) v = 0.0;
            else if (i < 26) v = xs[i];
            else {
                int src = i;
                if (do_aug) {
                    const int cb = (i - 26) / 7, off = (i - 26) % 7;
                    if (n < nmax) src = (cb < n) ? i : 10 + off;
                    else src = (cb < nmax - 1) ? i + 7 : 10 + off;
                }
                v = (src < 26) ? xs[src] : x[src];
            }
            x_out[i] = v;
        }
        if (tid == 0) {   // pose line (System.cc:371-374)
            const d3 pGk = mv33(tr33(RG), sub3(pk, pG));
            st3(pose_out, pGk); stq(pose_out + 3, qkG);
        }
        DBG_T(25);
    } else {
        const int c6 = 6 * n2;
        for (int c = (blockIdx.x - 1) * 256 + tid; c < c6; c += (gridDim.x - 1) * 256) {
            const int sc = aug_src(24 + c, n, nmax, do_aug);
            double col[24];
            const double* pc = P + (size_t)sc * ld;
#pragma unroll
            for (int k = 0; k < 24; ++k) col[k] = pc[k];
            for (int i = 0; i < 24; ++i) {
                double a = 0;
#pragma unroll
                for (int k = 0; k < 24; ++k) a += Vk[i][k] * col[k];
                P_out[(size_t)i + (size_t)(24 + c) * ld] = a;
                P_out[(size_t)(24 + c) + (size_t)i * ld] = a;
            }
            for (int r = 24; r < d2; ++r) P_out[(size_t)r + (size_t)(24 + c) * ld] = pc[aug_src(r, n, nmax, do_aug)];
        }
    }
}
