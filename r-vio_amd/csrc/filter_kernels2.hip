// filter_kernels2.hip — the filter path's kernels around the update:
//   propagate_kernel3 / propagate_kernel3b   PreIntegrator::propagate, one workgroup per instance (b: two workgroups per CU, batch handles)
//   feat_prop_kernel                          feat_build_kernel<16> + propagate as one more workgroup (pipelined single-stream path)
//   augcomp_kernel2                           augmentation / slide + composition (System.cc:279-365); bumps the filter's completion counter
// Rules applied (profiles/r01_*): no dynamic indexing of register arrays (it lands in scratch),
// __restrict__ on every pointer, sparsity of Phi / Vk exploited (the LDS port of ONE CU is the
// limiter), no loops over dependent global loads.
#pragma once
#include "rvio_dev.h"
#include "solve9.hip"   // the Cholesky role of the solve rides in the per-feature / propagate launches

// skew(v)[i][j] with v in LDS (dynamic indexing is fine there):  [[0,-z,y],[z,0,-x],[-y,x,0]]
__device__ __forceinline__ double skew_e(const double* v, int i, int j) {
    if (i == j) return 0.0;
    const double e = v[3 - i - j];
    return (j == (i + 1) % 3) ? -e : e;
}

// =============================================================== P1 propagate (v3)
// One workgroup, 256 threads.
//  A  lane s <-> IMU sample s: trig, dR, f1..f4 and the two a-dependent vectors (parallel)
//  B  the serial chain Rk <- dR Rk, dp, dv, pk, vk, gk in reference order (all threads, no broadcast)
//  C  the ten non-trivial 3x3 blocks of rows 9..17 of Phi = I + dt F (PreIntegrator.cc:123-132), thread <-> (sample, i, j)
//  D  (round 4) the chunk's transition composed first — suffix products of the Phi_s, one barrier per sample —, then applied to P once:
//     P <- Psi_c P Psi_c^T + sum_s S_s Q_s S_s^T  (rounds 1-3: sample by sample, four barriers each)
struct Prop3Sample { double dR[9], up[3], uv[3], w[3], dt, Rk[9], vk[3], gk[3]; };

// propagate's LDS: static in the kernels of their own, carved out of the launch's DYNAMIC LDS by the propagate workgroup of feat_prop_kernel (that
// workgroup builds no feature: the per-feature footprint is idle in it — so the fused launch needs max(per-feature, propagate), not the sum,
// and fits every window with the 16-sample chunk)
template <int PROP3_CH>
struct Prop3Lds {
    double Pl[24][25];
    double Psi[24][25];
    double PhiSx[2][PROP3_CH][9][25];       // (one array: the clone-column update at the end stages its columns in it)
    double PsiC[9][25];                     // rows 9..17 of the chunk's Psi_c = S_0 Phi_0
    double Nq[PROP3_CH][9][6];              // S_s[9..17, (theta, v)] Qd_s
    double Qd[PROP3_CH][6][6];              // the dense (theta, v) block of Q_s
    double vxs[PROP3_CH][9];
    Prop3Sample sm[PROP3_CH];
    double xs[26];
    double chain[9 + 3 * 5 + 1];            // the serial chain's state between the chunks' phase B
};

template <int PROP3_CH = 16>     // samples composed per chunk: 16 for one stream (one chunk at 200 Hz / 20 Hz), 8 for batch handles (47 instead of 86 KB of LDS: two workgroups per CU)
__device__ __forceinline__ void propagate_body(DevCfg cfg, FilterMeta* __restrict__ meta, int n, double* __restrict__ x,
                                               double* __restrict__ P, const rvio_imu* __restrict__ imu, int m, size_t bs, size_t imu_bs, Prop3Lds<PROP3_CH>& L) {
    meta = zoff(meta, bs); x = zoff(x, bs); P = zoff(P, bs); imu = zoff(imu, imu_bs);
    auto& Pl = L.Pl;
    auto& Psi = L.Psi;
    auto& PhiSx = L.PhiSx;
    double (*Phi9)[9][25] = PhiSx[0];
    double (*Sx)[9][25] = PhiSx[1];             // rows 9..17 of the suffix products S_s = Phi_{mc-1} ... Phi_{s+1}
    auto& PsiC = L.PsiC;
    auto& Nq = L.Nq;
    auto& Qd = L.Qd;
    auto& vxs = L.vxs;
    auto& sm = L.sm;
    auto& xs = L.xs;
    auto& chain = L.chain;
    const int tid = threadIdx.x;
    const int ld = cfg.dmax;
    if (tid == 0) { meta->n_good = 0; meta->n_rows = 0; meta->updated = 0; meta->trunc_at = -1; }
    for (int e = tid; e < 576; e += 256) {
        int i = e % 24, j = e / 24;
        Pl[i][j] = P[i + (size_t)j * ld];
        Psi[i][j] = (i == j) ? 1.0 : 0.0;
    }
    if (tid < 26) xs[tid] = x[tid];
    DBG_T(10); DBG_W(tid == 0, 26);
    __syncthreads();
    DBG_T(11);
    const d3 bg = ld3(xs + 20), ba = ld3(xs + 23);
    const d3 gR = ld3(xs + 7), vR = ld3(xs + 17);
    // the serial chain's state (Rk, dp, dv, pk, vk, gk, Dt) lives in LDS BETWEEN the chunks' phase B: held in registers across the whole kernel it
    // was 66 VGPRs of pressure on every other phase (the two-workgroups-per-CU form of batch handles spilled 916 B of scratch)
    if (tid == 0) {
        const m33 R0 = q2r(ldq(xs + 10));
#pragma unroll
        for (int k = 0; k < 9; ++k) chain[k] = R0.m[k];
        st3(chain + 9, mk3(0, 0, 0)); st3(chain + 12, mk3(0, 0, 0));      // dp, dv
        st3(chain + 15, ld3(xs + 14)); st3(chain + 18, vR); st3(chain + 21, gR);   // pk, vk, gk
        chain[24] = 0.0;                                                   // Dt
    }
    const m33 I = eye33();
    const double nG = cfg.gravity;
    // this thread's (row r9 of rows 9..17, column c9)
    const int r9 = tid / 24, c9 = tid % 24;
    for (int s0 = 0; s0 < m; s0 += PROP3_CH) {
        const int mc = (m - s0 < PROP3_CH) ? (m - s0) : PROP3_CH;
        for (int e = tid; e < mc * 9 * 25; e += 256) (&Phi9[0][0][0])[e] = 0.0;     // the structural zeros of rows 9..17 of Phi (phase C writes the others)
        // ---- A
        if (tid < mc) {
            const rvio_imu u = imu[s0 + tid];
            const d3 w = sub3(mk3(u.w[0], u.w[1], u.w[2]), bg), a = sub3(mk3(u.a[0], u.a[1], u.a[2]), ba);
            const double dt = u.dt, w1 = nrm3(w);
            const bool small = w1 < cfg.small_angle;
            const double wdt = w1 * dt, wdt2 = wdt * wdt;
            double sw, cw;
            sincos(wdt, &sw, &cw);
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
            Prop3Sample& q = sm[tid];
#pragma unroll
            for (int k = 0; k < 9; ++k) q.dR[k] = dR.m[k];
            st3(q.up, mv33(add33(add33(scl33(.5 * dt * dt, I), scl33(f1, wx)), scl33(f2, wx2)), a));
            st3(q.uv, mv33(add33(add33(scl33(dt, I), scl33(f3, wx)), scl33(f4, wx2)), a));
            st3(q.w, w); q.dt = dt;
        }
        __syncthreads();
        DBG_T(12);
        // ---- B: the serial chain, in ONE wave (the other three wait at the barrier: run by all four it was four times the instruction issue for
        // nothing — what the chain leaves behind is read from LDS; at 2048 instances per launch the kernel is issue bound); its state comes from
        // and returns to LDS
        if (tid < 64) {
            m33 Rk = ldm33(chain), RkT = tr33(Rk);
            d3 dp = ld3(chain + 9), dv = ld3(chain + 12), pk = ld3(chain + 15), vk = ld3(chain + 18), gk = ld3(chain + 21);
            double Dt = chain[24];
            for (int s = 0; s < mc; ++s) {
                Prop3Sample& q = sm[s];
                if (tid == 0) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) q.Rk[k] = Rk.m[k];
                    st3(q.vk, vk); st3(q.gk, gk);
                }
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
            // (one wave: every lane has read the chain's previous state before lane 0 overwrites it — program order within the wave)
            if (tid == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) chain[k] = Rk.m[k];
                st3(chain + 9, dp); st3(chain + 12, dv); st3(chain + 15, pk); st3(chain + 18, vk); st3(chain + 21, gk);
                chain[24] = Dt;
            }
        }
        __syncthreads();
        DBG_T(13);
        // ---- C: thread <-> (sample s, i, j)
        for (int e = tid; e < mc * 9; e += 256) {
            const int s = e / 9, i = (e % 9) / 3, j = e % 3;
            const Prop3Sample& q = sm[s];
            const double dt = q.dt, id = (i == j) ? 1.0 : 0.0;
            const double wxe = skew_e(q.w, i, j), vxe = skew_e(q.vk, i, j), gxe = skew_e(q.gk, i, j);
            const double rtv = q.Rk[i] * skew_e(q.vk, 0, j) + q.Rk[3 + i] * skew_e(q.vk, 1, j) + q.Rk[6 + i] * skew_e(q.vk, 2, j);   // (Rk^T [v]x)(i,j)
            double (*ph)[25] = Phi9[s];
            ph[i][9 + j] = id - dt * wxe;       ph[i][18 + j] = -dt * id;
            ph[3 + i][9 + j] = -dt * rtv;       ph[3 + i][12 + j] = id;           ph[3 + i][15 + j] = dt * q.Rk[3 * j + i];
            ph[6 + i][6 + j] = -dt * nG * q.Rk[3 * i + j];
            ph[6 + i][9 + j] = -dt * nG * gxe;  ph[6 + i][15 + j] = id - dt * wxe;
            ph[6 + i][18 + j] = -dt * vxe;      ph[6 + i][21 + j] = -dt * id;
            vxs[s][3 * i + j] = vxe;
        }
        __syncthreads();
        DBG_T(14);
        // ---- D (round 4): the chunk's samples are COMPOSED first and applied to P once.  Rounds 1-3 applied them one by one — P <- Phi_s P
        // Phi_s^T + Q_s with four barriers per sample: ten dependent steps, 25 of the kernel's 30 us.  Phi_s = I + (rows 9..17), and such
        // matrices are closed under multiplication, so with  S_s = Phi_{mc-1} ... Phi_{s+1}  (S_{mc-1} = I) and  Psi_c = S_0 Phi_0:
        //      P <- Psi_c P Psi_c^T + sum_s S_s Q_s S_s^T            (the same matrix as the sample-by-sample recursion, re-associated)
        // The suffix products are a chain of 9 x 9 x 24 products with ONE barrier per sample; everything else is parallel over the samples.
        if (tid < 216) Sx[mc - 1][r9][c9] = (c9 == 9 + r9) ? 1.0 : 0.0;
        __syncthreads();
        for (int s = mc - 2; s >= -1; --s) {       // s = -1: Psi_c
            double acc = 0;
            if (tid < 216) {
                const double (*S1)[25] = Sx[s + 1];
                const double (*ph)[25] = Phi9[s + 1];
                acc = (c9 >= 9 && c9 < 18) ? 0.0 : S1[r9][c9];
#pragma unroll
                for (int k = 0; k < 9; ++k) acc += S1[r9][9 + k] * ph[k][c9];
                if (s >= 0) Sx[s][r9][c9] = acc; else PsiC[r9][c9] = acc;
            }
            __syncthreads();
        }
        DBG_T(15);
        // Psi <- Psi_c Psi (rows 9..17; the other rows of both are identity rows) and X = rows 9..17 of Psi_c P
        double accS = 0, accP = 0;
        if (tid < 216) {
            accS = (c9 >= 9 && c9 < 18) ? 0.0 : PsiC[r9][c9];
#pragma unroll
            for (int k = 0; k < 9; ++k) accS += PsiC[r9][9 + k] * Psi[9 + k][c9];
#pragma unroll
            for (int k = 6; k < 24; ++k) accP += PsiC[r9][k] * Pl[k][c9];      // (columns 0..5 of rows 9..17 are structurally zero)
        }
        // Qd_s: the dense 6 x 6 block of Q_s = dt G Sigma G^T on (theta, v) = columns A6 = 9, 10, 11, 15, 16, 17 (PreIntegrator.cc:135-140: theta-theta
        // dt sg2 I, theta-v / v-theta dt sg2 [v]x terms, v-v dt sg2 [v]x [v]x^T + dt sa2 I), tabulated once per sample
        for (int e = tid; e < mc * 36; e += 256) {
            const int s = e / 36, rem = e - s * 36, a = rem / 6, b = rem - a * 6;
            const int ba = a / 3, ii = a - 3 * ba, bb = b / 3, jj = b - 3 * bb;
            const double dt = sm[s].dt;
            const double* vx = vxs[s];
            double q;
            if (ba == 0 && bb == 0) q = (ii == jj) ? dt * cfg.sg2 : 0.0;
            else if (ba == 0 && bb == 1) q = dt * cfg.sg2 * vx[3 * jj + ii];
            else if (ba == 1 && bb == 0) q = dt * cfg.sg2 * vx[3 * ii + jj];
            else {
                q = ((dt * vx[3 * ii]) * cfg.sg2) * vx[3 * jj] + ((dt * vx[3 * ii + 1]) * cfg.sg2) * vx[3 * jj + 1] + ((dt * vx[3 * ii + 2]) * cfg.sg2) * vx[3 * jj + 2];
                if (ii == jj) q += dt * cfg.sa2;
            }
            Qd[s][a][b] = q;
        }
        __syncthreads();
        if (tid < 216) { Psi[9 + r9][c9] = accS; Pl[9 + r9][c9] = accP; }
        // N_s = S_s[9..17, A6] Qd_s  (rows 18..23 of S_s are unit rows: zero in the A6 columns)
        for (int e = tid; e < mc * 54; e += 256) {
            const int s = e / 54, rem = e - s * 54, ip = rem / 6, b = rem - ip * 6;
            double acc = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) acc += Sx[s][ip][(a < 3 ? 9 : 12) + a] * Qd[s][a][b];
            Nq[s][ip][b] = acc;
        }
        __syncthreads();
        double accC = 0;
        if (tid < 216) {
#pragma unroll
            for (int k = 6; k < 24; ++k) accC += Pl[c9][k] * PsiC[r9][k];
        }
        // T = sum_s S_s Q_s S_s^T on rows / columns 9..23 (symmetric).  Threads 0..80: the 9 x 9 block; 81..134: the 9 x 6 block against the bias
        // random walks (Q_s[a][a] = dt swg2 for a = 18..20, dt swa2 for 21..23) and its mirror image; 135..140: the bias diagonal itself
        double accT = 0;
        if (tid < 81) {
            const int ti = tid / 9, tj = tid - 9 * ti;
            for (int s = 0; s < mc; ++s) {
                const double qg = sm[s].dt * cfg.swg2, qa = sm[s].dt * cfg.swa2;
                double t = 0;
#pragma unroll
                for (int b = 0; b < 6; ++b) t += Nq[s][ti][b] * Sx[s][tj][(b < 3 ? 9 : 12) + b];
#pragma unroll
                for (int a = 18; a < 24; ++a) t += (Sx[s][ti][a] * (a < 21 ? qg : qa)) * Sx[s][tj][a];
                accT += t;
            }
        } else if (tid < 135) {
            const int e = tid - 81, ti = e / 6, k = e - 6 * ti;
            for (int s = 0; s < mc; ++s) accT += Sx[s][ti][18 + k] * (sm[s].dt * (k < 3 ? cfg.swg2 : cfg.swa2));
        } else if (tid < 141) {
            const int k = tid - 135;
            for (int s = 0; s < mc; ++s) accT += sm[s].dt * (k < 3 ? cfg.swg2 : cfg.swa2);
        }
        __syncthreads();
        if (tid < 216) Pl[c9][9 + r9] = accC;
        __syncthreads();
        if (tid < 81) Pl[9 + tid / 9][9 + tid % 9] += accT;
        else if (tid < 135) { const int e = tid - 81, ti = e / 6, k = e - 6 * ti; Pl[9 + ti][18 + k] += accT; Pl[18 + k][9 + ti] += accT; }
        else if (tid < 141) Pl[18 + tid - 135][18 + tid - 135] += accT;
        __syncthreads();
        DBG_T(16);
    }
    if (tid == 0) {
        stq(x + 10, r2q(ldm33(chain)));
        st3(x + 14, ld3(chain + 15));
        st3(x + 17, ld3(chain + 18));
    }
    // P11 back (symmetrised, PreIntegrator.cc:192); P22 is untouched and already symmetric
    for (int e = tid; e < 576; e += 256) {
        int i = e % 24, j = e / 24;
        P[i + (size_t)j * ld] = .5 * (Pl[i][j] + Pl[j][i]);
    }
    // P12 = Psi P12, P21 = P12^T (PreIntegrator.cc:186-191).  Rows of Psi outside 9..17 are identity rows, so only rows 9..17 of each clone
    // column change.  The columns are staged in LDS (over the dead Phi / suffix-product buffers, in tiles when the window is long) with
    // COALESCED loads — a column is 24 contiguous doubles; one thread per column walking it was 64 cache lines per load instruction, ~24x the
    // bytes at 2048 instances per launch — then one thread per (column, three of the nine rows): 24 LDS reads, 3 outputs (+ mirror).  (One
    // thread per column also made the compiler keep all nine rows of Psi — 216 doubles — in registers across the column loop: 488 VGPRs in
    // the one-workgroup-per-CU forms, 916 B of scratch spills in the two-per-CU form of batch handles.)  Same sums in the same order per output.
    {
        double* stage = &PhiSx[0][0][0][0];
        constexpr int CAP_COLS = (2 * PROP3_CH * 9 * 25) / 25;     // columns of 25 doubles (24 + 1 pad: conflict-free when lanes differ in the column)
        const int c6 = 6 * n;
        for (int c0 = 0; c0 < c6; c0 += CAP_COLS) {
            const int nc = min(CAP_COLS, c6 - c0);
            __syncthreads();                                       // the buffers' previous readers are done
            for (int e = tid; e < 24 * nc; e += 256) { const int k = e % 24, c = e / 24; stage[c * 25 + k] = P[(size_t)k + (size_t)(24 + c0 + c) * ld]; }
            __syncthreads();
            for (int e = tid; e < 3 * nc; e += 256) {
                const int c = e % nc, r0 = 9 + 3 * (e / nc);
                const double* col = stage + c * 25;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double acc = 0;
#pragma unroll
                    for (int k = 0; k < 24; ++k) acc += Psi[r0 + r][k] * col[k];
                    P[(size_t)(r0 + r) + (size_t)(24 + c0 + c) * ld] = acc;
                    P[(24 + c0 + c) + (size_t)(r0 + r) * ld] = acc;
                }
            }
        }
    }
    DBG_T(17); DBG_W(tid == 0, 27);
}

__global__ __launch_bounds__(256) void propagate_kernel3(DevCfg cfg, FilterMeta* __restrict__ meta, int n, double* __restrict__ x,
                                                         double* __restrict__ P, const rvio_imu* __restrict__ imu, int m, size_t bs, size_t imu_bs) {
    __shared__ Prop3Lds<16> L;
    propagate_body<16>(cfg, meta, n, x, P, imu, m, bs, imu_bs, L);
}
// propagate with the Cholesky role of solve9 as a second workgroup (plain handle, 6n <= 96; staged entry points: rvio_hip_propagate[_dev],
// rvio_hip_frame_begin_dev): propagation rewrites the IMU block and the cross terms of P, the role reads the clone block only
template <int BS>
__global__ __launch_bounds__(256) void propagate_chol_kernel(DevCfg cfg, FilterMeta* __restrict__ meta, int n, double* __restrict__ x,
                                                             double* __restrict__ P, const rvio_imu* __restrict__ imu, int m, double* __restrict__ chol_scr) {
    if (blockIdx.x == 1) {
        __shared__ S9CholLds<2 * BS, 4> sh;
        s9_chol_role<BS>(cfg, n, P, chol_scr, sh);
        return;
    }
    __shared__ Prop3Lds<16> L;
    propagate_body<16>(cfg, meta, n, x, P, imu, m, 0, 0, L);
}
// batch handles: two workgroups per CU (256 VGPRs, part of the working set in scratch) — throughput, not latency
__global__ __launch_bounds__(256, 2) void propagate_kernel3b(DevCfg cfg, FilterMeta* __restrict__ meta, int n, double* __restrict__ x,
                                                             double* __restrict__ P, const rvio_imu* __restrict__ imu, int m, size_t bs, size_t imu_bs) {
    __shared__ Prop3Lds<8> L;
    propagate_body<8>(cfg, meta, n, x, P, imu, m, bs, imu_bs, L);
}

// PreIntegrator::propagate and the per-feature stage of Updater::update in ONE launch (single instance, pipelined whole-frame path):
// U1-U5 read only the clone states and P[24:,24:], which propagation does not touch (it rewrites the IMU state, P[0:24,0:24] and the
// cross terms P[0:24,24:] / P[24:,0:24]), so the two are independent; the last workgroup IS propagate_kernel3, the others ARE
// feat_build_kernel (256 threads).  Takes propagate's ~30 us off the filter stream's serial chain.
// chol_scr != NULL (round 5, solve9.hip at 6n <= 96): one more workgroup factors the clone block Pcc = L L^T into the solve's tile slab —
// the measurement-independent part of the solve, off the filter chain (its LDS: the launch's dynamic LDS, idle in that workgroup).
__global__ __launch_bounds__(256) void feat_prop_kernel(DevCfg cfg, int n, double* x, double* P,
                                                        const int* n_feat_ptr, const unsigned char* types, const int* lens, const float* meas,
                                                        double* Gshare, int* nrows_out, int* acc_out, int* ndof_out, double* gamma_out,
                                                        double* pfinv_out, double* tm_global, BatchIn bin,
                                                        FilterMeta* meta, const rvio_imu* imu, int m, double* chol_scr, int chol_nt, int shard_rank, int shard_world,
                                                        double* lit_rows) {
    DBG_R(blockIdx.x == 0, 0);
    DBG_W(blockIdx.x == 0 && threadIdx.x == 0, 28);
    extern __shared__ __align__(16) double fp_dyn[];
    if (blockIdx.x == gridDim.x - 1) { propagate_body<16>(cfg, meta, n, x, P, imu, m, 0, 0, *reinterpret_cast<Prop3Lds<16>*>(fp_dyn)); return; }   // (its LDS: the launch's dynamic LDS)
    if (chol_scr && blockIdx.x == gridDim.x - 2) {
        if (chol_nt == 4) s9_chol_role<2>(cfg, n, P, chol_scr, *reinterpret_cast<S9CholLds<4, 4>*>(fp_dyn));
        else s9_chol_role<3>(cfg, n, P, chol_scr, *reinterpret_cast<S9CholLds<6, 4>*>(fp_dyn));
        return;
    }
    feat_build_body<16>(cfg, n, x, P, n_feat_ptr, types, lens, meas, shard_rank, shard_world, Gshare, nrows_out, acc_out, ndof_out, gamma_out, pfinv_out, tm_global, 0, bin, meta, (int)blockIdx.x, nullptr, nullptr, lit_rows);
}

// =============================================================== S1 + S2 fused (v2): augmentation/slide + composition
// System.cc:279-365.  J P J^T with J = [I; rows 9..14] is a gather (out[a][b] = P[src(a)][src(b)]); composition
// multiplies the first 24 rows/columns by Vk, which is sparse: rows 0..8 have 4 / 9 / 6 non-zeros, rows 9..14 are
// zero, rows 15..23 are identity rows.  Out-of-place (reads cur, writes cur^1).
//   block 0      : the 24x24 corner Vk P11 Vk^T (symmetrised) + the state vector + the pose line
//   blocks 1..   : every other entry, one thread per entry (coalesced along the column)
__device__ __forceinline__ int aug_src2(int a, int n, int nmax, int do_aug) {
    if (a < 24 || !do_aug) return a;
    const int cb = (a - 24) / 6, off = (a - 24) % 6;
    if (n < nmax) return (cb < n) ? a : 9 + off;
    return (cb < nmax - 1) ? a + 6 : 9 + off;
}
__global__ __launch_bounds__(256) void augcomp_kernel2(DevCfg cfg, int n, int do_aug, const double* __restrict__ x, const double* __restrict__ P,
                                                       double* __restrict__ x_out, double* __restrict__ P_out, double* __restrict__ pose_out, size_t bs,
                                                       unsigned long long* done) {
    DBG_R(blockIdx.x == 0, 5);
    x = zoff(x, bs); P = zoff(P, bs); x_out = zoff(x_out, bs); P_out = zoff(P_out, bs); pose_out = zoff(pose_out, bs);
    __shared__ double Vk[24][25];
    __shared__ double P11[24][25];
    __shared__ double Tm[24][25];
    __shared__ double xs[26];
    __shared__ double xo[17];
    const int nmax = cfg.nmax, ld = cfg.dmax;
    const int n2 = do_aug ? ((n < nmax) ? n + 1 : nmax) : n;
    const int d2 = 24 + 6 * n2, xd2 = 26 + 7 * n2;
    const int tid = threadIdx.x;
    // (round 6) every global load of a workgroup is issued before the first barrier: the corner, the clone states the state copy moves, and — in the
    // gather workgroups — every entry that does not involve Vk (stored at once) plus the operands of the first one that does; Vk's structural zeros
    // are skipped in the corner's two products (a zero factor adds +-0 to a sum: the same bits, 9 terms instead of 24).  7.5 -> ~4.5 us in situ.
    if (tid < 26) xs[tid] = x[tid];
    for (int e = tid; e < 576; e += 256) Vk[e / 24][e % 24] = 0.0;
    const int gstride = ((int)gridDim.x - 1) * 256, e0 = ((int)blockIdx.x - 1) * 256 + tid, total = d2 * d2;
    double xv = 0.0, pr[9];
    bool pre = false;
    if (blockIdx.x == 0) {
        for (int e = tid; e < 576; e += 256) P11[e % 24][e / 24] = P[(size_t)(e % 24) + (size_t)(e / 24) * ld];
        static_assert(26 + 7 * (RVIO_MAX_LEN - 1) <= 256, "the state copy is one value per thread");
        for (int i = tid; i < xd2; i += 256) {
            int src = i;
            if (i >= 26 && do_aug) {
                const int cb = (i - 26) / 7, off = (i - 26) % 7;
                if (n < nmax) src = (cb < n) ? i : 10 + off;
                else src = (cb < nmax - 1) ? i + 7 : 10 + off;
            }
            xv = x[src];
        }
    } else {
        // entries with a >= 24 or b >= 24:  a = row, b = column (column-major: consecutive threads walk down a column)
        for (int e = e0; e < total; e += gstride) {
            const int a = e % d2, b = e / d2;
            if (a < 24 && b < 24) continue;
            if (a >= 24 && b >= 24) { P_out[(size_t)a + (size_t)b * ld] = P[(size_t)aug_src2(a, n, nmax, do_aug) + (size_t)aug_src2(b, n, nmax, do_aug) * ld]; continue; }
            const int i = (a < 24) ? a : b;                                   // row of Vk
            const double* pc = P + (size_t)aug_src2((a < 24) ? b : a, n, nmax, do_aug) * ld;   // source clone column (P symmetric)
            if (i >= 15) P_out[(size_t)a + (size_t)b * ld] = pc[i];
            else if (i >= 9) P_out[(size_t)a + (size_t)b * ld] = 0.0;
            else if (e == e0) {
                const int b3 = 3 * (i / 3);
                pr[0] = pc[b3]; pr[1] = pc[b3 + 1]; pr[2] = pc[b3 + 2];
#pragma unroll
                for (int k = 0; k < 6; ++k) pr[3 + k] = pc[9 + k];
                pre = true;
            }
        }
    }
    __syncthreads();
    if (tid < 64) {   // one wave builds Vk (System.cc:344-353) and the composed head of the state
        const q4 qG = ldq(xs), qk = ldq(xs + 10);
        const d3 pG = ld3(xs + 4), pk = ld3(xs + 14);
        const m33 Rk = q2r(qk);
        const d3 gk = unit3(mv33(Rk, ld3(xs + 7)));
        const d3 pkG = mv33(Rk, sub3(pG, pk));
        if (tid == 0) {
            const m33 spx = skew33(pkG), sgx = skew33(gk);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    Vk[i][j] = Rk.m[3 * i + j];           Vk[i][9 + j] = (i == j) ? 1.0 : 0.0;
                    Vk[3 + i][3 + j] = Rk.m[3 * i + j];   Vk[3 + i][9 + j] = spx.m[3 * i + j];   Vk[3 + i][12 + j] = -Rk.m[3 * i + j];
                    Vk[6 + i][6 + j] = Rk.m[3 * i + j];   Vk[6 + i][9 + j] = sgx.m[3 * i + j];
                }
#pragma unroll
            for (int i = 15; i < 24; ++i) Vk[i][i] = 1.0;
            if (blockIdx.x == 0) {
                const q4 qkG = qmul(qk, qG);
                stq(xo, qkG); st3(xo + 4, pkG); st3(xo + 7, gk);
                xo[10] = 0; xo[11] = 0; xo[12] = 0; xo[13] = 1; xo[14] = 0; xo[15] = 0; xo[16] = 0;
                const d3 pGk = mv33(tr33(q2r(qG)), sub3(pk, pG));   // pose line (System.cc:371-374)
                st3(pose_out, pGk); stq(pose_out + 3, qkG);
            }
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        // Tm = Vk P11, P11' = Tm Vk^T: rows 0..8 of Vk have non-zeros in their own 3-block and in columns 9..14 only, rows 9..14 are zero, rows 15..23
        // unit rows — the sums run over those columns in ascending order (the order of the full 24-term loop without its exact-zero terms)
        for (int e = tid; e < 576; e += 256) {
            const int i = e / 24, j = e % 24;
            double a = 0;
            if (i >= 15) a = P11[i][j];
            else if (i < 9) {
                const int b3 = 3 * (i / 3);
#pragma unroll
                for (int k = 0; k < 3; ++k) a += Vk[i][b3 + k] * P11[b3 + k][j];
#pragma unroll
                for (int k = 9; k < 15; ++k) a += Vk[i][k] * P11[k][j];
            }
            Tm[i][j] = a;
        }
        __syncthreads();
        for (int e = tid; e < 576; e += 256) {
            const int i = e / 24, j = e % 24;
            double a = 0;
            if (j >= 15) a = Tm[i][j];
            else if (j < 9) {
                const int b3 = 3 * (j / 3);
#pragma unroll
                for (int k = 0; k < 3; ++k) a += Tm[i][b3 + k] * Vk[j][b3 + k];
#pragma unroll
                for (int k = 9; k < 15; ++k) a += Tm[i][k] * Vk[j][k];
            }
            P11[i][j] = a;
        }
        __syncthreads();
        for (int e = tid; e < 576; e += 256) { int i = e / 24, j = e % 24; P_out[(size_t)i + (size_t)j * ld] = .5 * (P11[i][j] + P11[j][i]); }
        // state: augmentation (System.cc:282-287,303-306) then composition (System.cc:360-365)
        for (int i = tid; i < xd2; i += 256) x_out[i] = (i < 17) ? xo[i] : xv;
    } else {
        for (int e = e0; e < total; e += gstride) {
            const int a = e % d2, b = e / d2;
            if ((a < 24) == (b < 24)) continue;
            const int i = (a < 24) ? a : b;
            if (i >= 9) continue;
            const int b3 = 3 * (i / 3);
            if (!(pre && e == e0)) {
                const double* pc = P + (size_t)aug_src2((a < 24) ? b : a, n, nmax, do_aug) * ld;
                pr[0] = pc[b3]; pr[1] = pc[b3 + 1]; pr[2] = pc[b3 + 2];
#pragma unroll
                for (int k = 0; k < 6; ++k) pr[3 + k] = pc[9 + k];
            }
            double acc = Vk[i][b3] * pr[0] + Vk[i][b3 + 1] * pr[1] + Vk[i][b3 + 2] * pr[2];
            acc += Vk[i][9] * pr[3] + Vk[i][10] * pr[4] + Vk[i][11] * pr[5];
            if (b3 == 3) acc += Vk[i][12] * pr[6] + Vk[i][13] * pr[7] + Vk[i][14] * pr[8];
            P_out[(size_t)a + (size_t)b * ld] = acc;
        }
    }
    DBG_R(blockIdx.x == 0, 6);
    // done (single instance): every workgroup bumps the device-side completion counter of the filter chain — the frame's filter has
    // finished when all of them have (bookkeep_a_kernel of frame k+2 polls it instead of waiting for an event behind this kernel)
    if (done) stage_signal(done);
}
