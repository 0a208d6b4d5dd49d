// filter_kernels.hip — hand-written gfx950 kernels for the filter half of the
// R-VIO hot path.
//
//   feat_build_kernel    Updater::update per-feature loop  Updater.cc:109-455 (U1..U5) + the feature's share of the information block
//   geom4_kernel         U1 + U2 of a batch handle, four features per wave (one per 16-lane DPP row), ahead of feat_build_kernel<4>
//   gram_reduce_kernel / block_sum_kernel / gram_reduce_batch_kernel (batch handles: one workgroup per instance, the sum in LDS, stored tiles only)
//                        measurement compression (Updater.cc:469-536) in information form [A|b] = Hw^T [Hw | r], with the
//                        reference's rank truncation (Updater.cc:516-529) in its structural form   
//   gemm_T_kernel        FP64-MFMA (v_mfma_f64_16x16x4_f64) tiled GEMM:  T = s2 I + A Pcc
//   gemm_T_lds_kernel    the same for a batch handle (>= 128 instances, 6n <= 64): one workgroup per instance, operands staged in LDS
//   ug_kernel            U = Pc W, G = U A, P1 = P - G Pc^T   (FP64 MFMA, one 16-row strip / WG; U, G stored k-major for final_kernel)
//   final_kernel         P+ = sym( P1 - P1c G^T + s2 G U^T )   (Joseph form, Updater.cc:615-619)
//   joseph_batch_kernel  both stages for a batch handle (>= 128 instances, 6n <= 60): one workgroup per instance from P to P+, U / G / P1c in LDS
//   ug_lds_kernel, final_lds_kernel
//                        the same two stages for ONE instance with 6n <= 64: every operand of a workgroup staged in LDS by one batch of loads
//   joseph_lds_kernel    (round 4) both stages of one instance with 6n <= 60 in ONE launch: a workgroup per tile pair of P+ recomputes the strips it needs
// (propagate / augmentation + composition: filter_kernels2.hip; T, W = T^-1, dx, state injection: solve7.hip — solve6.hip behind
//  gemm_T_kernel for batch handles and as A/B forms)
//
// Design rules learnt from the first profile (profiles/r01_a): every kernel front-loads its
// global reads in one batch (a dependent global load after a kernel boundary costs 1-2 us),
// n_clones is a kernel argument (data-independent, mirrored on the host), wave reductions use
// DPP, serial chains are hoisted out of barrier-separated loops.
#include <type_traits>
#include "rvio_dev.h"
#include "../../include/rvio_hip.h"
#include "solve9.hip"   // (the dx / state-injection roles of the split solve ride in ug_tile_kernel<0>)

#include "literal.h"   // round 6: the literal Givens sweep + rank scan for small stacks (lit_decide / lit_finish)
__device__ const double kChi2Dev[500] = {
#include "chi2_table.inc"
};

typedef double d4 __attribute__((ext_vector_type(4)));

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
// reciprocal of a positive normal double: hardware estimate + two Newton steps (the IEEE division sequence is three times as long)
__device__ __forceinline__ double fast_rcp(double v) {
    double y = __builtin_amdgcn_rcp(v);
    y = fma(fma(-v, y, 1.0), y, y);
    return fma(fma(-v, y, 1.0), y, y);
}
// sin and cos of a bearing angle (round 6): one Cody-Waite step to |r| <= pi/4 and the two minimax kernels of the public fdlibm (k_sin.c / k_cos.c
// coefficients, < 1 ulp each) — ~40 instructions against the ~215 of the library's double-double sincos, which the LM loop ran once per iteration and
// the Jacobian phase twice (a quarter of an LM iteration).  The angles of a valid feature stay inside +-pi/2; anything the iteration throws outside
// 1e5 (or a NaN) takes the library routine, so a diverging feature ends the way it did.  Differences from the library: last-bit.
__device__ __forceinline__ void sincos_fast(double x, double* sp, double* cp) {
    if (!(fabs(x) < 1e5)) { sincos(x, sp, cp); return; }
    const double kf = rint(x * 6.36619772367581382433e-01);
    double r = fma(-kf, 1.57079632679489655800e+00, x);
    r = fma(-kf, 6.12323399573676603587e-17, r);
    const int q = (int)kf & 3;
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06); ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03); ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sr = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07); pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03); pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + (z * z) * pc);
    const double s = (q & 1) ? cr : sr, c = (q & 1) ? sr : cr;
    *sp = (q & 2) ? -s : s;
    *cp = ((q + 1) & 2) ? -c : c;
}

// U1 relative-pose chain (Updater.cc:114-141) of a track of at most 16 observations as a PREFIX SCAN inside one 16-lane DPP row (round 6): lane l
// holds clone l's rotation R_l = R(q_l) and position p_l, i.e. the affine map A_l(x) = R_l (x - p_l) = R_l x + c_l; the chain R_I(l) = R_l R_I(l-1),
// t_I(l) = R_l (t_I(l-1) - p_l) is the composition A_l o ... o A_0 applied to 0, and compositions associate: four Hillis-Steele steps (row_shr 1, 2, 4, 8
// through DPP) instead of up to 15 serial 3 x 3 products with an LDS round trip each (3.2 us of the per-feature stage's ~26 in situ).  The association
// differs from the serial order: rounding-level differences (1e-16), like the reference's own quaternion / rotation round trips.  Used by the single-
// stream kernel (wave 0, row 0) and by geom4_kernel (a feature per row): the same expressions in both.
template <int CTRL>
__device__ __forceinline__ void chain_step(m33& R, d3& c, bool take) {
    m33 Rp; d3 cp;
#pragma unroll
    for (int k = 0; k < 9; ++k) Rp.m[k] = dpp_f64<CTRL>(R.m[k]);
    cp.x = dpp_f64<CTRL>(c.x); cp.y = dpp_f64<CTRL>(c.y); cp.z = dpp_f64<CTRL>(c.z);
    if (take) { c = add3(mv33(R, cp), c); R = mul33(R, Rp); }     // (R, c) o (Rp, cp): the later map after the earlier one
}
__device__ __forceinline__ void pose_chain_row16(m33& R, d3& c, int l) {      // in: R_l, c_l = -R_l p_l; out: R_I(l), t_I(l)
    chain_step<0x111>(R, c, l >= 1);
    chain_step<0x112>(R, c, l >= 2);
    chain_step<0x114>(R, c, l >= 4);
    chain_step<0x118>(R, c, l >= 8);
}

// gamma = r^T S^-1 r by the square-root-free L D L^T of S, IN ONE WAVE: lane i holds row i of the lower triangle (lane rr: the residual as an extra
// row) in NMAX registers, the pivot row's entries reach the other lanes by v_readlane, no LDS traffic and no barrier between the rr pivots
// (round 6).  The thread-per-element form below it costs a barrier, five LDS loads per element and two reciprocals per pair of pivots on every
// wave of the workgroup: ~2200 wave-instructions per feature at rr = 17 against ~800 here — a fifth of the per-feature kernel's instructions in
// the batch, where it is instruction-issue bound, and ~1.5 us of the single stream's serial chain.  Same recurrence (columns unscaled:
// S[i][k] = l_ik d_k, the residual row carries w = L^-1 r, gamma = sum w_k^2 / d_k); the sums run pivot by pivot instead of two at a time.
template <int NMAX>
__device__ __forceinline__ double ldlt_gamma_wave(const double* S, int lds_s, int rr, int lane, bool* bad) {
    double row[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {                 // (clamped address + select: a predicated load costs an exec-mask save / restore each)
        const double v = S[min(lane, rr) * lds_s + min(j, rr)];
        row[j] = (lane <= rr && j <= lane && j < rr) ? v : 0.0;
    }
    double gl = 0;
    bool neg = false;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        if (k < rr) {                                // (uniform; fixed trip counts and constant register indices throughout: the rows stay in registers)
            const double dk = readlane_f64(row[k], k);
            neg |= !(dk > 0);
            const double lik = row[k] * fast_rcp(dk > 0 ? dk : 1e-300);      // l_ik (rows i > k); lane rr: w_k / d_k
            gl += row[k] * lik;                      // lane rr: w_k^2 / d_k
#pragma unroll
            for (int j = k + 1; j < NMAX; ++j) row[j] -= lik * readlane_f64(row[k], j);    // S[i][j] -= l_ik S[j][k]   (slots j > i, j >= rr: never read)
        }
    }
    *bad = neg;
    return readlane_f64(gl, rr);
}

// HOIST: k-values whose operand loads are in flight before the first MFMA of a gate tile (16: one stream, latency; 4: batch handles, 128 VGPRs)
template <int HOIST>
__device__ __forceinline__ void feat_build_body(DevCfg cfg, int n, const double* x, const double* P,
                                                const int* n_feat_ptr, const unsigned char* types, const int* lens, const float* meas,
                                                int shard_rank, int shard_world,
                                                double* Gshare, int* nrows_out, int* acc_out, int* ndof_out, double* gamma_out,
                                                double* pfinv_out, double* tm_global, size_t bs, BatchIn bin, FilterMeta* meta, const int f,
                                                const double* gpose = nullptr, const int* gvalid = nullptr, double* lit_rows = nullptr) {
    extern __shared__ __align__(16) double lds[];
    DBG_P0();
    const BatchIdx bi = batch_plain();
    x = zoffi(x, bs, bi.z); P = zoffi(P, bs, bi.z); Gshare = zoffi(Gshare, bs, bi.z); nrows_out = zoffi(nrows_out, bs, bi.z); acc_out = zoffi(acc_out, bs, bi.z);
    ndof_out = zoffi(ndof_out, bs, bi.z); gamma_out = zoffi(gamma_out, bs, bi.z); pfinv_out = zoffi(pfinv_out, bs, bi.z);
    if (tm_global) tm_global = zoffi(tm_global, bs, bi.z);
    if (gpose) { gpose = zoffi(gpose, bs, bi.z); gvalid = zoffi(gvalid, bs, bi.z); }
    if (lit_rows) lit_rows = zoffi(lit_rows, bs, bi.z);
    n_feat_ptr = zoffi(n_feat_ptr, bin.n_feat, bi.z); types = zoffi(types, bin.types, bi.z); lens = zoffi(lens, bin.len, bi.z); meas = zoffi(meas, bin.meas, bi.z);
    const int tid = threadIdx.x, T = blockDim.x;   // f: the feature slot of this pass
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
    // (count, header and observations of slot f go out together — every slot of the tables exists, a slot beyond the count is simply not used —: one
    //  round trip to L2 instead of three dependent ones at the head of every workgroup, ~0.8 us of the launch)
    const int n_feat = *n_feat_ptr;
    const unsigned char type = types[f];
    const int L = lens[f];
    const float* mz = meas + (size_t)f * cfg.max_len * 2;
    float mxv = 0, myv = 0;
    const int lane = threadIdx.x & 63;
    if (lane < cfg.max_len) { mxv = mz[2 * lane]; myv = mz[2 * lane + 1]; }
    const bool xpre_on = HOIST > 4 && !gpose && 7 * n <= (int)blockDim.x;    // one stream: the clone states ride in the same batch of loads
    double xpre = 0;
    if (xpre_on && (int)threadIdx.x < 7 * n) xpre = x[26 + threadIdx.x];
    // an update of at most LIT_FEATS features is not sharded (every rank builds every feature, block 0 alone is used: block_sum_kernel) and
    // exports the accepted features' rows: the reference's literal sweep may have to run on them (literal.h)
    const bool lit = lit_rows && n_feat <= LIT_FEATS;
    if (f >= n_feat || (!lit && (f % shard_world) != shard_rank)) {   // empty slot / another rank's feature: leave before touching anything else
        if (tid == 0) { nrows_out[f] = 0; acc_out[f] = 0; ndof_out[f] = 0; gamma_out[f] = 0; }
        return;
    }
    // a track the window cannot hold (caller-provided device tables, or a tracker that outlived a state reset): drop it and say so
    if (L < 2 || L > cfg.max_len || L - 1 > n || (type != '1' && type != '2')) {
        if (tid == 0) { nrows_out[f] = 0; acc_out[f] = 0; ndof_out[f] = 0; gamma_out[f] = 0; atomicOr(&zoffi(meta, bs, bi.z)->err, 2); }
        return;
    }
    const int nPh = L - 1;
    // batch handles: U1 + U2 were computed by geom4_kernel, four features per wave (the relative-pose chain and the triangulation are
    // single-wave phases with <= 11 of 64 lanes at work; here they are a third of the kernel's instructions) — fetch the poses and the triple
    if (gpose) { const double* gp = gpose + (size_t)f * (ML - 1) * 24; for (int e = tid; e < nPh * 24; e += T) pose[e] = gp[e]; }
    else if (xpre_on) { if (tid < 7 * n) xcl[tid] = xpre; }
    else for (int e = tid; e < 7 * n; e += T) xcl[e] = x[26 + e];
    const bool wave0 = tid < 64;
    const double sig = cfg.sigma_im, sig2 = sig * sig;
    const m33 Ric = ldm33(cfg.Ric), Rci = ldm33(cfg.Rci);
    const d3 tic = ld3(cfg.tic), tci = ld3(cfg.tci);
    DBG_T(30); DBG_P(30);
    __syncthreads();
    DBG_T(31); DBG_P(31);

    // ---- U1 relative-pose chain (Updater.cc:114-141).  R(q_i) for every clone in parallel (lane <-> clone), the chain
    // R_I(i) = R(q_i) R_I(i-1), t_I(i) = R(q_i) (t_I(i-1) - p_i) as a short serial product of 3x3 matrices, then the
    // camera-frame poses in parallel again.  The reference carries the chain as normalised quaternions and passes
    // R_c through RotToQuat/QuatToRot; both are the same rotations up to O(1e-16).
    if (wave0 && !gpose && nPh <= 15) {     // (every window up to 15 clones: the chain as a prefix scan inside the wave's first DPP row, no LDS in between)
        const double* rel = (type == '1') ? (xcl + 7 * n - 7 * nPh) : xcl;
        const int ll = lane < nPh ? lane : 0;
        m33 RI = q2r(ldq(rel + 7 * ll));
        d3 tI = scl3(-1.0, mv33(RI, ld3(rel + 7 * ll + 4)));
        pose_chain_row16(RI, tI, lane);
        if (lane < nPh) {
            double* o = pose + lane * 24;
            const m33 RciRI = mul33(Rci, RI);
            const m33 Rc = mul33(RciRI, Ric);
            const d3 tC = add3(add3(mv33(RciRI, tic), mv33(Rci, tI)), tci);
#pragma unroll
            for (int k = 0; k < 9; ++k) { o[k] = RI.m[k]; o[12 + k] = Rc.m[k]; }
            st3(o + 9, tI); st3(o + 21, tC);
        }
    } else
    if (wave0 && !gpose) {
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
    DBG_T(32); DBG_P(32);

    // ---- U2 inverse-depth LM triangulation (Updater.cc:143-269): lane i <-> observation i
    double phi = 0, psi = 0, rho = 0;
    bool valid = true;
    const float fx0 = __shfl(mxv, 0, 64), fy0 = __shfl(myv, 0, 64);
    if (gpose) {
        if (tid == 0) { misc[0] = pfinv_out[3 * f]; misc[1] = pfinv_out[3 * f + 1]; misc[2] = pfinv_out[3 * f + 2]; misc[3] = gvalid[f] ? 1.0 : 0.0; }
    } else
    if (wave0) {
        {   // one atan2 for both angles (even lanes phi, odd lanes psi), as the sincos below
            const double av = atan2((lane & 1) ? (double)fx0 : (double)fy0, (lane & 1) ? 1.0 : sqrt((double)fx0 * (double)fx0 + 1));
            phi = readlane_f64(av, 0); psi = readlane_f64(av, 1);
        }
        if (fabs(phi) > .5 * 3.14 || fabs(psi) > .5 * 3.14) valid = false;
        // a track of at most 16 observations: EVERY 16-lane row computes observation (lane & 15), and each row reduces its own three of the ten
        // sums (same DPP order inside a row: same bits) — three row reductions per iteration instead of ten
        const bool packed = L <= 16;
        const int ob = packed ? (lane & 15) : lane;
        const bool act = ob < L;
        const float mx = packed ? __shfl(mxv, ob, 64) : mxv, my = packed ? __shfl(myv, ob, 64) : myv;
        m33 Rc = eye33(); d3 tc = mk3(0, 0, 0);
        if (act && ob > 0) { Rc = ldm33(pose + (ob - 1) * 24 + 12); tc = ld3(pose + (ob - 1) * 24 + 21); }
        const double ri = 1. / sig2;
        double lambda = 0.01, lastCost = INFINITY;
        if (valid) {
            for (int it = 0; it < 10; ++it) {
                // one sincos for both angles: even lanes take phi, odd lanes psi (every lane would compute the same pair anyway)
                double sv, cv;
                sincos_fast((lane & 1) ? psi : phi, &sv, &cv);
                const double sph = readlane_f64(sv, 0), cph = readlane_f64(cv, 0), sps = readlane_f64(sv, 1), cps = readlane_f64(cv, 1);
                const d3 ep = mk3(cph * sps, sph, cph * cps);
                const double J00 = -sph * sps, J01 = cph * cps, J10 = cph, J20 = -sph * cps, J21 = -cph * sps;
                double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0, g0 = 0, g1 = 0, g2 = 0, cost = 0;
                if (act) {
                    d3 h = (ob == 0) ? ep : add3(mv33(Rc, ep), scl3(rho, tc));
                    // (reciprocal by estimate + two Newton steps; the five IEEE divisions of this block were a fifth of the iteration)
                    const double iz = fast_rcp(h.z);
                    const double Hp0[3] = {iz, 0, -(h.x * iz) * iz}, Hp1[3] = {0, iz, -(h.y * iz) * iz};
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
                    if (ob == 0) { H0[2] = 0; H1[2] = 0; }
                    else { H0[2] = Hp0[0] * tc.x + Hp0[2] * tc.z; H1[2] = Hp1[1] * tc.y + Hp1[2] * tc.z; }
                    const float px = (float)(h.x * iz), py = (float)(h.y * iz);    // cv::Point2f rounding (Updater.cc:197-202)
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
                if (packed) {    // row 0: cost c00 c01 | row 1: c02 c11 c12 | row 2: c22 g0 g1 | row 3: g2
                    const int rw = lane >> 4;
                    double q0 = rw == 0 ? cost : rw == 1 ? c02 : rw == 2 ? c22 : g2;
                    double q1 = rw == 0 ? c00 : rw == 1 ? c11 : g0;
                    double q2 = rw == 0 ? c01 : rw == 1 ? c12 : g1;
                    q0 = row16_allsum(q0); q1 = row16_allsum(q1); q2 = row16_allsum(q2);
                    cost = readlane_f64(q0, 0); c02 = readlane_f64(q0, 16); c22 = readlane_f64(q0, 32); g2 = readlane_f64(q0, 48);
                    c00 = readlane_f64(q1, 0); c11 = readlane_f64(q1, 16); g0 = readlane_f64(q1, 32);
                    c01 = readlane_f64(q2, 0); c12 = readlane_f64(q2, 16); g1 = readlane_f64(q2, 32);
                } else {
                    cost = wave_sum(cost);
                    c00 = wave_sum(c00); c01 = wave_sum(c01); c02 = wave_sum(c02);
                    c11 = wave_sum(c11); c12 = wave_sum(c12); c22 = wave_sum(c22);
                    g0 = wave_sum(g0); g1 = wave_sum(g1); g2 = wave_sum(g2);
                }
                if (cost <= lastCost) {
                    // damped normal equations, SPD 3x3: Cholesky solve (reference: colPivHouseholderQr, Updater.cc:239)
                    // L D L^T (square-root free): 3 reciprocals instead of 3 sqrt + 9 divisions
                    // A pivot that is zero against the largest diagonal entry is a dependent direction — the inverse-depth column is exactly
                    // zero when every relative translation of the track is (a platform at rest fed an empty IMU batch): the reference's
                    // rank-revealing QR leaves that component of the step at 0; a reciprocal here would turn it into NaN.
                    const double a00 = c00 + lambda * c00, a11 = c11 + lambda * c11, a22 = c22 + lambda * c22;
                    const double ptiny = 2.220446049250313e-16 * fmax(a00, fmax(a11, a22));
                    const double i0 = a00 > ptiny ? fast_rcp(a00) : 0.0, l10 = c01 * i0, l20 = c02 * i0;
                    const double dd1 = a11 - l10 * c01, i1 = dd1 > ptiny ? fast_rcp(dd1) : 0.0, l21 = (c12 - l20 * c01) * i1;
                    const double dd2 = a22 - l20 * c02 - l21 * (c12 - l20 * c01), i2 = dd2 > ptiny ? fast_rcp(dd2) : 0.0;
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
    DBG_T(33); DBG_P(33);
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
    {
        double sv, cv;
        sincos_fast((tid & 1) ? psi : phi, &sv, &cv);
        sph = readlane_f64(sv, 0); cph = readlane_f64(cv, 0); sps = readlane_f64(sv, 1); cps = readlane_f64(cv, 1);
    }
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
    DBG_T(34); DBG_P(34);
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
    DBG_T(35); DBG_P(35);
    if (lit) {   // the RAW block [Hx | r] (rows 0..M2-1, columns [cLo, cHi) and the residual column c6) and Hf, before the projection below: literal.h
        double* lr = lit_rows + (size_t)f * M2max * ldh;
        double* lh = lit_rows + (size_t)LIT_FEATS * M2max * ldh + (size_t)f * M2max * 3;
        const int wraw = cHi - cLo + 1;
        for (int e = tid; e < M2 * wraw; e += T) {
            const int i = e / wraw, k = e - i * wraw, col = (k < wraw - 1) ? cLo + k : c6;
            lr[(size_t)i * ldh + col] = Hx[(size_t)i * ldh + col];
        }
        for (int e = tid; e < M2 * 3; e += T) lh[e] = hf[e];
        __syncthreads();
    }
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
    DBG_T(36); DBG_P(36);
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
            // a dependent global load per k-step costs ~1.5 us here: all loads of 64 k-values go out before the first MFMA
            for (int k0 = 0; k0 < wa; k0 += 4 * HOIST) {
                double a[HOIST], b[HOIST];
#pragma unroll
                for (int u = 0; u < HOIST; ++u) {
                    const int k = k0 + 4 * u + lk;
                    a[u] = (aok && k < wa) ? ap[k] : 0.0;
                    b[u] = (bok && k < wa) ? bp[(size_t)k * ld] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < HOIST; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = it * 16 + lk + 4 * r;
                if (row < rr && bj < wa) Tm[(size_t)row * ldh + cLo + bj] = acc[r];
            }
        }
    }
    __syncthreads();
    DBG_T(37); DBG_P(37);
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
    DBG_T(38); DBG_P(38);
    // gamma = |r^T S^-1 r| by a square-root-free L D L^T of S with the residual row appended (reference:
    // colPivHouseholderQr().solve, Updater.cc:420).  Columns stay unscaled (S[i][k] = l_ik d_k), so the residual row
    // carries w = L^-1 r and gamma = sum_k w_k^2 / d_k.  Thread <-> element (i, j) of the lower triangle (+ residual row):
    // one rank-1 update step per barrier, every element touched once per step.
    double gam = 0;
    if (rr <= 20) {       // (tracks of at most 11 observations: every feature of the stock 10-clone window)
        if (wave0) {
            bool bad = false;
            const double g = (rr <= 12) ? ldlt_gamma_wave<12>(S, lds_s, rr, lane, &bad) : ldlt_gamma_wave<20>(S, lds_s, rr, lane, &bad);
            if (lane == 0) {
                misc[8] = fabs(g);
                // S_f = Hn Pcc Hn^T + s2 I is positive definite by construction; a non-positive pivot means the covariance handed in is not.
                // The feature is rejected (gamma overflows the table) where the reference's pivoted QR would return some finite gamma: say so.
                if (bad) atomicOr(&zoffi(meta, bs, bi.z)->err, 8);
            }
        }
    } else
    {
        // TWO pivots per barrier: a rank-1 step is latency (an LDS round trip, a reciprocal, a barrier: ~1000 cycles whatever the size), so
        // columns k and k+1 go together.  With a = S[k][k], b = S[k+1][k], c' = S[k+1][k+1] - b^2 / a and u_i = S[i][k+1] - S[i][k] b / a:
        //   S[i][j] -= S[i][k] S[j][k] / a + u_i u_j / c'        gamma += w_k^2 / a + u_w^2 / c'
        // — the two sequential steps written out.  All loads of a step are issued before its first store (the elements are distinct, but
        // the compiler cannot know and would otherwise order every slot's store before the next slot's loads).
        double gsum = 0;
        const int tot = (rr + 1) * rr;
        int ei[4], ej[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * T;
            const int i = e / rr, j = e - i * rr;
            const bool low = e < tot && j <= (i < rr ? i : rr - 1);
            ei[q] = low ? i : -1; ej[q] = j;
        }
        int k = 0;
        for (; k + 1 < rr; k += 2) {
            const double a = S[k * lds_s + k], b = S[(k + 1) * lds_s + k], c = S[(k + 1) * lds_s + k + 1];
            double sv[4], si0[4], si1[4], sj0[4], sj1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool on = ei[q] > k + 1 && ej[q] > k + 1;
                sv[q] = on ? S[ei[q] * lds_s + ej[q]] : 0.0;
                si0[q] = on ? S[ei[q] * lds_s + k] : 0.0; si1[q] = on ? S[ei[q] * lds_s + k + 1] : 0.0;
                sj0[q] = on ? S[ej[q] * lds_s + k] : 0.0; sj1[q] = on ? S[ej[q] * lds_s + k + 1] : 0.0;
            }
            const double ra = fast_rcp(a > 0 ? a : 1e-300), ba = b * ra, c2 = c - b * ba, rc = fast_rcp(c2 > 0 ? c2 : 1e-300);
            if (tid == 0) {
                const double w0 = S[rr * lds_s + k], w1 = S[rr * lds_s + k + 1] - w0 * ba;
                gsum += w0 * (w0 * ra) + w1 * (w1 * rc);
                // S_f = Hn Pcc Hn^T + s2 I is positive definite by construction; a non-positive pivot means the covariance handed in is not.
                // The feature is rejected (gamma overflows the table) where the reference's pivoted QR would return some finite gamma: say so.
                if (!(a > 0) || !(c2 > 0)) atomicOr(&zoffi(meta, bs, bi.z)->err, 8);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ei[q] > k + 1 && ej[q] > k + 1) {
                    const double ui = si1[q] - si0[q] * ba, uj = sj1[q] - sj0[q] * ba;
                    S[ei[q] * lds_s + ej[q]] = (sv[q] - (si0[q] * ra) * sj0[q]) - (ui * rc) * uj;
                }
            for (int e = tid + 4 * T; e < tot; e += T) {     // long tracks on narrow workgroups
                const int i = e / rr, j = e - i * rr;
                if (i > k + 1 && j > k + 1 && j <= (i < rr ? i : rr - 1)) {
                    const double ui = S[i * lds_s + k + 1] - S[i * lds_s + k] * ba, uj = S[j * lds_s + k + 1] - S[j * lds_s + k] * ba;
                    S[i * lds_s + j] = (S[i * lds_s + j] - (S[i * lds_s + k] * ra) * S[j * lds_s + k]) - (ui * rc) * uj;
                }
            }
            __syncthreads();
        }
        if (k < rr && tid == 0) {     // odd count: the last pivot only feeds gamma
            const double dk = S[k * lds_s + k], w = S[rr * lds_s + k];
            gsum += w * (w * fast_rcp(dk > 0 ? dk : 1e-300));
            if (!(dk > 0)) atomicOr(&zoffi(meta, bs, bi.z)->err, 8);
        }
        if (tid == 0) misc[8] = fabs(gsum);
    }
    DBG_T(39); DBG_P(39);
    __syncthreads();
    gam = misc[8];
    const bool accept = gam < kChi2Dev[rr - 1];
    if (tid == 0) { acc_out[f] = accept ? 1 : 0; ndof_out[f] = rr; gamma_out[f] = gam; nrows_out[f] = accept ? rr : 0; }
    if (accept) {
        // this feature's share of the information block, G_f = Hn^T [Hn | r] (rows 0..c6-1, columns 0..c6), on the FP64 matrix cores
        // straight from the LDS copy of Hn:  A[i = p][k = row] = Hn[row][p0 + i],  B[k = row][j = q] = Hn[row][q0 + j]; rows >= rr are zero.
        // gram_reduce_kernel adds the accepted features' shares in feature order (Updater.cc:469-536 in information form).
        // Only tiles (pt, qt) with qt >= pt whose columns meet the feature's range [cLo, cHi) — plus, in the tile row of such a pt, the
        // tile that holds the residual column c6 — are computed and stored (tile-granular: gram_reduce_kernel applies the same rule
        // and mirrors the lower triangle once at the end).
        double* out = Gshare + (size_t)f * ldh * ldh;
        const int gw = tid >> 6, gl = tid & 63, gi = gl & 15, gk = gl >> 4, nw = T >> 6;
        const int t0 = cLo >> 4, t1 = (cHi - 1) >> 4, tr = c6 >> 4;      // first / last tile of the range, tile of the residual column
        const int nts = t1 - t0 + 1, nq = nts + ((tr > t1) ? 1 : 0);
        for (int tile = gw; tile < nts * nq; tile += nw) {
            const int pt = t0 + tile / nq, qi = tile % nq, qt = (qi < nts) ? t0 + qi : tr;
            if (qt < pt) continue;   // the tiles on and above the diagonal only: the reduction mirrors the sum once
            const int p0 = pt * 16, q0 = qt * 16;
            const bool pok = p0 + gi < c6, qok = q0 + gi <= c6;
            d4 acc = {0, 0, 0, 0};
            for (int k0 = 0; k0 < rr; k0 += 4) {
                const int row = k0 + gk;
                const double* hr = Hn + (size_t)(row < rr ? row : 0) * ldh;
                const double a = (row < rr && pok) ? hr[p0 + gi] : 0.0;
                const double b = (row < rr && qok) ? hr[q0 + gi] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
            if (qok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pp = p0 + gk + 4 * r;
                    if (pp < c6) out[(size_t)pp * ldh + q0 + gi] = acc[r];
                }
            }
        }
    }
    DBG_T(40); DBG_P(40);
}

template <int HOIST>
__global__ __launch_bounds__(HOIST > 4 ? 256 : 1024) void feat_build_kernel(DevCfg cfg, int n, const double* x, const double* P,
                                  const int* n_feat_ptr, const unsigned char* types, const int* lens, const float* meas,
                                  int shard_rank, int shard_world,
                                  double* Gshare, int* nrows_out, int* acc_out, int* ndof_out, double* gamma_out,
                                  double* pfinv_out, double* tm_global, size_t bs, BatchIn bin, FilterMeta* meta,
                                  const double* gpose, const int* gvalid, double* lit_rows) {
    feat_build_body<HOIST>(cfg, n, x, P, n_feat_ptr, types, lens, meas, shard_rank, shard_world, Gshare, nrows_out, acc_out, ndof_out, gamma_out, pfinv_out,
                           tm_global, bs, bin, meta, (int)blockIdx.x, gpose, gvalid, lit_rows);
}

// =============================================================== U1 + U2, four features per wave (batch handles, max_len <= 16)
// The relative-pose chain and the inverse-depth triangulation of feat_build_kernel are single-wave phases with one lane per observation:
// at most 16 of 64 lanes work, and at B = 2048 the kernel is instruction-issue bound.  Here a wave carries FOUR features, one per 16-lane
// DPP row (lane l of row g <-> observation l of feature 4 blockIdx.x + g): the same expressions in the same order per feature — the
// row reductions are the DPP row rotations of row16_sum, the broadcasts come from the row's own lanes 0 / 1 —, so the results are those
// of feat_build_kernel bit for bit; the rows iterate until the last one has converged (a converged row is frozen).  Output: the camera /
// IMU poses of the chain (gpose[f][(L-1) x 24]), the triple (pfinv[f]) and the validity flag, which feat_build_kernel<4> then reads.
__device__ __forceinline__ double row_bcast(double v, int src) { return __shfl(v, (int)((threadIdx.x & 48) + src), 64); }
__device__ __forceinline__ double row16_sum_b(double v) {
    v += dpp_f64<0x128>(v);
    v += dpp_f64<0x124>(v);
    v += dpp_f64<0x122>(v);
    v += dpp_f64<0x121>(v);
    return row_bcast(v, 0);
}
#define GEOM4_ML 16
__global__ __launch_bounds__(64) void geom4_kernel(DevCfg cfg, int n, const double* __restrict__ x, const int* __restrict__ n_feat_ptr,
                                                   const unsigned char* __restrict__ types, const int* __restrict__ lens, const float* __restrict__ meas,
                                                   double* __restrict__ gpose, double* __restrict__ pfinv_out, int* __restrict__ gvalid, size_t bs, BatchIn bin) {
    const int z = blockIdx.z;
    x = zoffi(x, bs, z); gpose = zoffi(gpose, bs, z); pfinv_out = zoffi(pfinv_out, bs, z); gvalid = zoffi(gvalid, bs, z);
    n_feat_ptr = zoffi(n_feat_ptr, bin.n_feat, z); types = zoffi(types, bin.types, z); lens = zoffi(lens, bin.len, z); meas = zoffi(meas, bin.meas, z);
    __shared__ double xcl[7 * (GEOM4_ML - 1)];
    __shared__ double poses[4][(GEOM4_ML - 1) * 24];
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, ML = cfg.max_len;
    const int n_feat = *n_feat_ptr;
    if (4 * (int)blockIdx.x >= n_feat) return;                       // four empty slots
    const int f = 4 * blockIdx.x + g;
    unsigned char type = '1'; int L = 2;
    bool active = f < n_feat;
    if (active) { type = types[f]; L = lens[f]; }
    if (L < 2 || L > ML || L - 1 > n || (type != '1' && type != '2')) { active = false; L = 2; }   // (feat_build_kernel drops such a track and says so)
    const int nPh = L - 1;
    float mxv = 0, myv = 0;
    if (active && l < L) { const float* mz = meas + (size_t)f * ML * 2; mxv = mz[2 * l]; myv = mz[2 * l + 1]; }
    for (int e = lane; e < 7 * n; e += 64) xcl[e] = x[26 + e];
    const double sig = cfg.sigma_im, sig2 = sig * sig;
    const m33 Ric = ldm33(cfg.Ric), Rci = ldm33(cfg.Rci);
    const d3 tic = ld3(cfg.tic), tci = ld3(cfg.tci);
    __syncthreads();
    double* pose = poses[g];
    // ---- U1 (feat_build_kernel, same expressions: the chain as a prefix scan inside the feature's 16-lane row)
    const double* rel = (type == '1') ? (xcl + 7 * n - 7 * nPh) : xcl;
    {
        const int ll = (active && l < nPh) ? l : 0;
        m33 RI = q2r(ldq(rel + 7 * ll));
        d3 tI = scl3(-1.0, mv33(RI, ld3(rel + 7 * ll + 4)));
        pose_chain_row16(RI, tI, l);
        if (active && l < nPh) {
            double* o = pose + l * 24;
            const m33 RciRI = mul33(Rci, RI);
            const m33 Rc = mul33(RciRI, Ric);
            const d3 tC = add3(add3(mv33(RciRI, tic), mv33(Rci, tI)), tci);
#pragma unroll
            for (int k = 0; k < 9; ++k) { o[k] = RI.m[k]; o[12 + k] = Rc.m[k]; }
            st3(o + 9, tI); st3(o + 21, tC);
        }
    }
    __syncthreads();
    // ---- U2 (feat_build_kernel, same expressions; lane l of the row <-> observation l)
    const float fx0 = __shfl(mxv, lane & 48, 64), fy0 = __shfl(myv, lane & 48, 64);
    double phi, psi;
    {
        const double av = atan2((l & 1) ? (double)fx0 : (double)fy0, (l & 1) ? 1.0 : sqrt((double)fx0 * (double)fx0 + 1));
        phi = row_bcast(av, 0); psi = row_bcast(av, 1);
    }
    double rho = 0;
    bool valid = true;
    if (fabs(phi) > .5 * 3.14 || fabs(psi) > .5 * 3.14) valid = false;
    {
        const bool act = active && l < L;
        const float mx = mxv, my = myv;
        m33 Rc = eye33(); d3 tc = mk3(0, 0, 0);
        if (act && l > 0) { Rc = ldm33(pose + (l - 1) * 24 + 12); tc = ld3(pose + (l - 1) * 24 + 21); }
        const double ri = 1. / sig2;
        double lambda = 0.01, lastCost = INFINITY;
        bool done = !(active && valid);
        for (int it = 0; it < 10; ++it) {
            if (__builtin_amdgcn_readfirstlane(__any(!done)) == 0) break;
            double sv, cv;
            sincos_fast((l & 1) ? psi : phi, &sv, &cv);
            const double sph = row_bcast(sv, 0), cph = row_bcast(cv, 0), sps = row_bcast(sv, 1), cps = row_bcast(cv, 1);
            const d3 ep = mk3(cph * sps, sph, cph * cps);
            const double J00 = -sph * sps, J01 = cph * cps, J10 = cph, J20 = -sph * cps, J21 = -cph * sps;
            double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0, g0 = 0, g1 = 0, g2 = 0, cost = 0;
            if (act) {
                d3 h = (l == 0) ? ep : add3(mv33(Rc, ep), scl3(rho, tc));
                const double iz = fast_rcp(h.z);
                const double Hp0[3] = {iz, 0, -(h.x * iz) * iz}, Hp1[3] = {0, iz, -(h.y * iz) * iz};
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
                if (l == 0) { H0[2] = 0; H1[2] = 0; }
                else { H0[2] = Hp0[0] * tc.x + Hp0[2] * tc.z; H1[2] = Hp1[1] * tc.y + Hp1[2] * tc.z; }
                const float px = (float)(h.x * iz), py = (float)(h.y * iz);
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
            cost = row16_sum_b(cost);
            c00 = row16_sum_b(c00); c01 = row16_sum_b(c01); c02 = row16_sum_b(c02);
            c11 = row16_sum_b(c11); c12 = row16_sum_b(c12); c22 = row16_sum_b(c22);
            g0 = row16_sum_b(g0); g1 = row16_sum_b(g1); g2 = row16_sum_b(g2);
            if (!done) {
                if (cost <= lastCost) {
                    const double a00 = c00 + lambda * c00, a11 = c11 + lambda * c11, a22 = c22 + lambda * c22;
                    const double ptiny = 2.220446049250313e-16 * fmax(a00, fmax(a11, a22));
                    const double i0 = a00 > ptiny ? fast_rcp(a00) : 0.0, l10 = c01 * i0, l20 = c02 * i0;
                    const double dd1 = a11 - l10 * c01, i1 = dd1 > ptiny ? fast_rcp(dd1) : 0.0, l21 = (c12 - l20 * c01) * i1;
                    const double dd2 = a22 - l20 * c02 - l21 * (c12 - l20 * c01), i2 = dd2 > ptiny ? fast_rcp(dd2) : 0.0;
                    const double z0 = g0, z1 = g1 - l10 * z0, z2 = g2 - l20 * z0 - l21 * z1;
                    const double d2 = z2 * i2, d1 = z1 * i1 - l21 * d2, d0 = z0 * i0 - l10 * d1 - l20 * d2;
                    phi += d0; psi += d1; rho += d2;
                    if (fabs(lastCost - cost) < 1e-6 && d2 < 1e-6) done = true;
                    else { lambda *= .1; lastCost = cost; }
                } else { lambda *= 10; lastCost = cost; }
            }
        }
        if (valid && (fabs(phi) > .5 * 3.14 || fabs(psi) > .5 * 3.14 || isinf(rho) || rho < 0 || isnan(rho) || isnan(phi) || isnan(psi))) valid = false;
    }
    if (active) {
        if (l == 0) { pfinv_out[3 * f] = phi; pfinv_out[3 * f + 1] = psi; pfinv_out[3 * f + 2] = rho; gvalid[f] = valid ? 1 : 0; }
        double* gp = gpose + (size_t)f * (ML - 1) * 24;
        for (int e = l; e < nPh * 24; e += 16) gp[e] = pose[e];
    }
}

// =============================================================== U7 compression, information form (reduction stage)
// partial[f][p][q] = sum over the rows of accepted feature f of H[row][p] * H[row][q], q = 0..c6 (column c6 is the residual -> b),
// is produced by the epilogue of feat_build_kernel; the kernels below reduce it.
//
// The reference's leading-row rank scan (Updater.cc:516-529) in its structural form.  The Givens sweep treats exact zeros
// specially (makeGivens(0,q) swaps, makeGivens(p,0) is the identity), so as long as every accepted type-'1' feature starts behind
// the last column e2 = 6(ceil(L/2)-1)-1 of the type-'2' features the two families are never mixed while columns 0..e2 are swept.
// The type-'2' block has the scale gauge of a monocular window as null direction: its column e2 is dependent, a left-over row of
// rounding residue reaches position e2, the scan stops there (nRank = e2) and the type-'1' rows are discarded.  In every other
// constellation the scan only drops rows that are zero to rounding (derivation and the CPU proof against the literal sweep:
// oracle/filter.cpp above orc_update_local, tests/test_truncation.py).  So the type-'2' sum S2 and the type-'1' sum S1 are kept
// apart, and [A|b] = S2 if   min start column of type '1' > e2,   rows of type '2' >= e2+1,   the stack is tall (rows > 6n) and
// the Schur complement of S2[e2][e2] w.r.t. columns 0..e2-1 is < (1e-4)^2 (the scan's row-norm threshold);   else S2 + S1.
//
// block = the payload of one shard: part 0 = S2, part 1 = S1 (each c6 x ldh row-major inside an ldh x ldh square); the spare
// last row of part 0 carries {n_good, n_rows, rows of type '2', e2 (-1: none), min start column of type '1' (TR_NONE: none)}.
#define GRAM_MAX_FEATS 2048
#define TR_NONE 1000000000
__host__ __device__ inline int trunc_mmax(int max_len) { return 6 * ((max_len + 1) / 2 - 1); }   // largest e2 + 1
__host__ __device__ inline size_t trunc_lds_doubles(int max_len) { const size_t m = trunc_mmax(max_len); return m * (m | 1) + m + 8; }

// data written by other workgroups of the same launch: read it from L2 (agent scope), not through this CU's vector cache
__device__ __forceinline__ double ld_l2(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// true for exactly one workgroup of the launch (per instance): the one that finishes last.  cnt is re-armed for the next launch.
__device__ __forceinline__ bool last_block_done(int* cnt, int nblocks) {
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = atomicAdd(cnt, 1);
        s_last = (t == nblocks - 1);
        if (s_last) *cnt = 0;
    }
    __syncthreads();
    const bool last = s_last != 0;
    if (last) __threadfence();
    return last;
}

// The decision above and the final sum, by one workgroup of 256 threads: A holds S2 on entry and [A|b] on exit.  Msh: trunc_lds_doubles().
__device__ void trunc_finish(const DevCfg& cfg, int n, double* A, const double* S1, int good, int rows, int rows2, int e2, int smin, double* Msh) {
    const int c6 = 6 * n, ldh = cfg.ldh, tid = threadIdx.x;
    bool truncate = false;
    if (good > 2 && rows > c6 && e2 >= 0 && e2 < c6 && smin < TR_NONE && smin > e2 && rows2 >= e2 + 1) {
        // square-root-free unpivoted elimination of columns 0..e2-1 on the lower triangle of S2[0..e2][0..e2]; a column whose pivot has
        // cancelled to rounding level is skipped (the sweep's mixture row leaves the span of the later columns alone)
        const int m = e2 + 1, ldm = m | 1;
        double* M = Msh; double* d0 = Msh + (size_t)trunc_mmax(cfg.max_len) * (trunc_mmax(cfg.max_len) | 1);
        for (int e = tid; e < m * m; e += 256) { const int i = e / m, j = e - i * m; if (j <= i) M[i * ldm + j] = ld_l2(A + (size_t)j * ldh + i); }
        __syncthreads();
        for (int i = tid; i < m; i += 256) d0[i] = M[i * ldm + i];
        __syncthreads();
        for (int k = 0; k < e2; ++k) {
            const double dk = M[k * ldm + k];
            if (dk > 1e-12 * d0[k]) {
                const double rd = 1.0 / dk;
                for (int e = tid; e < m * m; e += 256) {
                    const int i = e / m, j = e - i * m;
                    if (i > k && j > k && j <= i) { const double f = M[i * ldm + k] * rd; M[i * ldm + j] -= f * M[j * ldm + k]; }
                }
            }
            __syncthreads();
        }
        truncate = !(M[e2 * ldm + e2] >= 1e-8);
    }
    // (only the tiles on and above the diagonal are filled in: the elimination above reads S2[i][j], j <= i, as S2[j][i])
    if (!truncate)
        for (int e = tid; e < c6 * ldh; e += 256) { const int q = e % ldh; if (q <= c6 && (q >> 4) >= ((e / ldh) >> 4)) A[e] = ld_l2(A + e) + ld_l2(S1 + e); }
    __threadfence();
    __syncthreads();
    for (int e = tid; e < c6 * ldh; e += 256) {          // A is symmetric: the lower tiles are the mirror image
        const int q = e % ldh, pr = e / ldh;
        if (q < c6 && (q >> 4) < (pr >> 4)) A[e] = ld_l2(A + (size_t)q * ldh + pr);
    }
    if (tid == 0) {
        double* mr = A + (size_t)ldh * (ldh - 1);
        mr[0] = (double)good; mr[1] = (double)rows; mr[2] = truncate ? (double)e2 : -1.0; mr[5] = -1.0;
    }
}

// block = this shard's [S2 | S1] + counters: the sums of the accepted features' shares G_f (written by feat_build_kernel to partial[f]) in
// ascending feature order — deterministic; also the all-gather payload of the sharded updater.  combine = 1 (unsharded update): the
// workgroup that finishes last turns part 0 into [A|b] in place (trunc_finish), so no further launch is needed.
// lit: the literal sweep for small stacks (literal.h) — rows exported by feat_build_kernel, the state of the array (lit.state == nullptr: in this
// launch's dynamic LDS, behind lit_aux_doubles()), the features handed to the update
struct LitArgs { double* rows; double* state; const int* n_feat; size_t lds_doubles; };   // lds_doubles: the dynamic LDS of the launch that may run lit_finish
__global__ __launch_bounds__(256) void gram_reduce_kernel(DevCfg cfg, int n, const double* partial, const int* nrows,
                                                          const unsigned char* types, const int* lens, double* block, int* cnt, int combine,
                                                          int wide, size_t bs, BatchIn bin, LitArgs lit) {
    extern __shared__ __align__(16) double g_dyn[];
    const BatchIdx bi = batch_plain();
    partial = zoffi(partial, bs, bi.z); nrows = zoffi(nrows, bs, bi.z); block = zoffi(block, bs, bi.z); cnt = zoffi(cnt, bs, bi.z);
    types = zoffi(types, bin.types, bi.z); lens = zoffi(lens, bin.len, bi.z);
    if (lit.rows) { lit.rows = zoffi(lit.rows, bs, bi.z); lit.n_feat = zoffi(lit.n_feat, bin.n_feat, bi.z); if (lit.state) lit.state = zoffi(lit.state, bs, bi.z); }
    const int c6 = 6 * n, ldh = cfg.ldh, Fu = cfg.Fu;
    const int total = c6 * ldh;
    const int nf_lit = (combine && lit.rows) ? *lit.n_feat : 0;      // (in flight with the list's loads)
    DBG_T(41); DBG_U(45);
    DBG_R(bi.x == 0, 1);
    // ascending list of the accepted features (wave ballots: order-preserving compaction); bit 30 marks type '2', bits 16..19 / 20..23 the
    // first / last 16-column tile of the feature's range (feat_build_kernel stores a share's tiles inside that range only)
    __shared__ int s_list[GRAM_MAX_FEATS], s_wtot[4], s_base, s_cnt[5], s_wc[4][5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    // (round 6: a feature's row count, length and type in ONE batch of unpredicated loads — the list and the counters below each took their own
    //  dependent round trips through these three arrays: 1.2 + 0.9 us of the stage)
    int good = 0, rows = 0, rows2 = 0, e2 = -1, smin = TR_NONE;       // this thread's part of the shard's counters: accepted features, their rows,
    for (int f0 = 0; f0 < Fu; f0 += 256) {                             // the rows / last column of type '2', the first column of type '1'
        const int f = f0 + tid, fc = f < Fu ? f : 0;
        const int r = nrows[fc], L = lens[fc];
        const bool t2 = types[fc] == '2';
        const bool flag = f < Fu && r > 0;
        const unsigned long long mask = __ballot(flag);
        if (lane == 0) s_wtot[wave] = __popcll(mask);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wtot[w];
        if (flag) {
            const int Lu = t2 ? (L + 1) / 2 : L, lo = t2 ? 0 : 6 * (n - (Lu - 1)), hi = lo + 6 * (Lu - 1);
            const int slot = off + __popcll(mask & ((1ull << lane) - 1ull));
            s_list[slot] = f | (t2 ? (1 << 30) : 0) | ((lo >> 4) << 16) | (((hi - 1) >> 4) << 20);
            good++; rows += r;
            if (t2) { rows2 += r; e2 = max(e2, 6 * ((L + 1) / 2 - 1) - 1); }
            else smin = min(smin, 6 * (n - (L - 1)));
        }
        __syncthreads();
        if (tid == 0) s_base += s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
        __syncthreads();
    }
    const int ng = s_base;
    const size_t gs = (size_t)ldh * ldh;
    double* S2 = block; double* S1 = block + gs;
    DBG_T(42); DBG_U(46);
    good = (int)wave_sum_i64(good); rows = (int)wave_sum_i64(rows); rows2 = (int)wave_sum_i64(rows2);
    for (int o = 32; o > 0; o >>= 1) { e2 = max(e2, __shfl_xor(e2, o, 64)); smin = min(smin, __shfl_xor(smin, o, 64)); }
    if (lane == 0) { s_wc[wave][0] = good; s_wc[wave][1] = rows; s_wc[wave][2] = rows2; s_wc[wave][3] = e2; s_wc[wave][4] = smin; }
    __syncthreads();
    if (tid < 5) {
        const int a0 = s_wc[0][tid], a1 = s_wc[1][tid], a2 = s_wc[2][tid], a3 = s_wc[3][tid];
        s_cnt[tid] = tid < 3 ? a0 + a1 + a2 + a3 : tid == 3 ? max(max(a0, a1), max(a2, a3)) : min(min(a0, a1), min(a2, a3));
    }
    __syncthreads();
    DBG_U(47);
    // a small stack whose rank decision the structure does not settle (literal.h): block 0 runs the reference's sweep + scan on the exported
    // rows and writes [A|b] itself; the shares are not needed.  (Unsharded update only: a shard's counters are partial — there the decision
    // is taken on the gathered whole, block_sum_kernel.)
    if (combine && lit.rows) {
        const int nf = nf_lit;
        if (lit_decide(lit.rows, n, nf, s_cnt[0], s_cnt[1], nrows, types, lens)) {
            if (bi.x != 0) return;
            lit_finish(cfg, n, nf, nrows, types, lens, lit.rows, block, s_cnt[0], s_cnt[1], lit.state ? lit.state : g_dyn + lit_aux_doubles(cfg.ldh, cfg.rho_max), lit.state == nullptr, g_dyn, lit.lds_doubles);
            return;
        }
    }
    // can the reference's rank scan cut the type-'1' rows off at all (structural precondition of trunc_finish)?  Almost never: then
    // [A|b] = S2 + S1 is written directly, mirror image included, and nobody has to wait for the last workgroup
    const bool cand = s_cnt[0] > 2 && s_cnt[1] > c6 && s_cnt[3] >= 0 && s_cnt[3] < c6 && s_cnt[4] < TR_NONE && s_cnt[4] > s_cnt[3] && s_cnt[2] >= s_cnt[3] + 1;
    const bool direct = combine && !cand;
    const bool split = !direct;                          // S2 and S1 are needed apart (a shard of the sharded updater, or a candidate)
    DBG_T(43); DBG_U(48);
    const int trq = c6 >> 4;                             // tile of the residual column
    // The shares were written by ~100 other CUs: every load here is a remote (fabric) round trip, and what one CU can keep in flight
    // bounds its rate.  So the reduction is spread wide: a workgroup covers only 64 consecutive elements, its four waves split the
    // feature list into four contiguous chunks, and the chunk sums are added in chunk order (deterministic).
    __shared__ double s_part[4][64][2];
    // sum of the shares t in [tb, te) at element e (p-tile pt, q-tile qt): a share contributes where feat_build_kernel stored it —
    // elsewhere its buffer holds stale numbers, which are loaded (no branch) and discarded.  16 loads in flight per round.
    auto accumulate = [&](int e, int pt, int qt, int tb, int te, double& a2, double& a1) {
        const double* pe = partial + e;
        for (int t = tb; t < te; t += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (t + u < te) ? pe[(size_t)(s_list[t + u] & 0xffff) * gs] : 0.0;     // (uniform predicate: the loads go out back to back)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (t + u < te) {
                    const int fl = s_list[t + u];
                    const unsigned lo = (fl >> 16) & 15, span = ((fl >> 20) & 15) - lo;
                    const bool in = (unsigned)pt - lo <= span && ((unsigned)qt - lo <= span || qt == trq);
                    const double w = in ? v[u] : 0.0;
                    if (!split) a2 += w;
                    else { const bool is2 = (fl >> 30) & 1; a2 += is2 ? w : 0.0; a1 += is2 ? 0.0 : w; }
                }
            }
        }
    };
    const ShardLayout SL = shard_layout(c6, cfg.max_len);
    auto store = [&](int e, int q, int pq, int pt, int qt, double a2, double a1) {
        if (direct) {
            const double v = a2 + a1;
            S2[e] = v;
            if (qt > pt && q < c6) S2[(size_t)q * ldh + pq] = v;
        } else if (combine) { S2[e] = a2; S1[e] = a1; }
        else {     // a shard's share in its wire format (rvio_dev.h shard_layout): the tiles that can be non-zero, upper triangle only
            const int w = (pq & 15) * 16 + (q & 15), o2 = shard_tile2(SL, pt, qt), o1 = shard_tile1(SL, pt, qt);
            if (o2 >= 0) block[o2 + w] = a2;
            if (o1 >= 0) block[o1 + w] = a1;
        }
    };
    if (wide) {
        // one stream: the shares were written by ~100 other CUs, every load is a remote (fabric) round trip and what one CU can keep in
        // flight bounds its rate.  A workgroup covers only 64 consecutive elements (the launch spreads over ~60 CUs), its four waves
        // split the feature list into four contiguous chunks, and the chunk sums are added in chunk order (deterministic).
        for (int e0 = bi.x * 64; e0 < total; e0 += gridDim.x * 64) {
            const int e = e0 + lane;
            const int q = e % ldh, pq = e / ldh, pt = pq >> 4, qt = q >> 4;
            const bool live = e < total && q <= c6 && qt >= pt;       // padding / lower tiles: the mirror image of the upper ones
            double a2 = 0, a1 = 0;
            if (live) accumulate(e, pt, qt, (ng * wave) / 4, (ng * (wave + 1)) / 4, a2, a1);
            s_part[wave][lane][0] = a2; s_part[wave][lane][1] = a1;
            __syncthreads();
            if (wave == 0 && live) {
                a2 = ((s_part[0][lane][0] + s_part[1][lane][0]) + s_part[2][lane][0]) + s_part[3][lane][0];
                a1 = ((s_part[0][lane][1] + s_part[1][lane][1]) + s_part[2][lane][1]) + s_part[3][lane][1];
                store(e, q, pq, pt, qt, a2, a1);
            }
            __syncthreads();
        }
    } else {
        // batch handles: one thread per element, every share in list order (few accepted features per instance, many instances)
        for (int e = bi.x * 256 + tid; e < total; e += gridDim.x * 256) {
            const int q = e % ldh, pq = e / ldh, pt = pq >> 4, qt = q >> 4;
            if (q > c6 || qt < pt) continue;                // padding / lower tiles: the mirror image of the upper ones (store)
            double a2 = 0, a1 = 0;
            accumulate(e, pt, qt, 0, ng, a2, a1);
            store(e, q, pq, pt, qt, a2, a1);
        }
    }
    DBG_T(44); DBG_U(49);
    if (direct) {
        if (bi.x == 0 && tid == 0) { double* mr = S2 + (size_t)ldh * (ldh - 1); mr[0] = s_cnt[0]; mr[1] = s_cnt[1]; mr[2] = -1.0; mr[5] = -1.0; }
        return;
    }
    if (!combine) {
        if (bi.x == 0 && tid == 0) { double* mr = block; mr[0] = s_cnt[0]; mr[1] = s_cnt[1]; mr[2] = s_cnt[2]; mr[3] = s_cnt[3]; mr[4] = s_cnt[4]; mr[5] = 0; mr[6] = 0; mr[7] = 0; }
        return;
    }
    if (last_block_done(cnt, gridDim.x)) trunc_finish(cfg, n, S2, S1, s_cnt[0], s_cnt[1], s_cnt[2], s_cnt[3], s_cnt[4], g_dyn);
}

// Batch handles, windows whose [A|b] fits in LDS (6n (6n+1) doubles <= 64 KB: cfg A, B): ONE workgroup per instance keeps the sum in LDS and
// walks the accepted features in ascending order, touching exactly the tiles feat_build_kernel stored (on and above the diagonal, inside
// the feature's column range, plus the residual column's tile) — the element-per-thread form above reads every share over the whole
// block, ~0.7 GB per launch at B = 2048 for 29 KB of result per instance.  Thread (r, c) owns element (16 pt + r, 16 qt + c) of EVERY tile,
// so the additions to one element happen in feature order in one thread: deterministic without a barrier between features.  The sum
// leaves LDS once, lower triangle mirrored.  A truncation candidate (rare) takes two passes (S2, then S1) and trunc_finish.
__host__ __device__ inline size_t gram_batch_lds_doubles(int max_len, int ldh) {
    const size_t a = (size_t)(ldh - 1) * ldh, t = trunc_lds_doubles(max_len);
    return a > t ? a : t;
}
template <int NT>      // row / column tiles of [A|b]: 4 for 6n <= 63, 6 for 6n <= 95
__global__ __launch_bounds__(256) void gram_reduce_batch_kernel(DevCfg cfg, int n, const double* __restrict__ partial, const int* __restrict__ nrows,
                                                                const unsigned char* __restrict__ types, const int* __restrict__ lens, double* __restrict__ block,
                                                                size_t bs, BatchIn bin) {
    extern __shared__ __align__(16) double gb_dyn[];
    const int z = blockIdx.z;
    partial = zoffi(partial, bs, z); nrows = zoffi(nrows, bs, z); block = zoffi(block, bs, z);
    types = zoffi(types, bin.types, z); lens = zoffi(lens, bin.len, z);
    const int c6 = 6 * n, ldh = cfg.ldh, Fu = cfg.Fu, total = c6 * ldh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_list[GRAM_MAX_FEATS], s_wtot[4], s_base, s_cnt[5];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int f0 = 0; f0 < Fu; f0 += 256) {      // ascending list of the accepted features (as in gram_reduce_kernel)
        const int f = f0 + tid;
        const bool flag = f < Fu && nrows[f] > 0;
        const unsigned long long mask = __ballot(flag);
        if (lane == 0) s_wtot[wave] = __popcll(mask);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wtot[w];
        if (flag) {
            const int L = lens[f];
            const bool t2 = types[f] == '2';
            const int Lu = t2 ? (L + 1) / 2 : L, lo = t2 ? 0 : 6 * (n - (Lu - 1)), hi = lo + 6 * (Lu - 1);
            s_list[off + __popcll(mask & ((1ull << lane) - 1ull))] = f | (t2 ? (1 << 30) : 0) | ((lo >> 4) << 16) | (((hi - 1) >> 4) << 20);
        }
        __syncthreads();
        if (tid == 0) s_base += s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
        __syncthreads();
    }
    const int ng = s_base;
    if (tid < 64) {
        int good = 0, rows = 0, rows2 = 0, e2 = -1, smin = TR_NONE;
        for (int f = tid; f < Fu; f += 64) {
            const int r = nrows[f];
            if (r > 0) {
                good++; rows += r;
                const int L = lens[f];
                if (types[f] == '2') { rows2 += r; e2 = max(e2, 6 * ((L + 1) / 2 - 1) - 1); }
                else smin = min(smin, 6 * (n - (L - 1)));
            }
        }
        good = (int)wave_sum_i64(good); rows = (int)wave_sum_i64(rows); rows2 = (int)wave_sum_i64(rows2);
        for (int o = 32; o > 0; o >>= 1) { e2 = max(e2, __shfl_xor(e2, o, 64)); smin = min(smin, __shfl_xor(smin, o, 64)); }
        if (tid == 0) { s_cnt[0] = good; s_cnt[1] = rows; s_cnt[2] = rows2; s_cnt[3] = e2; s_cnt[4] = smin; }
    }
    __syncthreads();
    const bool cand = s_cnt[0] > 2 && s_cnt[1] > c6 && s_cnt[3] >= 0 && s_cnt[3] < c6 && s_cnt[4] < TR_NONE && s_cnt[4] > s_cnt[3] && s_cnt[2] >= s_cnt[3] + 1;
    const size_t gs = (size_t)ldh * ldh;
    double* S2 = block; double* S1 = block + gs;
    double* acc = gb_dyn;                                  // [c6][ldh]
    const int r = tid >> 4, c = tid & 15, trq = c6 >> 4;
    // pass 0: every feature (direct) or the type-'2' features (candidate); pass 1 (candidate only): the type-'1' features
    for (int pass = 0; pass < (cand ? 2 : 1); ++pass) {
        // the thread's element of each of the NT (NT + 1) / 2 tiles on and above the diagonal, in registers: a feature's stored tiles are ALL in flight at once
        // (an LDS sum behind chunks of 8 tile slots had 7 + 3 loads in flight for a full-window feature, and a read-modify-write of LDS per slot)
        double a[NT * (NT + 1) / 2];
#pragma unroll
        for (int ti = 0; ti < NT * (NT + 1) / 2; ++ti) a[ti] = 0.0;
#pragma unroll 2
        for (int t = 0; t < ng; ++t) {
            const int fl = s_list[t];
            if (cand && (((fl >> 30) & 1) != (pass == 0))) continue;
            const double* sh = partial + (size_t)(fl & 0xffff) * gs;
            const int t0 = (fl >> 16) & 15, t1 = (fl >> 20) & 15;
            int ti = 0;
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) {
#pragma unroll
                for (int qt = pt; qt < NT; ++qt, ++ti) {
                    const int pp = 16 * pt + r, qq = 16 * qt + c;
                    const bool on = pt >= t0 && pt <= t1 && ((qt >= t0 && qt <= t1) || qt == trq) && pp < c6 && qq <= c6;
                    a[ti] += on ? sh[(size_t)pp * ldh + qq] : 0.0;      // (+ 0.0 leaves the sum as it is)
                }
            }
        }
        {
            int ti = 0;
#pragma unroll
            for (int pt = 0; pt < NT; ++pt) {
#pragma unroll
                for (int qt = pt; qt < NT; ++qt, ++ti) {
                    const int pp = 16 * pt + r, qq = 16 * qt + c;
                    if (pp < c6 && qq <= c6) acc[pp * ldh + qq] = a[ti];
                }
            }
        }
        __syncthreads();
        double* dst = pass == 0 ? S2 : S1;
        if (!cand) {
            // [A|b] = the sum, lower tiles mirrored (A is symmetric; column c6 = b has no mirror image)
            for (int e = tid; e < total; e += 256) {
                const int q = e % ldh, pq = e / ldh;
                if (q > c6) continue;
                dst[e] = (q < c6 && (q >> 4) < (pq >> 4)) ? acc[q * ldh + pq] : acc[e];
            }
        } else {
            for (int e = tid; e < total; e += 256) { const int q = e % ldh; if (q <= c6 && (q >> 4) >= ((e / ldh) >> 4)) dst[e] = acc[e]; }
        }
        __syncthreads();
    }
    if (!cand) {
        if (tid == 0) { double* mr = S2 + (size_t)ldh * (ldh - 1); mr[0] = s_cnt[0]; mr[1] = s_cnt[1]; mr[2] = -1.0; mr[5] = -1.0; }
        return;
    }
    __threadfence();
    __syncthreads();
    trunc_finish(cfg, n, S2, S1, s_cnt[0], s_cnt[1], s_cnt[2], s_cnt[3], s_cnt[4], gb_dyn);
}

// Batch handles whose share reduction is gram_reduce_batch_kernel (a throughput kernel: 64 VGPRs, eight waves per SIMD — the literal sweep inside it
// would cost it three quarters of that): the literal path as a launch of its own behind the reduction, one workgroup per instance, which leaves at
// once unless literal.h's decision holds for its instance (the counters come from the meta row the reduction wrote; the array's state lives in the slab).
__global__ __launch_bounds__(256) void lit_batch_kernel(DevCfg cfg, int n, const int* __restrict__ nrows, const unsigned char* __restrict__ types,
                                                        const int* __restrict__ lens, double* __restrict__ block, size_t bs, BatchIn bin, LitArgs lit) {
    extern __shared__ __align__(16) double lb_dyn[];
    const int z = blockIdx.z;
    nrows = zoffi(nrows, bs, z); block = zoffi(block, bs, z); types = zoffi(types, bin.types, z); lens = zoffi(lens, bin.len, z);
    lit.rows = zoffi(lit.rows, bs, z); lit.n_feat = zoffi(lit.n_feat, bin.n_feat, z); lit.state = zoffi(lit.state, bs, z);
    const double* mr = block + (size_t)cfg.ldh * (cfg.ldh - 1);
    const int good = (int)mr[0], rows = (int)mr[1], nf = *lit.n_feat;
    if (!lit_decide(lit.rows, n, nf, good, rows, nrows, types, lens)) return;
    lit_finish(cfg, n, nf, nrows, types, lens, lit.rows, block, good, rows, lit.state, false, lb_dyn, lit.lds_doubles);
}

// Gathered shards (rank-major, `block_stride` doubles apart) -> Ab = [A|b] + {n_good, n_rows, truncation column}: both parts are summed
// in rank order (Ab <- S2, Ab + ldh^2 <- S1), then the workgroup that finishes last applies trunc_finish.
// An update of at most LIT_FEATS features was not sharded (feat_build_body: every rank built every feature): block 0 alone is the whole
// of it, and the last workgroup may have to run the literal sweep on the (local, complete) exported rows (literal.h).
__global__ __launch_bounds__(256) void block_sum_kernel(DevCfg cfg, int n, const double* blocks, int world, size_t block_stride, double* Ab, int* cnt,
                                                        const int* nrows, const unsigned char* types, const int* lens, LitArgs lit) {
    extern __shared__ __align__(16) double g_dyn[];
    const int c6 = 6 * n, ldh = cfg.ldh;
    const int total = c6 * ldh;
    const size_t gs = (size_t)ldh * ldh;
    const int nf = lit.rows ? *lit.n_feat : LIT_FEATS + 1;
    if (nf <= LIT_FEATS) world = 1;
    const ShardLayout SL = shard_layout(c6, cfg.max_len);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int q = e % ldh, pq = e / ldh, pt = pq >> 4, qt = q >> 4;
        if (q > c6 || qt < pt) continue;     // the shards carry the tiles on and above the diagonal (wire format: rvio_dev.h shard_layout)
        const int w16 = (pq & 15) * 16 + (q & 15), o2 = shard_tile2(SL, pt, qt), o1 = shard_tile1(SL, pt, qt);
        double a2 = 0, a1 = 0;
        for (int w = 0; w < world; ++w) {
            if (o2 >= 0) a2 += blocks[(size_t)w * block_stride + o2 + w16];
            a1 += blocks[(size_t)w * block_stride + o1 + w16];
        }
        Ab[e] = a2; Ab[gs + e] = a1;
    }
    if (!last_block_done(cnt, gridDim.x)) return;
    __shared__ int s_cnt[5];
    if (threadIdx.x == 0) {
        int good = 0, rows = 0, rows2 = 0, e2 = -1, smin = TR_NONE;
        for (int w = 0; w < world; ++w) {
            const double* mr = blocks + (size_t)w * block_stride;
            good += (int)mr[0]; rows += (int)mr[1]; rows2 += (int)mr[2]; e2 = max(e2, (int)mr[3]); smin = min(smin, (int)mr[4]);
        }
        s_cnt[0] = good; s_cnt[1] = rows; s_cnt[2] = rows2; s_cnt[3] = e2; s_cnt[4] = smin;
    }
    __syncthreads();
    if (lit_decide(lit.rows, n, nf, s_cnt[0], s_cnt[1], nrows, types, lens)) {
        lit_finish(cfg, n, nf, nrows, types, lens, lit.rows, Ab, s_cnt[0], s_cnt[1], lit.state ? lit.state : g_dyn + lit_aux_doubles(cfg.ldh, cfg.rho_max), lit.state == nullptr, g_dyn, lit.lds_doubles);
        return;
    }
    trunc_finish(cfg, n, Ab, Ab + gs, s_cnt[0], s_cnt[1], s_cnt[2], s_cnt[3], s_cnt[4], g_dyn);
}

// =============================================================== FP64 MFMA GEMM:  T = s2 I + A Pcc
// One workgroup = 4 waves = a 32x32 output tile, each wave one 16x16 tile with v_mfma_f64_16x16x4_f64:
//   A operand lane l : A[i = l&15][k = l>>4]      B operand lane l : B[k = l>>4][j = l&15]
//   C/D      lane l : 4 values, row = (l>>4) + 4*r, col = l&15.
// A = Ab (row-major, ld = ldh), B = Pcc = P[24:,24:] (column-major, ld = dmax), T row-major ld = ldh.
__global__ __launch_bounds__(256) void gemm_T_kernel(DevCfg cfg, int n, const double* Ab, const double* P, double* Tm, size_t bs) {
    const BatchIdx bi = batch_remap();
    Ab = zoffi(Ab, bs, bi.z); P = zoffi(P, bs, bi.z); Tm = zoffi(Tm, bs, bi.z);
    const int c6 = 6 * n, ldh = cfg.ldh, ld = cfg.dmax;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = bi.y * 32 + (wave >> 1) * 16, j0 = bi.x * 32 + (wave & 1) * 16;
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

// T = s2 I + A Pcc for a batch handle (>= 128 instances, 6n <= 64): ONE workgroup per instance, both operands staged in LDS by coalesced loads
// (gemm_T_kernel reads its operands inside the k-loop with the row index along the lanes — 16 scattered 32-byte pieces per instruction, four dependent
// rounds per tile: 93 us per launch at B = 2048 for 2 us of matrix-core work).  Same MFMA order per tile, same closing expression.
// LDS: As[i][k], Bs[j][k] = Pcc[k][j], leading dimension 6n + 1: 2 * 6n (6n + 1) doubles (57 KB at 6n = 60, two workgroups per CU).
__global__ __launch_bounds__(256) void gemm_T_lds_kernel(DevCfg cfg, int n, const double* __restrict__ Ab, const double* __restrict__ P, double* __restrict__ Tm, size_t bs) {
    extern __shared__ __align__(16) double gt[];
    const int z = blockIdx.z;
    Ab = zoffi(Ab, bs, z); P = zoffi(P, bs, z); Tm = zoffi(Tm, bs, z);
    const int c6 = 6 * n, ldh = cfg.ldh, ld = cfg.dmax, LS = c6 + 1;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    double* const As = gt; double* const Bs = gt + c6 * LS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    {
        double va[16], vb[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {      // A: rows of 6n doubles; Pcc: columns of 6n doubles — both walks are contiguous in the source
            const int e = tid + u * 256, i = e / c6, k = e - i * c6;
            const bool ok = i < c6;
            va[u] = ok ? Ab[(size_t)i * ldh + k] : 0.0;
            vb[u] = ok ? P[(size_t)(24 + k) + (size_t)(24 + i) * ld] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int e = tid + u * 256, i = e / c6, k = e - i * c6; if (i < c6) { As[i * LS + k] = va[u]; Bs[i * LS + k] = vb[u]; } }
    }
    __syncthreads();
    const int nt1 = (c6 + 15) / 16;
    for (int t = wave; t < nt1 * nt1; t += 4) {
        const int it = t / nt1, jt = t - it * nt1, r = it * 16 + li, c = jt * 16 + li;
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = 32 * h + 4 * u + lk;
                av[u] = (r < c6 && k < c6) ? As[r * LS + k] : 0.0;
                bv[u] = (c < c6 && k < c6) ? Bs[c * LS + k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
        if (c < c6) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = it * 16 + lk + 4 * q;
                if (row < c6) Tm[(size_t)row * ldh + c] = acc[q] + ((row == c) ? s2 : 0.0);
            }
        }
    }
}

// =============================================================== U = Pc W, G = U A, P1 = P - G Pc^T
// One workgroup (4 waves) per 16-row strip of the d rows.  K H = [0 | G];  (I - K H) P = P1.
// LDS: Us[16][c6p], Gs[16][c6p] (row-major, c6p = c6 rounded up to 16, +1 pad).
__global__ __launch_bounds__(256) void ug_kernel(DevCfg cfg, int n, const double* P, const double* W, const double* Ab,
                                                 double* U, double* G, double* P1, size_t bs) {
    extern __shared__ __align__(16) double sh[];
    const BatchIdx bi = batch_remap();
    P = zoffi(P, bs, bi.z); W = zoffi(W, bs, bi.z); Ab = zoffi(Ab, bs, bi.z); U = zoffi(U, bs, bi.z); G = zoffi(G, bs, bi.z); P1 = zoffi(P1, bs, bi.z);
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax;
    const int c6t = (c6 + 15) / 16, dt = (d + 15) / 16;
    const int lds = c6t * 16 + 1;
    DBG_R(bi.x == 0, 3);
    double* Us = sh; double* Gs = sh + 16 * lds;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int i0 = bi.x * 16;
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
        }
    }
    __syncthreads();
    // U^T (k-major, leading dimension ld: UT[k][i] = U[i][k]) from the LDS strip — final_kernel reads U and G with the row index along the
    // lanes, which a row-major U made 64 scattered 8-byte loads per instruction; 16 consecutive rows of one k are one 128-byte segment
    for (int e = threadIdx.x; e < c6 * 16; e += 256) { const int k = e >> 4, r = e & 15; if (i0 + r < d) U[(size_t)k * ld + i0 + r] = Us[r * lds + k]; }
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
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < c6 * 16; e += 256) { const int k = e >> 4, r = e & 15; if (i0 + r < d) G[(size_t)k * ld + i0 + r] = Gs[r * lds + k]; }
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
            b1[u] = (bok && kok) ? G[(size_t)k * ld + bj] : 0.0;                   // G[j][k]  (G and U are stored k-major by ug_kernel)
            a2[u] = (aok && kok) ? G[(size_t)k * ld + ai] : 0.0;                   // G[i][k]
            b2[u] = (bok && kok) ? U[(size_t)k * ld + bj] : 0.0;                   // U[j][k]
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
__global__ __launch_bounds__(256) void final_kernel(DevCfg cfg, int n, const double* P1, const double* G, const double* U, double* Pout, size_t bs) {
    const BatchIdx bi = batch_remap();
    P1 = zoffi(P1, bs, bi.z); G = zoffi(G, bs, bi.z); U = zoffi(U, bs, bi.z); Pout = zoffi(Pout, bs, bi.z);
    __shared__ double tl[4][16][17];
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    const int nt = (d + 15) / 16, npair = nt * (nt + 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    DBG_R(bi.x == 0, 4);
    const int pr = bi.x * 4 + wave;
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

// =============================================================== one instance, 6n > 64: U, G, P1 and the Joseph form with one wave per TILE
// ug_kernel gives a 16-row strip of all three products to ONE workgroup (13 workgroups at 6n = 180, each walking ~120 tile products with cold operands:
// 78 us) and final_kernel both tiles and both sums of a tile pair to one wave (28 us).  A single instance has the whole chip to itself: here every
// output TILE of U, of G and of P1 is a wave of its own (three launches: each needs the one before complete), and a tile pair of P+ is a workgroup whose
// four waves take (X_IJ, X_JI) x (P1c G^T, G U^T).  Operands, the order of the MFMAs of a tile and the closing expressions are ug_kernel's /
// final_kernel's: the same bits.
// dx_scr != NULL (PH = 0 behind the split solve): ceil(d / 24) more workgroups at the end of the grid are the solve's dx / state-injection roles (solve9.hip)
template <int PH>
__global__ __launch_bounds__(256) void ug_tile_kernel(DevCfg cfg, int n, const double* __restrict__ P, const double* __restrict__ W, const double* __restrict__ Ab,
                                                      double* __restrict__ U, double* __restrict__ G, double* __restrict__ P1,
                                                      FilterMeta* __restrict__ meta, const double* __restrict__ x, double* __restrict__ x_out, const double* __restrict__ dx_scr, int dx_nt) {
    __shared__ double tl[4][16][17];
    if constexpr (PH == 0) {
        if (dx_scr) {
            const int n_roles = (24 + 6 * n + 23) / 24, first = (int)gridDim.x - n_roles;
            if ((int)blockIdx.x >= first) {
                __shared__ S9DxLds dxl;
                s9_dx_role(cfg, meta, n, Ab, x, P, dx_scr, x_out, dx_nt, (int)blockIdx.x - first, dxl);
                return;
            }
        }
    }
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax;
    const int c6t = (c6 + 15) / 16, dt = (d + 15) / 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int nj = PH == 2 ? dt : c6t;
    const int t = blockIdx.x * 4 + wave;
    if (t >= dt * nj) return;                                  // (wave-private LDS only: no workgroup barrier below)
    const int it = t / nj, jt = t - it * nj;
    const int i0 = it * 16, ai = i0 + li, bj = jt * 16 + li;
    const bool aok = ai < d, bok = bj < (PH == 2 ? d : c6);
    d4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < c6; k0 += 16) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 4 * u + lk;
            const bool kok = k < c6;
            if constexpr (PH == 0) {                           // U = Pc W
                a[u] = (aok && kok) ? P[(size_t)ai + (size_t)(24 + k) * ld] : 0.0;
                b[u] = (bok && kok) ? W[(size_t)k * ldh + bj] : 0.0;
            } else if constexpr (PH == 1) {                    // G = U A  (U k-major: U[k ld + i])
                a[u] = (aok && kok) ? U[(size_t)k * ld + ai] : 0.0;
                b[u] = (bok && kok) ? Ab[(size_t)k * ldh + bj] : 0.0;
            } else {                                           // G Pc^T
                a[u] = (aok && kok) ? G[(size_t)k * ld + ai] : 0.0;
                b[u] = (bok && kok) ? P[(size_t)bj + (size_t)(24 + k) * ld] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    if constexpr (PH == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + lk + 4 * r;
            if (row < d && bj < d) P1[(size_t)row + (size_t)bj * ld] = P[(size_t)row + (size_t)bj * ld] - acc[r];
        }
    } else {                                                   // k-major store (see ug_kernel): 16 consecutive rows of one k are one 128-byte segment
        double* out = PH == 0 ? U : G;
#pragma unroll
        for (int r = 0; r < 4; ++r) tl[wave][lk + 4 * r][li] = acc[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = jt * 16 + lk + 4 * r, row = i0 + li;  // element (row, kk) = tile (li, lk + 4 r)
            if (kk < c6 && row < d) out[(size_t)kk * ld + row] = tl[wave][li][lk + 4 * r];
        }
    }
}

// one workgroup per unordered tile pair (I <= J) of P+; wave w: tile (w & 2 ? (J, I) : (I, J)), sum (w & 1 ? G U^T : P1c G^T)
__global__ __launch_bounds__(256) void final_tile_kernel(DevCfg cfg, int n, const double* __restrict__ P1, const double* __restrict__ G, const double* __restrict__ U,
                                                         double* __restrict__ Pout) {
    __shared__ double s2acc[2][4][64];                          // the G U^T sums of the two tiles, in accumulator layout
    __shared__ double tl[16][17];
    const int c6 = 6 * n, d = 24 + c6, ld = cfg.dmax;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    const int nt = (d + 15) / 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    int I = 0, rem = blockIdx.x;
    while (rem >= nt - I) { rem -= nt - I; ++I; }
    const int J = I + rem;
    const int second = wave >> 1, which = wave & 1;
    const bool live = !(second && I == J);
    const int i0 = (second ? J : I) * 16, j0 = (second ? I : J) * 16;
    const int ai = i0 + li, bj = j0 + li;
    const bool aok = ai < d, bok = bj < d;
    d4 acc = {0, 0, 0, 0};
    if (live) {
        for (int k0 = 0; k0 < c6; k0 += 16) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 4 * u + lk;
                const bool kok = k < c6;
                if (which == 0) {
                    a[u] = (aok && kok) ? P1[(size_t)ai + (size_t)(24 + k) * ld] : 0.0;   // P1c[i][k]
                    b[u] = (bok && kok) ? G[(size_t)k * ld + bj] : 0.0;                   // G[j][k]
                } else {
                    a[u] = (aok && kok) ? G[(size_t)k * ld + ai] : 0.0;                   // G[i][k]
                    b[u] = (bok && kok) ? U[(size_t)k * ld + bj] : 0.0;                   // U[j][k]
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
        if (which == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s2acc[second][r][lane] = acc[r];
        }
    }
    __syncthreads();
    d4 out = {0, 0, 0, 0};
    if (live && which == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + lk + 4 * r;
            const double p1 = (row < d && bj < d) ? P1[(size_t)row + (size_t)bj * ld] : 0.0;
            const double acc2 = s2acc[second][r][lane];
            out[r] = p1 - acc[r] + s2 * acc2;
        }
        if (second) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tl[lk + 4 * r][li] = out[r];
        }
    }
    __syncthreads();
    if (wave == 0) {
        if (I == J) {                                          // X_JI = X_IJ: the transpose is this wave's own tile
#pragma unroll
            for (int r = 0; r < 4; ++r) tl[lk + 4 * r][li] = out[r];
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = lk + 4 * r, row = I * 16 + rr, col = J * 16 + li;
            const double v = .5 * (out[r] + tl[li][rr]);       // X_JI[col_local][row_local]
            if (row < d && col < d) {
                Pout[(size_t)row + (size_t)col * ld] = v;
                if (I != J) Pout[(size_t)col + (size_t)row * ld] = v;
            }
        }
    }
}

// =============================================================== batch handles, 6n <= 60: U, G, P1 and the Joseph form in ONE kernel
// ug_kernel + final_kernel move P1, U and G through HBM / L2 between them and fetch every MFMA operand from global memory inside the k-loop (at B = 2048:
// 0.55 ms of the 1.94 ms batched frame for 125 us of matrix-core work).  Here ONE workgroup of JB_WAVES waves owns an instance from P to P+: W (then A), Pc,
// U, G live in LDS (row-major, leading dimension 6n + 1), the tiles of P (then P1) stay in the registers of the wave that needs them again — fetched with the
// kernel's first loads, long before their use —, P1c takes Pc's place once every tile of P1 exists.  The operands, the order of the MFMAs inside a tile and
// the closing expressions are those of ug_kernel / final_kernel, so the result is theirs.  LDS: (3 d + 6n)(6n + 1) doubles = 152 KB at 6n = 60 — one
// workgroup per CU, 8 instances per CU at B = 2048.
#ifndef JB_WAVES
#define JB_WAVES 12
#endif
#define JB_THREADS (64 * JB_WAVES)
#define JB_PAIRS ((21 + JB_WAVES - 1) / JB_WAVES)            /* tile pairs of P per wave (nt <= 6: at most 21) */
#define JB_NWA ((60 * 60 + JB_THREADS - 1) / JB_THREADS)     /* elements of W / A per thread */
#define JB_NPC ((60 * 84 + JB_THREADS - 1) / JB_THREADS)     /* elements of Pc per thread */
#define JB_TL_DOUBLES (JB_WAVES * 16 * 17)
__global__ __launch_bounds__(JB_THREADS) void joseph_batch_kernel(DevCfg cfg, int n, const double* __restrict__ P, const double* __restrict__ W,
                                                                   const double* __restrict__ Ab, double* __restrict__ Pout, size_t bs) {
    extern __shared__ __align__(16) double jl[];
    const int z = blockIdx.z;          // (grid (1, 1, B) like every batch kernel) nothing is shared between instances: the round-robin of workgroups over the XCDs is the mapping wanted
    P = zoffi(P, bs, z); W = zoffi(W, bs, z); Ab = zoffi(Ab, bs, z); Pout = zoffi(Pout, bs, z);
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax, LS = c6 + 1;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    double* const R0 = jl;                 // Pc[r][k] = P[r][24 + k], r < d;  later P1c[r][k]
    double* const R1 = R0 + d * LS;        // U[r][k]
    double* const R2 = R1 + d * LS;        // G[r][k]
    double* const R3 = R2 + d * LS;        // W[k][j], then A[k][j], then the waves' transposition tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    // the unordered tile pairs (I <= J) of P: pair wave + JB_WAVES s belongs to this wave in the P1 stage and in the closing stage
    const int nt = (d + 15) / 16, npair = nt * (nt + 1) / 2;
    int pi[JB_PAIRS], pj[JB_PAIRS];
    d4 pa[JB_PAIRS], pb[JB_PAIRS];         // tile (I, J) and tile (J, I) of P, then of P1
#pragma unroll
    for (int s = 0; s < JB_PAIRS; ++s) {
        int I = 0, rem = wave + JB_WAVES * s;
        if (rem < npair) while (rem >= nt - I) { rem -= nt - I; ++I; }
        pi[s] = I; pj[s] = I + rem;
    }
    // rows i0 .. i0+15 of an LDS matrix as MFMA operand (A: [i][k]; B of X Y^T: [k][j] = Y[j][k]), k = 32 h .. 32 h + 31
    auto rows8 = [&](const double* R, int i0, int h, double (&v)[8]) {
        const int r = i0 + li;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = 32 * h + 4 * u + lk; v[u] = (r < d && k < c6) ? R[r * LS + k] : 0.0; }
    };
    // columns j0 .. j0+15 of W / A (row-major [k][j], 6n columns) as B operand
    auto cols8 = [&](const double* R, int j0, int h, double (&v)[8]) {
        const int c = j0 + li;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = 32 * h + 4 * u + lk; v[u] = (c < c6 && k < c6) ? R[k * LS + c] : 0.0; }
    };
    {
        double vw[JB_NWA], va[JB_NWA], vp[JB_NPC];
#pragma unroll
        for (int u = 0; u < JB_NWA; ++u) {
            const int e = tid + u * JB_THREADS, k = e / c6, j = e - k * c6;
            const bool ok = k < c6;
            vw[u] = ok ? W[(size_t)k * ldh + j] : 0.0;
            va[u] = ok ? Ab[(size_t)k * ldh + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < JB_NPC; ++u) {      // column-major source: consecutive threads walk down a column of P
            const int e = tid + u * JB_THREADS, k = e / d, r = e - k * d;
            vp[u] = (k < c6) ? P[(size_t)r + (size_t)(24 + k) * ld] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < JB_PAIRS; ++s) {    // the wave's tiles of P, needed in the third stage: in flight from here
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool on = wave + JB_WAVES * s < npair;
                { const int row = pi[s] * 16 + lk + 4 * q, c = pj[s] * 16 + li; pa[s][q] = (on && row < d && c < d) ? P[(size_t)row + (size_t)c * ld] : 0.0; }
                { const int row = pj[s] * 16 + lk + 4 * q, c = pi[s] * 16 + li; pb[s][q] = (on && row < d && c < d) ? P[(size_t)row + (size_t)c * ld] : 0.0; }
            }
        }
#pragma unroll
        for (int u = 0; u < JB_NWA; ++u) { const int e = tid + u * JB_THREADS, k = e / c6, j = e - k * c6; if (k < c6) R3[k * LS + j] = vw[u]; }
#pragma unroll
        for (int u = 0; u < JB_NPC; ++u) { const int e = tid + u * JB_THREADS, k = e / d, r = e - k * d; if (k < c6) R0[r * LS + k] = vp[u]; }
        __syncthreads();
        const int ntj = (c6 + 15) / 16, ntile = nt * ntj;
        // U = Pc W
#pragma unroll 1
        for (int t = wave; t < ntile; t += JB_WAVES) {
            const int it = t / ntj, jt = t - it * ntj, i0 = it * 16, c = jt * 16 + li;
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double av[8], bv[8];
                rows8(R0, i0, h, av); cols8(R3, jt * 16, h, bv);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int row = i0 + lk + 4 * q; if (row < d && c < c6) R1[row * LS + c] = acc[q]; }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < JB_NWA; ++u) { const int e = tid + u * JB_THREADS, k = e / c6, j = e - k * c6; if (k < c6) R3[k * LS + j] = va[u]; }
        __syncthreads();
        // G = U A
#pragma unroll 1
        for (int t = wave; t < ntile; t += JB_WAVES) {
            const int it = t / ntj, jt = t - it * ntj, i0 = it * 16, c = jt * 16 + li;
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double av[8], bv[8];
                rows8(R1, i0, h, av); cols8(R3, jt * 16, h, bv);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int row = i0 + lk + 4 * q; if (row < d && c < c6) R2[row * LS + c] = acc[q]; }
        }
        __syncthreads();
    }
    // X Y^T, tile (i0, j0): rows i0.. of X times rows j0.. of Y
    auto xyT = [&](const double* X, int i0, const double* Y, int j0) -> d4 {
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double av[8], bv[8];
            rows8(X, i0, h, av); rows8(Y, j0, h, bv);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
        return acc;
    };
    // P1 = P - G Pc^T
#pragma unroll
    for (int s = 0; s < JB_PAIRS; ++s) {
        if (wave + JB_WAVES * s < npair) {
            const int I = pi[s], J = pj[s];
            const d4 acc = xyT(R2, I * 16, R0, J * 16);
            if (I != J) {
                const d4 acc2 = xyT(R2, J * 16, R0, I * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) pb[s][q] = pb[s][q] - acc2[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) pa[s][q] = pa[s][q] - acc[q];
            if (I == J) pb[s] = pa[s];
        }
    }
    __syncthreads();            // every read of Pc is done: P1c = P1[:, 24:] takes its place
#pragma unroll
    for (int s = 0; s < JB_PAIRS; ++s) {
        if (wave + JB_WAVES * s < npair) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                { const int row = pi[s] * 16 + lk + 4 * q, c = pj[s] * 16 + li; if (row < d && c < d && c >= 24) R0[row * LS + c - 24] = pa[s][q]; }
                if (pi[s] != pj[s]) { const int row = pj[s] * 16 + lk + 4 * q, c = pi[s] * 16 + li; if (row < d && c < d && c >= 24) R0[row * LS + c - 24] = pb[s][q]; }
            }
        }
    }
    __syncthreads();
    // X = P1 - P1c G^T + s2 G U^T per tile;  P+ = .5 (X + X^T)
    double* const tl = R3 + wave * (16 * 17);
#pragma unroll
    for (int s = 0; s < JB_PAIRS; ++s) {
        if (wave + JB_WAVES * s < npair) {
            const int I = pi[s], J = pj[s];
            d4 xij, xji;
            {
                const d4 acc = xyT(R0, I * 16, R2, J * 16);               // P1c_I G_J^T
                const d4 acc2 = xyT(R2, I * 16, R1, J * 16);              // G_I U_J^T
#pragma unroll
                for (int q = 0; q < 4; ++q) xij[q] = pa[s][q] - acc[q] + s2 * acc2[q];
            }
            xji = xij;
            if (I != J) {
                const d4 acc = xyT(R0, J * 16, R2, I * 16);               // P1c_J G_I^T
                const d4 acc2 = xyT(R2, J * 16, R1, I * 16);              // G_J U_I^T
#pragma unroll
                for (int q = 0; q < 4; ++q) xji[q] = pb[s][q] - acc[q] + s2 * acc2[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) tl[(lk + 4 * q) * 17 + li] = xji[q];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = lk + 4 * q, row = I * 16 + rr, col = J * 16 + li;
                const double v = .5 * (xij[q] + tl[li * 17 + rr]);     // X_JI[col_local][row_local]
                if (row < d && col < d) {
                    Pout[(size_t)row + (size_t)col * ld] = v;
                    if (I != J) Pout[(size_t)col + (size_t)row * ld] = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// =============================================================== single instance, 6n <= 64: the same two stages with every operand staged in LDS
// ug_kernel / final_kernel fetch their MFMA operands from global memory inside the k-loop: 16 dependent rounds of cold loads (the operands
// were written by another CU a few microseconds earlier) per tile, 14 / 11 us per launch for 2-3 us of matrix-core work.  With one
// instance per launch the operands of a workgroup fit its LDS: ONE batch of coalesced loads, then every tile from LDS.
#define UGL_LS 65
#define UGL_LDS_DOUBLES (2 * 64 * UGL_LS + 88 * UGL_LS + 2 * 16 * UGL_LS)
// U = Pc W, G = U A, P1 = P - G Pc^T for one 16-row strip (grid: ceil(d / 16) workgroups); same arithmetic and outputs as ug_kernel
__global__ __launch_bounds__(256) void ug_lds_kernel(DevCfg cfg, int n, const double* __restrict__ P, const double* __restrict__ W, const double* __restrict__ Ab,
                                                     double* __restrict__ U, double* __restrict__ G, double* __restrict__ P1) {
    extern __shared__ __align__(16) double ul[];
    constexpr int LS = UGL_LS;
    double* const Wl = ul;                    // W[k][j], zero-padded to 64 x 64
    double* const Al = Wl + 64 * LS;          // A[k][j]
    double* const Pcl = Al + 64 * LS;         // Pc[r][k] = P[r][24 + k], r < d (<= 88), k < 64
    double* const Us = Pcl + 88 * LS;         // 16 x LS
    double* const Gs = Us + 16 * LS;
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int i0 = blockIdx.x * 16;
    DBG_R(blockIdx.x == 0, 3);
    {
        double vw[16], va[16], vp[22];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + u * 256, k = e >> 6, j = e & 63;
            const bool ok = k < c6 && j < c6;
            vw[u] = ok ? W[(size_t)k * ldh + j] : 0.0;
            va[u] = ok ? Ab[(size_t)k * ldh + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 22; ++u) {      // column-major source: consecutive threads walk down a column of P
            const int e = tid + u * 256, k = e / 88, r = e - k * 88;
            vp[u] = (k < c6 && r < d) ? P[(size_t)r + (size_t)(24 + k) * ld] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int e = tid + u * 256; Wl[(e >> 6) * LS + (e & 63)] = vw[u]; Al[(e >> 6) * LS + (e & 63)] = va[u]; }
#pragma unroll
        for (int u = 0; u < 22; ++u) { const int e = tid + u * 256, k = e / 88, r = e - k * 88; if (k < 64) Pcl[r * LS + k] = vp[u]; }
    }
    __syncthreads();
    const int ntj = (c6 + 15) / 16, nti = (d + 15) / 16;
    const int r = i0 + li;
    // U strip
    for (int jt = wave; jt < ntj; jt += 4) {
        const int c = jt * 16 + li;
        double av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int k = 4 * u + lk; av[u] = (r < d) ? Pcl[r * LS + k] : 0.0; bv[u] = Wl[k * LS + c]; }
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = lk + 4 * q;
            Us[row * LS + c] = acc[q];
            if (i0 + row < d && c < c6) U[(size_t)(i0 + row) * ldh + c] = acc[q];
        }
    }
    __syncthreads();
    // G strip
    for (int jt = wave; jt < ntj; jt += 4) {
        const int c = jt * 16 + li;
        double av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int k = 4 * u + lk; av[u] = (k < c6) ? Us[li * LS + k] : 0.0; bv[u] = Al[k * LS + c]; }
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = lk + 4 * q;
            Gs[row * LS + c] = acc[q];
            if (i0 + row < d && c < c6) G[(size_t)(i0 + row) * ldh + c] = acc[q];
        }
    }
    __syncthreads();
    // P1 strip = P - G Pc^T
    for (int jt = wave; jt < nti; jt += 4) {
        const int c = jt * 16 + li;
        double av[16], bv[16], p0[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = i0 + lk + 4 * q; p0[q] = (row < d && c < d) ? P[(size_t)row + (size_t)c * ld] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int k = 4 * u + lk; av[u] = (k < c6) ? Gs[li * LS + k] : 0.0; bv[u] = (c < d) ? Pcl[c * LS + k] : 0.0; }
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = i0 + lk + 4 * q; if (row < d && c < d) P1[(size_t)row + (size_t)c * ld] = p0[q] - acc[q]; }
    }
}

// Joseph form P+ = sym(P1 - P1c G^T + s2 G U^T): one wave per unordered tile pair, four pairs per workgroup; same arithmetic as final_kernel
#define FNL_LDS_DOUBLES (3 * 88 * UGL_LS + 4 * 16 * 17)
__global__ __launch_bounds__(256) void final_lds_kernel(DevCfg cfg, int n, const double* __restrict__ P1, const double* __restrict__ G, const double* __restrict__ U,
                                                        double* __restrict__ Pout) {
    extern __shared__ __align__(16) double fl[];
    constexpr int LS = UGL_LS;
    double* const P1c = fl;                   // P1[r][24 + k]
    double* const Gl = P1c + 88 * LS;
    double* const Ul = Gl + 88 * LS;
    double (*const tl)[16][17] = reinterpret_cast<double (*)[16][17]>(Ul + 88 * LS);
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    DBG_R(blockIdx.x == 0, 4);
    {
        double vg[22], vu[22], vp[22];
#pragma unroll
        for (int u = 0; u < 22; ++u) {
            const int e = tid + u * 256;
            const int rr = e >> 6, k = e & 63;            // row-major sources (G, U)
            const bool ok = rr < d && k < c6;
            vg[u] = ok ? G[(size_t)rr * ldh + k] : 0.0;
            vu[u] = ok ? U[(size_t)rr * ldh + k] : 0.0;
            const int k2 = e / 88, r2 = e - k2 * 88;      // column-major source (P1)
            vp[u] = (k2 < c6 && r2 < d) ? P1[(size_t)r2 + (size_t)(24 + k2) * ld] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 22; ++u) {
            const int e = tid + u * 256;
            const int rr = e >> 6, k = e & 63;
            if (rr < 88) { Gl[rr * LS + k] = vg[u]; Ul[rr * LS + k] = vu[u]; }
            const int k2 = e / 88, r2 = e - k2 * 88;
            if (k2 < 64) P1c[r2 * LS + k2] = vp[u];
        }
    }
    __syncthreads();
    const int nt = (d + 15) / 16, npair = nt * (nt + 1) / 2;
    const int pr = blockIdx.x * 4 + wave;
    if (pr >= npair) return;
    int I = 0, rem = pr;
    while (rem >= nt - I) { rem -= nt - I; ++I; }
    const int J = I + rem;
    auto xtile = [&](int i0, int j0) -> d4 {
        const int r = i0 + li, c = j0 + li;
        double a1[16], b1[16], p1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = i0 + lk + 4 * q; p1[q] = (row < d && c < d) ? P1[(size_t)row + (size_t)c * ld] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int k = 4 * u + lk; a1[u] = (r < d) ? P1c[r * LS + k] : 0.0; b1[u] = (c < d) ? Gl[c * LS + k] : 0.0; }
        d4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int k = 4 * u + lk; a1[u] = (r < d) ? Gl[r * LS + k] : 0.0; b1[u] = (c < d) ? Ul[c * LS + k] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc2, 0, 0, 0);
        d4 out;
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = p1[q] - acc[q] + s2 * acc2[q];
        return out;
    };
    const d4 xij = xtile(I * 16, J * 16);
    d4 xji = xij;
    if (I != J) xji = xtile(J * 16, I * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) tl[wave][lk + 4 * q][li] = xji[q];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = lk + 4 * q, row = I * 16 + rr, col = J * 16 + li;
        const double v = .5 * (xij[q] + tl[wave][li][rr]);
        if (row < d && col < d) {
            Pout[(size_t)row + (size_t)col * ld] = v;
            if (I != J) Pout[(size_t)col + (size_t)row * ld] = v;
        }
    }
}

// =============================================================== round 4: the two stages above in ONE launch (single instance, 6n <= 60)
// ug_lds_kernel + final_lds_kernel are two launches on the filter stream's serial chain (9.8 + 9.8 us and the boundary between them; U, G and P1
// travel through L2 in between).  Here one workgroup owns an unordered tile pair (I, J) of P+ from the operands to the result: it forms the
// strips it needs itself — U_I, U_J = Pc W;  G_I, G_J = U A;  P1c_I, P1c_J = Pc - G Pcc (the clone columns of P1 = P - G Pc^T) — and then
//      X(I, J) = (P(I, J) - G_I Pc_J^T) - P1c_I G_J^T + s2 G_I U_J^T,        P+(I, J) = .5 (X(I, J) + X(J, I)^T)
// exactly as the two kernels do (same operand order in every product, the same rounded intermediate P1 tile), so the result is theirs bit for
// bit.  The strips are recomputed by every pair that touches them (~3.5x the products of the two-kernel form, spread over 21 CUs instead of 6).
// Measured (tools/solve_probe.py, full-load update at cfg B): 16.8 us against 9.2 + 8.9 us for the two launches — the workgroup is bound by
// its 480 FP64 MFMAs on four SIMDs (64 cycles each) plus the operand reads in front of every chain (loads 10.5 k cycles, the three strip
// rounds ~7 k each, the closing stage 4.3 k) —, so what the fusion buys the pipelined frame is one launch boundary and one launch less for
// the host: +1-2 % frames/s in same-box A/Bs (8.39 / 8.42 k against 8.32 / 8.30 k; a box whose host is the limit: 7.58 / 7.70 against 7.56 / 7.53).
#define JL_LS 61
#define JL_LDS_DOUBLES (3 * 60 * JL_LS + 8 * 16 * JL_LS + 16)
__global__ __launch_bounds__(256) void joseph_lds_kernel(DevCfg cfg, int n, const double* __restrict__ P, const double* __restrict__ W, const double* __restrict__ Ab,
                                                         double* __restrict__ Pout, FilterMeta* __restrict__ meta, const double* __restrict__ x, double* __restrict__ x_out,
                                                         const double* __restrict__ dx_scr, int n_pairs, int dx_nt) {
    // dx_scr != NULL (round 6): the workgroups behind the n_pairs tile pairs are the solve's dx / state-injection roles (solve9.hip s9_dx_role): dx = Pc y
    // needs W complete exactly like U = Pc W, and nothing in this launch needs dx — 4.7 us less on the filter chain's serial path
    if (dx_scr && (int)blockIdx.x >= n_pairs) {
        __shared__ S9DxLds dxl;
        s9_dx_role(cfg, meta, n, Ab, x, P, dx_scr, x_out, dx_nt, (int)blockIdx.x - n_pairs, dxl);
        return;
    }
    extern __shared__ __align__(16) double jl[];
    constexpr int LS = JL_LS;
    double* const Wl = jl;                         // W[k][j]      (c6 x c6)
    double* const Al = Wl + 60 * LS;               // A[k][j]
    double* const Ccl = Al + 60 * LS;              // Pcc[c][k] = P[24 + c][24 + k]
    // strips I (s = 0) and J (s = 1), 16 x LS each, strip s at base + s * 16 * LS
    double* const PcS = Ccl + 60 * LS;             // Pc strips: Pc[r][k] = P[r][24 + k]
    double* const Us = PcS + 2 * 16 * LS;
    double* const Gs = Us + 2 * 16 * LS;
    double* const Qs = Gs + 2 * 16 * LS;           // P1c strips
    constexpr int SS = 16 * LS;                    // strip stride
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    DBG_R(blockIdx.x == 0, 3);
    const int nt = (d + 15) / 16;
    int I = 0, rem = blockIdx.x;
    while (rem >= nt - I) { rem -= nt - I; ++I; }
    const int J = I + rem;
    // (first rows of strip I and strip J; selected with a compare, never indexed with a run-time value: a two-element array indexed by `sidx & 1` lived in
    //  SCRATCH — a private-memory load in front of every strip load and operand fetch of this kernel, round 6)
    const int r0[2] = {I * 16, J * 16};
    auto r0of = [&](int sidx) { return sidx ? J * 16 : I * 16; };
    DBG_T(20);
    double p0[4] = {0, 0, 0, 0}, p0t[4] = {0, 0, 0, 0};   // wave 0: the P tiles of the closing stage, in flight from the start — p0[q] = P(I16 + lk + 4q, J16 + li),
    if (wave == 0) {                                        // p0t[q] = P(J16 + li, I16 + lk + 4q) (the (J, I) tile in the transposed lane layout)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = r0[0] + lk + 4 * q, c = r0[1] + li;
            const bool ok = row < d && c < d;
            p0[q] = ok ? P[(size_t)row + (size_t)c * ld] : 0.0;
            p0t[q] = ok ? P[(size_t)c + (size_t)row * ld] : 0.0;
        }
    }
    // ONE batch of coalesced loads: W and the two Pc strips first (what U = Pc W needs), then A and Pcc (column-major source) — those two stay in
    // registers through the first strip round and go to LDS behind it (round 6; vmcnt retires in issue order).
    // NO predicated load anywhere in this kernel (round 6): `ok ? mem[i] : 0.0` compiles to an exec-mask save / restore around every single load — ten
    // instructions per LDS read, 403 such sequences, ~3 k of a strip round's ~6 k cycles.  Global loads take a clamped address and a select; the LDS
    // operands need no mask at all: every buffer is zero beyond 6n columns / d rows (the staging writes zeros there, the strips' outputs start zeroed
    // and are stored up to column 60), so the products add the same exact zeros the masks produced.  Columns 60..63 of a tile's B operand read past
    // the row (finite or not: a column of the product nobody stores; the A operand is always initialised).
    auto ldz = [](const double* __restrict__ p, size_t i, bool ok) { const double v = p[ok ? i : 0]; return ok ? v : 0.0; };
    double vw[15], va[15], vc[15], vs[8];
#pragma unroll
    for (int u = 0; u < 15; ++u) {
        const int e = tid + u * 256, k = e / 60, j = e - k * 60;            // 3600 = 60 x 60 elements
        vw[u] = ldz(W, (size_t)k * ldh + j, e < 3600 && k < c6 && j < c6);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = tid + u * 256, sidx = e / 960, ee = e - sidx * 960, k = ee >> 4, rr = ee & 15;     // 2 strips x 60 columns x 16 rows
        const int row = r0of(sidx & 1) + rr;
        vs[u] = ldz(P, (size_t)row + (size_t)(24 + k) * ld, e < 1920 && k < c6 && row < d);
    }
#pragma unroll
    for (int u = 0; u < 15; ++u) {
        const int e = tid + u * 256, k = e / 60, j = e - k * 60;
        va[u] = ldz(Ab, (size_t)k * ldh + j, e < 3600 && k < c6 && j < c6);
    }
#pragma unroll
    for (int u = 0; u < 15; ++u) {
        const int e = tid + u * 256, k = e / 60, j = e - k * 60;
        vc[u] = ldz(P, (size_t)(24 + j) + (size_t)(24 + k) * ld, e < 3600 && k < c6 && j < c6);      // element (row c = j, k) of Pcc read down column k: consecutive threads, consecutive rows
    }
    for (int e = tid; e < 6 * SS; e += 256) Us[e] = 0.0;     // U, G, P1c strips (contiguous): zero beyond the columns the rounds store
#pragma unroll
    for (int u = 0; u < 15; ++u) {
        const int e = tid + u * 256, k = e / 60, j = e - k * 60;
        if (e < 3600) Wl[k * LS + j] = vw[u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = tid + u * 256, sidx = e / 960, ee = e - sidx * 960, k = ee >> 4, rr = ee & 15;
        if (e < 1920) PcS[(sidx & 1) * SS + rr * LS + k] = vs[u];
    }
    __syncthreads();
    DBG_T(21);
    const int ntj = (c6 + 15) / 16;
    // strip products: out[s][row][c] = sum_k X[s][row][k] * Y(k, c) for the 2 strips x ntj column tiles.  A wave takes tiles `wave` and `wave + 4`
    // TOGETHER: all operands of both in flight, then the two accumulation chains interleaved (a lone chain of dependent FP64 MFMAs leaves the
    // matrix pipe idle between issues; the pipe of one SIMD — 64 cycles per 16x16x4 — is what bounds this kernel).  Yt: Y(k, c) = Ym[c * LS + k]
    auto strips = [&](const double* X, const double* Ym, bool Yt, double* out, const double* sub) {
        const int t0 = wave, t1 = wave + 4, ntile = 2 * ntj;
        const bool v0 = t0 < ntile, v1 = t1 < ntile;      // (a tile past the last one: computed on clamped operands, not stored)
        const int s0 = min(t0 / ntj, 1), c0 = (t0 % ntj) * 16 + li, s1 = min(t1 / ntj, 1), c1 = (t1 % ntj) * 16 + li;
        double a0[15], b0[15], a1[15], b1[15];     // 6n <= 60: fifteen k-blocks (a sixteenth would add zeros)
#pragma unroll
        for (int u = 0; u < 15; ++u) {
            const int k = 4 * u + lk;
            a0[u] = X[s0 * SS + li * LS + k];
            b0[u] = Yt ? Ym[c0 * LS + k] : Ym[k * LS + c0];
            a1[u] = X[s1 * SS + li * LS + k];
            b1[u] = Yt ? Ym[c1 * LS + k] : Ym[k * LS + c1];
        }
        d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 15; ++u) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc1, 0, 0, 0);
        }
        const bool st0 = v0 && c0 < 60, st1 = v1 && c1 < 60;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = lk + 4 * q;
            if (st0) out[s0 * SS + row * LS + c0] = sub ? (sub[s0 * SS + row * LS + min(c0, 59)] - acc0[q]) : acc0[q];
            if (st1) out[s1 * SS + row * LS + c1] = sub ? (sub[s1 * SS + row * LS + min(c1, 59)] - acc1[q]) : acc1[q];
        }
    };
    strips(PcS, Wl, false, Us, nullptr);             // U = Pc W
#pragma unroll
    for (int u = 0; u < 15; ++u) {
        const int e = tid + u * 256, k = e / 60, j = e - k * 60;
        if (e < 3600) { Al[k * LS + j] = va[u]; Ccl[j * LS + k] = vc[u]; }
    }
    __syncthreads();
    DBG_T(22);
    strips(Us, Al, false, Gs, nullptr);              // G = U A
    __syncthreads();
    DBG_T(23);
    strips(Gs, Ccl, true, Qs, PcS);                  // P1c = Pc - G Pcc^T  (P1[r][24 + c] = P[r][24 + c] - sum_k G[r][k] Pc[24 + c][k])
    __syncthreads();
    DBG_T(24);
    // The closing stage: six 16 x 16 x 60 products — for X(I, J): a = G_I Pc_J^T, b = P1c_I G_J^T, c = G_I U_J^T; for X(J, I) the same with
    // the strips exchanged — spread over the four waves (products w and w + 4), parked in LDS (W's buffer: W is dead), combined by wave 0 in
    // the two kernels' order:  x = ((p0 - a) - b) + s2 c,  P+(I, J) = .5 (x_IJ + x_JI^T).   I == J: three products, x_JI = x_IJ.
    double (*const xt)[16][17] = reinterpret_cast<double (*)[16][17]>(Wl);
    const int nprod = (I == J) ? 3 : 6;
    auto product = [&](int pidx) -> d4 {
        const int sa = pidx / 3, sb = sa ^ 1, kind = pidx - 3 * sa;       // rows of strip sa, columns of strip sb (rows past d are zero in every strip)
        const double* Xa = (kind == 1 ? Qs : Gs) + sa * SS;
        const double* Yb = (kind == 0 ? PcS : (kind == 1 ? Gs : Us)) + sb * SS;
        double a1[15], b1[15];
#pragma unroll
        for (int u = 0; u < 15; ++u) { const int k = 4 * u + lk; a1[u] = Xa[li * LS + k]; b1[u] = Yb[li * LS + k]; }
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 15; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc, 0, 0, 0);
        return acc;
    };
    {
        const bool two = wave + 4 < nprod;
        d4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
        if (wave < nprod) pa = product(wave);
        if (two) pb = product(wave + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (wave < nprod) xt[wave][lk + 4 * q][li] = pa[q];
            if (two) xt[wave + 4][lk + 4 * q][li] = pb[q];
        }
    }
    __syncthreads();
    DBG_T(25);
    if (wave == 0) {
        const int o = (I == J) ? 0 : 3;              // tiles of X(J, I) (X(I, I) itself on the diagonal)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = lk + 4 * q, row = I * 16 + rr, col = J * 16 + li;
            const double p1 = p0[q] - xt[0][rr][li];
            const double xij = p1 - xt[1][rr][li] + s2 * xt[2][rr][li];
            const double p1t = p0t[q] - xt[o][li][rr];
            const double xji = p1t - xt[o + 1][li][rr] + s2 * xt[o + 2][li][rr];      // element (row li of strip J, column rr of strip I) of X(J, I)
            const double v = .5 * (xij + xji);
            if (row < d && col < d) {
                Pout[(size_t)row + (size_t)col * ld] = v;
                if (I != J) Pout[(size_t)col + (size_t)row * ld] = v;
            }
        }
    }
}
