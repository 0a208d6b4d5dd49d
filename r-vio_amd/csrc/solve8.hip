// solve8.hip — W = (s2 I + A Pcc)^-1, y = W b, dx = Pc y, state injection (Updater.cc:540-613) WITHOUT A PIVOT SEARCH, generation 8
// (one instance, 6n <= 64).  MEASURED, NOT ADOPTED (round 4): compiled into the instrumented build only (-DRVIO_DBG_CLOCKS, RVIO_SOLVE8=1;
// tools/solve_probe.py) — correct (the whole GPU suite passed with it as the default: 186 tests) but not faster than solve7.
//
// The idea.  solve7 inverts the non-symmetric T = s2 I + A Pcc by Gauss-Jordan elimination with partial pivoting.  T is a product of two
// symmetric positive definite matrices, T = (s2 Pcc^-1 + A) Pcc, and SPD matrices need no pivoting (growth factor 1):
//      B  = s2 Pcc^-1 + A              SPD
//      W0 = Pcc^-1 B^-1                = T^-1 up to the rounding of the two inversions (relative ~eps cond(Pcc), ~1e-11; cond(Pcc) ~2e5)
//      R  = I - T W0,   W = W0 + W0 R  one Newton-Schulz step against T itself: what is left is ~eps cond(T), cond(T) <= ~1e2 on every
//                                      sequence tried (tools/update_forms_study.py) = the accuracy of the reference's partial-pivot LU
// Both inversions are the SAME unpivoted register-tableau Gauss-Jordan (gj_spd below: lane <-> row, wave <-> every NW-th pair of columns,
// publications through an LDS ring and flags like solve7, pivots on the diagonal): Pcc^-1 does not depend on this frame's measurements and
// runs as one more workgroup of the per-feature launch (pinv_role in feat_prop_kernel: off the serial chain), B^-1 is the one elimination
// left on the chain.  An exactly singular Pcc (a clone augmented from a zeroed pose: an empty IMU batch) is handled by a 1e-30 variance in
// the inversion only; the Newton-Schulz loop is residual-controlled and flags a start that is no inverse at all.
//
// The measurement (MI355X, full-load update at cfg B, shader cycles of thread 0, tools/solve_probe.py):
//                                   solve7 (pivoted)     solve8
//   loads + T (+ tableau)                 13.9 k           11.9 k
//   elimination, 60 columns               58.9 k           41.4 k     <- the search was ~300 of ~980 cycles per column, not 650:
//   W0 = Pcc^-1 B^-1                        —               5.8 k        the cross-wave hand-over and the 16-column update are the rest
//   Newton-Schulz (2 products) + W out      5.0 k (W out)  15.6 k
//   y, dx                                   2.8 k           2.3 k
//   kernel, wall clock                     35.8 us         34.2 us
// The three extra 64^3 products cost one CU's FP64 matrix pipe what the missing search saves (16 tiles x 16 MFMA x 32 cycles / 4 SIMDs = 2 k
// cycles each at best, ~5 k measured with the operand reads and the barrier), and the inverse role makes the per-feature launch carry 18 KB
// more LDS.  Dropping the refinement would leave -8 us but dx errors of 4e-13 instead of 1e-14 per update.  Pipelined frame: 7.55 k frames/s
// with solve8 against 7.80 k with solve7.  Kept as the record of the experiment and as the A/B form.
#pragma once
#include <type_traits>
#include <utility>
#include "rvio_dev.h"
#include "solve7.hip"

#define S8_LS 65           // LDS row stride of a staged 64 x 64 operand
#define S8_RING 16         // publication slots: >= NW + 2 (a wave lags the publisher by less than NW + 1 steps), power of two
#ifndef S8_NW
#define S8_NW 8            // waves of solve8_kernel / of the inverse role (64 columns: four pairs of columns per wave)
#endif
#define S8_CPW (64 / S8_NW)
#define S8_PINV_LDS_DOUBLES (64 * S8_LS)
#define S8_LDS_DOUBLES (4 * 64 * S8_LS)

struct S8Ring { double f[S8_RING][2][64]; int flag[S8_RING]; };

// In-place inverse of the SPD 64 x 64 matrix held as mcol[cc] = column 2 ((cc / 2) NW + wv) + (cc & 1), this lane's row: unpivoted
// Gauss-Jordan, TWO columns per publication (the algebra of solve7 with the identity as pivot order).  On exit
//     inverse[k][j] = mcol(column j)[row k] * ipiv[k],   ipiv (LDS, 64 doubles) = the reciprocals of the pivots.
// nact (a multiple of 2, <= 64): columns / rows >= nact must be identity rows and columns (padding) and are skipped.
// A non-positive or NaN pivot sets the sticky error bit 1 (the singular-pivot flag of solve7).
template <int CPW, int NW>
__device__ __forceinline__ void gj_spd(double (&mcol)[CPW], int nact, int lane, int wv, S8Ring& rg, double* ipiv, FilterMeta* meta) {
    auto rcp_nr = [](double v) { double y = __builtin_amdgcn_rcp(v); y = fma(fma(-v, y, 1.0), y, y); return fma(fma(-v, y, 1.0), y, y); };
    auto elim = [&](double& col, double f, int p) { col -= f * readlane_f64(col, p); };
    auto publish_pair = [&](auto C0tag, int sp) {
        constexpr int C0 = decltype(C0tag)::value;
        const int pa = 2 * sp, pb = 2 * sp + 1;
        const double da = readlane_f64(mcol[C0], pa);
        const double ipa = rcp_nr(da);
        const double fa = (lane == pa) ? 0.0 : mcol[C0] * ipa;
        elim(mcol[C0 + 1], fa, pa);
        const double db = readlane_f64(mcol[C0 + 1], pb);
        const double ipb = rcp_nr(db);
        const double fb = (lane == pb) ? 0.0 : mcol[C0 + 1] * ipb;
        const int slot = sp & (S8_RING - 1);
        rg.f[slot][0][lane] = fa; rg.f[slot][1][lane] = fb;
        if (lane == 0) {
            ipiv[pa] = ipa; ipiv[pb] = ipb;
            if (!(da > 0.0) || !(db > 0.0)) atomicOr(&meta->err, 1);
        }
        // LDS operations of one wave complete in order: the flag becomes visible after the data
        __hip_atomic_store(&rg.flag[slot], sp + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if (wv == 0) publish_pair(std::integral_constant<int, 0>{}, 0);
    const int npair = nact / 2;
    bool done = false;
    s7_for<0, CPW / 2>([&](auto Htag) {
        constexpr int C = 2 * decltype(Htag)::value;
        for (int w = 0; w < NW && !done; ++w) {
            const int sp = (C / 2) * NW + w;
            if (sp >= npair) { done = true; break; }
            const int slot = sp & (S8_RING - 1);
            const int pa = 2 * sp, pb = 2 * sp + 1;
            double fa, fb;
            for (;;) {
                const int fl = __hip_atomic_load(&rg.flag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                fa = *(volatile double*)&rg.f[slot][0][lane]; fb = *(volatile double*)&rg.f[slot][1][lane];
                __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the reads were issued in this order and complete in order
                if (__builtin_amdgcn_readfirstlane(fl) == sp + 1) break;
                if (NW > 4) __builtin_amdgcn_s_sleep(1);   // more than one wave per SIMD: a spinning wave must not take the publisher's issue slots
            }
            auto both = [&](double& col) { elim(col, fa, pa); elim(col, fb, pb); };
            const bool own_next = (sp + 1 < npair) && (wv == ((w + 1 < NW) ? w + 1 : 0));
            const bool next_same = (w + 1 < NW);
            if (own_next) {
                if (next_same) { both(mcol[C]); both(mcol[C + 1]); publish_pair(std::integral_constant<int, C>{}, sp + 1); }
                else if constexpr (C + 3 < CPW) { both(mcol[C + 2]); both(mcol[C + 3]); publish_pair(std::integral_constant<int, C + 2>{}, sp + 1); }
            }
            const bool own = (wv == w);
            s7_for<0, CPW>([&](auto Itag) {
                constexpr int I = decltype(Itag)::value;
                if constexpr (I == C) {
                    if (own) { mcol[I] = (lane == pa) ? 1.0 : -fa; elim(mcol[I], fb, pb); }
                    else if (!(own_next && next_same)) both(mcol[I]);
                } else if constexpr (I == C + 1) {
                    if (own) mcol[I] = (lane == pb) ? 1.0 : -fb;
                    else if (!(own_next && next_same)) both(mcol[I]);
                } else if constexpr (I == C + 2 || I == C + 3) {
                    if (!(own_next && !next_same)) both(mcol[I]);
                } else both(mcol[I]);
            });
        }
    });
}

// one 16 x 16 tile of X Y on the FP64 matrix cores: acc[r] = sum_k X(i0 + lk + 4 r, k) Y(k, j0 + li), both operands in LDS with arbitrary strides
// (xs_r / xs_k: row / k stride of X; ys_k / ys_c: k / column stride of Y), k = 0 .. 63, every operand in flight before the first MFMA
__device__ __forceinline__ s7_d4 s8_tile(const double* X, int xs_r, int xs_k, const double* Y, int ys_k, int ys_c, int i0, int j0, int li, int lk) {
    const double* ap = X + (i0 + li) * xs_r;
    const double* bp = Y + (j0 + li) * ys_c;
    double av[16], bv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { av[u] = ap[(4 * u + lk) * xs_k]; bv[u] = bp[(4 * u + lk) * ys_k]; }
    s7_d4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    return acc;
}

// ---------------------------------------------------------------- Pcc^-1 (off the serial chain)
// One workgroup of 64 NW threads: Pinv (64 x 64 row-major, rows / columns >= 6n hold the identity) = inverse of the clone block of P.
// lds: S8_PINV_LDS_DOUBLES doubles of dynamic LDS (any kernel's extern array).
template <int NW>
__device__ __forceinline__ void pinv_role(const DevCfg& cfg, int n, const double* __restrict__ P, double* __restrict__ Pinv, FilterMeta* meta, double* lds) {
    constexpr int CPW = 64 / NW, NT = 64 * NW;
    __shared__ S8Ring rg;
    __shared__ double s_ip[64];
    const int c6 = 6 * n, ld = cfg.dmax;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid >= NT) return;                                   // (hosted by a kernel with more threads)
    double* Ps = lds;                                        // Ps[i][j] = Pcc[i][j] (symmetric), identity on the padding
    if (tid < S8_RING) rg.flag[tid] = 0;
    {
        constexpr int NB = 64 * 64 / NT;
        double v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {                       // column-major source: consecutive threads walk down a column
            const int e = tid + u * NT, j = e >> 6, i = e & 63;
            v[u] = (i < c6 && j < c6) ? P[(size_t)(24 + i) + (size_t)(24 + j) * ld] : ((i == j) ? 1.0 : 0.0);
            // A clone whose covariance is EXACTLY zero (augmentation copies the current relative pose, which composition has just zeroed, when an
            // empty IMU batch left it there: Tracker / System keep running when the IMU stream ends) has zero rows and columns in Pcc.  T = s2 I +
            // A Pcc does not mind; the SPD form does.  A variance of 1e-30 in its place is invisible in T W0 (the refinement below uses the true
            // T) and keeps every pivot positive: the inverse carries 1e30 there, B = s2 Pcc^-1 + A ~5e24, and their product is 1 / s2 again.
            if (i == j && i < c6 && v[u] == 0.0) v[u] = 1e-30;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) { const int e = tid + u * NT, j = e >> 6, i = e & 63; Ps[i * S8_LS + j] = v[u]; }
    }
    // (hosted inside feat_prop_kernel: named barrier-free alternative is not needed — every thread of THIS workgroup takes this path)
    __syncthreads();
    double mcol[CPW];
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) mcol[cc] = Ps[lane * S8_LS + 2 * ((cc >> 1) * NW + wv) + (cc & 1)];
    __syncthreads();                                         // flags cleared, tableau loaded
    gj_spd<CPW, NW>(mcol, (c6 + 1) & ~1, lane, wv, rg, s_ip, meta);
    __syncthreads();
    const double ip = (lane < ((c6 + 1) & ~1)) ? s_ip[lane] : 1.0;
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) Ps[lane * S8_LS + 2 * ((cc >> 1) * NW + wv) + (cc & 1)] = mcol[cc] * ip;
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += NT) {                // symmetrised on the way out (the elimination's two triangles differ by rounding)
        const int i = e >> 6, j = e & 63;
        Pinv[e] = 0.5 * (Ps[i * S8_LS + j] + Ps[j * S8_LS + i]);
    }
}
__global__ __launch_bounds__(64 * S8_NW) void pinv_kernel(DevCfg cfg, int n, const double* __restrict__ P, double* __restrict__ Pinv, FilterMeta* meta) {
    extern __shared__ __align__(16) double s8_pl[];
    pinv_role<S8_NW>(cfg, n, P, Pinv, meta, s8_pl);
}

// ---------------------------------------------------------------- the solve on the chain
// Inputs: Ab = [A | b] (row-major, ld = ldh; its spare row carries n_good, n_rows, trunc_at), P (propagated), Pinv (pinv_role, same Pcc).
// Outputs: Wout (c6 x c6, ld = ldh), x_out (injected state), FilterMeta counters — the interface of solve7_kernel.
template <int GJW>      // waves that run the elimination (the others wait at the barrier behind it: they would only share SIMDs with the publisher)
__global__ __launch_bounds__(64 * S8_NW) void solve8_kernel(DevCfg cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Ab,
                                                            const double* __restrict__ x, const double* __restrict__ P,
                                                            const double* __restrict__ Pinv, double* __restrict__ Wout, double* __restrict__ x_out) {
    constexpr int NW = S8_NW, CPW = 64 / GJW, NT = 64 * NW, LS = S8_LS;
    extern __shared__ __align__(16) double s8_dyn[];
    double* const X0 = s8_dyn;                 // A, then B^-1, then the Newton-Schulz iterates alternate between X0 and X1
    double* const X1 = s8_dyn + 64 * LS;       // Pcc (as rows of its transpose = itself), then W0
    double* const X2 = s8_dyn + 2 * 64 * LS;   // Pcc^-1, then R = I - T W
    double* const X3 = s8_dyn + 3 * 64 * LS;   // T
    __shared__ S8Ring rg;
    __shared__ double s_ip[64], s_b[64], s_y[64], s_rm[S8_NW];
    __shared__ double s_part[4 * (24 + 64)];
    __shared__ double s_dx[24 + 64];
    DBG_W(threadIdx.x == 0, 45);
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax, xd = 26 + 7 * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_good = (int)Ab[(size_t)ldh * (ldh - 1)], n_rows = (int)Ab[(size_t)ldh * (ldh - 1) + 1];
    const bool upd = n_good > 2;                       // Updater.cc:460
    DBG_R(true, 2);
    if (tid == 0) { meta->n_good = n_good; meta->n_rows = n_rows; meta->updated = upd ? 1 : 0; meta->trunc_at = (int)Ab[(size_t)ldh * (ldh - 1) + 2]; }
    if (!upd) {                                        // pass-through (Updater.cc:621-627): W = 0 => U = G = 0 => P+ = P exactly
        for (int e = tid; e < c6 * c6; e += NT) Wout[(size_t)(e / c6) * ldh + (e % c6)] = 0.0;
        for (int i = tid; i < xd; i += NT) x_out[i] = x[i];
        return;
    }
    DBG_T(56);
    if (tid < S8_RING) rg.flag[tid] = 0;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    // ---- ONE batch of coalesced loads: A, Pcc, Pcc^-1 (zero-padded to 64 x 64; identity / s2 on the padding's diagonals), b
    const int dx_pt = tid / d, dx_i = tid - dx_pt * d;   // dx = Pc y: row dx_i, column share dx_pt (16 columns each)
    double pcv[16];
    {
        constexpr int NB = 64 * 64 / NT;
        double va[NB], vp[NB], vi[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = tid + u * NT, a = e >> 6, k = e & 63;
            const bool ok = a < c6 && k < c6;
            va[u] = ok ? Ab[(size_t)a * ldh + k] : 0.0;
            vp[u] = ok ? P[(size_t)(24 + k) + (size_t)(24 + a) * ld] : ((a == k) ? 1.0 : 0.0);
            vi[u] = Pinv[e];
        }
        if (tid < 64) s_b[tid] = (tid < c6) ? Ab[(size_t)tid * ldh + c6] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {                   // this thread's share of Pc for dx = Pc y at the very end: in flight from the start
            const int k = dx_pt * 16 + u;
            pcv[u] = (dx_pt < 4 && dx_i < d && k < c6) ? P[(size_t)dx_i + (size_t)(24 + k) * ld] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) { const int e = tid + u * NT, a = e >> 6, k = e & 63; X0[a * LS + k] = va[u]; X1[a * LS + k] = vp[u]; X2[a * LS + k] = vi[u]; }
    }
    __syncthreads();
    DBG_T(57);
    const int li = lane & 15, lk = lane >> 4;
    // ---- T = s2 I + A Pcc  -> X3 (Pcc symmetric: Y(k, j) = X1[j][k]);   the tableau of B = s2 Pcc^-1 + A from X2, X0
    for (int t = wv; t < 16; t += NW) {
        const int i0 = (t >> 2) * 16, j0 = (t & 3) * 16;
        const s7_d4 acc = s8_tile(X0, LS, 1, X1, 1, LS, i0, j0, li, lk);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = i0 + lk + 4 * r, col = j0 + li; X3[row * LS + col] = acc[r] + ((row == col) ? s2 : 0.0); }
    }
    double mcol[CPW];
    if (wv < GJW) {
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) { const int j = 2 * ((cc >> 1) * GJW + wv) + (cc & 1); mcol[cc] = s2 * X2[lane * LS + j] + X0[lane * LS + j]; }
    }
    __syncthreads();                                    // T complete, flags cleared, tableau loaded (X0 is free from here)
    DBG_T(58);
    if (wv < GJW) gj_spd<CPW, GJW>(mcol, (c6 + 1) & ~1, lane, wv, rg, s_ip, meta);
    __syncthreads();
    DBG_T(60);
    if (wv < GJW) {   // B^-1 -> X0 (rows scaled by the pivots' reciprocals; on the padding B = s2 I: its inverse is 1 / s2 there)
        const double ip = (lane < ((c6 + 1) & ~1)) ? s_ip[lane] : 1.0 / (s2 * s2);   // (padding: the untouched diagonal entry is s2, its inverse 1 / s2)
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) X0[lane * LS + 2 * ((cc >> 1) * GJW + wv) + (cc & 1)] = mcol[cc] * ip;
    }
    __syncthreads();
    // ---- W0 = Pcc^-1 B^-1 -> X1   (B^-1 symmetric up to rounding: Y(k, j) = X0[k][j])
    for (int t = wv; t < 16; t += NW) {
        const int i0 = (t >> 2) * 16, j0 = (t & 3) * 16;
        const s7_d4 acc = s8_tile(X2, LS, 1, X0, LS, 1, i0, j0, li, lk);
#pragma unroll
        for (int r = 0; r < 4; ++r) X1[(i0 + lk + 4 * r) * LS + j0 + li] = acc[r];
    }
    __syncthreads();
    DBG_T(63);
    // ---- Newton-Schulz against T itself: R = I - T W0 -> X2, W0 <- W0 + W0 R (into the idle buffer), until |R|max^2 is below double precision.
    // One step in every run so far (|R|max ~ 1e-11: the rounding of two unpivoted inversions, ~eps cond(Pcc)); an ill-conditioned Pcc takes a
    // second or third; |R|max >= 0.5 means the start was no inverse at all (Pcc not positive definite): sticky error bit 1, as solve7 flags
    // a singular pivot.
    double* Wc = X1; double* Wn = X0;
    for (int it = 0; it < 4; ++it) {
        double rm = 0;
        for (int t = wv; t < 16; t += NW) {
            const int i0 = (t >> 2) * 16, j0 = (t & 3) * 16;
            const s7_d4 acc = s8_tile(X3, LS, 1, Wc, LS, 1, i0, j0, li, lk);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + lk + 4 * r, col = j0 + li;
                const double rv = ((row == col) ? 1.0 : 0.0) - acc[r];
                X2[row * LS + col] = rv;
                if (row < c6 && col < c6) rm = fmax(rm, (rv == rv) ? fabs(rv) : 1e300);
            }
        }
        for (int o = 32; o > 0; o >>= 1) rm = fmax(rm, __shfl_xor(rm, o, 64));
        if (lane == 0) s_rm[wv] = rm;
        __syncthreads();
        double rmax = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) rmax = fmax(rmax, s_rm[w]);
        if (it == 0 && tid == 0 && !(rmax < 0.5)) atomicOr(&meta->err, 1);
        for (int t = wv; t < 16; t += NW) {
            const int i0 = (t >> 2) * 16, j0 = (t & 3) * 16;
            const s7_d4 acc = s8_tile(Wc, LS, 1, X2, LS, 1, i0, j0, li, lk);
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = i0 + lk + 4 * r, col = j0 + li; Wn[row * LS + col] = Wc[row * LS + col] + acc[r]; }
        }
        __syncthreads();
        double* tsw = Wc; Wc = Wn; Wn = tsw;
        if (rmax * rmax < 1e-17) break;
    }
    // ---- W -> global (the interface of ug_lds_kernel), coalesced rows
    for (int e = tid; e < c6 * c6; e += NT) { const int row = e / c6, col = e - row * c6; Wout[(size_t)row * ldh + col] = Wc[row * LS + col]; }
    DBG_T(61);
    // y = W b: four threads per row, partial sums in a fixed order
    {
        const int row = tid >> 2, q = tid & 3;
        if (row < 64) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += Wc[row * LS + 16 * q + k] * s_b[16 * q + k];
            s_part[q * 64 + row] = acc;
        }
    }
    __syncthreads();
    if (tid < 64) s_y[tid] = ((s_part[tid] + s_part[64 + tid]) + s_part[128 + tid]) + s_part[192 + tid];
    __syncthreads();
    // dx = K r = Pc y (Updater.cc:544): four column shares of 16 per row (Pc prefetched with the kernel's first loads), added in a fixed order
    {
        if (dx_pt < 4) {
            double acc = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += pcv[u] * s_y[min(dx_pt * 16 + u, 63)];
            s_part[dx_pt * d + dx_i] = acc;
        }
        __syncthreads();
        if (tid < d) s_dx[tid] = ((s_part[tid] + s_part[d + tid]) + s_part[2 * d + tid]) + s_part[3 * d + tid];
    }
    __syncthreads();
    DBG_T(62);
    // state injection (Updater.cc:546-613)
    const double* dx = s_dx;
    if (tid == 0) {
        stq(x_out, qmul(small_q(dx[0], dx[1], dx[2]), ldq(x)));
        for (int i = 0; i < 6; ++i) x_out[4 + i] = dx[3 + i] + x[4 + i];
        st3(x_out + 7, unit3(ld3(x_out + 7)));
        stq(x_out + 10, qmul(small_q(dx[9], dx[10], dx[11]), ldq(x + 10)));
        for (int i = 0; i < 12; ++i) x_out[14 + i] = dx[12 + i] + x[14 + i];
    }
    for (int p = tid - 64; p >= 0 && p < n; p += NT - 64) {
        stq(x_out + 26 + 7 * p, qmul(small_q(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), ldq(x + 26 + 7 * p)));
        for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
    }
    DBG_W(tid == 0, 46);
    DBG_W(tid == NT - 1, 47);
    DBG_R(true, 7);
}
