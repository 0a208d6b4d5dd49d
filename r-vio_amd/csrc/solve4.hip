// solve4.hip — W = T^-1, y = W b, dx = Pc y, state injection  (Updater.cc:540-613), generation 4.
//
// One workgroup of 256 threads (one wave per SIMD).  IN-PLACE Gauss-Jordan inversion of T with partial pivoting on
// the tableau M = [T | b]  (c6 x (c6+1)):
//   * no row swaps (step k uses the not-yet-used row p_k with the largest |M[i][k]|), deferred pivot scaling;
//   * every other row i:  f = M[i][k]/piv;  M[i][j] -= f M[p][j] (j != k);  M[i][k] = -f;  then M[p][k] := 1;
//   * lane <-> column (NCH chunks of 64 columns, a template parameter so that cfg A/B/D compile to ONE chunk... two for
//     c6 > 63, three for c6 > 127), wave <-> every 4th row, 8 rows in flight per batch;
//   * the arg-max for column k+1 is folded into the elimination of step k: ONE barrier per column.
// Result: T^-1[k][p_j] = M[p_k][j] / piv_k,  y[k] = M[p_k][c6] / piv_k.
// The first profile of generation 3 showed the loop to be INSTRUCTION bound (~80 instructions per row); this version
// keeps the per-row work to two LDS reads, a multiply, NCH fused multiply-adds/stores and a 4-instruction candidate update.
#pragma once
#include "rvio_dev.h"

#define SOLVE4_T 256
#define SOLVE4_NW 4

template <bool USE_LDS, int NCH>
__device__ __forceinline__ void solve4_body(const DevCfg& cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Tg,
                                            const double* __restrict__ Ab, const double* __restrict__ x, const double* __restrict__ P,
                                            double* __restrict__ Wout, double* __restrict__ x_out, double* __restrict__ Mg, double* sh) {
    __shared__ int s_prow[6 * RVIO_MAX_LEN], s_invp[6 * RVIO_MAX_LEN];
    __shared__ double s_ipiv[6 * RVIO_MAX_LEN];
    __shared__ double s_y[6 * RVIO_MAX_LEN];
    __shared__ double s_dx[24 + 6 * RVIO_MAX_LEN];
    __shared__ double s_cv[2][SOLVE4_NW];   // per-wave candidate |value| for the next pivot (double-buffered by step parity)
    __shared__ int s_ci[2][SOLVE4_NW];
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax, xd = 26 + 7 * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NC = c6 + 1;
    const int ldm = USE_LDS ? (NC | 1) : (2 * ldh);
    double* M = USE_LDS ? sh : Mg;
    const int n_good = (int)Ab[(size_t)ldh * (ldh - 1)], n_rows = (int)Ab[(size_t)ldh * (ldh - 1) + 1];
    const bool upd = n_good > 2;                       // Updater.cc:460
    if (tid == 0) { meta->n_good = n_good; meta->n_rows = n_rows; meta->updated = upd ? 1 : 0; meta->trunc_at = (int)Ab[(size_t)ldh * (ldh - 1) + 2]; }
    if (!upd) {                                        // pass-through (Updater.cc:621-627): W = 0 => U = G = 0 => P+ = P exactly
        for (int e = tid; e < c6 * c6; e += SOLVE4_T) Wout[(size_t)(e / c6) * ldh + (e % c6)] = 0.0;
        for (int i = tid; i < xd; i += SOLVE4_T) x_out[i] = x[i];
        return;
    }
    for (int i = wv; i < c6; i += SOLVE4_NW)
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int j = lane + 64 * u;
            if (j < NC) M[i * ldm + j] = (j < c6) ? Tg[(size_t)i * ldh + j] : Ab[(size_t)i * ldh + c6];
        }
    const int nrw = (c6 > wv) ? (c6 - wv + SOLVE4_NW - 1) / SOLVE4_NW : 0;   // rows of this wave: i = wv + NW*q, q < nrw
    const int rstride = SOLVE4_NW * ldm, rbase = wv * ldm;
    unsigned long long usedmask = 0;                                        // bit q: row wv + NW*q was a pivot row already
    __syncthreads();
    {   // first pivot: arg-max of column 0 over this wave's rows
        double best = -1.0; int bi = 0;
        for (int i = wv; i < c6; i += SOLVE4_NW) { const double v = fabs(M[i * ldm]); if (v > best) { best = v; bi = i; } }
        if (lane == 0) { s_cv[0][wv] = best; s_ci[0][wv] = bi; }
    }
    __syncthreads();
    int ppr = -1;
    for (int k = 0; k < c6; ++k) {
        const int par = k & 1;
        if (k == 30) DBG_T(50);
        // the previous pivot row's column entry becomes 1 (stored form of 1/piv) only now, after the barrier
        if (ppr >= 0 && (ppr & (SOLVE4_NW - 1)) == wv && lane == ((k - 1) & 63)) M[ppr * ldm + (k - 1)] = 1.0;
        double best = s_cv[par][0]; int pr = s_ci[par][0];
#pragma unroll
        for (int w = 1; w < SOLVE4_NW; ++w) { const double v = s_cv[par][w]; const int ii = s_ci[par][w]; if (v > best || (v == best && ii < pr)) { best = v; pr = ii; } }
        ppr = pr;
        if (k == 30) DBG_T(51);
        const int prow = pr * ldm;
        const double ipiv = 1.0 / M[prow + k];
        if (tid == 0) { s_prow[k] = pr; s_invp[pr] = k; s_ipiv[k] = ipiv; if (!(best > 0)) atomicOr(&meta->err, 1); }
        double prv[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) { const int j = lane + 64 * u; prv[u] = (j < NC) ? M[prow + j] : 0.0; }
        const int qpr = ((pr & (SOLVE4_NW - 1)) == wv) ? (pr / SOLVE4_NW) : -1;   // index of the pivot row among this wave's rows
        if (qpr >= 0) usedmask |= 1ull << qpr;
        if (k == 30) DBG_T(52);
        const int uk1 = (k + 1) >> 6, lk1 = (k + 1) & 63;      // chunk / lane that owns column k+1
        const int uk = k >> 6, lk0 = k & 63;
        // Branch-free elimination: dead rows (pivot row, tail of the last batch) and out-of-range lanes store into a dump
        // slot behind the tableau instead of being masked, row offsets advance by addition, the candidate for the next
        // pivot is kept with selects.  `dump` lives in the slack the host allocates after the tableau.
        double nbest = -1.0; int nbq = 0;
        const int dump = c6 * ldm + lane;
        bool colok[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) colok[u] = lane + 64 * u < NC;
        for (int q0 = 0; q0 < nrw; q0 += 8) {
            double fb[8], mv[8][NCH];
            int ro = rbase + q0 * rstride;
            const int rlast = rbase + (nrw - 1) * rstride;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int rr_ = (ro < rlast) ? ro : rlast;             // clamp: tail slots re-read the last row, their stores are dumped
                fb[b] = M[rr_ + k];
#pragma unroll
                for (int u = 0; u < NCH; ++u) mv[b][u] = M[rr_ + lane + 64 * u];
                ro += rstride;
            }
            ro = rbase + q0 * rstride;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int q = q0 + b;
                const bool live = (q < nrw) & (q != qpr);
                const double f = fb[b] * ipiv;
                double cand = 0;
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    double nvv = mv[b][u] - f * prv[u];
                    nvv = (u == uk && lane == lk0) ? -f : nvv;
                    M[(live & colok[u]) ? (ro + lane + 64 * u) : dump] = nvv;
                    cand = (u == uk1) ? nvv : cand;
                }
                const double av = fabs(cand);
                const bool take = live & !((usedmask >> q) & 1ull) & (av > nbest);
                nbest = take ? av : nbest; nbq = take ? q : nbq;
                ro += rstride;
            }
        }
        if (k == 30) DBG_T(53);
        // the lane that owns column k+1 publishes this wave's candidate
        if (k + 1 < c6 && lane == lk1) { s_cv[par ^ 1][wv] = nbest; s_ci[par ^ 1][wv] = wv + SOLVE4_NW * nbq; }
        __syncthreads();
        if (k == 30) DBG_T(54);
        if (k == 31) DBG_T(55);
    }
    if ((ppr & (SOLVE4_NW - 1)) == wv && lane == ((c6 - 1) & 63)) M[ppr * ldm + (c6 - 1)] = 1.0;
    __syncthreads();
    // read the result out: W[k][p_j] = M[p_k][j] * ipiv_k ;  y[k] = M[p_k][c6] * ipiv_k
    for (int k = wv; k < c6; k += SOLVE4_NW) {
        const int ro = s_prow[k] * ldm; const double ip = s_ipiv[k];
        for (int c = lane; c < c6; c += 64) Wout[(size_t)k * ldh + c] = M[ro + s_invp[c]] * ip;
        if (lane == 0) s_y[k] = M[ro + c6] * ip;
    }
    __syncthreads();
    // dx = K r = Pc y   (Updater.cc:544)
    for (int i = tid; i < d; i += SOLVE4_T) {
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
    for (int p = tid - 64; p >= 0 && p < n; p += SOLVE4_T - 64) {
        stq(x_out + 26 + 7 * p, qmul(small_q(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), ldq(x + 26 + 7 * p)));
        for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
    }
}

template <int NCH>
__global__ __launch_bounds__(SOLVE4_T) void solve4_kernel_lds(DevCfg cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Tg,
                                                               const double* __restrict__ Ab, const double* __restrict__ x, const double* __restrict__ P,
                                                               double* __restrict__ Wout, double* __restrict__ x_out) {
    extern __shared__ __align__(16) double sh[];
    solve4_body<true, NCH>(cfg, meta, n, Tg, Ab, x, P, Wout, x_out, nullptr, sh);
}
__global__ __launch_bounds__(SOLVE4_T) void solve4_kernel_glb(DevCfg cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Tg,
                                                               const double* __restrict__ Ab, const double* __restrict__ x, const double* __restrict__ P,
                                                               double* __restrict__ Wout, double* __restrict__ x_out, double* __restrict__ Mg) {
    solve4_body<false, 3>(cfg, meta, n, Tg, Ab, x, P, Wout, x_out, Mg, nullptr);
}
