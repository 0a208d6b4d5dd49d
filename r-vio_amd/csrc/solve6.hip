// solve6.hip — W = T^-1, y = W b, dx = Pc y, state injection  (Updater.cc:540-613), generation 6.
//
// In-place Gauss-Jordan inversion of T = s2 I + A Pcc with partial pivoting on the LDS tableau M = [T | b], one
// workgroup of 4 waves (one per SIMD), ONE barrier per column.  With a single wave per SIMD every instruction costs
// ~5 issue cycles, so the step is written to need ~5 instructions per row:
//   * row stride = 64*NCH + 1 doubles: every lane of every row has its own slot, so all stores are unconditional
//     (columns >= c6+1 and the padding rows q >= nrw just hold garbage that is never read back as data);
//   * the pivot row is neutralised by a zero multiplier instead of being skipped;
//   * column k (new value -M[i][k]/piv) and the search for the next pivot are two lane-parallel passes
//     (lane <-> row of this wave) instead of per-row work;
//   * pivot candidates are 64-bit keys (bits of |value| with the 7 low mantissa bits replaced by 127 - row): the
//     cross-wave combine is an unsigned max; the candidate's reciprocal is published with it, so the serial chain of
//     a step contains no division.
// No row swaps, deferred pivot scaling (see solve4.hip for the algebra):  T^-1[k][p_j] = M[p_k][j] / piv_k.
// c6 <= 64*NCH - 1 and ceil(c6/4) <= RPW <= 32; larger windows use solve4_kernel_glb.
#pragma once
#include "rvio_dev.h"

#define SOLVE6_T 256
#define SOLVE6_NW 4

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    int lo = (int)(v & 0xffffffffull), hi = (int)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
template <int L>
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), L), hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), L);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

template <int NCH, int RPW>
__global__ __launch_bounds__(SOLVE6_T) void solve6_kernel(DevCfg cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Tg,
                                                           const double* __restrict__ Ab, const double* __restrict__ x, const double* __restrict__ P,
                                                           double* __restrict__ Wout, double* __restrict__ x_out) {
    extern __shared__ __align__(16) double M[];
    __shared__ int s_prow[6 * RVIO_MAX_LEN], s_invp[6 * RVIO_MAX_LEN];
    __shared__ double s_ipiv[6 * RVIO_MAX_LEN];
    __shared__ double s_y[6 * RVIO_MAX_LEN];
    __shared__ double s_dx[24 + 6 * RVIO_MAX_LEN];
    __shared__ unsigned long long s_key[2][SOLVE6_NW];
    __shared__ double s_rcp[2][SOLVE6_NW];
    constexpr int LDM = 64 * NCH + 1;                   // odd: conflict-free column walks; every lane owns a slot
    constexpr int RS = SOLVE6_NW * LDM;                 // distance between consecutive rows of one wave
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax, xd = 26 + 7 * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n_good = (int)Ab[(size_t)ldh * (ldh - 1)], n_rows = (int)Ab[(size_t)ldh * (ldh - 1) + 1];
    const bool upd = n_good > 2;                       // Updater.cc:460
    if (tid == 0) { meta->n_good = n_good; meta->n_rows = n_rows; meta->updated = upd ? 1 : 0; }
    if (!upd) {                                        // pass-through (Updater.cc:621-627): W = 0 => U = G = 0 => P+ = P exactly
        for (int e = tid; e < c6 * c6; e += SOLVE6_T) Wout[(size_t)(e / c6) * ldh + (e % c6)] = 0.0;
        for (int i = tid; i < xd; i += SOLVE6_T) x_out[i] = x[i];
        return;
    }
    const int nrw = (c6 > wv) ? (c6 - wv + SOLVE6_NW - 1) / SOLVE6_NW : 0;   // rows of this wave: i = wv + 4 q, q < nrw <= RPW
    double* Mw = M + wv * LDM;                          // this wave's row q lives at Mw + q*RS
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int i = wv + SOLVE6_NW * q;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int j = lane + 64 * u;
            double v = 0.0;
            if (q < nrw && j <= c6) v = (j < c6) ? Tg[(size_t)i * ldh + j] : Ab[(size_t)i * ldh + c6];
            Mw[q * RS + j] = v;
        }
    }
    unsigned long long usedmask = 0;                    // bit q: row wv + 4 q was a pivot row already (uniform per wave)
    __syncthreads();
    {   // first pivot candidate of this wave: lane <-> row, column 0
        unsigned long long key = 0;
        double cv = 1.0;
        if (lane < nrw) { cv = Mw[lane * RS]; key = ((unsigned long long)__double_as_longlong(fabs(cv)) & ~127ull) | (unsigned long long)(127 - (wv + SOLVE6_NW * lane)); }
        unsigned long long m = key;
        m = umax64(m, dpp_u64<0x128>(m)); m = umax64(m, dpp_u64<0x124>(m)); m = umax64(m, dpp_u64<0x122>(m)); m = umax64(m, dpp_u64<0x121>(m));
        const unsigned long long m0 = readlane_u64<0>(m);
        const unsigned long long m1 = readlane_u64<16>(m);
        const unsigned long long best = (RPW > 16) ? umax64(m0, m1) : m0;
        if (best == 0) { if (lane == 0) { s_key[0][wv] = 0; s_rcp[0][wv] = 1.0; } }
        else if (key == best) { s_key[0][wv] = best; s_rcp[0][wv] = 1.0 / cv; }
    }
    __syncthreads();
    int ppr = -1;
    for (int k = 0; k < c6; ++k) {
        const int par = k & 1;
        // previous pivot row: its column entry becomes 1 (stored form of 1/piv) only now, after the barrier
        if (ppr >= 0 && (ppr & 3) == wv && lane == 0) M[ppr * LDM + (k - 1)] = 1.0;
        // row loads do not depend on the new pivot: issue them first
        double fb[RPW], mv[RPW][NCH];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            fb[q] = Mw[q * RS + k];
#pragma unroll
            for (int u = 0; u < NCH; ++u) mv[q][u] = Mw[q * RS + lane + 64 * u];
        }
        const double ck = (lane < RPW) ? Mw[lane * RS + k] : 0.0;      // lane <-> row: old column k (for the fix-up pass)
        // combine the four per-wave candidates: unsigned max of the keys (ties -> smaller row)
        unsigned long long kb = s_key[par][0]; int wb = 0;
#pragma unroll
        for (int w = 1; w < SOLVE6_NW; ++w) { const unsigned long long kw = s_key[par][w]; const bool b = kw > kb; kb = b ? kw : kb; wb = b ? w : wb; }
        const int pr = (kb >> 7) ? 127 - (int)(kb & 127ull) : 0;        // all-zero column: flagged below, keep addresses sane
        const double ipiv = s_rcp[par][wb];
        ppr = pr;
        double prv[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) prv[u] = M[pr * LDM + lane + 64 * u];
        if (tid == 0) { s_prow[k] = pr; s_invp[pr] = k; s_ipiv[k] = ipiv; if ((kb >> 7) == 0) meta->err |= 1; }
        const int qpr = ((pr & 3) == wv) ? (pr >> 2) : -1;
        if (qpr >= 0) usedmask |= 1ull << qpr;
        __builtin_amdgcn_sched_barrier(0);
        // ---- elimination: M[i][:] -= (M[i][k]/piv) * M[p][:]   (pivot row: multiplier 0)
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const double f = (q == qpr) ? 0.0 : fb[q] * ipiv;
#pragma unroll
            for (int u = 0; u < NCH; ++u) Mw[q * RS + lane + 64 * u] = mv[q][u] - f * prv[u];
        }
        // ---- column k of every non-pivot row: -M[i][k]/piv   (lane <-> row)
        if (lane < RPW && lane != qpr) Mw[lane * RS + k] = -(ck * ipiv);
        // ---- next pivot: arg-max over this wave's unused rows of |M[i][k+1]|, published with its reciprocal
        if (k + 1 < c6) {
            unsigned long long key = 0;
            double cv = 1.0;
            if (lane < nrw && !((usedmask >> lane) & 1ull)) {
                cv = Mw[lane * RS + k + 1];
                key = ((unsigned long long)__double_as_longlong(fabs(cv)) & ~127ull) | (unsigned long long)(127 - (wv + SOLVE6_NW * lane));
            }
            unsigned long long m = key;
            m = umax64(m, dpp_u64<0x128>(m)); m = umax64(m, dpp_u64<0x124>(m)); m = umax64(m, dpp_u64<0x122>(m)); m = umax64(m, dpp_u64<0x121>(m));
            const unsigned long long m0 = readlane_u64<0>(m);
            const unsigned long long m1 = readlane_u64<16>(m);
            const unsigned long long best = (RPW > 16) ? umax64(m0, m1) : m0;
            if (best == 0) { if (lane == 0) { s_key[par ^ 1][wv] = 0; s_rcp[par ^ 1][wv] = 1.0; } }
            else if (key == best) { s_key[par ^ 1][wv] = best; s_rcp[par ^ 1][wv] = 1.0 / cv; }
        }
        __syncthreads();
    }
    if ((ppr & 3) == wv && lane == 0) M[ppr * LDM + (c6 - 1)] = 1.0;
    __syncthreads();
    // read the result out: W[k][p_j] = M[p_k][j] * ipiv_k ;  y[k] = M[p_k][c6] * ipiv_k
    for (int k = wv; k < c6; k += SOLVE6_NW) {
        const int ro = s_prow[k] * LDM; const double ip = s_ipiv[k];
        for (int c = lane; c < c6; c += 64) Wout[(size_t)k * ldh + c] = M[ro + s_invp[c]] * ip;
        if (lane == 0) s_y[k] = M[ro + c6] * ip;
    }
    __syncthreads();
    // dx = K r = Pc y   (Updater.cc:544)
    for (int i = tid; i < d; i += SOLVE6_T) {
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
    for (int p = tid - 64; p >= 0 && p < n; p += SOLVE6_T - 64) {
        stq(x_out + 26 + 7 * p, qmul(small_q(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), ldq(x + 26 + 7 * p)));
        for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
    }
}
