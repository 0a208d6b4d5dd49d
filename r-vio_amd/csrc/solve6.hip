// solve6.hip — W = T^-1, y = W b, dx = Pc y, state injection  (Updater.cc:540-613), generation 6.
//
// In-place Gauss-Jordan inversion of T = s2 I + A Pcc with partial pivoting on the LDS tableau M = [T | b], one
// workgroup of NW waves, ONE barrier per column.  A wave that is alone on its SIMD pays ~5-10 cycles per instruction,
// so the step is written to need ~5 instructions per row and NW = 8 puts two waves on every SIMD:
//   * row stride = 64*NCH + 1 doubles: every lane of every row has its own slot, so all stores are unconditional
//     (columns >= c6+1 and the padding rows q >= nrw just hold garbage that is never read back as data);
//   * the pivot row is neutralised by a zero multiplier instead of being skipped;
//   * the multipliers f_i = M[i][k]/piv, the new column k (-f_i) and the search for the next pivot are lane-parallel
//     passes (lane <-> row of this wave); a row's multiplier reaches the row update through v_readlane (SGPR operand);
//   * pivot candidates are 32-bit keys: the high word of |value| (sign, exponent, 20 mantissa bits) with its 7 low bits
//     replaced by 127 - row.  Keys compare as unsigned integers (one v_max_u32 per DPP step, ties -> smaller row), i.e.
//     partial pivoting that resolves magnitudes to 2^-13 relative: the chosen pivot is within 0.013 % of the column
//     maximum, which leaves the growth bound of partial pivoting unchanged for all practical purposes and is
//     deterministic (replicas of the multi-GPU path stay bit-identical);
//   * the candidate's reciprocal is published with its key, so the serial chain of a step contains no division.
// No row swaps (step k uses the not-yet-used row p_k with the largest |M[i][k]|), deferred pivot scaling — every other row i: f = M[i][k] / piv,
// M[i][j] -= f M[p][j] (j != k), M[i][k] = -f, then M[p][k] := 1 —:  T^-1[k][p_j] = M[p_k][j] / piv_k,  y[k] = M[p_k][c6] / piv_k.
// c6 <= 64*NCH - 1 and ceil(c6/NW) <= RPW <= 16; larger windows use solve7_kernel (every window does on a plain handle).
#pragma once
#include "rvio_dev.h"

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned umax32(unsigned a, unsigned b) { return a > b ? a : b; }

template <int NCH, int RPW, int NW>
__global__ __launch_bounds__(64 * NW) void solve6_kernel(DevCfg cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Tg,
                                                         const double* __restrict__ Ab, const double* __restrict__ x, const double* __restrict__ P,
                                                         double* __restrict__ Wout, double* __restrict__ x_out, size_t bs) {
    meta = zoff(meta, bs); Tg = zoff(Tg, bs); Ab = zoff(Ab, bs); x = zoff(x, bs); P = zoff(P, bs); Wout = zoff(Wout, bs); x_out = zoff(x_out, bs);
    static_assert(NW == 4 || NW == 8 || NW == 16, "wave count");
    static_assert(RPW <= 16, "the per-wave pivot search reduces one DPP row");
    extern __shared__ __align__(16) double M[];
    __shared__ int s_prow[6 * RVIO_MAX_LEN], s_invp[6 * RVIO_MAX_LEN];
    __shared__ double s_ipiv[6 * RVIO_MAX_LEN];
    __shared__ double s_y[6 * RVIO_MAX_LEN];
    __shared__ double s_dx[24 + 6 * RVIO_MAX_LEN];
    __shared__ unsigned s_key[2][NW];
    __shared__ double s_rcp[2][NW];
    constexpr int NT = 64 * NW;
    constexpr int LDM = 64 * NCH + 1;                   // odd: conflict-free column walks; every lane owns a slot
    constexpr int RS = NW * LDM;                        // distance between consecutive rows of one wave
    constexpr int WSH = (NW == 16) ? 4 : (NW == 8) ? 3 : 2;
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax, xd = 26 + 7 * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane < RPW ? lane : 0;             // lane <-> row passes: surplus lanes alias row 0 (reads only)
    const int n_good = (int)Ab[(size_t)ldh * (ldh - 1)], n_rows = (int)Ab[(size_t)ldh * (ldh - 1) + 1];
    const bool upd = n_good > 2;                       // Updater.cc:460
    if (tid == 0) { meta->n_good = n_good; meta->n_rows = n_rows; meta->updated = upd ? 1 : 0; meta->trunc_at = (int)Ab[(size_t)ldh * (ldh - 1) + 2]; }
    if (!upd) {                                        // pass-through (Updater.cc:621-627): W = 0 => U = G = 0 => P+ = P exactly
        for (int e = tid; e < c6 * c6; e += NT) Wout[(size_t)(e / c6) * ldh + (e % c6)] = 0.0;
        for (int i = tid; i < xd; i += NT) x_out[i] = x[i];
        return;
    }
    const int nrw = (c6 > wv) ? (c6 - wv + NW - 1) / NW : 0;   // rows of this wave: i = wv + NW q, q < nrw <= RPW
    double* Mw = M + wv * LDM;                          // this wave's row q lives at Mw + q*RS
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int i = wv + NW * q;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int j = lane + 64 * u;
            double v = 0.0;
            if (q < nrw && j <= c6) v = (j < c6) ? Tg[(size_t)i * ldh + j] : Ab[(size_t)i * ldh + c6];
            Mw[q * RS + j] = v;
        }
    }
    const unsigned rowtag = 127u - (unsigned)(wv + NW * lane);          // lane <-> row: low 7 key bits
    unsigned usedmask = 0;                              // bit q: row wv + NW q was a pivot row already (uniform per wave)
    __syncthreads();
    int ppr = -1;
    for (int k = -1; k < c6; ++k) {
        // step k = -1 only runs the pivot search (column 0); steps 0..c6-1 eliminate column k and search column k+1
        const int par = k & 1;
        int qpr = -1;
        if (k >= 0) {
            // previous pivot row: its column entry becomes 1 (stored form of 1/piv) only now, after the barrier
            if (ppr >= 0 && (ppr & (NW - 1)) == wv && lane == 0) M[ppr * LDM + (k - 1)] = 1.0;
            if (k == 30) DBG_T(50);
            if (k == 31) DBG_T(55);
            // the candidate keys head the serial chain of the step: read them first
            const unsigned kq = s_key[par][lane & (NW - 1)];
            const double rq = s_rcp[par][lane & (NW - 1)];
            const double ck = Mw[lrow * RS + k];                        // lane <-> row: old column k
            double mv[RPW][NCH];
#pragma unroll
            for (int q = 0; q < RPW; ++q)
#pragma unroll
                for (int u = 0; u < NCH; ++u) mv[q][u] = Mw[q * RS + lane + 64 * u];
            __builtin_amdgcn_sched_barrier(0);
            // combine the per-wave candidates: lane w (mod NW) holds (key, 1/value) of wave w; unsigned max
            unsigned kb = umax32(kq, dpp_u32<0xB1>(kq));                // quad_perm [1,0,3,2]
            kb = umax32(kb, dpp_u32<0x4E>(kb));                         // quad_perm [2,3,0,1]
            if (NW >= 8) kb = umax32(kb, dpp_u32<0x124>(kb));           // row_ror:4 (the pattern has period NW)
            if (NW == 16) kb = umax32(kb, dpp_u32<0x128>(kb));          // row_ror:8
            kb = (unsigned)__builtin_amdgcn_readfirstlane((int)kb);
            const int pr = (kb >> 7) ? 127 - (int)(kb & 127u) : 0;      // all-zero column: flagged below, keep addresses sane
            const double ipiv = readlane_f64(rq, pr & (NW - 1));        // row pr belongs to wave pr mod NW
            ppr = pr;
            if (k == 30) DBG_T(51);
            double prv[NCH];
#pragma unroll
            for (int u = 0; u < NCH; ++u) prv[u] = M[pr * LDM + lane + 64 * u];
            if (tid == 0) { s_prow[k] = pr; s_invp[pr] = k; s_ipiv[k] = ipiv; if ((kb >> 7) == 0) atomicOr(&meta->err, 1); }
            qpr = ((pr & (NW - 1)) == wv) ? (pr >> WSH) : -1;
            if (qpr >= 0) usedmask |= 1u << qpr;
            // multipliers, lane <-> row: f_i = M[i][k]/piv, 0 for the pivot row
            const double fcol = (lane == qpr) ? 0.0 : ck * ipiv;
            if (k == 30) DBG_T(52);
            __builtin_amdgcn_sched_barrier(0);
            // ---- elimination: M[i][:] -= f_i * M[p][:]
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const double f = readlane_f64(fcol, q);
#pragma unroll
                for (int u = 0; u < NCH; ++u) Mw[q * RS + lane + 64 * u] = mv[q][u] - f * prv[u];
            }
            // ---- column k of every non-pivot row: -M[i][k]/piv   (lane <-> row)
            if (lane < RPW && lane != qpr) Mw[lane * RS + k] = -fcol;
            if (k == 30) DBG_T(53);
        }
        // ---- next pivot: arg-max over this wave's unused rows of |M[i][k+1]|, published with its reciprocal
        if (k + 1 < c6) {
            unsigned key = 0;
            double cv = 1.0;
            if (lane < nrw && !((usedmask >> lane) & 1u)) {
                cv = Mw[lane * RS + k + 1];
                key = ((unsigned)__double2hiint(fabs(cv)) & ~127u) | rowtag;
            }
            const double rc = 1.0 / cv;                                 // every lane: overlaps the reduction below
            unsigned m = key;
            m = umax32(m, dpp_u32<0x128>(m));                           // row_ror:8,4,2,1: every lane of the row ends with the max
            m = umax32(m, dpp_u32<0x124>(m)); m = umax32(m, dpp_u32<0x122>(m)); m = umax32(m, dpp_u32<0x121>(m));
            const unsigned best = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
            if (best == 0) { if (lane == 0) { s_key[par ^ 1][wv] = 0; s_rcp[par ^ 1][wv] = 1.0; } }
            else if (key == best) { s_key[par ^ 1][wv] = best; s_rcp[par ^ 1][wv] = rc; }
        }
        if (k == 30) DBG_T(54);
        __syncthreads();
    }
    if ((ppr & (NW - 1)) == wv && lane == 0) M[ppr * LDM + (c6 - 1)] = 1.0;
    __syncthreads();
    // read the result out: W[k][p_j] = M[p_k][j] * ipiv_k ;  y[k] = M[p_k][c6] * ipiv_k
    for (int k = wv; k < c6; k += NW) {
        const int ro = s_prow[k] * LDM; const double ip = s_ipiv[k];
        for (int c = lane; c < c6; c += 64) Wout[(size_t)k * ldh + c] = M[ro + s_invp[c]] * ip;
        if (lane == 0) s_y[k] = M[ro + c6] * ip;
    }
    __syncthreads();
    // dx = K r = Pc y   (Updater.cc:544)
    for (int i = tid; i < d; i += NT) {
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
    for (int p = tid - 64; p >= 0 && p < n; p += NT - 64) {
        stq(x_out + 26 + 7 * p, qmul(small_q(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), ldq(x + 26 + 7 * p)));
        for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
    }
}
