// solve9.hip — W = (s2 I + A Pcc)^-1, y = W b, dx = Pc y, state injection (Updater.cc:540-613), generation 9 (round 5):
// the solve as blocked factorisations of SYMMETRIC POSITIVE DEFINITE matrices on the FP64 matrix cores.
//
// solve7 eliminates the non-symmetric T = s2 I + A Pcc column by column (Gauss-Jordan, partial pivoting): ~980 cycles per column on
// ONE CU, every column a pivot search + a cross-wave hand-over + a rank-1 update on the vector ALU (35.7 us at 6n = 60, 796 us at 180).
// Here the same W comes from pieces that never need a pivot search (tools/solve9_model.py is the NumPy model of this file):
//
//     Pcc = L L^T                     blocked Cholesky (positive-SEMI-definite safe: a clone with zero covariance — an IMU stream
//                                     that ended — gives a zero column of L, exactly as the reference's pivoted LU tolerates it)
//     M   = s2 I + L^T A L            symmetric, every eigenvalue >= s2: elimination in natural order is unconditionally stable
//     Mi  = M^-1                      blocked symmetric sweep in its Cholesky form (per step: Z = F old, S -= Z^T Z, new = F^T Z)
//     W   = (I - (A L) Mi L^T) / s2   Woodbury; no inverse of L or Pcc is ever formed, cond(Pcc) does not enter
//
// cond(M) = cond(T) <= 5e3 on every sequence of the test suite (median 1.5): against numpy.linalg.inv(T) the model sits at
// 1e-15 .. 8e-14 (5e-12 in U = Pc W at rest), and at LAPACK's own residual on a synthetic cond(T) = 7e8 case.
//
// Layout.  Everything is 16 x 16 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64: lane (li = lane & 15, lk = lane >> 4),
// register r <-> element (4 r + lk, li).  A tile in that layout IS the B operand of four consecutive MFMAs, and it is the A operand
// of its TRANSPOSE — so every product is written as X^T Y of stored tiles (the matrices involved are symmetric, or kept in both
// orientations: G = L^T beside L, Q^T beside Q) and operands never go through a layout conversion.  NW = WGR^2 waves; wave (a, b)
// owns the BS x BS tiles of block row a, block column b (NT = WGR BS tiles per side, 16 NT >= 6n).
//   * one in-wave primitive on the 16 x 16 diagonal tile: forward elimination of [Mkk | I] without pivoting (16 steps; the pivot
//     row travels through the wave's LDS scratch) -> F = Lkk^-1 and F^T.  Every wave runs it redundantly on the published tile:
//     no second barrier, no hand-over of the result.
//   * Cholesky step k: row panel G(k, j) = F S(k, j), L(j, k) = S(k, j)^T F^T, trailing S(i, j) -= G(k, i)^T G(k, j)  (i <= j).
//   * sweep step k:    Z_j = F S(k, j);  S(i, j) -= Z_i^T Z_j;  S(k, j) = F^T Z_j;  S(i, k) = Z_i^T F;  S(k, k) = -F^T F.
//   * the GEMM phases (Q = A L, M, X = Mi G, W) read tiles from a scratch slab in L2 (written by the phase before, one barrier).
// One workgroup per instance, one launch; the row panel of a step crosses waves through LDS (double-buffered: one barrier per step).
#pragma once
#include "rvio_dev.h"

typedef double s9_d4 __attribute__((ext_vector_type(4)));
#define S9_TILE 256                                  // doubles of a stored tile: [r][lane]

__device__ __forceinline__ bool lane_is(int li, int lk, int a, int b) { return li == a && lk == b; }
struct S9Wave {                                      // per-wave LDS scratch of the in-wave primitive
    double rowb[2][128];                              // pivot row of [M | E], by step parity
    double tb[16 * 17 + 64 + 192];                   // transposition of a tile (272), then 3 x 64 slots for stores that are not meant to be read
};
typedef __attribute__((address_space(3))) double s9_lds_t;      // LDS-qualified: ds_read / ds_write instead of flat accesses
typedef volatile s9_lds_t* s9_vlp;

// acc += X^T Y for tiles in the accumulator layout
__device__ __forceinline__ s9_d4 s9_tn(const s9_d4& x, const s9_d4& y, s9_d4 acc) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], y[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], y[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], y[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], y[3], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ s9_d4 s9_zero() { s9_d4 z = {0.0, 0.0, 0.0, 0.0}; return z; }
// Scratch tiles are written and read by waves of ONE workgroup, i.e. of one CU: __syncthreads() (a workgroup-scope release / acquire
// + s_barrier) between the phase that writes and the phase that reads is all the ordering they need, and the loads are served by the
// CU's own L1 / the XCD's L2.  (Agent-scope loads — what a slab shared between CUs would need — leave the XCD: measured 4x slower.)
__device__ __forceinline__ s9_d4 s9_ldg(const double* t, int lane) {
    s9_d4 v = {t[lane], t[64 + lane], t[128 + lane], t[192 + lane]};
    return v;
}
__device__ __forceinline__ void s9_stg(double* t, int lane, const s9_d4& v) {
    t[lane] = v[0]; t[64 + lane] = v[1]; t[128 + lane] = v[2]; t[192 + lane] = v[3];
}
// a tile of the row panel in LDS (written before a workgroup barrier, read after it)
__device__ __forceinline__ s9_d4 s9_lds(const double* t_, int lane) {
    const s9_lds_t* t = (const s9_lds_t*)t_;
    s9_d4 v = {t[lane], t[64 + lane], t[128 + lane], t[192 + lane]};
    return v;
}
__device__ __forceinline__ void s9_sts(double* t_, int lane, const s9_d4& v) {
    s9_lds_t* t = (s9_lds_t*)t_;
    t[lane] = v[0]; t[64 + lane] = v[1]; t[128 + lane] = v[2]; t[192 + lane] = v[3];
}
// The wave-private LDS scratch is only ever touched through VOLATILE LDS pointers: the compiler keeps those accesses in program
// order, and the LDS operations of one wave complete in order — a store followed by loads of other lanes' slots needs nothing else.
// the transpose of a tile, through the wave's scratch
__device__ __forceinline__ s9_d4 s9_transpose_tb(const s9_d4& v, double* tb_, int li, int lk) {
    s9_vlp tb = (s9_vlp)tb_;
#pragma unroll
    for (int r = 0; r < 4; ++r) tb[(4 * r + lk) * 17 + li] = v[r];
    s9_d4 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = tb[li * 17 + 4 * r + lk];
    return t;
}

// The in-wave primitive (run by ONE wave per step; the others wait for its result at a barrier — run redundantly by all 16 waves it is
// bound by instruction issue, four waves to a SIMD).  m: a symmetric positive (semi-)definite 16 x 16 tile.  Blocked forward
// elimination of [m | I] in natural order, FOUR rows per step (rows 4 b .. 4 b + 3 are register b of every lane):
//     a    = the 4 x 4 diagonal block of the current tile,   F4 = chol(a)^-1 (lower triangular, formed in registers by every lane)
//     Z    = F4 m[blk, :]   (= the block row of the tile's Cholesky factor),   Ze = F4 E[blk, :]   (= rows blk of F, final)
//     m   -= Z^T Z,  E -= Z^T Ze  on the rows below the block:  ONE v_mfma_f64_16x16x4 each — the A operand of lane (li, lk) is
//            -Z(lk, li) (zero for the rows up to the block), the B operand is Z(lk, li) resp. Ze(lk, li): what the lane holds anyway.
// E ends as F = (chol m)^-1.  A pivot <= tol * ref[p] (ref: the ORIGINAL diagonal of the matrix the tile belongs to; nullptr: plain
// positivity) is a zero direction: no elimination with it, row p of F is zero (the convention of a semi-definite Cholesky).
// returns F and F^T; bad |= 1 when a pivot is not positive (or NaN) where it has to be.
// Serial chain per block step: one LDS round trip (block rows of m and E out, 10 + 8 doubles back), four inverse square roots
// (v_rsq_f64 + two Newton steps) with a handful of FMAs between them, two MFMAs.  Four block steps per tile.
__device__ __forceinline__ double s9_rsqrt(double t) {
    double y = __builtin_amdgcn_rsq(t);
    const double h = 0.5 * t;
    y = fma(y, fma(-(h * y), y, 0.5), y);
    y = fma(y, fma(-(h * y), y, 0.5), y);
    return y;
}
__device__ __forceinline__ void s9_factor(s9_d4 m, const double* ref_, double tol, S9Wave* ws, int li, int lk, s9_d4& F, s9_d4& Ft, int& bad) {
    const s9_lds_t* ref = (const s9_lds_t*)ref_;
    const int lane = li + 16 * lk;
    s9_d4 e;
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = (4 * r + lk == li) ? 1.0 : 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        s9_vlp rb = (s9_vlp)ws->rowb[b & 1];
        rb[lane] = m[b];                                // R(lk, li)  = m(4 b + lk, li)
        rb[64 + lane] = e[b];                           // Re(lk, li) = E(4 b + lk, li)
        const double a00 = rb[0 * 16 + 4 * b + 0];
        const double a10 = rb[1 * 16 + 4 * b + 0], a11 = rb[1 * 16 + 4 * b + 1];
        const double a20 = rb[2 * 16 + 4 * b + 0], a21 = rb[2 * 16 + 4 * b + 1], a22 = rb[2 * 16 + 4 * b + 2];
        const double a30 = rb[3 * 16 + 4 * b + 0], a31 = rb[3 * 16 + 4 * b + 1], a32 = rb[3 * 16 + 4 * b + 2], a33 = rb[3 * 16 + 4 * b + 3];
        double Rv[4], Ev[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) { Rv[v] = rb[v * 16 + li]; Ev[v] = rb[64 + v * 16 + li]; }
        double lim[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) lim[k] = ref_ ? tol * ref[4 * b + k] : 0.0;
        auto pivot = [&](double t, int k) {             // 1 / sqrt(t), or 0 for a zero direction
            const bool ok = t > lim[k] && t > 0.0;      // false for NaN too
            if (!ok && (ref_ == nullptr || !(t >= -lim[k]))) bad |= 1;
            return ok ? s9_rsqrt(t) : 0.0;
        };
        // chol(a) through its inverse diagonal i_k = 1 / l_kk
        const double i0 = pivot(a00, 0);
        const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
        const double i1 = pivot(fma(-l10, l10, a11), 1);
        const double l21 = fma(-l20, l10, a21) * i1, l31 = fma(-l30, l10, a31) * i1;
        const double i2 = pivot(fma(-l21, l21, fma(-l20, l20, a22)), 2);
        const double l32 = fma(-l31, l21, fma(-l30, l20, a32)) * i2;
        const double i3 = pivot(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a33))), 3);
        // F4 = chol(a)^-1
        const double f00 = i0, f11 = i1, f22 = i2, f33 = i3;
        const double f10 = -(l10 * f00) * i1;
        const double f21 = -(l21 * f11) * i2;
        const double f32 = -(l32 * f22) * i3;
        const double f20 = -fma(l21, f10, l20 * f00) * i2;
        const double f31 = -fma(l32, f21, l31 * f11) * i3;
        const double f30 = -fma(l32, f20, fma(l31, f10, l30 * f00)) * i3;
        // row lk of F4
        const double g0 = lk == 0 ? f00 : lk == 1 ? f10 : lk == 2 ? f20 : f30;
        const double g1 = lk == 0 ? 0.0 : lk == 1 ? f11 : lk == 2 ? f21 : f31;
        const double g2 = lk <= 1 ? 0.0 : lk == 2 ? f22 : f32;
        const double g3 = lk == 3 ? f33 : 0.0;
        const double z = fma(g3, Rv[3], fma(g2, Rv[2], fma(g1, Rv[1], g0 * Rv[0])));
        const double ze = fma(g3, Ev[3], fma(g2, Ev[2], fma(g1, Ev[1], g0 * Ev[0])));
        const double za = (li > 4 * b + 3) ? -z : 0.0;
        m = __builtin_amdgcn_mfma_f64_16x16x4f64(za, z, m, 0, 0, 0);
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(za, ze, e, 0, 0, 0);
        e[b] = ze;
    }
    F = e;
    Ft = s9_transpose_tb(F, ws->tb, li, lk);
}

// sum_{k0 <= k < k1} X_k^T Y_k with the operands of step k + 1 in flight while step k multiplies
template <class FX, class FY>
__device__ __forceinline__ s9_d4 s9_gemm(int k0, int k1, FX fx, FY fy) {
    s9_d4 acc = s9_zero();
    if (k0 >= k1) return acc;
    s9_d4 xa = fx(k0), xb = fy(k0);
#pragma unroll 2
    for (int k = k0; k < k1; ++k) {
        s9_d4 na = xa, nb = xb;
        if (k + 1 < k1) { na = fx(k + 1); nb = fy(k + 1); }
        acc = s9_tn(xa, xb, acc);
        xa = na; xb = nb;
    }
    return acc;
}

// A wave's BS x BS block of a product: per k the BS y-tiles are loaded once and each x-tile once (2 BS tile loads for BS^2 products —
// a single CU takes 64 B per clock from its L1: at 12 tiles per side the tile-at-a-time loop above is bound by exactly that).
// The phases of the one-workgroup kernel are MATRIX-PIPE bound: gfx950 issues one v_mfma_f64_16x16x4 per SIMD every 64 cycles (78.6 TFLOP/s over 256 x 4 SIMDs
// = 32 flop per clock each), i.e. 256 cycles per tile product — a k step of 4 products per wave, 2-3 waves to a SIMD, is 2-3 k cycles of pipe time, which is what the
// stamps show (2.7-4.7 k per step at 6n = 84).  Measured and NOT adopted there (round 5): the operands of step k + 1 in flight while step k multiplies (64.4
// against 61.5 us), the loads of two / three steps issued together (62.2 us / spills) — the loads were never the bound
template <int BS, class FX, class FY, class FOK>
__device__ __forceinline__ void s9_gemm_block(int k0, int k1, s9_d4 (&acc)[BS * BS], FX fx, FY fy, FOK ok) {
#pragma unroll 1
    for (int k = k0; k < k1; ++k) {
        s9_d4 y[BS];
#pragma unroll
        for (int qj = 0; qj < BS; ++qj) y[qj] = fy(k, qj);
#pragma unroll
        for (int qi = 0; qi < BS; ++qi) {
            const s9_d4 xq = fx(k, qi);
#pragma unroll
            for (int qj = 0; qj < BS; ++qj)
                if (ok(k, qi, qj)) acc[qi * BS + qj] = s9_tn(xq, y[qj], acc[qi * BS + qj]);
        }
    }
}

// LDS of the Cholesky phase (P0 + P1): static in solve9_kernel, carved out of the launch's dynamic LDS by the role workgroup of feat_prop_kernel
template <int NT, int NW>
struct S9CholLds {
    double rowp[2][NT][S9_TILE];                       // [0]: the row panel of a step, [1]: its Z tiles (F times the panel)
    S9Wave ws;                                         // the in-wave primitive runs in wave 0
    double F[2][S9_TILE];                              // its result: F and F^T of the step's diagonal tile
    double tb[NW][16 * 17];                            // per-wave transposition scratch (L from G; Q^T in the solve)
    double ref[16 * NT];                               // the original diagonal of Pcc (the scale of "zero" for its Cholesky)
    int bad;
};

// P0 + P1: the clone block of P -> L (bL: tiles (i, k), i >= k) and G = L^T (bG: tiles (k, j), j >= k) in the tile slab.
// Depends on P only — NOT on the measurements: the pipelined path runs it as one more workgroup of the per-feature launch
// (feat_prop_kernel: propagation leaves the clone block alone, so it is the Pcc the solve will see), off the filter chain.
// Every thread of a workgroup of 64 WGR^2 threads calls it; returns (in every thread) whether a pivot was not positive where it had to be.
template <int BS, int WGR>
__device__ __forceinline__ int s9_cholesky(const DevCfg& cfg, int n, const double* __restrict__ P, double* __restrict__ scr, S9CholLds<WGR * BS, WGR * WGR>& sh) {
    constexpr int NW = WGR * WGR, NT = WGR * BS, NTH = 64 * NW, TS = BS * BS, NP = 16 * NT;
    const int c6 = 6 * n, ld = cfg.dmax;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
    const int wa = wv / WGR, wb = wv % WGR;
    double* bL = scr;
    double* bG = scr + (size_t)NT * NT * S9_TILE;
    auto tile = [&](double* b, int i, int j) { return b + (size_t)(i * NT + j) * S9_TILE; };
    // ---- P0: the clone block of P into the tableau (symmetric: element (row, col) is read as P[col + row ld], coalesced along li); identity beyond 6n
    s9_d4 S[TS];
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const int i = wa * BS + s / BS, j = wb * BS + s % BS;
        if (i > j) { S[s] = s9_zero(); continue; }     // (never read: one CU pulls P at ~10 B per clock — at 6n = 180 the 66 dead tiles were 12 us of the load)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + 4 * r + lk, col = 16 * j + li;
            S[s][r] = (row < c6 && col < c6) ? P[(size_t)(24 + col) + (size_t)(24 + row) * ld] : (row == col ? 1.0 : 0.0);
        }
    }
    for (int i = tid; i < NP; i += NTH) sh.ref[i] = (i < c6) ? P[(size_t)(24 + i) * (ld + 1)] : 1.0;
    if (tid == 0) sh.bad = 0;
    int bad = 0;
    __syncthreads();
    DBG_T(31);

    // ---- P1: Pcc = L L^T.  Only tiles i <= j are live (the row panel of step k is S(k, j >= k)).
    // Per step three barriers: panel -> LDS | one wave factors the diagonal tile | every Z_j = F S(k, j) is formed ONCE (wave j - k mod NW) and
    // published | the updates read their two operands from LDS.  (Each wave forming the Z tiles of its own block row and column itself — 2 BS
    // products beside BS^2 updates, 2 BS tiles more in registers — spilled at BS = 3 and put 1 MB of scratch traffic into every step.)
    double (*rowp)[S9_TILE] = sh.rowp[0];
    double (*zb)[S9_TILE] = sh.rowp[1];
#pragma unroll 1
    for (int k = 0; k < NT; ++k) {
        if (wa == k / BS) {
#pragma unroll
            for (int s = 0; s < TS; ++s) {
                const int i = wa * BS + s / BS, j = wb * BS + s % BS;
                if (i == k && j >= k) s9_sts(rowp[j], lane, S[s]);
            }
        }
        __syncthreads();
        if (k == 0) DBG_T(40);
        if (wv == 0) {
            s9_d4 F, Ft;
            s9_factor(s9_lds(rowp[k], lane), sh.ref + 16 * k, 1e-12, &sh.ws, li, lk, F, Ft, bad);
            s9_sts(sh.F[1], lane, Ft);
        }
        __syncthreads();
        if (k == 0) DBG_T(41);
        {   // G(k, j) = F S(k, j), j >= k
            const s9_d4 Ft = s9_lds(sh.F[1], lane);
            for (int j = k + wv; j < NT; j += NW) s9_sts(zb[j], lane, s9_tn(Ft, s9_lds(rowp[j], lane), s9_zero()));
        }
        __syncthreads();
        s9_d4 Zc[BS];
#pragma unroll
        for (int q = 0; q < BS; ++q) {
            const int jc = wb * BS + q;
            Zc[q] = (jc >= k) ? s9_lds(zb[jc], lane) : s9_zero();
        }
#pragma unroll
        for (int qi = 0; qi < BS; ++qi) {
            const int i = wa * BS + qi;
            s9_d4 nz = s9_zero();
            if (i > k) {
                const s9_d4 zr = s9_lds(zb[i], lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) nz[r] = -zr[r];
            }
#pragma unroll
            for (int qj = 0; qj < BS; ++qj) {
                const int s = qi * BS + qj, j = wb * BS + qj;
                if (i == k && j >= k) {
                    s9_d4 g = Zc[qj];
                    if (j == k) {                          // the diagonal tile of G = Lkk^T: exact zeros below the diagonal
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (4 * r + lk > li) g[r] = 0.0;
                    }
                    s9_stg(tile(bG, k, j), lane, g);
                    s9_stg(tile(bL, j, k), lane, s9_transpose_tb(g, sh.tb[wv], li, lk));
                } else if (i > k && j >= i) S[s] = s9_tn(nz, Zc[qj], S[s]);
            }
        }
        if (k == 0) DBG_T(42);
    }
    if (bad) atomicOr(&sh.bad, 1);
    __syncthreads();
    return sh.bad;
}

// The Cholesky phase as a ROLE of another launch (256 threads = 2 x 2 waves): one more workgroup of the per-feature launch (pipelined
// frame) or of propagate's launch (staged entry points) factors the clone block while that launch does its own work; the verdict goes
// to the slab's last word.  Same tiles, same order of operations per tile as solve9_kernel's own phase: the same bits.
#define S9_SLAB_DOUBLES(NT) ((size_t)5 * (NT) * (NT) * S9_TILE + 8 + (size_t)16 * (NT) * (NT))   // five tile buffers, the verdict words, the by-tile-column shares of y (split form)
#define S9_YP_OFF(NT) ((size_t)5 * (NT) * (NT) * S9_TILE + 8)
template <int BS>
__device__ __forceinline__ void s9_chol_role(const DevCfg& cfg, int n, const double* __restrict__ P, double* __restrict__ scr, S9CholLds<2 * BS, 4>& sh) {
    const int bad = s9_cholesky<BS, 2>(cfg, n, P, scr, sh);
    if (threadIdx.x == 0) scr[(size_t)5 * (2 * BS) * (2 * BS) * S9_TILE] = bad ? 1.0 : 0.0;
}

// The blocked symmetric sweep of the tableau S (wave (a, b) of WGR x WGR holds the BS x BS tiles of block row a, block column b): S ends as -M^-1.
// Per step k: the row panel S(k, :) goes to LDS, ONE wave factors the diagonal tile (F = chol(S(k, k))^-1), the Z_j = F S(k, j) are formed once
// each and published, every wave updates its tiles from them.  s_rowp: [2][NT][S9_TILE] (the panel; the Z tiles), s_F: [2][S9_TILE].
// Measured and NOT adopted (round 5): LOOK-AHEAD of the factor — block row k + 1 updated first and published, wave 0 factoring its diagonal tile beside
// the other waves' trailing update: 20.8 k cycles per step at 6n = 180 against 18 k (10.5 / 10.3 k at 120, 9.4 / 8.8 k at 84).  The factor wave shares its
// matrix pipe with three updating waves (its eight dependent MFMAs queue behind theirs) and still has its own tiles to update afterwards: the step ends
// with that tail.  A wave that owns no tiles was tried next, together with keeping only the UPPER TRIANGLE of the symmetric tableau (78 of 144 tiles at
// 6n = 180, dealt to ten waves so that the four matrix pipes carry 21 / 21 / 18 / 18 tiles; row panels completed by transposes; results identical): 19.5 k
// cycles per step against 18 k, 10.3 k against 10.3 k at 6n = 120.  Halving the tiles per matrix pipe changed nothing: the trailing update of a step is not
// bound by the pipe's throughput but by each wave's own chain (operand tiles from LDS, four dependent MFMAs per tile, two tiles of the tableau in
// scratch at BS = 3) — the step gets shorter with fewer tiles per WAVE, i.e. with more than one workgroup, not with fewer tiles per pipe.
template <int BS, int WGR>
__device__ __forceinline__ void s9_sweep(s9_d4 (&S)[BS * BS], double (*s_rowp)[WGR * BS][S9_TILE], double (*s_F)[S9_TILE], S9Wave* ws, int& bad) {
    constexpr int NT = WGR * BS, NW = WGR * WGR;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
    const int wa = wv / WGR, wb = wv % WGR;
    double (*rowp)[S9_TILE] = s_rowp[0];
    double (*zb)[S9_TILE] = s_rowp[1];
#pragma unroll 1
    for (int k = 0; k < NT; ++k) {
        if (wa == k / BS) {
#pragma unroll
            for (int s = 0; s < BS * BS; ++s) {
                const int i = wa * BS + s / BS, j = wb * BS + s % BS;
                if (i == k) s9_sts(rowp[j], lane, S[s]);
            }
        }
        __syncthreads();
        if (k == 1) DBG_T(43);
        DBG_T(46 + k);
        if (wv == 0) {
            s9_d4 F, Ft;
            s9_factor(s9_lds(rowp[k], lane), nullptr, 0.0, ws, li, lk, F, Ft, bad);
            s9_sts(s_F[0], lane, F);
            s9_sts(s_F[1], lane, Ft);
        }
        __syncthreads();
        if (k == 1) DBG_T(44);
        {   // Z_j = F S(k, j): each tile once
            const s9_d4 Ft = s9_lds(s_F[1], lane);
            for (int j = wv; j < NT; j += NW) s9_sts(zb[j], lane, s9_tn(Ft, s9_lds(rowp[j], lane), s9_zero()));
        }
        __syncthreads();
        s9_d4 Zc[BS];
#pragma unroll
        for (int q = 0; q < BS; ++q) Zc[q] = s9_lds(zb[wb * BS + q], lane);
#pragma unroll
        for (int qi = 0; qi < BS; ++qi) {
            const int i = wa * BS + qi;
            s9_d4 nz = s9_lds(zb[i], lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) nz[r] = -nz[r];
#pragma unroll
            for (int qj = 0; qj < BS; ++qj) {
                const int s = qi * BS + qj, j = wb * BS + qj;
                if (i == k || j == k) {                    // (wave-uniform; F is fetched where it is needed: nothing of it stays in registers)
                    const s9_d4 F = s9_lds(s_F[0], lane);
                    const s9_d4 t = (i == k && j == k) ? s9_tn(F, F, s9_zero()) : (i == k) ? s9_tn(F, Zc[qj], s9_zero()) : s9_tn(nz, F, s9_zero());
#pragma unroll
                    for (int r = 0; r < 4; ++r) S[s][r] = (i == k && j != k) ? t[r] : -t[r];   // -F^T F | F^T Z_j | Z_i^T F = -((-Z_i)^T F)
                } else S[s] = s9_tn(nz, Zc[qj], S[s]);
            }
        }
        if (k == 1) DBG_T(45);
    }
}

// BS x BS tiles per wave, WGR x WGR waves: NT = WGR * BS tiles per side.  scr: 5 * NT^2 * 256 (+ 8) doubles per instance.
// PRE: L and G are in the slab already (s9_cholesky ran in the per-feature launch of this very update); *chol_bad: what it returned.
template <int BS, int WGR, bool PRE = false>
__global__ __launch_bounds__(64 * WGR * WGR) void solve9_kernel(DevCfg cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Ab,
                                                                const double* __restrict__ x, const double* __restrict__ P, double* __restrict__ scr,
                                                                double* __restrict__ Wout, double* __restrict__ x_out, size_t bs, size_t scr_bs, int defer_dx = 0) {
    // defer_dx (round 6; one instance, the frame's update): the kernel ends with W and the by-tile-column shares of y = W b in the slab; dx = Pc y and the
    // state injection are role workgroups of the Joseph launch right behind it (s9_dx_role), as in the split form and in solve9_small_kernel
    meta = zoff(meta, bs); Ab = zoff(Ab, bs); x = zoff(x, bs); P = zoff(P, bs); Wout = zoff(Wout, bs); x_out = zoff(x_out, bs);
    scr = (double*)((char*)scr + (size_t)blockIdx.z * scr_bs);
    constexpr int NW = WGR * WGR, NT = WGR * BS, NTH = 64 * NW, TS = BS * BS, NP = 16 * NT;
    __shared__ S9CholLds<NT, NW> sh;
    // 6n <= 96 (NT = 6): Q = A L, later X = Mi G, stays in LDS (72 KB): the y operands of M = L^T Q and of W = (I - Q X) / s2 come from there (-1.7 us at 6n = 84)
    constexpr bool QLDS = (NT == 6);
    __shared__ double s_q[QLDS ? NT * NT : 1][S9_TILE];
    __shared__ double s_b[NP], s_y[NP], s_yp[NT][NP];
    __shared__ double s_dx[24 + 6 * RVIO_MAX_LEN];
    __shared__ double s_part[4 * (24 + 6 * RVIO_MAX_LEN)];
    __shared__ int s_bad;
    double (*s_rowp)[NT][S9_TILE] = sh.rowp;
    double (*s_F)[S9_TILE] = sh.F;
    double (*s_tb)[16 * 17] = sh.tb;
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax, xd = 26 + 7 * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
    const int wa = wv / WGR, wb = wv % WGR;             // this wave's block row / block column
    const int n_good = (int)Ab[(size_t)ldh * (ldh - 1)], n_rows = (int)Ab[(size_t)ldh * (ldh - 1) + 1];
    const bool upd = n_good > 2;                       // Updater.cc:460
    DBG_R(true, 2);
    if (tid == 0) { meta->n_good = n_good; meta->n_rows = n_rows; meta->updated = upd ? 1 : 0; meta->trunc_at = (int)Ab[(size_t)ldh * (ldh - 1) + 2]; s_bad = 0; }
    if (!upd) {                                        // pass-through (Updater.cc:621-627): W = 0 => U = G = 0 => P+ = P exactly
        for (int e = tid; e < c6 * c6; e += NTH) Wout[(size_t)(e / c6) * ldh + (e % c6)] = 0.0;
        if (!defer_dx) for (int i = tid; i < xd; i += NTH) x_out[i] = x[i];      // (deferred: the roles pass the state through)
        return;
    }
    DBG_T(30);
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    S9Wave* ws = &sh.ws;
    double* bL = scr;                                   // L(i, k), i >= k
    double* bG = scr + (size_t)NT * NT * S9_TILE;       // G = L^T: G(k, j), j >= k
    double* bQ = scr + (size_t)2 * NT * NT * S9_TILE;   // Q = A L;   later X = Mi G
    double* bQt = scr + (size_t)3 * NT * NT * S9_TILE;  // Q^T
    double* bMi = scr + (size_t)4 * NT * NT * S9_TILE;  // M^-1
    auto tile = [&](double* b, int i, int j) { return b + (size_t)(i * NT + j) * S9_TILE; };
    // A(i, j) as a tile, straight from the information block (read-only input: ordinary loads); zero beyond 6n
    auto ld_A = [&](int i, int j) {
        s9_d4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + 4 * r + lk, col = 16 * j + li;
            v[r] = (row < c6 && col < c6) ? Ab[(size_t)row * ldh + col] : 0.0;
        }
        return v;
    };
    for (int i = tid; i < NP; i += NTH) s_b[i] = (i < c6) ? Ab[(size_t)i * ldh + c6] : 0.0;
    int bad = 0;
    s9_d4 S[TS];
    if constexpr (PRE) {
        // the factor of Pcc came with the per-feature launch (its verdict travels in the slab's last word); an earlier KERNEL wrote the tiles: plain loads see them
        if (tid == 0 && scr[(size_t)5 * NT * NT * S9_TILE] != 0.0) s_bad = 1;
        __syncthreads();
    } else {
        if (s9_cholesky<BS, WGR>(cfg, n, P, scr, sh)) bad = 1;
    }

    DBG_T(32);
    // ---- P2: Q = A L  (Q(i, j) = sum_{k >= j} A(k, i)^T L(k, j)); Q and Q^T to the slab
    {
        s9_d4 acc[TS];
#pragma unroll
        for (int s = 0; s < TS; ++s) acc[s] = s9_zero();
        s9_gemm_block<BS>(wb * BS, NT, acc, [&](int k, int qi) { return ld_A(k, wa * BS + qi); },
                          [&](int k, int qj) { return s9_ldg(tile(bL, k, wb * BS + qj), lane); }, [&](int k, int, int qj) { return k >= wb * BS + qj; });
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            const int i = wa * BS + s / BS, j = wb * BS + s % BS;
            if constexpr (QLDS) s9_sts(s_q[i * NT + j], lane, acc[s]);
            else s9_stg(tile(bQ, i, j), lane, acc[s]);
            s9_stg(tile(bQt, j, i), lane, s9_transpose_tb(acc[s], s_tb[wv], li, lk));
        }
    }
    __syncthreads();

    DBG_T(33);
    // ---- P3: M = s2 I + L^T Q  (M(i, j) = sum_{k >= i} L(k, i)^T Q(k, j)) into the tableau
#pragma unroll
    for (int s = 0; s < TS; ++s) S[s] = s9_zero();
    s9_gemm_block<BS>(wa * BS, NT, S, [&](int k, int qi) { return s9_ldg(tile(bL, k, wa * BS + qi), lane); },
                      [&](int k, int qj) { if constexpr (QLDS) return s9_lds(s_q[k * NT + wb * BS + qj], lane); else return s9_ldg(tile(bQ, k, wb * BS + qj), lane); },
                      [&](int k, int qi, int) { return k >= wa * BS + qi; });
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const int i = wa * BS + s / BS, j = wb * BS + s % BS;
        if (i == j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * r + lk == li) S[s][r] += s2;
        }
    }
    __syncthreads();                                   // (the row-panel buffers of P1 are idle again)

    DBG_T(34);
    // ---- P4: the symmetric sweep; the tableau ends as -M^-1
    s9_sweep<BS, WGR>(S, s_rowp, s_F, ws, bad);
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const int i = wa * BS + s / BS, j = wb * BS + s % BS;
        s9_d4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = -S[s][r];
        s9_stg(tile(bMi, i, j), lane, v);
    }
    __syncthreads();

    DBG_T(35);
    // ---- P5: X = Mi G  (X(i, j) = sum_{k <= j} Mi(k, i)^T G(k, j)) over Q's buffer (Q itself is dead: W reads Q^T)
#pragma unroll
    for (int s = 0; s < TS; ++s) S[s] = s9_zero();
    s9_gemm_block<BS>(0, wb * BS + BS, S, [&](int k, int qi) { return s9_ldg(tile(bMi, k, wa * BS + qi), lane); },
                      [&](int k, int qj) { return s9_ldg(tile(bG, k, wb * BS + qj), lane); }, [&](int k, int, int qj) { return k <= wb * BS + qj; });
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        if constexpr (QLDS) s9_sts(s_q[(wa * BS + s / BS) * NT + wb * BS + s % BS], lane, S[s]);      // (Q's last reader was P3, two barriers ago)
        else s9_stg(tile(bQ, wa * BS + s / BS, wb * BS + s % BS), lane, S[s]);
    }
    __syncthreads();

    DBG_T(36);
    // ---- P6: W = (I - Q X) / s2  (W(i, j) = (delta - sum_k Qt(k, i)^T X(k, j)) / s2), stored row-major like solve7's; y = W b by tile
    const double is2 = 1.0 / s2;
#pragma unroll
    for (int s = 0; s < TS; ++s) S[s] = s9_zero();
    s9_gemm_block<BS>(0, NT, S, [&](int k, int qi) { return s9_ldg(tile(bQt, k, wa * BS + qi), lane); },
                      [&](int k, int qj) { if constexpr (QLDS) return s9_lds(s_q[k * NT + wb * BS + qj], lane); else return s9_ldg(tile(bQ, k, wb * BS + qj), lane); },
                      [&](int, int, int) { return true; });
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const int i = wa * BS + s / BS, j = wb * BS + s % BS;
        const double bj = s_b[16 * j + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + 4 * r + lk, col = 16 * j + li;
            const double w = (((row == col) ? 1.0 : 0.0) - S[s][r]) * is2;
            if (row < c6 && col < c6) Wout[(size_t)row * ldh + col] = w;
            // this tile's share of y[row]: the 16 columns summed over the 16 lanes of a DPP row (fixed order)
            double t = w * bj;
            t += __shfl_xor(t, 1, 16); t += __shfl_xor(t, 2, 16); t += __shfl_xor(t, 4, 16); t += __shfl_xor(t, 8, 16);
            if (li == 0) s_yp[j][16 * i + 4 * r + lk] = t;
        }
    }
    if (bad) atomicOr(&s_bad, 1);
    __syncthreads();
    if (tid == 0 && s_bad) atomicOr(&meta->err, 1);
    if (defer_dx) {      // the shares of y go to the slab in the layout of the split form (S9_YP_OFF: [tile column][row]); the sweep's verdict word is reported above
        double* ypg = scr + S9_YP_OFF(NT);
        for (int e = tid; e < NT * NP; e += NTH) ypg[e] = s_yp[e / NP][e % NP];
        if (tid == 0) scr[(size_t)5 * NT * NT * S9_TILE + 1] = 0.0;
        DBG_T(37); DBG_T(38);
        DBG_R(true, 7);
        return;
    }
    for (int i = tid; i < NP; i += NTH) { double acc = s_yp[0][i]; for (int j = 1; j < NT; ++j) acc += s_yp[j][i]; s_y[i] = acc; }
    __syncthreads();

    DBG_T(37);
    // ---- dx = K r = Pc y (Updater.cc:544): NTH / d threads per row, each a contiguous share of the columns; partial sums added in a fixed order
    {
        const int np = max(1, min(4, NTH / d)), share = (c6 + np - 1) / np;
        const int pt = tid / d, i = tid - pt * d;
        if (pt < np) {
            double acc = 0;
            const int k1 = min(c6, (pt + 1) * share);
#pragma unroll 8
            for (int k = pt * share; k < k1; ++k) acc += P[(size_t)i + (size_t)(24 + k) * ld] * s_y[k];
            s_part[pt * d + i] = acc;
        }
        __syncthreads();
        if (tid < d) { double acc = s_part[tid]; for (int q = 1; q < np; ++q) acc += s_part[q * d + tid]; s_dx[tid] = acc; }
    }
    __syncthreads();
    // ---- state injection (Updater.cc:546-613)
    const double* dx = s_dx;
    if (tid == 0) {
        stq(x_out, qmul(small_q(dx[0], dx[1], dx[2]), ldq(x)));
        for (int i = 0; i < 6; ++i) x_out[4 + i] = dx[3 + i] + x[4 + i];
        st3(x_out + 7, unit3(ld3(x_out + 7)));
        stq(x_out + 10, qmul(small_q(dx[9], dx[10], dx[11]), ldq(x + 10)));
        for (int i = 0; i < 12; ++i) x_out[14 + i] = dx[12 + i] + x[14 + i];
    }
    for (int p = tid - 64; p >= 0 && p < n; p += NTH - 64) {
        stq(x_out + 26 + 7 * p, qmul(small_q(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), ldq(x + 26 + 7 * p)));
        for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
    }
    DBG_T(38);
    DBG_R(true, 7);
}

// =============================================================== long windows (NT >= 8: 6n > 96): the phases as launches of their own
// In ONE workgroup the four product phases (Q = A L, M, X = Mi G, W) are 537 k of the 1.28 M cycles at 6n = 180: 144 output tiles on the four matrix
// pipes of one CU.  They are plain tile products with no dependence inside a phase: as launches of their own — one wave per output tile, four to a
// workgroup, 36 workgroups at 6n = 180 — each takes a few microseconds, and the kernel boundary is the only synchronisation the slab needs (plain
// loads see what an earlier KERNEL wrote).  Same tiles, same k order, same MFMA sequence per tile as solve9_kernel: the same bits.
//   chol (one workgroup; depends on P only) -> prod<0> Q, Q^T -> prod<1> M -> sweep (one workgroup) -> prod<2> X -> prod<3> W, shares of y -> dx
template <int BS, int WGR>
__global__ __launch_bounds__(64 * WGR * WGR) void solve9_chol_kernel(DevCfg cfg, int n, const double* __restrict__ P, double* __restrict__ scr) {
    constexpr int NT = WGR * BS;
    __shared__ S9CholLds<NT, WGR * WGR> sh;
    const int bad = s9_cholesky<BS, WGR>(cfg, n, P, scr, sh);
    if (threadIdx.x == 0) scr[(size_t)5 * NT * NT * S9_TILE] = bad ? 1.0 : 0.0;
}

template <int PH>
__global__ __launch_bounds__(256) void solve9_prod_kernel(DevCfg cfg, int n, const double* __restrict__ Ab, double* __restrict__ scr, double* __restrict__ Wout, int NT) {
    __shared__ double s_tb[4][16 * 17];
    const int c6 = 6 * n, ldh = cfg.ldh;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
    const int t = blockIdx.x * 4 + wv;
    if (t >= NT * NT) return;                                    // (no workgroup barrier below)
    const int i = t / NT, j = t - i * NT;
    if ((int)Ab[(size_t)ldh * (ldh - 1)] <= 2) {                 // pass-through (Updater.cc:460, 621-627): W = 0 => U = G = 0 => P+ = P exactly
        if constexpr (PH == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i + 4 * r + lk, col = 16 * j + li;
                if (row < c6 && col < c6) Wout[(size_t)row * ldh + col] = 0.0;
            }
        }
        return;
    }
    double* bL = scr;
    double* bG = scr + (size_t)NT * NT * S9_TILE;
    double* bQ = scr + (size_t)2 * NT * NT * S9_TILE;
    double* bQt = scr + (size_t)3 * NT * NT * S9_TILE;
    double* bMi = scr + (size_t)4 * NT * NT * S9_TILE;
    auto tile = [&](double* b, int a, int c) { return b + (size_t)(a * NT + c) * S9_TILE; };
    if constexpr (PH == 0) {                                     // Q(i, j) = sum_{k >= j} A(k, i)^T L(k, j); Q and Q^T
        auto ld_A = [&](int a, int c) {
            s9_d4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * a + 4 * r + lk, col = 16 * c + li;
                v[r] = (row < c6 && col < c6) ? Ab[(size_t)row * ldh + col] : 0.0;
            }
            return v;
        };
        const s9_d4 acc = s9_gemm(j, NT, [&](int k) { return ld_A(k, i); }, [&](int k) { return s9_ldg(tile(bL, k, j), lane); });
        s9_stg(tile(bQ, i, j), lane, acc);
        s9_stg(tile(bQt, j, i), lane, s9_transpose_tb(acc, s_tb[wv], li, lk));
    } else if constexpr (PH == 1) {                              // M(i, j) = s2 delta + sum_{k >= i} L(k, i)^T Q(k, j)
        s9_d4 acc = s9_gemm(i, NT, [&](int k) { return s9_ldg(tile(bL, k, i), lane); }, [&](int k) { return s9_ldg(tile(bQ, k, j), lane); });
        if (i == j) {
            const double s2 = cfg.sigma_im * cfg.sigma_im;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * r + lk == li) acc[r] += s2;
        }
        s9_stg(tile(bMi, i, j), lane, acc);
    } else if constexpr (PH == 2) {                              // X(i, j) = sum_{k <= j} Mi(k, i)^T G(k, j), over Q's buffer (W reads Q^T)
        const s9_d4 acc = s9_gemm(0, j + 1, [&](int k) { return s9_ldg(tile(bMi, k, i), lane); }, [&](int k) { return s9_ldg(tile(bG, k, j), lane); });
        s9_stg(tile(bQ, i, j), lane, acc);
    } else {                                                     // W(i, j) = (delta - sum_k Qt(k, i)^T X(k, j)) / s2; this tile's share of y = W b
        const s9_d4 acc = s9_gemm(0, NT, [&](int k) { return s9_ldg(tile(bQt, k, i), lane); }, [&](int k) { return s9_ldg(tile(bQ, k, j), lane); });
        const double is2 = 1.0 / (cfg.sigma_im * cfg.sigma_im);
        double* yp = scr + S9_YP_OFF(NT);
        const int col = 16 * j + li;
        const double bj = col < c6 ? Ab[(size_t)col * ldh + c6] : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + 4 * r + lk;
            const double w = (((row == col) ? 1.0 : 0.0) - acc[r]) * is2;
            if (row < c6 && col < c6) Wout[(size_t)row * ldh + col] = w;
            double tt = w * bj;
            tt += __shfl_xor(tt, 1, 16); tt += __shfl_xor(tt, 2, 16); tt += __shfl_xor(tt, 4, 16); tt += __shfl_xor(tt, 8, 16);
            if (li == 0) yp[(size_t)j * 16 * NT + row] = tt;
        }
    }
}

template <int BS, int WGR>
__global__ __launch_bounds__(64 * WGR * WGR) void solve9_sweep_kernel(DevCfg cfg, const double* __restrict__ Ab, double* __restrict__ scr) {
    constexpr int NT = WGR * BS, TS = BS * BS;
    __shared__ double s_rowp[2][NT][S9_TILE];
    __shared__ double s_F[2][S9_TILE];
    __shared__ S9Wave ws;
    __shared__ int s_bad;
    const int ldh = cfg.ldh;
    if ((int)Ab[(size_t)ldh * (ldh - 1)] <= 2) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wv / WGR, wb = wv % WGR;
    double* bMi = scr + (size_t)4 * NT * NT * S9_TILE;
    DBG_T(34);
    if (tid == 0) s_bad = 0;
    s9_d4 S[TS];
#pragma unroll
    for (int s = 0; s < TS; ++s) S[s] = s9_ldg(bMi + (size_t)((wa * BS + s / BS) * NT + wb * BS + s % BS) * S9_TILE, lane);
    int bad = 0;
    s9_sweep<BS, WGR>(S, s_rowp, s_F, &ws, bad);
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        s9_d4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = -S[s][r];
        s9_stg(bMi + (size_t)((wa * BS + s / BS) * NT + wb * BS + s % BS) * S9_TILE, lane, v);
    }
    if (bad) atomicOr(&s_bad, 1);
    __syncthreads();
    if (tid == 0) scr[(size_t)5 * NT * NT * S9_TILE + 1] = s_bad ? 1.0 : 0.0;
    DBG_T(35);
}

// y = sum of the shares, dx = K r = Pc y (Updater.cc:544), state injection (Updater.cc:546-613); the pass-through of an update without rows
__global__ __launch_bounds__(1024) void solve9_dx_kernel(DevCfg cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Ab, const double* __restrict__ x,
                                                         const double* __restrict__ P, const double* __restrict__ scr, double* __restrict__ Wout,
                                                         double* __restrict__ x_out, int NT) {
    constexpr int NTH = 1024;
    __shared__ double s_y[16 * 12];
    __shared__ double s_dx[24 + 6 * RVIO_MAX_LEN];
    __shared__ double s_part[4 * (24 + 6 * RVIO_MAX_LEN)];
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax, xd = 26 + 7 * n, NP = 16 * NT;
    const int tid = threadIdx.x;
    const int n_good = (int)Ab[(size_t)ldh * (ldh - 1)], n_rows = (int)Ab[(size_t)ldh * (ldh - 1) + 1];
    const bool upd = n_good > 2;                       // Updater.cc:460
    DBG_R(true, 2);
    if (tid == 0) { meta->n_good = n_good; meta->n_rows = n_rows; meta->updated = upd ? 1 : 0; meta->trunc_at = (int)Ab[(size_t)ldh * (ldh - 1) + 2]; }
    if (!upd) {                                        // pass-through (Updater.cc:621-627; solve9_prod_kernel<3> wrote W = 0)
        for (int i = tid; i < xd; i += NTH) x_out[i] = x[i];
        return;
    }
    DBG_T(37);
    const double* yp = scr + S9_YP_OFF(NT);
    if (tid == 0 && (scr[(size_t)5 * NT * NT * S9_TILE] != 0.0 || scr[(size_t)5 * NT * NT * S9_TILE + 1] != 0.0)) atomicOr(&meta->err, 1);
    for (int i = tid; i < NP; i += NTH) { double acc = yp[i]; for (int j = 1; j < NT; ++j) acc += yp[(size_t)j * NP + i]; s_y[i] = acc; }
    __syncthreads();
    {
        const int np = max(1, min(4, NTH / d)), share = (c6 + np - 1) / np;
        const int pt = tid / d, i = tid - pt * d;
        if (pt < np) {
            double acc = 0;
            const int k1 = min(c6, (pt + 1) * share);
#pragma unroll 8
            for (int k = pt * share; k < k1; ++k) acc += P[(size_t)i + (size_t)(24 + k) * ld] * s_y[k];
            s_part[pt * d + i] = acc;
        }
        __syncthreads();
        if (tid < d) { double acc = s_part[tid]; for (int q = 1; q < np; ++q) acc += s_part[q * d + tid]; s_dx[tid] = acc; }
    }
    __syncthreads();
    const double* dx = s_dx;
    if (tid == 0) {
        stq(x_out, qmul(small_q(dx[0], dx[1], dx[2]), ldq(x)));
        for (int i = 0; i < 6; ++i) x_out[4 + i] = dx[3 + i] + x[4 + i];
        st3(x_out + 7, unit3(ld3(x_out + 7)));
        stq(x_out + 10, qmul(small_q(dx[9], dx[10], dx[11]), ldq(x + 10)));
        for (int i = 0; i < 12; ++i) x_out[14 + i] = dx[12 + i] + x[14 + i];
    }
    for (int p = tid - 64; p >= 0 && p < n; p += NTH - 64) {
        stq(x_out + 26 + 7 * p, qmul(small_q(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), ldq(x + 26 + 7 * p)));
        for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
    }
    DBG_T(38);
    DBG_R(true, 7);
}

// The same as ROLE workgroups of the launch that follows the solve on the chain (ug_tile_kernel<0>: U = Pc W needs W complete, exactly like dx = Pc W b):
// role r of ceil(d / 24) takes 24 rows of dx — the IMU block, or four clones — and injects that part of the state; 256 threads, row = tid % 24 along the
// lanes (a column of Pc is contiguous), ten column groups summed in a fixed order.  Takes solve9_dx_kernel (one CU pulling all of Pc: 11 us at 6n = 180)
// and its launch off the filter chain.  role 0 also reports (meta, the factorisations' verdicts).
struct S9DxLds { double y[16 * 12]; double part[10][24]; double dx[24]; };
__device__ __forceinline__ void s9_dx_role(const DevCfg& cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Ab, const double* __restrict__ x,
                                           const double* __restrict__ P, const double* __restrict__ scr, double* __restrict__ x_out, int NT, int role, S9DxLds& L) {
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax, NP = 16 * NT;
    const int tid = threadIdx.x;
    const int n_good = (int)Ab[(size_t)ldh * (ldh - 1)], n_rows = (int)Ab[(size_t)ldh * (ldh - 1) + 1];
    const bool upd = n_good > 2;                       // Updater.cc:460
    if (role == 0 && tid == 0) {
        meta->n_good = n_good; meta->n_rows = n_rows; meta->updated = upd ? 1 : 0; meta->trunc_at = (int)Ab[(size_t)ldh * (ldh - 1) + 2];
        if (upd && (scr[(size_t)5 * NT * NT * S9_TILE] != 0.0 || scr[(size_t)5 * NT * NT * S9_TILE + 1] != 0.0)) atomicOr(&meta->err, 1);
    }
    const int x0 = role == 0 ? 0 : 26 + 28 * (role - 1), x1 = role == 0 ? 26 : min(26 + 7 * n, x0 + 28);   // this role's part of the state vector
    if (!upd) {                                        // pass-through (Updater.cc:621-627)
        for (int i = x0 + tid; i < x1; i += blockDim.x) x_out[i] = x[i];
        return;
    }
    const double* yp = scr + S9_YP_OFF(NT);
    for (int i = tid; i < NP; i += blockDim.x) { double acc = yp[i]; for (int j = 1; j < NT; ++j) acc += yp[(size_t)j * NP + i]; L.y[i] = acc; }
    __syncthreads();
    const int r0 = 24 * role, rl = tid % 24, grp = tid / 24, share = (c6 + 9) / 10;
    if (grp < 10) {
        double acc = 0;
        const int k1 = min(c6, (grp + 1) * share);
        if (r0 + rl < d) {
#pragma unroll 6
            for (int k = grp * share; k < k1; ++k) acc += P[(size_t)(r0 + rl) + (size_t)(24 + k) * ld] * L.y[k];
        }
        L.part[grp][rl] = acc;
    }
    __syncthreads();
    if (tid < 24) { double acc = L.part[0][tid]; for (int q = 1; q < 10; ++q) acc += L.part[q][tid]; L.dx[tid] = acc; }
    __syncthreads();
    const double* dx = L.dx;                            // rows 24 role .. of K r (Updater.cc:544)
    if (role == 0) {                                    // state injection (Updater.cc:546-613), the IMU block
        if (tid == 0) {
            stq(x_out, qmul(small_q(dx[0], dx[1], dx[2]), ldq(x)));
            for (int i = 0; i < 6; ++i) x_out[4 + i] = dx[3 + i] + x[4 + i];
            st3(x_out + 7, unit3(ld3(x_out + 7)));
            stq(x_out + 10, qmul(small_q(dx[9], dx[10], dx[11]), ldq(x + 10)));
            for (int i = 0; i < 12; ++i) x_out[14 + i] = dx[12 + i] + x[14 + i];
        }
    } else if (tid < 4) {                               // ... four clones
        const int p = 4 * (role - 1) + tid;
        if (p < n) {
            stq(x_out + 26 + 7 * p, qmul(small_q(dx[6 * tid], dx[6 * tid + 1], dx[6 * tid + 2]), ldq(x + 26 + 7 * p)));
            for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[6 * tid + 3 + i] + x[26 + 7 * p + 4 + i];
        }
    }
}

// =============================================================== 6n <= 64 with the Cholesky factor already in the slab: everything in LDS
// The pipelined frame's solve at the headline window (6n = 60).  The generic kernel above sends every phase's tiles through the L2 slab
// (one L2 round trip per product step, ~1 k cycles, four or five steps per phase) and reads Pc for dx = Pc y from HBM at the very end; here
//   * L and G come from the slab ONCE (written by the role workgroup of the per-feature launch) and every later tile lives in LDS,
//   * 16 waves, one 16 x 16 tile each, no register blocking needed,
//   * W^T = (I - L Mi (G A)) / s2 — the transposed Woodbury form: with R = G A beside Q = A L no tile is ever transposed
//       Q = A L, R = G A;   M = s2 I + L^T Q;   Mi = M^-1 (sweep);   V = Mi R;   W^T = (I - L V) / s2,   y = W b
//   * the IMU rows and the clone block of P (dx = Pc y) are fetched into registers at the start and parked in LDS once Q and R are dead.
// LDS (dynamic): L -> V 32 KB | G (10 tiles) 20 KB | Q -> Mi 32 KB | R 32 KB | row panel 16 KB | F 4 KB | in-wave scratch | vectors.
// Same arithmetic per tile as solve9_kernel<1, 4, true> up to the association of the Woodbury product (W^T instead of W): results agree to rounding.
struct S9SmallLds {
    double L[16][S9_TILE];                             // L(i, k) at [4 i + k]; later V(i, j)
    double G[10][S9_TILE];                             // G(k, j), k <= j, at [j (j + 1) / 2 + k]
    double Q[16][S9_TILE];                             // Q(i, j); later Mi(i, j); later (with R) the parked Pc
    double R[16][S9_TILE];                             // R(i, j) = (G A)(i, j)
    double rowp[2][4][S9_TILE];
    double F[2][S9_TILE];
    double Z[4][S9_TILE];                              // Z(t) = F^-T-products of the sweep's row panel: formed ONCE per step (four waves), read by all
    S9Wave ws;
    double b[64], y[64], yp[4][64];
    double dx[24 + 64];
    double part[4 * (24 + 64)];
    int bad;
};
__global__ __launch_bounds__(1024) void solve9_small_kernel(DevCfg cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Ab,
                                                            const double* __restrict__ x, const double* __restrict__ P, const double* __restrict__ scr,
                                                            double* __restrict__ Wout, double* __restrict__ x_out, double* __restrict__ yp_out) {
    // yp_out != NULL (round 6, the frame's update): the kernel ends with W and the four row-tile shares of y = W b in the slab — dx = Pc y and the state
    // injection (4.7 of its 29 us: nothing the Joseph stage needs) are role workgroups of the Joseph launch right behind it (s9_dx_role), as in the split form
    constexpr int NT = 4, NTH = 1024;
    extern __shared__ __align__(16) double s9s_dyn[];
    S9SmallLds& sh = *reinterpret_cast<S9SmallLds*>(s9s_dyn);
    const int c6 = 6 * n, d = 24 + c6, ldh = cfg.ldh, ld = cfg.dmax, xd = 26 + 7 * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
    const int ti = wv >> 2, tj = wv & 3;                // this wave's tile
    const int n_good = (int)Ab[(size_t)ldh * (ldh - 1)], n_rows = (int)Ab[(size_t)ldh * (ldh - 1) + 1];
    const bool upd = n_good > 2;                       // Updater.cc:460
    DBG_R(true, 2);
    if (tid == 0) { meta->n_good = n_good; meta->n_rows = n_rows; meta->updated = upd ? 1 : 0; meta->trunc_at = (int)Ab[(size_t)ldh * (ldh - 1) + 2]; sh.bad = 0; }
    if (!upd) {                                        // pass-through (Updater.cc:621-627): W = 0 => U = G = 0 => P+ = P exactly
        for (int e = tid; e < c6 * c6; e += NTH) Wout[(size_t)(e / c6) * ldh + (e % c6)] = 0.0;
        if (!yp_out) for (int i = tid; i < xd; i += NTH) x_out[i] = x[i];      // (deferred: the roles pass the state through)
        return;
    }
    DBG_T(30);
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    auto ld_A = [&](int i, int j) {                     // A(i, j) as a tile, zero beyond 6n
        s9_d4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + 4 * r + lk, col = 16 * j + li;
            v[r] = (row < c6 && col < c6) ? Ab[(size_t)row * ldh + col] : 0.0;
        }
        return v;
    };
    // ---- everything this workgroup will ever read from memory, in flight at once, every byte ONCE: A (tile (ti, tj) per wave, staged in the Q region
    // — one CU takes 64 B per clock from its L1: sixteen waves fetching the eight A tiles each of them multiplies with was 256 KB for a 29 KB matrix),
    // the factor, Pc, b
    const s9_d4 a0 = ld_A(ti, tj);
    const bool lowt = ti >= tj;                         // tile (ti, tj) of L exists; its transpose position holds G(tj, ti)
    s9_d4 l0 = s9_zero(), g0 = s9_zero();
    if (lowt) { l0 = s9_ldg(scr + (size_t)(ti * NT + tj) * S9_TILE, lane); g0 = s9_ldg(scr + (size_t)NT * NT * S9_TILE + (size_t)(tj * NT + ti) * S9_TILE, lane); }
    constexpr int PCN = (88 * 64 + NTH - 1) / NTH;      // Pc: d <= 88 rows x c6 <= 64 columns, column k of Pc at [k][88]
    double pc[PCN];
#pragma unroll
    for (int u = 0; u < PCN; ++u) {
        const int e = tid + u * NTH, k = e / 88, i = e - k * 88;
        pc[u] = (!yp_out && k < c6 && i < d) ? P[(size_t)i + (size_t)(24 + k) * ld] : 0.0;
    }
    if (tid < 64) sh.b[tid] = (tid < c6) ? Ab[(size_t)tid * ldh + c6] : 0.0;
    if (tid == 0 && scr[(size_t)5 * NT * NT * S9_TILE] != 0.0) sh.bad = 1;     // the Cholesky role's verdict
    if (lowt) { s9_sts(sh.L[ti * NT + tj], lane, l0); s9_sts(sh.G[ti * (ti + 1) / 2 + tj], lane, g0); }
    s9_sts(sh.Q[wv], lane, a0);
    __syncthreads();
    DBG_T(32);
    // ---- Q = A L (Q(i, j) = sum_{k >= j} A(k, i)^T L(k, j)),  R = G A (R(i, j) = sum_{k >= i} L(k, i)^T A(k, j)); A leaves the Q region behind a barrier
    {
        s9_d4 q = s9_zero(), rr = s9_zero();
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            if (k >= tj) q = s9_tn(s9_lds(sh.Q[k * NT + ti], lane), s9_lds(sh.L[k * NT + tj], lane), q);
            if (k >= ti) rr = s9_tn(s9_lds(sh.L[k * NT + ti], lane), s9_lds(sh.Q[k * NT + tj], lane), rr);
        }
        __syncthreads();
        s9_sts(sh.Q[wv], lane, q);
        s9_sts(sh.R[wv], lane, rr);
    }
    __syncthreads();
    DBG_T(33);
    // ---- M = s2 I + L^T Q into the tableau
    s9_d4 S = s9_zero();
#pragma unroll
    for (int k = 0; k < NT; ++k)
        if (k >= ti) S = s9_tn(s9_lds(sh.L[k * NT + ti], lane), s9_lds(sh.Q[k * NT + tj], lane), S);
    if (ti == tj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (4 * r + lk == li) S[r] += s2;
    }
    DBG_T(34);
    // ---- the symmetric sweep (as in solve9_kernel); the tableau ends as -M^-1
    int bad = 0;
#pragma unroll 1
    for (int k = 0; k < NT; ++k) {
        double (*rowp)[S9_TILE] = sh.rowp[k & 1];
        if (ti == k) s9_sts(rowp[tj], lane, S);
        __syncthreads();
        s9_d4 F, Ft;
        if (wv == 0) {
            s9_factor(s9_lds(rowp[k], lane), nullptr, 0.0, &sh.ws, li, lk, F, Ft, bad);
            s9_sts(sh.F[0], lane, F);
            s9_sts(sh.F[1], lane, Ft);
        }
        __syncthreads();
        // (round 6) Z(t) = Ft^T rowp[t], t = 0..3, formed ONCE by waves 0..3 (one per SIMD) and shared through LDS: every wave used to form its own Zr = Z(ti)
        // and Zc = Z(tj) — 32 tile products per step for 4 distinct results, and the FP64 matrix pipe of a SIMD (64 cycles per 16x16x4, four waves to a
        // SIMD) was the step: 12 MFMAs x 4 waves = 3 k cycles.  Now 4 + 4 x 4 = 1.3 k and a barrier.  Same products on the same operands: same bits.
        F = s9_lds(sh.F[0], lane);
        if (wv < NT) {
            Ft = s9_lds(sh.F[1], lane);
            s9_sts(sh.Z[wv], lane, s9_tn(Ft, s9_lds(rowp[wv], lane), s9_zero()));
        }
        __syncthreads();
        if (ti == k && tj == k) {
            const s9_d4 dd = s9_tn(F, F, s9_zero());
#pragma unroll
            for (int r = 0; r < 4; ++r) S[r] = -dd[r];
        } else if (ti == k) S = s9_tn(F, s9_lds(sh.Z[tj], lane), s9_zero());
        else if (tj == k) S = s9_tn(s9_lds(sh.Z[ti], lane), F, s9_zero());
        else {
            const s9_d4 Zr = s9_lds(sh.Z[ti], lane);
            s9_d4 nz;
#pragma unroll
            for (int r = 0; r < 4; ++r) nz[r] = -Zr[r];
            S = s9_tn(nz, s9_lds(sh.Z[tj], lane), S);
        }
    }
    {
        s9_d4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = -S[r];
        s9_sts(sh.Q[wv], lane, v);                      // Mi over Q (every wave left Q behind before the sweep's first barrier)
    }
    __syncthreads();
    DBG_T(35);
    // ---- V = Mi R (V(i, j) = sum_k Mi(k, i)^T R(k, j)) over L (dead since M)
    {
        s9_d4 v = s9_zero();
#pragma unroll
        for (int k = 0; k < NT; ++k) v = s9_tn(s9_lds(sh.Q[k * NT + ti], lane), s9_lds(sh.R[k * NT + tj], lane), v);
        s9_sts(sh.L[wv], lane, v);
    }
    __syncthreads();
    DBG_T(36);
    // park Pc in the dead Q | R region (64 KB >= 88 x 64 doubles) for dx
    double* pcs = &sh.Q[0][0];
    if (!yp_out) {
#pragma unroll
        for (int u = 0; u < PCN; ++u) { const int e = tid + u * NTH; if (e < 88 * 64) ((s9_lds_t*)pcs)[e] = pc[u]; }
    }
    // ---- W^T = (I - L V) / s2  (Wt(i, j) = (delta - sum_{k <= i} G(k, i)^T V(k, j)) / s2);  W(16 j + b, 16 i + a) = Wt(a, b);  y = W b by tile
    {
        const double is2 = 1.0 / s2;
        s9_d4 acc = s9_zero();
#pragma unroll
        for (int k = 0; k < NT; ++k)
            if (k <= ti) acc = s9_tn(s9_lds(sh.G[ti * (ti + 1) / 2 + k], lane), s9_lds(sh.L[k * NT + tj], lane), acc);
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = 16 * ti + 4 * r + lk, bcol = 16 * tj + li;      // Wt(a, bcol) = W(bcol, a)
            const double w = (((a == bcol) ? 1.0 : 0.0) - acc[r]) * is2;
            if (a < c6 && bcol < c6) Wout[(size_t)bcol * ldh + a] = w;
            t = fma(w, sh.b[a], t);                     // this tile's share of y[bcol]: over its 16 rows a — four in the lane, then the four lane groups (fixed order)
        }
        t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
        if (lk == 0) { if (yp_out) yp_out[ti * 64 + 16 * tj + li] = t; else sh.yp[ti][16 * tj + li] = t; }
    }
    if (bad) atomicOr(&sh.bad, 1);
    __syncthreads();
    if (tid == 0 && sh.bad) atomicOr(&meta->err, 1);
    if (yp_out) {      // the roles of the Joseph launch sum the shares in the same fixed order, form dx = Pc y from P itself and inject the state
        if (tid == 0) yp_out[-7] = 0.0;     // (the sweep's verdict word of the slab, S9_YP_OFF - 7: reported through meta->err above)
        DBG_T(37); DBG_T(38);
        DBG_R(true, 7);
        return;
    }
    if (tid < 64) sh.y[tid] = ((sh.yp[0][tid] + sh.yp[1][tid]) + sh.yp[2][tid]) + sh.yp[3][tid];
    __syncthreads();
    DBG_T(37);
    // ---- dx = K r = Pc y (Updater.cc:544) from the parked Pc: 4 shares of the columns per row, summed in a fixed order
    {
        const int np = 4, share = (c6 + np - 1) / np;
        const int pt = tid / 88, i = tid - pt * 88;
        if (pt < np && i < d) {
            double acc = 0;
            const int k1 = min(c6, (pt + 1) * share);
            for (int k = pt * share; k < k1; ++k) acc += ((const s9_lds_t*)pcs)[k * 88 + i] * sh.y[k];
            sh.part[pt * 88 + i] = acc;
        }
        __syncthreads();
        if (tid < d) sh.dx[tid] = ((sh.part[tid] + sh.part[88 + tid]) + sh.part[2 * 88 + tid]) + sh.part[3 * 88 + tid];
    }
    __syncthreads();
    // ---- state injection (Updater.cc:546-613)
    const double* dx = sh.dx;
    if (tid == 0) {
        stq(x_out, qmul(small_q(dx[0], dx[1], dx[2]), ldq(x)));
        for (int i = 0; i < 6; ++i) x_out[4 + i] = dx[3 + i] + x[4 + i];
        st3(x_out + 7, unit3(ld3(x_out + 7)));
        stq(x_out + 10, qmul(small_q(dx[9], dx[10], dx[11]), ldq(x + 10)));
        for (int i = 0; i < 12; ++i) x_out[14 + i] = dx[12 + i] + x[14 + i];
    }
    for (int p = tid - 64; p >= 0 && p < n; p += NTH - 64) {
        stq(x_out + 26 + 7 * p, qmul(small_q(dx[24 + 6 * p], dx[24 + 6 * p + 1], dx[24 + 6 * p + 2]), ldq(x + 26 + 7 * p)));
        for (int i = 0; i < 3; ++i) x_out[26 + 7 * p + 4 + i] = dx[24 + 6 * p + 3 + i] + x[26 + 7 * p + 4 + i];
    }
    DBG_T(38);
    DBG_R(true, 7);
}
