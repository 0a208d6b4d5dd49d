// literal.h — the reference's measurement compression, LITERALLY: sequential Givens QR of the stacked [Hw | r] in the reference's
// row order + the leading-row rank scan (Updater.cc:493-529), on the device, for the small stacks where the structural form of the
// rank decision (filter_kernels.hip, trunc_finish) is not certain (round 6).
//
// Why.  With R = s2 I the update only needs [A|b] = Hn^T [Hn | rn]; the device forms it as a sum of per-feature Gram shares and
// reproduces the reference's truncation `nRank = leading rows of R with norm >= 1e-4` from the STRUCTURE of the stack.  That rule is
// exact on every simulated sequence, but random sweeps against the reference's own Updater::update (tests/test_truncation.py,
// tests/test_ref_pins.py, round 5) found stacks of a handful of features it does not cover: a COLUMN GAP behind an over-determined
// block (rounding residue of the block is compacted into the gap, the scan stops there and every later feature is thrown away: up to
// 4e-4 of state on the stock motion) and barely tall stacks whose last rows of R are weak without being residue.  Where the residue
// rows sit when a gap column is swept depends on the sweep's row order, which only the sweep itself knows — so those stacks get the
// sweep itself.
//
// When (lit_decide; mirrored by oracle/filter.cpp:orc_update_local): the update was handed at most LIT_FEATS features, more than two
// were accepted, the stack is tall (rows > 6n), and  (a) a greedy count of the accepted features' rows against their column spans
// finds a column no feature can fill while an over-determined group precedes it (`_gap_stop` of the tests + the over-determination
// flag; the type-'2' block counts with rank e2: the scale gauge of a monocular window), or  (b) the stack is barely tall
// (rows - 6n <= LIT_SLACK).  Everything else keeps the information form (and the structural rule of trunc_finish for the many-feature
// case it was derived on).
//
// How.  feat_build_body exports every feature's RAW block [Hx | r | Hf] (before its own nullspace projection, which uses three
// Householder reflectors: another orthonormal basis of the same space — equal information, but rows of R that sit within rounding
// of the scan's threshold can fall on the other side of it) when n_feat <= LIT_FEATS (a few KB per feature).  ONE workgroup of 256
// threads first repeats the reference's OWN nullspace sweep on the accepted features' blocks (Updater.cc:370-402: Givens rotations
// column by column, rows bottom-up, applied to Hf, Hx and r; one wave per feature, the block in LDS), so that the stack it then
// compresses is the reference's stack row for row.  The same workgroup (the workgroup that finishes the Gram reduction: gram_reduce_kernel's block 0,
// lit_batch_kernel, block_sum_kernel's last block) then runs the compression sweep as a SYSTOLIC ARRAY: cell n holds the running row of
// column n's bottom-up chain (Updater.cc:498-511: m = M-1 .. n+1, rows (m-1, m)); the stack enters cell 0 from the bottom, one row
// per step; each rotation keeps the upper result as the new running row and hands the lower (zeroed) row to cell n+1, which is two
// steps behind — exactly the order of operations of the sequential loops, M + 2N steps deep instead of M N.  makeGivens is Eigen's
// (its exact-zero cases decide where structurally empty rows travel: oracle/refshim/mini_eigen.hpp is the specification) and the
// arithmetic is not contracted into FMAs, so that given the same rows the device takes the same branches.  Then the scan, and
// [A|b] = Rn^T [Rn | zn] of the nRank leading rows goes to the same solve as ever.
#pragma once

// (experiments only: rvio_hip_debug_literal_force — every update the literal sweep CAN take (<= LIT_FEATS features, > 2 accepted, tall) takes it)
__device__ int g_lit_force = 0;
#define LIT_FEATS 24      // an update handed more features than this never takes the literal path (M <= LIT_FEATS * rho_max rows)
#define LIT_SLACK 0       // "barely tall" (rows - 6n <= LIT_SLACK) as a second trigger: 0 = off — see the header
#define LIT_SPARE 48      // the gap trigger applies to stacks with few rows to spare only: rows - 6n <= LIT_SPARE (see lit_decide)
#define LIT_RING 32       // stack rows staged in LDS (two blocks of LIT_RING / 2)

// state of the array: U (running rows), X[2] (rows in flight between cells, double-buffered), each `tri` doubles: cell n owns
// columns n..Nc (Nc = the residual), offset n (Nc + 1) - n (n - 1) / 2
__host__ __device__ inline size_t lit_tri(int c6) { return (size_t)c6 * (c6 + 1) / 2 + c6; }
// LDS of lit_finish besides the state: ring of stack rows, (c, s) pairs of two steps, the level-to-level hand-over of two steps, row norms, the row map
__host__ __device__ inline size_t lit_aux_doubles(int ldh, int rho_max) { return (size_t)LIT_RING * ldh + 4 * (size_t)ldh + 2 * 256 + ldh + (size_t)(LIT_FEATS * rho_max + 1) / 2 + 8; }
__host__ __device__ inline size_t lit_state_doubles(int c6) { return 3 * lit_tri(c6); }
// a feature's raw block in LDS for the nullspace sweep: 2 max_len rows of [Hx columns + residual | Hf (3)]
__host__ __device__ inline size_t lit_slab_doubles(int ldh, int rho_max) { return (size_t)(rho_max + 2) * (ldh + 3); }
// the export buffer: LIT_FEATS blocks of 2 max_len rows x ldh, then the Hf blocks (2 max_len x 3 each)
// + the projected blocks the nullspace sweep leaves (rho_max rows x ldh each): the raw blocks stay as exported, a second run on the same export
// (rvio_hip_debug_time_kernel) finds what the first one found
__host__ __device__ inline size_t lit_rows_doubles(int ldh, int rho_max) { return (size_t)LIT_FEATS * (rho_max + 2) * (ldh + 3) + (size_t)LIT_FEATS * rho_max * ldh; }
__host__ __device__ inline double* lit_proj_of(double* lit_rows, int ldh, int rho_max) { return lit_rows + (size_t)LIT_FEATS * (rho_max + 2) * (ldh + 3); }
__host__ __device__ inline const double* lit_hf_of(const double* lit_rows, int ldh, int rho_max) { return lit_rows + (size_t)LIT_FEATS * (rho_max + 2) * ldh; }

// Eigen::JacobiRotation<double>::makeGivens(p, q) (real case): the rotation G with G^T [p; q] = [r; 0].  The cases are Eigen's — its exact
// zeros decide where structurally empty rows travel — and so are the formula (t = q / p, u = +-sqrt(1 + t^2), c = 1 / u, s = -t c, and its
// mirror image) and the arithmetic: IEEE division and square root, no contraction.  On windows with exactly duplicated clone poses whole rows
// of the stack are equal, the rotations produce EXACT zeros and the next makeGivens branches on them; hardware reciprocal / rsqrt estimates
// refined by Newton steps (<= 2 ulp, three times shorter — this sits on the serial chain of every step) were tried twice and moved nRank from
// 39 to 38 on such stacks (1.2e-4 / 2.9e-4 of state: tests/test_gpu_literal.py, sweep_wider trials 493 and 1055).
// Straight-line code (selects): one quotient, one root, one reciprocal whatever the case.
__device__ __forceinline__ void lit_givens(double p, double q, double& c, double& s) {
#pragma clang fp contract(off)
    const bool pbig = fabs(p) > fabs(q);
    const double num = pbig ? q : p, den = pbig ? p : q;
    const double t = num / ((den == 0.0) ? 1.0 : den);
    double u = sqrt(1.0 + t * t);
    if (den < 0.0) u = -u;
    const double r = 1.0 / u;
    const double cb = pbig ? r : -t * (-r), sb = pbig ? -t * r : -r;     // |p| > |q|: c = 1/u, s = -t c;  else: s = -1/u, c = -t s
    const bool qz = q == 0.0, pz = p == 0.0;
    c = qz ? (p < 0.0 ? -1.0 : 1.0) : (pz ? 0.0 : cb);
    s = qz ? 0.0 : (pz ? (q < 0.0 ? 1.0 : -1.0) : sb);
}

// (a) of the header, by ONE thread: the accepted features' rows by start column (type '2': columns 0..e2, 2 ceil(L/2) - 3 rows, rank
// <= e2; type '1': columns 6 (n - L + 1) .. 6n - 1, 2 L - 3 rows), greedy fill in the order of the start columns
__device__ inline bool lit_gap_trigger(int n, int n_feat, const int* nrows, const unsigned char* types, const int* lens, int* rows_k, int* end_k) {
    // rows_k / end_k [40] (LDS): group k <-> start column 6 k (k = 0: the type-'2' block and full-window type-'1' features)
    const int ng = n + 1 < 40 ? n + 1 : 40;
    for (int k = 0; k < ng; ++k) { rows_k[k] = 0; end_k[k] = -1; }
    bool gauge0 = false;           // group 0 holds type-'2' rows only: its rank is one short of its span
    bool any1_0 = false;
    for (int f = 0; f < n_feat; ++f) {
        const int r = nrows[f];
        if (r <= 0) continue;
        const int L = lens[f];
        int k, e;
        if (types[f] == '2') { const int Lu = (L + 1) / 2; k = 0; e = 6 * (Lu - 1) - 1; gauge0 = true; }
        else { k = n - (L - 1); e = 6 * n - 1; if (k == 0) any1_0 = true; }
        if (k < 0 || k >= ng) continue;
        rows_k[k] += r; end_k[k] = max(end_k[k], e);
    }
    const bool only2 = gauge0 && !any1_0 && rows_k[0] > 0;     // a type-'2' block alone at column 0
    int p = 0, done = 0; bool over = false;
    for (int k = 0; k < ng; ++k) {
        if (rows_k[k] == 0) continue;
        if (p < 6 * k) {
            // the gap right behind a lone type-'2' block is the constellation the structural rule was derived on and is proven on (trunc_finish,
            // conditions (a)-(d): every simulated sequence; full-load frames have it, with thousands of rows) — it stays there
            if (!(only2 && done == 1)) return over;       // (behind that one the count goes on: later rows move up into the gap, a second gap may follow)
        }
        const int cap = (k == 0 && only2 && end_k[0] < 6 * n - 1) ? end_k[0] : end_k[k] + 1;
        if (p + rows_k[k] > cap) over = true;
        p = min(p + rows_k[k], cap);
        ++done;
    }
    return false;
}

// the decision (every thread of the calling workgroup gets the same answer; good / rows: the counters of the whole update)
__device__ inline bool lit_decide(const double* lit_rows, int n, int n_feat, int good, int rows, const int* nrows, const unsigned char* types, const int* lens) {
    __shared__ int s_lit, s_rows_k[40], s_end_k[40];
    const int c6 = 6 * n;
    if (!lit_rows || n_feat > LIT_FEATS || good <= 2 || rows <= c6) return false;     // (uniform: no barrier below is skipped by a part of the workgroup)
    if (rows - c6 <= LIT_SLACK) return true;
    if (__hip_atomic_load(&g_lit_force, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;   // (uniform)
    // a gap stops the scan only when the stack has few rows to spare: with many, informative rows move up into the gap and the residue of the
    // over-determined group ends at the bottom (every exception of the sweeps: rows - 6n <= 21; the one trigger on the stock sequence: 390 spare
    // rows, literal result = information form to 1e-16 — and 0.8 ms of sweep)
    if (rows - c6 > LIT_SPARE) return false;
    if (threadIdx.x == 0) s_lit = lit_gap_trigger(n, n_feat, nrows, types, lens, s_rows_k, s_end_k) ? 1 : 0;
    __syncthreads();
    const bool go = s_lit != 0;
    __syncthreads();
    return go;
}

template <bool LDS> struct LitAS { typedef double* P; };
template <> struct LitAS<true> { typedef __attribute__((address_space(3))) double* P; };
typedef __attribute__((address_space(3))) double* LitLP;

// The array's M + Nc - 1 steps, GENERAL form (any window; state U, X[2] in LDS — typed pointers — or in global memory): thread <-> (column c,
// cell group g), cells nn = g, g + G, .. <= min(c, Nc - 1), in chunks of LIT_CH with every load of a chunk issued before its first store; two
// barriers per step (apply; form the rotations of the next step).  Bound by instruction issue: ~25 instructions per element of the array on
// four waves, ~1.5 us per step at 6n = 60 — the form below takes over wherever it fits.
#define LIT_CH 8
template <bool LDS, class REFILL>
__device__ __forceinline__ void lit_sweep(double* st_, double* ring_, double* cs_, int M, int Nc, int c6, int ldh, REFILL refill) {
#pragma clang fp contract(off)
    typedef typename LitAS<LDS>::P SP;
    typedef LitLP LP;
    const int tid = threadIdx.x, T = blockDim.x, W = Nc + 1;
    const size_t tri = lit_tri(c6);
    SP U = (SP)st_; SP X0 = (SP)(st_ + tri); SP X1 = (SP)(st_ + 2 * tri);
    LP ring = (LP)ring_; LP cs = (LP)cs_;
    auto off = [&](int nn) { return nn * W - nn * (nn - 1) / 2; };      // cell nn owns columns nn..Nc
    const int G = max(1, T / W), c = tid % W, g = tid / W;
    const bool live = g < G;
    const int ncell = min(c, Nc - 1) + 1;
    const int HB = LIT_RING / 2;
    const int t_end = M + Nc - 2;                        // cell Nc-1 takes its last input (the row at position Nc-1) at step 2 (Nc-1) + (M - Nc)
    for (int t = 0; t <= t_end; ++t) {
        SP Xin = (t & 1) ? X1 : X0; SP Xout = (t & 1) ? X0 : X1;
        LP csn = cs + (size_t)(t & 1) * 2 * ldh;
        if (live) {
            // cells active at step t: input i = t - 2 nn in [0, M - 1 - nn]
            const int n_hi = min(t >> 1, ncell - 1), n_lo = max(0, t - (M - 1));
            const double rin = ring[(size_t)(((M - 1 - t) % LIT_RING + LIT_RING) % LIT_RING) * ldh + c];      // cell 0's input (unused otherwise)
            for (int n0 = n_lo + ((g - n_lo) % G + G) % G; n0 <= n_hi; n0 += LIT_CH * G) {
                double xin[LIT_CH], y[LIT_CH], cc[LIT_CH], ss[LIT_CH];
#pragma unroll
                for (int u = 0; u < LIT_CH; ++u) {
                    const int nn = n0 + u * G;
                    if (nn <= n_hi) {
                        const int o = off(nn) + (c - nn);
                        xin[u] = (nn == 0) ? rin : Xin[o];
                        y[u] = U[o]; cc[u] = csn[2 * nn]; ss[u] = csn[2 * nn + 1];
                    } else { xin[u] = 0; y[u] = 0; cc[u] = 1; ss[u] = 0; }
                }
#pragma unroll
                for (int u = 0; u < LIT_CH; ++u) {
                    const int nn = n0 + u * G;
                    if (nn <= n_hi) {
                        const int o = off(nn) + (c - nn);
                        if (t == 2 * nn) U[o] = xin[u];  // its first input: the running row
                        else {
                            U[o] = cc[u] * xin[u] - ss[u] * y[u];        // block.applyOnTheLeft(0, 1, G.adjoint()): upper row (m-1) <- c x - s y, lower row (m) <- s x + c y
                            if (nn + 1 < ncell) Xout[off(nn + 1) + (c - nn - 1)] = ss[u] * xin[u] + cc[u] * y[u];
                        }
                    }
                }
            }
        }
        __syncthreads();
        // the rotations of step t + 1: cell nn turns (p = its next input at column nn, q = its running row at column nn)
        if (tid < Nc) {
            const int nn = tid, i = t + 1 - 2 * nn;
            if (i >= 1 && i <= M - 1 - nn) {
                const double p = (nn == 0) ? ring[(size_t)((M - 2 - t) % LIT_RING) * ldh] : Xout[off(nn)];
                double cc, ss;
                lit_givens(p, U[off(nn)], cc, ss);
                LP o = cs + (size_t)((t + 1) & 1) * 2 * ldh + 2 * nn;
                o[0] = cc; o[1] = ss;
            }
        }
        if ((t % HB) == HB - 1) refill(M - 1 - ((t / HB) + 2) * HB);      // the block consumed during the last LIT_RING / 2 steps is dead: its slots take the block after next
        __syncthreads();
    }
}

// The same steps with the array in REGISTERS (no state in memory until the end): thread <-> (column c, level j) holds K consecutive cells of
// its column, from the top — slot k <-> cell nhi - k, nhi = min(c, Nc - 1) - j K — as the running entries uu[k] and the entries in flight
// xin[k].  A rotation's lower result moves from slot k to slot k - 1 (the next cell, same column) inside the thread; only slot 0's crosses
// to the thread one level up, through LDS.  The thread of a pivot column (level 0, c < Nc) owns both operands of its cell's next rotation —
// the entry its slot 1 just emitted and its slot 0's new running entry — and forms it on the spot: ONE barrier per step, ~8 instructions
// per element.  Threads needed: sum over levels j (j K < Nc) of Nc - j K + 1.
__host__ __device__ inline int lit_reg_threads(int Nc, int K) { const int L = (Nc + K - 1) / K; return L * (Nc + 1) - K * L * (L - 1) / 2; }
template <int K, bool LDS, class REFILL>
__device__ __forceinline__ void lit_sweep_reg(double* st_, double* ring_, double* cs_, double* bnd_, int M, int Nc, int c6, int ldh, REFILL refill) {
#pragma clang fp contract(off)
    typedef typename LitAS<LDS>::P SP;
    typedef LitLP LP;
    const int tid = threadIdx.x, T = blockDim.x, W = Nc + 1;
    LP ring = (LP)ring_; LP cs = (LP)cs_; LP bnd = (LP)bnd_;      // bnd[2][T]
    int j = 0, base = 0, c = -1, cnt = 0;
    for (;;) {
        cnt = (j * K < Nc) ? (Nc - j * K + 1) : 0;
        if (cnt == 0) break;
        if (tid < base + cnt) { c = j * K + (tid - base); break; }
        base += cnt; ++j;
    }
    const bool live = c >= 0;
    const int ncell = live ? min(c, Nc - 1) + 1 : 0, nhi = ncell - 1 - j * K;     // top cell of this thread (>= 0 when live)
    const bool below = live && nhi - K >= 0;             // the thread one level down exists: its slot 0 feeds this thread's slot K - 1
    const int tid_below = base + cnt + (c - (j + 1) * K);
    const bool diag = live && j == 0 && c < Nc;          // slot 0 is cell c at its pivot column
    double uu[K], xin[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { uu[k] = 0.0; xin[k] = 0.0; }
    const int HB = LIT_RING / 2;
    const int t_end = M + Nc - 2;
    for (int t = 0; t <= t_end; ++t) {
        LP csn = cs + (size_t)(t & 1) * 2 * ldh; LP csx = cs + (size_t)((t + 1) & 1) * 2 * ldh;
        LP bcur = bnd + (size_t)(t & 1) * T; LP bnxt = bnd + (size_t)((t + 1) & 1) * T;
        if (live) {
            // Straight-line code (selects, no branches): a branch per slot kept every slot's rotation load inside its own exec-masked block —
            // K LDS round trips in a row, 5.6 k clocks per step.  Here the K rotation loads go out together and the slots' arithmetic pipelines.
            if (below) xin[K - 1] = bcur[tid_below];
            const double rin = ring[(size_t)(((M - 1 - t) % LIT_RING + LIT_RING) % LIT_RING) * ldh + c];      // cell 0's input (unused otherwise)
            double cc[K], ss[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { const int nc2 = 2 * max(nhi - k, 0); cc[k] = csn[nc2]; ss[k] = csn[nc2 + 1]; }
            double dn0 = 0.0; bool em0 = false;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int nn = nhi - k, i = t - 2 * nn;
                const bool on = nn >= 0 && i >= 0 && i <= M - 1 - nn, rot = on && i >= 1;
                const double x = (nn == 0) ? rin : xin[k], y = uu[k];
                const double up = cc[k] * x - ss[k] * y;     // block.applyOnTheLeft(0, 1, G.adjoint()): upper row (m-1) <- c x - s y, lower row (m) <- s x + c y
                const double dn = ss[k] * x + cc[k] * y;
                uu[k] = rot ? up : (on ? x : y);             // (i == 0: its first input is the running row)
                if (k > 0) xin[k - 1] = rot ? dn : xin[k - 1];   // (slot k - 1 went through before: its input of the NEXT step)
                else { dn0 = dn; em0 = rot; }
            }
            if (em0 && j > 0) bnxt[tid] = dn0;
            if (diag) {                                  // the rotation cell c applies at step t + 1
                const int i1 = t + 1 - 2 * c;
                if (i1 >= 1 && i1 <= M - 1 - c) {
                    const double p = (c == 0) ? ring[(size_t)((M - 2 - t) % LIT_RING) * ldh] : xin[0];
                    double gc, gs;
                    lit_givens(p, uu[0], gc, gs);
                    csx[2 * c] = gc; csx[2 * c + 1] = gs;
                }
            }
        }
        if ((t % HB) == HB - 1) { __syncthreads(); refill(M - 1 - ((t / HB) + 2) * HB); }      // the block consumed during the last LIT_RING / 2 steps is dead
        __syncthreads();
    }
    // the running rows are R: into the state buffer, for the scan and the Gram product
    SP U = (SP)st_;
    if (live) {
#pragma unroll
        for (int k = 0; k < K; ++k) { const int nn = nhi - k; if (nn >= 0) U[nn * W - nn * (nn - 1) / 2 + (c - nn)] = uu[k]; }
    }
    __syncthreads();
}

// Updater.cc:370-402 on ONE feature's raw block, by one wave: M2 rows of [Hx (columns lo..hi-1) | r] in rows[.][ldh] and Hf in hf[.][3]
// (global, written by feat_build_body); N = 3, or 2 where the reference found Hf's third column short (the per-feature kernel took that
// decision for the gate: N = M2 - accepted rows).  Lane <-> up to three columns of [Hx | r]; every lane carries the three columns of
// Hf itself (the rotations come from them: no hand-over between lanes).  slab: lit_slab_doubles() of LDS, private to the wave.
// On return out[0 .. M2-N-1] hold the projected rows (rows N.. of the swept block): tempHx_, tempr_ of Updater.cc:407-409.
__device__ void lit_nullspace_wave(const double* rows, const double* hf, double* out, int M2, int N, int lo, int hi, int c6, int ldh, double* slab_) {
#pragma clang fp contract(off)
    typedef __attribute__((address_space(3))) double* LP;
    LP slab = (LP)slab_;
    const int lane = threadIdx.x & 63, wa = hi - lo, Wf = wa + 1, ls = ldh + 3;
    int col[3]; bool on[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int k = lane + 64 * j; on[j] = k < Wf; col[j] = k < wa ? lo + k : c6; }
    for (int i = 0; i < M2; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) if (on[j]) slab[(size_t)i * ls + lane + 64 * j] = rows[(size_t)i * ldh + col[j]];
        if (lane < 3) slab[(size_t)i * ls + ldh + lane] = hf[3 * i + lane];
    }
    __builtin_amdgcn_wave_barrier();
    for (int n = 0; n < N; ++n) {
        double ru[3], rh[3];
        LP last = slab + (size_t)(M2 - 1) * ls;
#pragma unroll
        for (int j = 0; j < 3; ++j) { ru[j] = on[j] ? last[lane + 64 * j] : 0.0; rh[j] = last[ldh + j]; }
        for (int m = M2 - 1; m > n; --m) {
            LP up = slab + (size_t)(m - 1) * ls;
            LP dn = slab + (size_t)m * ls;
            double lu[3], lh[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { lu[j] = on[j] ? up[lane + 64 * j] : 0.0; lh[j] = up[ldh + j]; }
            double c, s;
            lit_givens(lh[n], rh[n], c, s);              // makeGivens(tempHf(m-1, n), tempHf(m, n))
#pragma unroll
            for (int j = 0; j < 3; ++j) {                // rows (m-1, m) of Hx and r: x <- c x - s y, y <- s x + c y
                const double x = lu[j], y = ru[j];
                ru[j] = c * x - s * y;
                if (on[j]) dn[lane + 64 * j] = s * x + c * y;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {                // Hf: columns n..N-1 only (Updater.cc:391)
                if (j >= n && j < N) {
                    const double x = lh[j], y = rh[j];
                    rh[j] = c * x - s * y;
                    if (lane == 0) dn[ldh + j] = s * x + c * y;
                } else { if (lane == 0) dn[ldh + j] = rh[j]; rh[j] = lh[j]; }     // (untouched columns: the rows stay where they are)
            }
            __builtin_amdgcn_wave_barrier();
        }
        LP top = slab + (size_t)n * ls;
#pragma unroll
        for (int j = 0; j < 3; ++j) { if (on[j]) top[lane + 64 * j] = ru[j]; if (lane == 0) top[ldh + j] = rh[j]; }
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = N; i < M2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) if (on[j]) out[(size_t)(i - N) * ldh + col[j]] = slab[(size_t)i * ls + lane + 64 * j];
}

// The scan (Updater.cc:516-523): leading rows of R with norm >= 1e-4 (Ho.row(i): the 6n columns, not the residual), and
// [A|b] = Rn^T [Rn | zn] of those nRank rows, both triangles, columns / rows beyond Nc zero.  Also leaves the row norms the scan saw in the unused
// second part of the block (diagnostic).  Returns nRank.
template <bool LDS>
__device__ __forceinline__ int lit_scan_gram(double* st_, double* nrm, double* A, int Nc, int c6, int ldh, int* s_rank) {
    typedef typename LitAS<LDS>::P SP;
    SP U = (SP)st_;
    const int tid = threadIdx.x, T = blockDim.x, W = Nc + 1;
    auto off = [&](int nn) { return nn * W - nn * (nn - 1) / 2; };
    for (int i = tid; i < Nc; i += T) {
        double s = 0;
        const int o = off(i);
        for (int k = 0; k < Nc - i; ++k) s += U[o + k] * U[o + k];
        nrm[i] = sqrt(s);
    }
    __syncthreads();
    if (tid == 0) {
        int r = 0;
        while (r < Nc && !(nrm[r] < 1e-4)) ++r;
        *s_rank = r;
    }
    for (int i = tid; i < Nc; i += T) A[(size_t)ldh * ldh + i] = nrm[i];
    __syncthreads();
    const int nRank = *s_rank;
    for (int e = tid; e < c6 * ldh; e += T) {
        const int q = e % ldh, p = e / ldh;
        if (q > c6) continue;
        double v = 0;
        const int qc = (q == c6) ? Nc : q;               // column of the array
        if (p < Nc && (q == c6 || q < Nc)) {
            const int lim = min(nRank, min(p, qc) + 1);
#pragma unroll 4
            for (int i = 0; i < lim; ++i) { const int o = off(i) - i; v = fma(U[o + p], U[o + qc], v); }
        }
        A[e] = v;
    }
    return nRank;
}

// phase stamps (100 MHz wall clock) behind the diagnostic row norms in the unused second part of the block: tools/lit_probe.py
#define LIT_STAMP(k) do { if (threadIdx.x == 0) A[(size_t)cfg.ldh * cfg.ldh + 200 + (k)] = (double)wall_clock64(); } while (0)
// The two sweeps + scan + [A|b], by one workgroup of 256 threads.  A: the [A|b] block (c6 x ldh row-major inside ldh x ldh; the spare last
// row = the meta row).  st: lit_state_doubles(c6) doubles (LDS or global — generic pointer), aux: lit_aux_doubles() doubles of LDS.
// lds_cap: doubles of LDS behind aux that the nullspace sweep may use for its slabs (everything else of this function starts after it).
__device__ void lit_finish(const DevCfg& cfg, int n, int n_feat, const int* nrows, const unsigned char* types, const int* lens,
                           double* lit_rows, double* A, int good, int rows, double* st, bool st_in_lds, double* aux, size_t lds_cap) {
#pragma clang fp contract(off)
    const int c6 = 6 * n, ldh = cfg.ldh, rho_max = cfg.rho_max, tid = threadIdx.x, T = blockDim.x;
    __shared__ int s_f[LIT_FEATS], s_lo[LIT_FEATS], s_hi[LIT_FEATS], s_off[LIT_FEATS + 1], s_ng, s_nc, s_rank;
    double* ring = aux;                                  // [LIT_RING][ldh]: row m of the stack at slot m % LIT_RING, columns 0..Nc-1 + residual at Nc
    double* cs = ring + (size_t)LIT_RING * ldh;          // [2][ldh][2]
    double* bnd = cs + 4 * (size_t)ldh;                  // [2][256]
    double* nrm = bnd + 2 * 256;                         // [ldh]
    int* rowmap = (int*)(nrm + ldh);                     // [M]: (slot << 8) | local row
    if (tid == 0) {
        int ng = 0, M = 0, Nc = 0;
        for (int f = 0; f < n_feat && f < LIT_FEATS; ++f) {
            const int r = nrows[f];
            if (r <= 0) continue;
            const int L = lens[f];
            const bool t2 = types[f] == '2';
            const int Lu = t2 ? (L + 1) / 2 : L, lo = t2 ? 0 : 6 * (n - (Lu - 1)), hi = lo + 6 * (Lu - 1);
            s_f[ng] = f; s_lo[ng] = lo; s_hi[ng] = hi; s_off[ng] = M;
            M += r; Nc = max(Nc, hi); ++ng;
        }
        s_off[ng] = M; s_ng = ng; s_nc = Nc;             // Nc: columns 0..Nc-1 are swept (the trailing all-zero columns are dropped, Updater.cc:482-491)
    }
    __syncthreads();
    const int ng = s_ng, Nc = s_nc, M = s_off[ng];
    LIT_STAMP(0);
    const int RB = rho_max + 2;                          // rows of a feature's block in the export buffer
    double* proj = lit_proj_of(lit_rows, ldh, rho_max);  // the projected rows (rho_max x ldh per feature)
    {   // (0) the reference's nullspace sweep on the accepted features' raw blocks: one wave per feature, as many at a time as slabs fit
        const size_t slab = lit_slab_doubles(ldh, rho_max);
        const int nslab = (int)min((size_t)(T >> 6), lds_cap / slab), wv = tid >> 6;
        const double* hf_all = lit_hf_of(lit_rows, ldh, rho_max);
        if (wv < nslab)
            for (int s = wv; s < ng; s += nslab) {
                const int f = s_f[s], rr = s_off[s + 1] - s_off[s], L = lens[f];
                const int M2 = 2 * ((types[f] == '2') ? (L + 1) / 2 : L);
                lit_nullspace_wave(lit_rows + (size_t)f * RB * ldh, hf_all + (size_t)f * RB * 3, proj + (size_t)f * rho_max * ldh, M2, M2 - rr, s_lo[s], s_hi[s], c6, ldh,
                                   aux + (size_t)wv * slab);
            }
        __threadfence_block();
        __syncthreads();
    }
    LIT_STAMP(1);
    for (int m = tid; m < M; m += T) {
        int s = 0;
        while (s + 1 < ng && s_off[s + 1] <= m) ++s;
        rowmap[m] = (s << 8) | (m - s_off[s]);
    }
    const int W = Nc + 1;                                // columns of a stack row as the array sees it: 0..Nc-1, residual
    const size_t tri = lit_tri(c6);
    auto off = [&](int nn) { return nn * W - nn * (nn - 1) / 2; };      // cell nn owns columns nn..Nc
    // element (m, c) of the stack
    auto stack_at = [&](int m, int c) -> double {
        const int rm = rowmap[m], s = rm >> 8, loc = rm & 255;
        const double* row = proj + ((size_t)s_f[s] * rho_max + loc) * ldh;
        if (c == Nc) return row[c6];
        return (c >= s_lo[s] && c < s_hi[s]) ? row[c] : 0.0;
    };
    __syncthreads();
    // stage the first LIT_RING rows from the bottom of the stack
    for (int e = tid; e < LIT_RING * W; e += T) {
        const int m = M - 1 - e / W, c = e % W;
        if (m >= 0) ring[(size_t)(m % LIT_RING) * ldh + c] = stack_at(m, c);
    }
    __syncthreads();
    // the sweep itself (state in LDS: typed LDS pointers — as generic pointers every access is a flat instruction whose store fences the next load)
    auto refill = [&](int mb2) {                         // rows mb2, mb2 - 1, .. (LIT_RING / 2 of them) into their ring slots
        const int HBr = LIT_RING / 2;
        if (mb2 >= 0)
            for (int e = tid; e < HBr * W; e += T) {
                const int m = mb2 - e / W;
                if (m >= 0) ring[(size_t)(m % LIT_RING) * ldh + e % W] = stack_at(m, e % W);
            }
    };
    LIT_STAMP(2);
    if (tid < Nc) { cs[2 * tid] = 1.0; cs[2 * tid + 1] = 0.0; cs[2 * ldh + 2 * tid] = 1.0; cs[2 * ldh + 2 * tid + 1] = 0.0; }
    __syncthreads();
    if (T >= 256 && lit_reg_threads(Nc, 12) <= T) { if (st_in_lds) lit_sweep_reg<12, true>(st, ring, cs, bnd, M, Nc, c6, ldh, refill); else lit_sweep_reg<12, false>(st, ring, cs, bnd, M, Nc, c6, ldh, refill); }
    else if (T >= 256 && lit_reg_threads(Nc, 24) <= T) { if (st_in_lds) lit_sweep_reg<24, true>(st, ring, cs, bnd, M, Nc, c6, ldh, refill); else lit_sweep_reg<24, false>(st, ring, cs, bnd, M, Nc, c6, ldh, refill); }
    else if (T >= 256 && lit_reg_threads(Nc, 40) <= T) { if (st_in_lds) lit_sweep_reg<40, true>(st, ring, cs, bnd, M, Nc, c6, ldh, refill); else lit_sweep_reg<40, false>(st, ring, cs, bnd, M, Nc, c6, ldh, refill); }
    else if (st_in_lds) lit_sweep<true>(st, ring, cs, M, Nc, c6, ldh, refill);
    else lit_sweep<false>(st, ring, cs, M, Nc, c6, ldh, refill);
    LIT_STAMP(3);
    int nRank;
    if (st_in_lds) nRank = lit_scan_gram<true>(st, nrm, A, Nc, c6, ldh, &s_rank);
    else nRank = lit_scan_gram<false>(st, nrm, A, Nc, c6, ldh, &s_rank);
    if (tid == 0) {
        double* mr = A + (size_t)ldh * (ldh - 1);
        mr[0] = (double)good; mr[1] = (double)rows; mr[2] = nRank < Nc ? (double)nRank : -1.0; mr[5] = (double)nRank;
    }
    LIT_STAMP(4);
}
