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

#define LIT_FEATS 24      // an update handed more features than this never takes the literal path (M <= LIT_FEATS * rho_max rows)
#define LIT_SLACK 8       // "barely tall": rows - 6n <= LIT_SLACK
#define LIT_RING 32       // stack rows staged in LDS (two blocks of LIT_RING / 2)

// state of the array: U (running rows), X[2] (rows in flight between cells, double-buffered), each `tri` doubles: cell n owns
// columns n..Nc (Nc = the residual), offset n (Nc + 1) - n (n - 1) / 2
__host__ __device__ inline size_t lit_tri(int c6) { return (size_t)c6 * (c6 + 1) / 2 + c6; }
// LDS of lit_finish besides the state: ring of stack rows, (c, s) pairs of two steps, row norms, the row map
__host__ __device__ inline size_t lit_aux_doubles(int ldh, int rho_max) { return (size_t)LIT_RING * ldh + 4 * (size_t)ldh + ldh + (size_t)(LIT_FEATS * rho_max + 1) / 2 + 8; }
__host__ __device__ inline size_t lit_state_doubles(int c6) { return 3 * lit_tri(c6); }
// a feature's raw block in LDS for the nullspace sweep: 2 max_len rows of [Hx columns + residual | Hf (3)]
__host__ __device__ inline size_t lit_slab_doubles(int ldh, int rho_max) { return (size_t)(rho_max + 2) * (ldh + 3); }
// the export buffer: LIT_FEATS blocks of 2 max_len rows x ldh, then the Hf blocks (2 max_len x 3 each)
__host__ __device__ inline size_t lit_rows_doubles(int ldh, int rho_max) { return (size_t)LIT_FEATS * (rho_max + 2) * (ldh + 3); }
__host__ __device__ inline const double* lit_hf_of(const double* lit_rows, int ldh, int rho_max) { return lit_rows + (size_t)LIT_FEATS * (rho_max + 2) * ldh; }

// Eigen::JacobiRotation<double>::makeGivens(p, q) (real case): the rotation G with G^T [p; q] = [r; 0]
__device__ __forceinline__ void lit_givens(double p, double q, double& c, double& s) {
#pragma clang fp contract(off)
    if (q == 0.0) { c = p < 0.0 ? -1.0 : 1.0; s = 0.0; }
    else if (p == 0.0) { c = 0.0; s = q < 0.0 ? 1.0 : -1.0; }
    else if (fabs(p) > fabs(q)) {
        const double t = q / p;
        double u = sqrt(1.0 + t * t);
        if (p < 0.0) u = -u;
        c = 1.0 / u; s = -t * c;
    } else {
        const double t = p / q;
        double u = sqrt(1.0 + t * t);
        if (q < 0.0) u = -u;
        s = -1.0 / u; c = -t * s;
    }
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
    int p = 0; bool over = false;
    for (int k = 0; k < ng; ++k) {
        if (rows_k[k] == 0) continue;
        const int s = 6 * k;
        if (p < s) return over;
        const int cap = (k == 0 && gauge0 && !any1_0 && end_k[k] < 6 * n - 1) ? end_k[k] : end_k[k] + 1;
        if (p + rows_k[k] > cap) over = true;
        p = min(p + rows_k[k], cap);
    }
    return false;
}

// the decision (every thread of the calling workgroup gets the same answer; good / rows: the counters of the whole update)
__device__ inline bool lit_decide(const double* lit_rows, int n, int n_feat, int good, int rows, const int* nrows, const unsigned char* types, const int* lens) {
    __shared__ int s_lit, s_rows_k[40], s_end_k[40];
    const int c6 = 6 * n;
    if (!lit_rows || n_feat > LIT_FEATS || good <= 2 || rows <= c6) return false;     // (uniform: no barrier below is skipped by a part of the workgroup)
    if (rows - c6 <= LIT_SLACK) return true;
    if (threadIdx.x == 0) s_lit = lit_gap_trigger(n, n_feat, nrows, types, lens, s_rows_k, s_end_k) ? 1 : 0;
    __syncthreads();
    const bool go = s_lit != 0;
    __syncthreads();
    return go;
}

// Updater.cc:370-402 on ONE feature's raw block, by one wave: M2 rows of [Hx (columns lo..hi-1) | r] in rows[.][ldh] and Hf in hf[.][3]
// (global, written by feat_build_body); N = 3, or 2 where the reference found Hf's third column short (the per-feature kernel took that
// decision for the gate: N = M2 - accepted rows).  Lane <-> up to three columns of [Hx | r]; every lane carries the three columns of
// Hf itself (the rotations come from them: no hand-over between lanes).  slab: lit_slab_doubles() of LDS, private to the wave.
// On return rows[0 .. M2-N-1] hold the projected rows (rows N.. of the swept block): tempHx_, tempr_ of Updater.cc:407-409.
__device__ void lit_nullspace_wave(double* rows, const double* hf, int M2, int N, int lo, int hi, int c6, int ldh, double* slab) {
#pragma clang fp contract(off)
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
        const double* last = slab + (size_t)(M2 - 1) * ls;
#pragma unroll
        for (int j = 0; j < 3; ++j) { ru[j] = on[j] ? last[lane + 64 * j] : 0.0; rh[j] = last[ldh + j]; }
        for (int m = M2 - 1; m > n; --m) {
            const double* up = slab + (size_t)(m - 1) * ls;
            double* dn = slab + (size_t)m * ls;
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
        double* top = slab + (size_t)n * ls;
#pragma unroll
        for (int j = 0; j < 3; ++j) { if (on[j]) top[lane + 64 * j] = ru[j]; if (lane == 0) top[ldh + j] = rh[j]; }
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = N; i < M2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) if (on[j]) rows[(size_t)(i - N) * ldh + col[j]] = slab[(size_t)i * ls + lane + 64 * j];
}

// The two sweeps + scan + [A|b], by one workgroup of 256 threads.  A: the [A|b] block (c6 x ldh row-major inside ldh x ldh; the spare last
// row = the meta row).  st: lit_state_doubles(c6) doubles (LDS or global — generic pointer), aux: lit_aux_doubles() doubles of LDS.
// lds_cap: doubles of LDS behind aux that the nullspace sweep may use for its slabs (everything else of this function starts after it).
__device__ void lit_finish(const DevCfg& cfg, int n, int n_feat, const int* nrows, const unsigned char* types, const int* lens,
                           double* lit_rows, double* A, int good, int rows, double* st, double* aux, size_t lds_cap) {
#pragma clang fp contract(off)
    const int c6 = 6 * n, ldh = cfg.ldh, rho_max = cfg.rho_max, tid = threadIdx.x, T = blockDim.x;
    __shared__ int s_f[LIT_FEATS], s_lo[LIT_FEATS], s_hi[LIT_FEATS], s_off[LIT_FEATS + 1], s_ng, s_nc, s_rank;
    double* ring = aux;                                  // [LIT_RING][ldh]: row m of the stack at slot m % LIT_RING, columns 0..Nc-1 + residual at Nc
    double* cs = ring + (size_t)LIT_RING * ldh;          // [2][ldh][2]
    double* nrm = cs + 4 * (size_t)ldh;                  // [ldh]
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
    const int RB = rho_max + 2;                          // rows of a feature's block in the export buffer
    {   // (0) the reference's nullspace sweep on the accepted features' raw blocks: one wave per feature, as many at a time as slabs fit
        const size_t slab = lit_slab_doubles(ldh, rho_max);
        const int nslab = (int)min((size_t)(T >> 6), lds_cap / slab), wv = tid >> 6;
        const double* hf_all = lit_hf_of(lit_rows, ldh, rho_max);
        if (wv < nslab)
            for (int s = wv; s < ng; s += nslab) {
                const int f = s_f[s], rr = s_off[s + 1] - s_off[s], L = lens[f];
                const int M2 = 2 * ((types[f] == '2') ? (L + 1) / 2 : L);
                lit_nullspace_wave(lit_rows + (size_t)f * RB * ldh, hf_all + (size_t)f * RB * 3, M2, M2 - rr, s_lo[s], s_hi[s], c6, ldh, aux + (size_t)wv * slab);
            }
        __threadfence_block();
        __syncthreads();
    }
    for (int m = tid; m < M; m += T) {
        int s = 0;
        while (s + 1 < ng && s_off[s + 1] <= m) ++s;
        rowmap[m] = (s << 8) | (m - s_off[s]);
    }
    const int W = Nc + 1;                                // columns of a stack row as the array sees it: 0..Nc-1, residual
    const size_t tri = lit_tri(c6);
    double* U = st; double* X0 = st + tri; double* X1 = st + 2 * tri;
    auto off = [&](int nn) { return nn * W - nn * (nn - 1) / 2; };      // cell nn owns columns nn..Nc
    // element (m, c) of the stack
    auto stack_at = [&](int m, int c) -> double {
        const int rm = rowmap[m], s = rm >> 8, loc = rm & 255;
        const double* row = lit_rows + ((size_t)s_f[s] * RB + loc) * ldh;
        if (c == Nc) return row[c6];
        return (c >= s_lo[s] && c < s_hi[s]) ? row[c] : 0.0;
    };
    __syncthreads();
    // stage the first LIT_RING rows from the bottom of the stack
    for (int e = tid; e < LIT_RING * W; e += T) {
        const int m = M - 1 - e / W, c = e % W;
        if (m >= 0) ring[(size_t)(m % LIT_RING) * ldh + c] = stack_at(m, c);
    }
    // thread <-> (column c, cell group g): cells n = g, g + G, ... <= min(c, Nc - 1)
    const int G = max(1, T / W), c = tid % W, g = tid / W;
    const bool live = g < G;
    const int ncell = min(c, Nc - 1) + 1;
    if (tid < Nc) { cs[2 * tid] = 1.0; cs[2 * tid + 1] = 0.0; cs[2 * ldh + 2 * tid] = 1.0; cs[2 * ldh + 2 * tid + 1] = 0.0; }
    __syncthreads();
    const int HB = LIT_RING / 2;
    const int t_end = M + Nc - 2;                        // cell Nc-1 takes its last input (the row at position Nc-1) at step 2 (Nc-1) + (M - Nc)
    for (int t = 0; t <= t_end; ++t) {
        double* Xin = (t & 1) ? X1 : X0; double* Xout = (t & 1) ? X0 : X1;
        const double* csn = cs + (size_t)(t & 1) * 2 * ldh;
        if (live) {
            // cells active at step t: input i = t - 2 nn in [0, M - 1 - nn]
            const int n_hi = min(t >> 1, ncell - 1), n_lo = max(0, t - (M - 1));
            int nn = n_lo + ((g - n_lo) % G + G) % G;
            for (; nn <= n_hi; nn += G) {
                const int i = t - 2 * nn, o = off(nn) + (c - nn);
                const double xin = (nn == 0) ? ring[(size_t)((M - 1 - t) % LIT_RING) * ldh + c] : Xin[o];
                if (i == 0) { U[o] = xin; continue; }
                const double cc = csn[2 * nn], ss = csn[2 * nn + 1], y = U[o];
                U[o] = cc * xin - ss * y;                // block.applyOnTheLeft(0, 1, G.adjoint()): upper row (m-1) <- c x - s y, lower row (m) <- s x + c y
                if (nn + 1 < ncell) Xout[off(nn + 1) + (c - nn - 1)] = ss * xin + cc * y;
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
                double* o = cs + (size_t)((t + 1) & 1) * 2 * ldh + 2 * nn;
                o[0] = cc; o[1] = ss;
            }
        }
        if ((t % HB) == HB - 1) {                        // the block consumed during the last LIT_RING / 2 steps is dead: its slots take the block after next
            const int mb2 = M - 1 - ((t / HB) + 2) * HB; // (a plain copy through registers: one exposed L2 round trip per LIT_RING / 2 steps)
            if (mb2 >= 0)
                for (int e = tid; e < HB * W; e += T) {
                    const int m = mb2 - e / W;
                    if (m >= 0) ring[(size_t)(m % LIT_RING) * ldh + e % W] = stack_at(m, e % W);
                }
        }
        __syncthreads();
    }
    // the scan (Updater.cc:516-523): leading rows with norm >= 1e-4 (Ho.row(i): the 6n columns, not the residual)
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
        s_rank = r;
    }
    for (int i = tid; i < Nc; i += T) A[(size_t)ldh * ldh + i] = nrm[i];     // (diagnostic: the row norms the scan saw, in the unused second part of the block)
    __syncthreads();
    const int nRank = s_rank;
    // [A|b] = Rn^T [Rn | zn], both triangles, columns / rows beyond Nc zero
    for (int e = tid; e < c6 * ldh; e += T) {
        const int q = e % ldh, p = e / ldh;
        if (q > c6) continue;
        double v = 0;
        const int qc = (q == c6) ? Nc : q;               // column of the array
        if (p < Nc && (q == c6 || q < Nc)) {
            const int lim = min(nRank, min(p, qc) + 1);
            for (int i = 0; i < lim; ++i) { const int o = off(i) - i; v = fma(U[o + p], U[o + qc], v); }
        }
        A[e] = v;
    }
    if (tid == 0) {
        double* mr = A + (size_t)ldh * (ldh - 1);
        mr[0] = (double)good; mr[1] = (double)rows; mr[2] = nRank < Nc ? (double)nRank : -1.0; mr[5] = (double)nRank;
    }
}
