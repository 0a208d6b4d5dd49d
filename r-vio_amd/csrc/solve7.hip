// solve7.hip — T = s2 I + A Pcc, W = T^-1, y = W b, dx = Pc y, state injection  (Updater.cc:540-613), generation 7.
//
// One launch replaces gemm_T_kernel + solve6_kernel.  The in-place Gauss-Jordan inversion (partial pivoting, no row swaps,
// deferred pivot scaling: the algebra written out in solve6.hip) runs with the tableau in REGISTERS instead of LDS:
//   * lane <-> row (RPL rows per lane: i = lane + 64 r), wave <-> every NW-th PAIR of columns (columns 2q, 2q+1 live in wave q % NW,
//     registers 2 (q / NW), + 1); every wave also carries the right-hand side b as one more column, so nobody owns it;
//   * DATAFLOW instead of barriers: the wave that owns the next pair of columns brings those two up to date first, takes BOTH pivots
//     (the first one's elimination reaches the second column inside the wave) and publishes (pivot rows, their reciprocals, the
//     multipliers f_i of every row, twice) into an LDS ring slot, then raises the slot's flag;
//     every wave consumes the steps in order at its own pace (acquire-load of the flag, one LDS read), so the chain of pivot
//     decisions is never held up by the other waves' 16 column updates — they run beside it.  With cyclic ownership a wave is the
//     publisher every NW-th step only and has caught up by then;
//   * the pivot row reaches the other rows through v_readlane (an SGPR operand of the FMA), never through memory;
//   * pivot candidates are 32-bit keys (high word of |value| with the low 8 bits replaced by 255 - row): unsigned max, ties -> smaller
//     row, magnitudes resolved to 2^-12 relative — the chosen pivot is within 0.025 % of the column maximum, deterministic.
// solve6_kernel spent ~1900 cycles per column on five dependent LDS round trips behind a barrier; this one spends one, unsynchronised.
// The prologue forms T on the FP64 matrix cores (the former gemm_T_kernel).  6n <= 64: A and Pcc are staged in LDS with one batch
// of coalesced loads, the tiles go to LDS and from there into the register layout; longer windows use the T scratch buffer in L2.
// Result as before: T^-1[k][p_j] = M[p_k][j] / piv_k,  y[k] = M[p_k][b] / piv_k.
#pragma once
#include <type_traits>
#include <utility>
#include "rvio_dev.h"

template <int CTRL>
__device__ __forceinline__ unsigned s7_dpp(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned s7_max(unsigned a, unsigned b) { return a > b ? a : b; }
typedef double s7_d4 __attribute__((ext_vector_type(4)));
// compile-time loop: f(std::integral_constant<int, I>) for I = 0..N-1 (register arrays must be indexed by constants)
template <int I, int N, class F>
__device__ __forceinline__ void s7_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); s7_for<I + 1, N>(f); }
}

#define S7_RING 32     // slots of the publication ring: >= NW + 2 (a wave lags the publisher by less than NW + 1 steps)

// STAGE (6n <= 64 only): T's operands through LDS (one stream: latency); false: T through the scratch buffer in L2 like the longer windows —
// 11 KB of LDS instead of 112 KB, eight workgroups per CU: the throughput form batch handles use (with RING = 8 publication slots).
template <int RPL, int CPW, int NW, bool STAGE_ = (RPL == 1), int RING = S7_RING>
__global__ __launch_bounds__(64 * NW) void solve7_kernel(DevCfg cfg, FilterMeta* __restrict__ meta, int n, const double* __restrict__ Ab,
                                                         const double* __restrict__ x, const double* __restrict__ P, double* __restrict__ Tscr,
                                                         double* __restrict__ Wout, double* __restrict__ x_out, size_t bs) {
    meta = zoff(meta, bs); Ab = zoff(Ab, bs); x = zoff(x, bs); P = zoff(P, bs); Tscr = zoff(Tscr, bs); Wout = zoff(Wout, bs); x_out = zoff(x_out, bs);
    static_assert(RPL >= 1 && RPL <= 3, "rows per lane");
    static_assert(CPW % 2 == 0, "a wave owns pairs of columns");
    static_assert(RING >= NW + 2 && (RING & (RING - 1)) == 0, "ring depth");
    static_assert(!STAGE_ || RPL == 1, "staging needs 6n <= 64");
    constexpr int NT = 64 * NW, NR = 64 * RPL;
    constexpr bool STAGE = STAGE_;                      // 6n <= 64: T through LDS
    constexpr int LS = 65;                              // LDS row stride of the staged 64 x 64 operands
    extern __shared__ __align__(16) double s7_dyn[];    // STAGE: As | Ps | Ts (3 x 64 x LS) | Pt (24 x 64); As is reused for the dx partial sums
    __shared__ double s_f[RING][2][NR];
    __shared__ int s_p[RING][2], s_flag[RING];
    __shared__ int s_prow[6 * RVIO_MAX_LEN], s_invp[NR];
    __shared__ double s_ipiv[6 * RVIO_MAX_LEN];
    __shared__ double s_y[6 * RVIO_MAX_LEN];
    __shared__ double s_dx[24 + 6 * RVIO_MAX_LEN];
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
    if (tid < RING) s_flag[tid] = 0;
    const double s2 = cfg.sigma_im * cfg.sigma_im;
    double mcol[CPW][RPL], mbv[RPL];                    // the tableau: register column cc of this wave, this lane's RPL rows; the right-hand side
    // ---- T = s2 I + A Pcc on the matrix cores: A = Ab (row-major, ld = ldh), B = Pcc = P[24:,24:] (column-major, ld = dmax)
    if constexpr (STAGE) {
        double* As = s7_dyn; double* Ps = s7_dyn + 64 * LS; double* Ts = s7_dyn + 2 * 64 * LS; double* Pt = s7_dyn + 3 * 64 * LS;
        // ONE batch of coalesced loads, all in flight before the first store: As[i][k] = A[i][k], Ps[j][k] = Pcc[k][j] (zero-padded to
        // 64 x 64), Pt[j][i] = P[i][24 + j] for the 24 IMU rows (dx = Pc y at the end reads Pc from LDS)
        {
            constexpr int NB = 64 * 64 / NT, NB2 = (24 * 64 + NT - 1) / NT;
            double va[NB], vp[NB], vt[NB2];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int e = tid + u * NT, a = e >> 6, k = e & 63;
                const bool ok = a < c6 && k < c6;
                va[u] = ok ? Ab[(size_t)a * ldh + k] : 0.0;
                vp[u] = ok ? P[(size_t)(24 + k) + (size_t)(24 + a) * ld] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NB2; ++u) {
                const int e = tid + u * NT, j = e / 24, i = e - j * 24;
                vt[u] = (e < 24 * 64 && j < c6) ? P[(size_t)i + (size_t)(24 + j) * ld] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) { const int e = tid + u * NT, a = e >> 6, k = e & 63; As[a * LS + k] = va[u]; Ps[a * LS + k] = vp[u]; }
#pragma unroll
            for (int u = 0; u < NB2; ++u) { const int e = tid + u * NT; if (e < 24 * 64) Pt[e] = vt[u]; }
        }
        __syncthreads();
        const int nt = (c6 + 15) / 16, li = lane & 15, lk = lane >> 4;
        for (int t = wv; t < nt * nt; t += NW) {
            const int i0 = (t / nt) * 16, j0 = (t % nt) * 16;
            const double* ap = As + (i0 + li) * LS;
            const double* bp = Ps + (j0 + li) * LS;
            s7_d4 acc = {0, 0, 0, 0};
            double av[16], bv[16];                      // every operand of the tile in flight before the first MFMA
#pragma unroll
            for (int u = 0; u < 16; ++u) { av[u] = ap[4 * u + lk]; bv[u] = bp[4 * u + lk]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + lk + 4 * r, col = j0 + li;
                Ts[row * LS + col] = acc[r] + ((row == col && row < c6) ? s2 : 0.0);
            }
        }
        __syncthreads();
        DBG_T(57);
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int j = 2 * ((cc >> 1) * NW + wv) + (cc & 1);
            mcol[cc][0] = (lane < c6 && j < c6) ? Ts[lane * LS + j] : 0.0;
        }
        mbv[0] = (lane < c6) ? Ab[(size_t)lane * ldh + c6] : 0.0;
    } else {
        const int nt = (c6 + 15) / 16, li = lane & 15, lk = lane >> 4;
        for (int t = wv; t < nt * nt; t += NW) {
            const int i0 = (t / nt) * 16, j0 = (t % nt) * 16;
            const int ai = i0 + li, bj = j0 + li;
            const bool aok = ai < c6, bok = bj < c6;
            const double* ap = Ab + (size_t)ai * ldh;
            const double* bp = P + 24 + (size_t)(24 + bj) * ld;
            s7_d4 acc = {0, 0, 0, 0};
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
            if (bj < c6) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + lk + 4 * r;
                    if (row < c6) Tscr[(size_t)row * ldh + bj] = acc[r] + ((row == bj) ? s2 : 0.0);
                }
            }
        }
        __threadfence();
        __syncthreads();
        DBG_T(57);
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = lane + 64 * r;
#pragma unroll
            for (int cc = 0; cc < CPW; ++cc) {
                const int j = 2 * ((cc >> 1) * NW + wv) + (cc & 1);
                mcol[cc][r] = (i < c6 && j < c6) ? __hip_atomic_load(Tscr + (size_t)i * ldh + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
            mbv[r] = (i < c6) ? Ab[(size_t)i * ldh + c6] : 0.0;
        }
    }
    unsigned used = 0;                                  // bit r: row lane + 64 r has been a pivot row
    DBG_T(58);

    // ---- elimination, TWO columns per publication.  Column j lives in wave (j/2) % NW, register 2*((j/2)/NW) + (j&1): a wave owns
    // pairs of adjacent columns, so the owner of pair s+1 can take both pivots of that pair back to back — first pivot, its elimination
    // applied to the pair's second column inside the wave, second pivot — and hand both over with ONE LDS publication: half the
    // cross-wave round trips on the serial chain of the inversion.
    // search: arg-max of this wave's register column CC over the unused rows, its reciprocal, every row's multiplier
    auto search = [&](auto CCtag, double (&fo)[RPL], int& p_o, double& ip_o, bool& sing) {
        constexpr int CC = decltype(CCtag)::value;
        unsigned key = 0;
        double rc[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = lane + 64 * r;
            const double cv = mcol[CC][r];
            // reciprocal of the candidate: hardware estimate + two Newton steps (full double precision for normal numbers; the
            // IEEE division sequence is three times as long and sits on the serial chain of the elimination)
            double y0 = __builtin_amdgcn_rcp(cv);
            y0 = fma(fma(-cv, y0, 1.0), y0, y0);
            rc[r] = fma(fma(-cv, y0, 1.0), y0, y0);
            if (i < c6 && !((used >> r) & 1u)) key = s7_max(key, ((unsigned)__double2hiint(fabs(cv)) & ~255u) | (255u - (unsigned)i));
        }
        unsigned b = key;
        b = s7_max(b, s7_dpp<0x128>(b)); b = s7_max(b, s7_dpp<0x124>(b)); b = s7_max(b, s7_dpp<0x122>(b)); b = s7_max(b, s7_dpp<0x121>(b));
        const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)b, 0), b1 = (unsigned)__builtin_amdgcn_readlane((int)b, 16);
        const unsigned b2 = (unsigned)__builtin_amdgcn_readlane((int)b, 32), b3 = (unsigned)__builtin_amdgcn_readlane((int)b, 48);
        const unsigned best = s7_max(s7_max(b0, b1), s7_max(b2, b3));
        sing = (best >> 8) == 0;
        const int pi = sing ? 0 : 255 - (int)(best & 255u);          // all-zero column: flagged by the caller, keep the indices sane
        const int lp = pi & 63, rp = pi >> 6;
        double ipiv = readlane_f64(rc[0], lp);
        if (RPL > 1 && rp == 1) ipiv = readlane_f64(rc[RPL > 1 ? 1 : 0], lp);
        if (RPL > 2 && rp == 2) ipiv = readlane_f64(rc[RPL > 2 ? 2 : 0], lp);
#pragma unroll
        for (int r = 0; r < RPL; ++r) fo[r] = (lane + 64 * r == pi) ? 0.0 : mcol[CC][r] * ipiv;
        p_o = pi; ip_o = ipiv;
    };
    // one elimination step on register column CC (or the right-hand side): m -= f * (pivot row's entry)
    auto elim_v = [&](double (&col)[RPL], const double (&fr)[RPL], int p) {
        const int lp = p & 63, rp = p >> 6;
        double pr = readlane_f64(col[0], lp);
        if (RPL > 1 && rp == 1) pr = readlane_f64(col[RPL > 1 ? 1 : 0], lp);
        if (RPL > 2 && rp == 2) pr = readlane_f64(col[RPL > 2 ? 2 : 0], lp);
#pragma unroll
        for (int r = 0; r < RPL; ++r) col[r] -= fr[r] * pr;
    };
    // both pivots of the pair held in registers (C0, C0 + 1) = columns (2 sp, 2 sp + 1), published as pair sp
    auto publish_pair = [&](auto C0tag, int sp) {
        constexpr int C0 = decltype(C0tag)::value;
        double fa[RPL], fb[RPL];
        int pa, pb; double ipa, ipb; bool sa, sb;
        search(std::integral_constant<int, C0>{}, fa, pa, ipa, sa);
        if (lane == (pa & 63)) used |= 1u << (pa >> 6);
        elim_v(mcol[C0 + 1], fa, pa);
        search(std::integral_constant<int, C0 + 1>{}, fb, pb, ipb, sb);
        const int slot = sp & (RING - 1);
#pragma unroll
        for (int r = 0; r < RPL; ++r) { s_f[slot][0][lane + 64 * r] = fa[r]; s_f[slot][1][lane + 64 * r] = fb[r]; }
        if (lane == 0) {
            s_p[slot][0] = pa; s_p[slot][1] = pb;
            s_prow[2 * sp] = pa; s_prow[2 * sp + 1] = pb; s_invp[pa] = 2 * sp; s_invp[pb] = 2 * sp + 1; s_ipiv[2 * sp] = ipa; s_ipiv[2 * sp + 1] = ipb;
            if (sa || sb) atomicOr(&meta->err, 1);
        }
        // LDS operations of one wave complete in order: the flag becomes visible after the data
        __hip_atomic_store(&s_flag[slot], sp + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    __syncthreads();                                    // flags cleared, tableau loaded
    if (wv == 0) publish_pair(std::integral_constant<int, 0>{}, 0);
    DBG_T(59);

    // pair sp = (C/2)*NW + w: registers (C, C+1) of wave w.  No barrier: every wave follows the flags.
    const int npair = c6 / 2;
    bool done = false;
    s7_for<0, CPW / 2>([&](auto Htag) {
        constexpr int C = 2 * decltype(Htag)::value;
        for (int w = 0; w < NW && !done; ++w) {
            const int sp = (C / 2) * NW + w;
            if (sp >= npair) { done = true; break; }
            if (sp == 15) DBG_T(50);
            if (sp == 16) DBG_T(52);
            const int slot = sp & (RING - 1);
            // flag, pivot rows and multipliers are read together; the data is valid if the flag (written last by the publisher) matches
            int pa, pb;
            double fa[RPL], fb[RPL];
            for (;;) {
                const int fl = __hip_atomic_load(&s_flag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int qa = *(volatile int*)&s_p[slot][0], qb = *(volatile int*)&s_p[slot][1];
#pragma unroll
                for (int r = 0; r < RPL; ++r) { fa[r] = *(volatile double*)&s_f[slot][0][lane + 64 * r]; fb[r] = *(volatile double*)&s_f[slot][1][lane + 64 * r]; }
                __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the reads were issued in this order and complete in order
                if (__builtin_amdgcn_readfirstlane(fl) == sp + 1) { pa = __builtin_amdgcn_readfirstlane(qa); pb = __builtin_amdgcn_readfirstlane(qb); break; }
                if (NW > 4) __builtin_amdgcn_s_sleep(2);   // more than one wave per SIMD: a spinning wave must not take the issue slots of the publisher
            }
            if (lane == (pa & 63)) used |= 1u << (pa >> 6);
            if (lane == (pb & 63)) used |= 1u << (pb >> 6);
            auto both = [&](double (&col)[RPL]) { elim_v(col, fa, pa); elim_v(col, fb, pb); };
            // the next pair lives in wave (w+1) % NW, registers (C, C+1) — or (C+2, C+3) when the ownership wraps: that wave brings
            // those two columns up to date first and publishes the next two pivots before it touches its other columns
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
                    if (own) {   // column 2 sp itself: -f_a (1 in its pivot row, the stored form of 1/piv), then an ordinary column of step b
#pragma unroll
                        for (int r = 0; r < RPL; ++r) mcol[I][r] = (lane + 64 * r == pa) ? 1.0 : -fa[r];
                        elim_v(mcol[I], fb, pb);
                    } else if (!(own_next && next_same)) both(mcol[I]);
                } else if constexpr (I == C + 1) {
                    if (own) {   // column 2 sp + 1 (step a reached it when the pair was published): -f_b, 1 in its pivot row
#pragma unroll
                        for (int r = 0; r < RPL; ++r) mcol[I][r] = (lane + 64 * r == pb) ? 1.0 : -fb[r];
                    } else if (!(own_next && next_same)) both(mcol[I]);
                } else if constexpr (I == C + 2 || I == C + 3) {
                    if (!(own_next && !next_same)) both(mcol[I]);
                } else both(mcol[I]);
            });
            both(mbv);
            if (sp == 15) DBG_T(51);
        }
    });
    __syncthreads();
    DBG_T(60);
    // ---- read the result out: W[k][p_j] = M[p_k][j] * ipiv_k ;  y[k] = M[p_k][b] * ipiv_k   (row i = p_k <=> k = invp[i])
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int i = lane + 64 * r;
        if (i < c6) {
            const int kk = s_invp[i];
            const double ip = s_ipiv[kk];
#pragma unroll
            for (int cc = 0; cc < CPW; ++cc) {
                const int j = 2 * ((cc >> 1) * NW + wv) + (cc & 1);
                if (j < c6) Wout[(size_t)kk * ldh + s_prow[j]] = mcol[cc][r] * ip;
            }
            if (wv == 0) s_y[kk] = mbv[r] * ip;
        }
    }
    __syncthreads();
    DBG_T(61);
    // dx = K r = Pc y   (Updater.cc:544): NT / d threads per row, each a contiguous share of the columns; partial sums added in a fixed order
    {
        double* part = STAGE ? s7_dyn : &s_f[0][0][0];         // (As is idle now; the ring is idle too: 2 NR RING >= 1024 doubles >= np d)
        const int np = max(1, min(4, NT / d)), share = (c6 + np - 1) / np;
        const int pt = tid / d, i = tid - pt * d;
        if (pt < np) {
            double acc = 0;
            const int k1 = min(c6, (pt + 1) * share);
            if constexpr (STAGE) {   // Pc from LDS: rows 0..23 = Pt[k][i], rows 24.. = Pcc[i-24][k] = Ps[k][i-24]
                const double* src = (i < 24) ? (s7_dyn + 3 * 64 * LS + i) : (s7_dyn + 64 * LS + (i - 24));
                const int st = (i < 24) ? 24 : LS;
                for (int k = pt * share; k < k1; ++k) acc += src[k * st] * s_y[k];
            } else
                for (int k = pt * share; k < k1; ++k) acc += P[(size_t)i + (size_t)(24 + k) * ld] * s_y[k];
            part[pt * d + i] = acc;
        }
        __syncthreads();
        if (tid < d) { double acc = part[tid]; for (int q = 1; q < np; ++q) acc += part[q * d + tid]; s_dx[tid] = acc; }
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
