// detector.hip — T7: FeatureDetector::DetectWithSubPix (FeatureDetector.cc:55-75) =
//   cv::goodFeaturesToTrack(im, corners, nFeatures, qualityLevel, s*minDistance)   + cv::cornerSubPix(win 7x7, 30 it, 0.01)
// restated for the device, bit-exact against oracle/detector.cpp (same float/double expression order, no contraction; this
// file is included inside the fp-contract(off) region).  s = 1 on the first image, 2 on refills (Tracker.cc:207,350): the
// first-image flag lives on the device, so the kernels pick s themselves.
//   mineig_kernel     Sobel 3x3 (scaled) -> products -> 3x3 box (double) -> lambda_min, + image maximum (atomic on an ordered key)
//   nms_kernel        threshold at max*quality, strict 3x3 local maxima away from the border -> candidate list + per-cell buckets
//   greedy_kernel     OpenCV's sequential min-distance selection in descending-strength order, computed as the
//                     lexicographically-first maximal independent set by priority rounds: a candidate is dropped once a
//                     STRONGER candidate within the distance is taken, and taken once all of those are dropped; the
//                     fixpoint equals the sequential result.  Rank-by-counting keeps the strongest nFeatures, in order.
//   neigh_kernel      (many workgroups) the stronger neighbours of every candidate -> lists; greedy_kernel (one workgroup)
//                     runs the rounds on those lists with the candidate states in LDS, then ranks
//   subpix_kernel     one workgroup of 4 waves per corner: the neighbourhood cached in LDS, 17x17 bilinear patch, one
//                     window term per thread, double sums in the oracle's canonical tree order, 2x2 solve, <= 30 iterations
//   subpix_kernel16   the throughput form (four corners per wave, one 16-lane DPP row each: batch handles); subpix_generic_kernel: any half-window 1..15 other than the stock 7
//   mineig_nms_strip_kernel: the fused pass of batch handles (one wave per strip of 60 x 16 pixels, rows walked with the state in registers)
#pragma once

#define DET_NBCAP 64          // stronger-neighbour list capacity per candidate (global memory)
#define DET_FAST_N 2048       // up to this many candidates neigh_kernel buckets them in LDS
#define DET_FAST_CELLS 4096   // ... given at most this many grid cells

struct DetDev {
    const int* first;        // Tracker's mbIsTheFirstImage (device)
    float* eig;              // W*H
    int* maxkey;             // ordered-int key of the image maximum
    int* counters;           // [0] n candidates, [1] n accepted
    int* cell_cnt;           // [cells at s=1]
    unsigned long long* cell_ent;   // bucketed candidate keys  [(W+cell)*(H+cell)]
    int* cell_ci;                   // ... and their index in `cand`
    int* nb;                        // general path: stronger neighbours within the distance, per candidate  [n_cap][DET_NBCAP]
    int* nb_cnt;                    // list length, or -1 if it overflowed (that candidate scans its cells every round)
    int n_cap;                      // candidates with a neighbour list (the rest scan)
    unsigned long long* prov;       // provisional candidates of the fused min-eigenvalue + local-maximum pass: every strict 3x3 local maximum, the
                                    // image-wide threshold not applied yet (it needs the maximum of the whole map) [W*H]; count in counters[3]
    unsigned long long* cand;       // flat candidate keys      [W*H]
    unsigned long long* acc;        // accepted keys            [W*H]
    unsigned char* state;           // general path, per candidate: 1 undecided, 2 taken, 3 dropped
    float* raw_xy;           // goodFeaturesToTrack output [F][2]
    float* xy;               // after cornerSubPix         [F][2]   (one of two buffers, by frame parity: see rvio_hip.hip)
    int* n_out;              // number of output corners (same parity)
    const float* spmask;     // (2 sp_win + 1)^2 Gaussian window of cornerSubPix (host-computed: expf is glibc's)
    int sp_win;              // cornerSubPix half-window floor(nMinDist / 2) (FeatureDetector.cc:68): 7 for the stock 15 px
    int W, H, F;
    float min_dist;          // Tracker.nMinDist
    double quality;          // (double)(float)Tracker.nQualLvl
    int max_cells;
};

// batched launches: every buffer of instance z lies z * bs bytes behind instance 0's (the cornerSubPix window is shared)
__device__ __forceinline__ void det_shift(DetDev& d, size_t off) {
    zmove(d.first, off); zmove(d.eig, off); zmove(d.maxkey, off); zmove(d.counters, off); zmove(d.cell_cnt, off); zmove(d.cell_ent, off);
    zmove(d.cell_ci, off); zmove(d.nb, off); zmove(d.nb_cnt, off); zmove(d.prov, off); zmove(d.cand, off); zmove(d.acc, off); zmove(d.state, off);
    zmove(d.raw_xy, off); zmove(d.xy, off); zmove(d.n_out, off);
}
__device__ __forceinline__ int f2ord(float f) { const int b = __float_as_int(f); return b >= 0 ? b : (b ^ 0x7fffffff); }
__device__ __forceinline__ float ord2f(int k) { return __int_as_float(k >= 0 ? k : (k ^ 0x7fffffff)); }

struct __attribute__((packed)) U32u { unsigned v; };   // a dword at any byte address
#define DET_TW 64
#define DET_TH 8
#define DET_T (DET_TW * DET_TH)
__global__ __launch_bounds__(DET_T) void mineig_kernel(const uint8_t* __restrict__ src, int stride, DetDev d, size_t src_bs, size_t bs, int dbg_tag) {
    DBG_I(blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0, dbg_tag, 1);
    src = zoff(src, src_bs); det_shift(d, (size_t)blockIdx.z * bs);
    __shared__ float sdx[DET_TH + 2][DET_TW + 2], sdy[DET_TH + 2][DET_TW + 2];
    __shared__ int s_max[DET_TH];
    const int W = d.W, H = d.H;
    const int tid = threadIdx.x, x0 = blockIdx.x * DET_TW, y0 = blockIdx.y * DET_TH;
    if (blockIdx.x == 0 && blockIdx.y == 0) {          // per-frame reset of the detector's counters (nothing in this kernel reads them)
        for (int i = tid; i < d.max_cells; i += DET_T) d.cell_cnt[i] = 0;
        if (tid < 3) d.counters[tid] = 0;
    }
    const double scale = 1.0 / (4.0 * 3.0 * 255.0);
    const float k1 = (float)scale, k0 = (float)(2.0 * scale);
    // gradients on the (TW+2) x (TH+2) halo; positions outside the image take the gradient AT the reflected position
    for (int e = tid; e < (DET_TW + 2) * (DET_TH + 2); e += DET_T) {
        const int ly = e / (DET_TW + 2), lx = e % (DET_TW + 2);
        const int gy = reflect1(y0 + ly - 1, H), gx = reflect1(x0 + lx - 1, W);
        float dxv = 0.f, dyv = 0.f;
        if (y0 + ly - 1 < H + 1 && x0 + lx - 1 < W + 1) {
            const uint8_t* r0 = src + (size_t)reflect1(gy - 1, H) * stride;
            const uint8_t* r1 = src + (size_t)gy * stride;
            const uint8_t* r2 = src + (size_t)reflect1(gy + 1, H) * stride;
            const int xl = reflect1(gx - 1, W), xr = reflect1(gx + 1, W);
            const float a00 = r0[xl], a01 = r0[gx], a02 = r0[xr], a10 = r1[xl], a12 = r1[xr], a20 = r2[xl], a21 = r2[gx], a22 = r2[xr];
            const float rr0 = a02 - a00, rr1 = a12 - a10, rr2 = a22 - a20;
            dxv = k0 * rr1 + k1 * (rr0 + rr2);
            const float q0 = k0 * a01 + k1 * (a00 + a02);
            const float q2 = k0 * a21 + k1 * (a20 + a22);
            dyv = q2 - q0;
        }
        sdx[ly][lx] = dxv; sdy[ly][lx] = dyv;
    }
    __syncthreads();
    const int lx = tid & 63, ly = tid >> 6, x = x0 + lx, y = y0 + ly;
    int key = (int)0x80000000;
    if (x < W && y < H) {
        float cov[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double col[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double v[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float gx = sdx[ly + i][lx + j], gy = sdy[ly + i][lx + j];
                    const float p = (k == 0) ? gx * gx : (k == 1 ? gx * gy : gy * gy);
                    v[i] = (double)p;
                }
                col[j] = (v[0] + v[1]) + v[2];
            }
            cov[k] = (float)((col[0] + col[1]) + col[2]);
        }
        const float a = cov[0] * 0.5f, b = cov[1], c = cov[2] * 0.5f;
        const float ev = (a + c) - sqrtf((a - c) * (a - c) + b * b);
        d.eig[(size_t)y * W + x] = ev;
        key = f2ord(ev);
    }
    // block maximum -> global
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(key, o); key = other > key ? other : key; }
    if ((tid & 63) == 0) s_max[tid >> 6] = key;
    __syncthreads();
    if (tid == 0) {
        int m = s_max[0];
        for (int k = 1; k < DET_TH; ++k) m = s_max[k] > m ? s_max[k] : m;
        atomicMax(d.maxkey, m);
    }
}

// ---------------------------------------------------------------- round 4: min-eigenvalue map + 3x3 local maxima in ONE pass
// goodFeaturesToTrack thresholds the map at quality x its MAXIMUM and keeps the strict 3x3 local maxima (cv::dilate + compare).  The two
// kernels of rounds 1-3 wrote the whole float map (1.44 MB at 752 x 480) for the second one to read back (1.07 MB after the threshold test):
// 2.5 MB of HBM / L2 round trip per frame for a result of ~2 k candidates.  The local-maximum test does not need the maximum; only the
// threshold does.  So: one workgroup computes the map on its 64 x 16 tile PLUS a one-pixel ring (the neighbours' values, recomputed —
// the same arithmetic, hence the same bits), keeps it in LDS, appends every local maximum (value != 0) to a provisional list and folds the
// tile's maximum into the image maximum; nms_threshold_kernel then applies  value > quality x maximum  to that list (a few thousand
// entries instead of the 361 k-pixel map) and fills the candidate list and the cell buckets exactly as nms_kernel did.  The map itself
// is no longer stored (batch handles of >= 8 instances run the same pass in its strip form, mineig_nms_strip_kernel);
// rvio_hip_get_corners(eig) recomputes it on demand with mineig_kernel.
// Arithmetic = mineig_kernel's, in the separable form of round 2's throughput kernel: raw pixels staged once (one byte load per pixel),
// Scharr-scaled Sobel gradients in float, the three product planes' vertical three-sums (p(i) + p(i+1)) + p(i+2) formed once per column in
// double and shared through LDS, a pixel adds three of them left to right: the canonical additions in the canonical order.
#define DET_FH 16                                   // tile rows of the fused pass
// pyr_signal (round 6, one pipelined stream): the counter klt_kernel3 polls for "this image chain's pyramid is complete" — the pyramid launch precedes this one on
// the same queue, so its stores are complete and visible device-wide when this kernel starts: workgroup 0 just bumps the counter (no launch of its own on the image chain)
__global__ __launch_bounds__(DET_T) void mineig_nms_kernel(const uint8_t* __restrict__ src, int stride, DetDev d, size_t src_bs, size_t bs, int dbg_tag, unsigned long long* pyr_signal) {
    DBG_I(blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0, dbg_tag, 1);
    if (pyr_signal && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(pyr_signal, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    src = zoff(src, src_bs); det_shift(d, (size_t)blockIdx.z * bs);
    constexpr int TH = DET_FH, TW = DET_TW;
    constexpr int EW = TW + 2, EH = TH + 2;          // map incl. the ring:    rows y0-1 .. y0+TH,   columns x0-1 .. x0+TW
    constexpr int GW = TW + 4, GH = TH + 4;          // gradients:             rows y0-2 .. y0+TH+1, columns x0-2 .. x0+TW+1
    constexpr int RW = TW + 6, RH = TH + 6;          // raw pixels:            rows y0-3 .. y0+TH+2, columns x0-3 .. x0+TW+2
    __shared__ unsigned char raw[RH][RW + 2];
    __shared__ float sdx[GH][GW], sdy[GH][GW];
    __shared__ double cs[3][EH][GW];                 // cs[k][i][c] = (p_k(i, c) + p_k(i+1, c)) + p_k(i+2, c) on gradient rows i .. i+2, gradient column c
    __shared__ float se[EH][EW];
    __shared__ int s_max[DET_T / 64];
    const int W = d.W, H = d.H;
    const int tid = threadIdx.x, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    if (blockIdx.x == 0 && blockIdx.y == 0) {        // per-frame reset of the detector's counters (nothing in this kernel reads them; [3] is the provisional count)
        for (int i = tid; i < d.max_cells; i += DET_T) d.cell_cnt[i] = 0;
        if (tid < 3) d.counters[tid] = 0;
    }
    // raw[tr][tc] = pixel (y0 - 3 + tr, x0 - 3 + tc) clamped into the image (positions the reflections below never address hold a copy of an edge pixel)
    for (int e = tid; e < RH * RW; e += DET_T) {
        const int tr = e / RW, tc = e % RW;
        const int yy = min(max(y0 - 3 + tr, 0), H - 1), xx = min(max(x0 - 3 + tc, 0), W - 1);
        raw[tr][tc] = src[(size_t)yy * stride + xx];
    }
    __syncthreads();
    const double scale = 1.0 / (4.0 * 3.0 * 255.0);
    const float k1 = (float)scale, k0 = (float)(2.0 * scale);
    // gradient at grid position (ly, lx) <-> pixel (y0 - 2 + ly, x0 - 2 + lx); outside the image: the gradient AT the reflected position
    // (cornerMinEigenVal's BORDER_DEFAULT box filter over a Sobel image computed with BORDER_DEFAULT), as mineig_kernel does
    for (int e = tid; e < GW * GH; e += DET_T) {
        const int ly = e / GW, lx = e % GW;
        const int py = y0 - 2 + ly, px = x0 - 2 + lx;
        float dxv = 0.f, dyv = 0.f;
        if (py >= -1 && py < H + 1 && px >= -1 && px < W + 1) {
            const int gy = reflect1(py, H), gx = reflect1(px, W);
            const unsigned char* r0 = raw[reflect1(gy - 1, H) - y0 + 3];
            const unsigned char* r1 = raw[gy - y0 + 3];
            const unsigned char* r2 = raw[reflect1(gy + 1, H) - y0 + 3];
            const int xl = reflect1(gx - 1, W) - x0 + 3, xc = gx - x0 + 3, xr = reflect1(gx + 1, W) - x0 + 3;
            const float a00 = r0[xl], a01 = r0[xc], a02 = r0[xr], a10 = r1[xl], a12 = r1[xr], a20 = r2[xl], a21 = r2[xc], a22 = r2[xr];
            const float rr0 = a02 - a00, rr1 = a12 - a10, rr2 = a22 - a20;
            dxv = k0 * rr1 + k1 * (rr0 + rr2);
            const float q0 = k0 * a01 + k1 * (a00 + a02);
            const float q2 = k0 * a21 + k1 * (a20 + a22);
            dyv = q2 - q0;
        }
        sdx[ly][lx] = dxv; sdy[ly][lx] = dyv;
    }
    __syncthreads();
    // vertical three-sums: item = (gradient column c, pair of map rows 2g, 2g+1) — map row i sums gradient rows i, i+1, i+2
    for (int it = tid; it < GW * (EH / 2); it += DET_T) {
        const int c = it % GW, r0 = 2 * (it / GW);
        double p[3][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gx = sdx[r0 + r][c], gy = sdy[r0 + r][c];
            const float pxx = gx * gx, pxy = gx * gy, pyy = gy * gy;
            p[0][r] = (double)pxx; p[1][r] = (double)pxy; p[2][r] = (double)pyy;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            cs[k][r0][c] = (p[k][0] + p[k][1]) + p[k][2];
            cs[k][r0 + 1][c] = (p[k][1] + p[k][2]) + p[k][3];
        }
    }
    __syncthreads();
    // the map on the tile and its ring: map position (ey, ex) <-> pixel (y0 - 1 + ey, x0 - 1 + ex); -inf outside the image (never a neighbour
    // that matters: candidates keep one pixel away from the border)
    int key = (int)0x80000000;
    for (int e = tid; e < EW * EH; e += DET_T) {
        const int ey = e / EW, ex = e % EW;
        const int y = y0 - 1 + ey, x = x0 - 1 + ex;
        float ev = -INFINITY;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            float cov[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) cov[k] = (float)((cs[k][ey][ex] + cs[k][ey][ex + 1]) + cs[k][ey][ex + 2]);
            const float a = cov[0] * 0.5f, b = cov[1], c = cov[2] * 0.5f;
            ev = (a + c) - sqrtf((a - c) * (a - c) + b * b);
            if (ex >= 1 && ex <= TW && ey >= 1 && ey <= TH) { const int k2 = f2ord(ev); key = k2 > key ? k2 : key; }   // the tile's own pixels feed the image maximum
        }
        se[ey][ex] = ev;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(key, o); key = other > key ? other : key; }
    if ((tid & 63) == 0) s_max[tid >> 6] = key;
    __syncthreads();
    if (tid == 0) {
        int m = s_max[0];
        for (int k = 1; k < DET_T / 64; ++k) m = s_max[k] > m ? s_max[k] : m;
        atomicMax(d.maxkey, m);
    }
    // strict 3x3 local maxima of the tile's pixels, one pixel away from the image border (goodFeaturesToTrack: dilate, compare, skip the border).
    // Collected in LDS first: ONE global atomic per workgroup reserves the tile's range of the provisional list (an atomic per local maximum —
    // ~15 k per image on one counter — cost the image chain 40 us)
    __shared__ unsigned long long s_list[TW * TH];
    __shared__ int s_n, s_base;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int lx = tid & 63, x = x0 + lx;
#pragma unroll
    for (int rr = 0; rr < TH / (DET_T / 64); ++rr) {
        const int ly = (tid >> 6) + (DET_T / 64) * rr, y = y0 + ly;
        if (x < 1 || y < 1 || x >= W - 1 || y >= H - 1) continue;
        const float v = se[ly + 1][lx + 1];
        if (v == 0.f) continue;
        float m = v;
        m = fmaxf(m, se[ly][lx]);     m = fmaxf(m, se[ly][lx + 1]);     m = fmaxf(m, se[ly][lx + 2]);
        m = fmaxf(m, se[ly + 1][lx]);                                   m = fmaxf(m, se[ly + 1][lx + 2]);
        m = fmaxf(m, se[ly + 2][lx]); m = fmaxf(m, se[ly + 2][lx + 1]); m = fmaxf(m, se[ly + 2][lx + 2]);
        if (v != m) continue;
        const int idx = y * W + x;
        s_list[atomicAdd(&s_n, 1)] = ((unsigned long long)(unsigned)__float_as_int(v) << 32) | (unsigned)idx;
    }
    __syncthreads();
    const int cnt = s_n;
    if (tid == 0 && cnt > 0) s_base = atomicAdd(&d.counters[3], cnt);
    __syncthreads();
    for (int i = tid; i < cnt; i += DET_T) d.prov[s_base + i] = s_list[i];
}

// ---------------------------------------------------------------- round 4, throughput form of the fused pass (batched launches)
// One WAVE per strip of DET_SW columns x DET_SH rows, lane = image column, rows walked top to bottom with everything a row needs from the
// rows above it kept in registers:
//   raw row r      -> per column  rr = right - left,  q = k0 * centre + k1 * (left + right)            (3 byte reads from the staged tile)
//   gradient row g = r - 1:   dx = k0 * rr(g) + k1 * (rr(g-1) + rr(g+1)),  dy = q(g+1) - q(g-1)  ->  the three products, in double
//   column sums  c = g - 1:   cs_k = (p_k(c-1) + p_k(c)) + p_k(c+1)                                     (the canonical vertical three-sums)
//   map row c:                cov_k = (cs_k[x-1] + cs_k[x]) + cs_k[x+1]  — the neighbours' sums through LDS (one ds_read2 per plane) — eigenvalue
//   local maxima  n = c - 1:  the 3 x 3 test on map rows n-1, n, n+1 (left / right through LDS), appended to the wave's list by ballot
// i.e. mineig_kernel's additions in mineig_kernel's order (hence the same bits) at about half the instructions of the tiled form:
// no gradient, product or sum is computed twice (the tiled kernels recompute the vertical sums' inputs for every pair of rows and re-read
// nine column sums per pixel), nothing but the raw tile, one row of column sums and one row of the map goes through LDS, and the map never
// goes to HBM (mineig_kernel4 wrote 183 MB per 128 images, nms_kernel4 read 107 MB back).  Borders as cornerMinEigenVal does them
// (BORDER_REFLECT_101 on the pixels for the Sobel rows / columns, and on the GRADIENT products for the box sums): row -1 of anything is
// row 1, row H is row H - 2, column -1 is column 1, column W is column W - 2 — two virtual steps behind the last image row flush the
// pipeline.  A wave keeps two columns of halo on either side (its lanes 0, 1, 62, 63 only feed their neighbours).
// Strip height and list size are measured choices (profiles/r05_strip_ab.md, 128 batched streams, shipping-flag builds back to back on one box):
// the LDS list of a wave's maxima at 128 instead of 256 entries (1 KB instead of 2 KB of the workgroup's LDS: more strips resident per CU)
// 144.0 -> 146.4 k frames/s at every strip height; taller strips (fewer halo rows per output row) are SLOWER — 64 rows 140.5 k, 96 rows
// 138.6 k against 144.0 k at 32: the row walk is a dependent chain per wave, the chip wants more waves, not fewer instructions per wave.
// The loop is not unrolled (the compiler declines: #pragma unroll 2 gives the same code).
#ifndef DET_SH
#define DET_SH 16                       // rows a wave walks (6 rows of halo per strip)
#endif
#ifndef DET_STRIP_UNROLL
#define DET_STRIP_UNROLL 1
#endif
#define DET_STR2(x) #x
#define DET_STR(x) DET_STR2(x)
#define DET_SW 60
#ifndef DET_SL
#define DET_SL 128                      // the wave's list of local maxima in LDS: flushed to the provisional list when a row might not fit
#endif
__global__ __launch_bounds__(64) void mineig_nms_strip_kernel(const uint8_t* __restrict__ src, int stride, DetDev d, size_t src_bs, size_t bs) {
    src = zoff(src, src_bs); det_shift(d, (size_t)blockIdx.z * bs);
    constexpr int TR = DET_SH + 6, TWB = 68;                          // raw tile: rows ys-3 .. ys+SH+2, columns X0-3 .. X0+64 (17 dwords)
    __shared__ __align__(16) unsigned char raw[TR * TWB];
    __shared__ double xcs[2][3][66];                                  // lane L's column sums at [L + 1], by row parity
    __shared__ float xe[2][66];                                       // ... and its map value
    __shared__ unsigned long long s_list[DET_SL];
    const int W = d.W, H = d.H;
    const int lane = threadIdx.x;
    const int X0 = blockIdx.x * DET_SW, ys = blockIdx.y * DET_SH, ye = min(ys + DET_SH, H);
    const int x = X0 - 2 + lane;
    if (blockIdx.x == 0 && blockIdx.y == 0) {        // per-frame reset of the detector's counters (nothing in this kernel reads them; [3] is the provisional count)
        for (int i = lane; i < d.max_cells; i += 64) d.cell_cnt[i] = 0;
        if (lane < 3) d.counters[lane] = 0;
    }
    // ---- the raw tile (clamped coordinates: positions the reflections below never address hold a copy of an edge pixel)
    const int ty0 = ys - 3, tx0 = X0 - 3;
    if (tx0 >= 0 && tx0 + TWB <= W) {
        for (int e = lane; e < TR * (TWB / 4); e += 64) {
            const int tr = e / (TWB / 4), q = e % (TWB / 4);
            const int yy = min(max(ty0 + tr, 0), H - 1);
            ((unsigned*)raw)[e] = ((const U32u*)(src + (size_t)yy * stride + tx0))[q].v;
        }
    } else {
        for (int e = lane; e < TR * TWB; e += 64) {
            const int tr = e / TWB, tc = e % TWB;
            const int yy = min(max(ty0 + tr, 0), H - 1), xx = min(max(tx0 + tc, 0), W - 1);
            raw[e] = src[(size_t)yy * stride + xx];
        }
    }
    __syncthreads();
    const double scale = 1.0 / (4.0 * 3.0 * 255.0);
    const float k1 = (float)scale, k0 = (float)(2.0 * scale);
    const bool edge = X0 - 2 <= 0 || X0 - 2 + 63 >= W - 1;           // the wave holds image column 0 or W - 1 (wave-uniform)
    const bool isL = x == 0, isR = x == W - 1;
    const bool own = lane >= 2 && lane < 2 + DET_SW && x < W;        // (x >= 0 for these lanes)
    const bool nmsx = own && x >= 1 && x < W - 1;
    // sliding state: rr / q of raw rows (r-2, r-1, r); products of gradient rows (g-2, g-1, g); map rows (c-2, c-1, c) with their row maxima
    float rrA = 0, rrB = 0, rrC = 0, qA = 0, qB = 0, qC = 0;
    double pA[3] = {0, 0, 0}, pB[3] = {0, 0, 0}, pC[3] = {0, 0, 0};
    float eA = 0, eB = 0, eC = 0, lB = 0, rB = 0, mA = 0, mB = 0, mC = 0;    // l / r: the row's left / right neighbours; m: max of the row's three
    int key = (int)0x80000000;
    int n_list = 0;
    auto flush = [&]() {                   // ONE global atomic reserves the range (wave-uniform control flow)
        int base = 0;
        if (lane == 0) base = atomicAdd(&d.counters[3], n_list);
        base = __builtin_amdgcn_readfirstlane(base);
        // s_list was filled through ballot slots by other lanes of this wave: in-order LDS of one wave + these fences (no reliance on alias analysis)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < n_list; i += 64) d.prov[base + i] = s_list[i];
        n_list = 0;
    };
    const int r_lo = max(ys - 3, 0), r_hi = min(ye + 2, H + 1);
    const int g_min = (ys - 3 <= 0) ? 0 : r_lo + 1;                    // first gradient row this strip forms
    const int c_min = (g_min == 0) ? 0 : g_min + 1;                    // first row of column sums / of the map
_Pragma(DET_STR(unroll DET_STRIP_UNROLL))
    for (int r = r_lo; r <= r_hi; ++r) {
        // ---- stage A: raw row r -> rr, q; gradient row g = r - 1 -> products
        bool have_p = false;
        if (r < H) {
            const unsigned char* t = raw + (r - ty0) * TWB + lane;     // tile column lane <-> image column x - 1
            float a0 = (float)t[0], a1 = (float)t[1], a2 = (float)t[2];
            if (edge) { const float l0 = a0; if (isL) a0 = a2; if (isR) a2 = l0; }
            rrA = rrB; rrB = rrC; qA = qB; qB = qC;
            rrC = a2 - a0;
            qC = k0 * a1 + k1 * (a0 + a2);
        }
        const int g = r - 1;
        if (g >= g_min && g <= H - 1) {
            float rrU, rrM, rrD, qU, qD;
            if (r < H) { rrU = (g == 0) ? rrC : rrA; rrM = rrB; rrD = rrC; qU = (g == 0) ? qC : qA; qD = qC; }
            else { rrU = rrB; rrM = rrC; rrD = rrB; qU = qB; qD = qB; }      // g = H - 1: the row below is row H - 2
            const float gx = k0 * rrM + k1 * (rrU + rrD);
            const float gy = qD - qU;
            const float pxx = gx * gx, pxy = gx * gy, pyy = gy * gy;
#pragma unroll
            for (int k = 0; k < 3; ++k) { pA[k] = pB[k]; pB[k] = pC[k]; }
            pC[0] = (double)pxx; pC[1] = (double)pxy; pC[2] = (double)pyy;
            have_p = true;
        }
        // ---- stage B: column sums of row c = g - 1 (c = H - 1: one step behind the last gradient row, nothing new above)
        const int c = g - 1;
        if (!(c >= c_min && c <= H - 1 && (have_p || c == H - 1))) continue;
        double cs[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (have_p) cs[k] = (((c == 0) ? pC[k] : pA[k]) + pB[k]) + pC[k];
            else cs[k] = (pB[k] + pC[k]) + pB[k];                      // c = H - 1: the row below is row H - 2
        }
        // ---- stage C: the neighbours' sums through LDS, covariation, eigenvalue
        // (a lane's own slot and its neighbours' are different addresses: without the barrier the compiler moves the loads above the stores.
        //  One wave per workgroup: the barrier costs a wait for the LDS queue.  Two buffers: the next row's stores cannot overtake these loads.)
        double (*xc)[66] = xcs[c & 1];
        float* xr = xe[c & 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) xc[k][lane + 1] = cs[k];
        __syncthreads();
        float cov[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double cl = xc[k][lane], cr = xc[k][lane + 2];
            if (edge) { const double l0 = cl; if (isL) cl = cr; if (isR) cr = l0; }
            cov[k] = (float)((cl + cs[k]) + cr);
        }
        const float a = cov[0] * 0.5f, b = cov[1], cc = cov[2] * 0.5f;
        const float ev = (a + cc) - sqrtf((a - cc) * (a - cc) + b * b);
        if (own && c >= ys && c < ye) { const int k2 = f2ord(ev); key = k2 > key ? k2 : key; }     // the strip's own pixels feed the image maximum
        // ---- stage D: the 3 x 3 test on map row n = c - 1
        xr[lane + 1] = ev;
        __syncthreads();
        const float el = xr[lane], er = xr[lane + 2];
        eA = eB; eB = eC; eC = ev;
        const float lOld = lB, rOld = rB;
        mA = mB; mB = mC;
        lB = el; rB = er;                                              // (row c's neighbours: row n's next step)
        mC = fmaxf(fmaxf(el, ev), er);
        const int n = c - 1;
        if (n >= max(ys, 1) && n <= min(ye - 1, H - 2)) {
            // row n = the middle row: eB is its value, lOld / rOld its neighbours (saved one step ago), mA / mC the rows above / below
            const float v = eB;
            float m = fmaxf(v, mA);
            m = fmaxf(m, lOld); m = fmaxf(m, rOld);
            m = fmaxf(m, mC);
            const bool hit = nmsx && v != 0.f && v == m;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
            if (hit) s_list[n_list + __popcll(bal & ((1ull << lane) - 1ull))] = ((unsigned long long)(unsigned)__float_as_int(v) << 32) | (unsigned)(n * W + x);
            n_list += __popcll(bal);
            if (n_list + 64 > DET_SL) flush();                        // (a map of equal values makes every pixel a "maximum")
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(key, o); key = other > key ? other : key; }
    if (lane == 0) atomicMax(d.maxkey, key);
    if (n_list > 0) flush();
}

__device__ __forceinline__ void det_geometry(const DetDev& d, float* md, int* cell, int* gw, int* gh) {
    const float s = (*d.first) ? 1.f : 2.f;
    *md = s * d.min_dist;                                    // s*mnMinDistance (int * float)
    *cell = (int)rintf(*md);                                 // cvRound(minDistance)
    *gw = (d.W + *cell - 1) / *cell; *gh = (d.H + *cell - 1) / *cell;
}

// The image-wide threshold on the provisional list (value > quality x maximum: goodFeaturesToTrack's cv::threshold THRESH_TOZERO):
// the survivors are the candidates; list and cell buckets as nms_kernel filled them (the order inside `cand` is an atomic append, as it was).
#define NMS_T 256
__global__ __launch_bounds__(NMS_T) void nms_threshold_kernel(DetDev d, size_t bs) {
    det_shift(d, (size_t)blockIdx.z * bs);
    const int W = d.W;
    const int np = d.counters[3];
    const float mx = ord2f(*d.maxkey);
    const float thr = (float)((double)mx * d.quality);
    float md; int cell, gw, gh;
    det_geometry(d, &md, &cell, &gw, &gh);
    for (int i = blockIdx.x * NMS_T + threadIdx.x; i < np; i += gridDim.x * NMS_T) {
        const unsigned long long key = d.prov[i];
        const float v = __int_as_float((int)(key >> 32));
        if (!(v > thr)) continue;
        const int idx = (int)(key & 0xffffffffull), x = idx % W, y = idx / W;
        const int ci = atomicAdd(&d.counters[0], 1);
        d.cand[ci] = key;
        const int c = (y / cell) * gw + (x / cell);
        const size_t slot = (size_t)c * cell * cell + atomicAdd(&d.cell_cnt[c], 1);
        d.cell_ent[slot] = key;
        d.cell_ci[slot] = ci;
    }
}

// The two-pass form of rounds 1-3 (the map through HBM), kept for A/B timing in the instrumented build and for rvio_hip_get_corners(eig)
#ifdef RVIO_DBG_CLOCKS   // (the two-pass form of rounds 1-3, RVIO_DET_TWO_PASS: instrumented build only)
__global__ __launch_bounds__(DET_T) void nms_kernel(DetDev d, size_t bs) {
    det_shift(d, (size_t)blockIdx.z * bs);
    const int W = d.W, H = d.H;
    const int x = blockIdx.x * DET_TW + (threadIdx.x & 63), y = blockIdx.y * DET_TH + (threadIdx.x >> 6);
    if (x < 1 || y < 1 || x >= W - 1 || y >= H - 1) return;
    const float mx = ord2f(*d.maxkey);
    const float thr = (float)((double)mx * d.quality);
    const float* e = d.eig + (size_t)y * W + x;
    const float v = e[0];
    if (!(v > thr) || v == 0.f) return;
    float m = v;
    m = fmaxf(m, e[-W - 1]); m = fmaxf(m, e[-W]); m = fmaxf(m, e[-W + 1]);
    m = fmaxf(m, e[-1]);     m = fmaxf(m, e[1]);
    m = fmaxf(m, e[W - 1]);  m = fmaxf(m, e[W]);  m = fmaxf(m, e[W + 1]);
    if (v != m) return;
    float md; int cell, gw, gh;
    det_geometry(d, &md, &cell, &gw, &gh);
    const int idx = y * W + x;
    const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(v) << 32) | (unsigned)idx;   // v > 0: bits are ordered
    const int ci = atomicAdd(&d.counters[0], 1);
    d.cand[ci] = key;
    const int c = (y / cell) * gw + (x / cell);
    const size_t slot = (size_t)c * cell * cell + atomicAdd(&d.cell_cnt[c], 1);
    d.cell_ent[slot] = key;
    d.cell_ci[slot] = ci;
}
#endif

#define NEIGH_T 1024
#define NEIGH_BLOCKS 8
#define NEIGH_BLOCKS_WIDE 2     // batch handles (measured at 128 streams: 8 -> 134.5 k frames/s, 4 -> 135.3, 2 -> 136.0, 1 -> 135.9)
// LDS of neigh_kernel's bucket build: keys 8 B, packed xy 4, cell 2, slot 2, order 2 per candidate + cell starts
#define NEIGH_LDS (DET_FAST_N * (8 + 4 + 2 + 2 + 2) + (DET_FAST_CELLS + 2) * 4)

// Every candidate collects its STRONGER neighbours within the distance — only those decide its fate: it is dropped when
// one of them is taken, taken when all of them are dropped.  OpenCV searches the 3x3 grid cells around the candidate.
// Up to DET_FAST_N candidates every workgroup rebuilds compact cell buckets in its LDS (a few microseconds) and walks
// them; denser candidate sets walk the global buckets nms_kernel filled.  Lists go to d.nb (global, L2 resident).
__global__ __launch_bounds__(NEIGH_T) void neigh_kernel(DetDev d, size_t bs) {
    det_shift(d, (size_t)blockIdx.z * bs);
    extern __shared__ __align__(16) unsigned char ndyn[];
    __shared__ int s_w[16];
    const int W = d.W, tid = threadIdx.x;
    float md; int cell, gw, gh;
    det_geometry(d, &md, &cell, &gw, &gh);
    const double md2 = (double)md * (double)md;
    const int n = d.counters[0], ncell = gw * gh;
    if (n <= DET_FAST_N && ncell <= DET_FAST_CELLS) {
        unsigned long long* fkey = (unsigned long long*)ndyn;                           // [N]
        unsigned int* fxy = (unsigned int*)(fkey + DET_FAST_N);                          // [N] x | y << 16
        int* cstart = (int*)(fxy + DET_FAST_N);                                          // [CELLS + 2]
        unsigned short* fcell = (unsigned short*)(cstart + DET_FAST_CELLS + 2);          // [N]
        unsigned short* fslot = fcell + DET_FAST_N;                                      // [N]
        unsigned short* order = fslot + DET_FAST_N;                                      // [N]
        for (int c = tid; c < ncell; c += NEIGH_T) cstart[c] = 0;
        __syncthreads();
        for (int c = tid; c < n; c += NEIGH_T) {
            const unsigned long long key = d.cand[c];
            const int idx = (int)(key & 0xffffffffull), x = idx % W, y = idx / W;
            const int cc = (y / cell) * gw + (x / cell);
            fkey[c] = key; fxy[c] = (unsigned)x | ((unsigned)y << 16); fcell[c] = (unsigned short)cc;
            fslot[c] = (unsigned short)atomicAdd(&cstart[cc], 1);
        }
        __syncthreads();
        {   // exclusive scan of the bucket counts (`per` consecutive cells per thread, wave scans, 16 wave totals)
            const int per = (ncell + NEIGH_T - 1) / NEIGH_T, b0 = tid * per, lane = tid & 63, wv = tid >> 6;
            int sum = 0;
            for (int k = 0; k < per; ++k) if (b0 + k < ncell) sum += cstart[b0 + k];
            const int inc = wave_incl_scan(sum);
            if (lane == 63) s_w[wv] = inc;
            __syncthreads();
            int run = inc - sum;
            for (int w = 0; w < wv; ++w) run += s_w[w];
            for (int k = 0; k < per; ++k) if (b0 + k < ncell) { const int c = cstart[b0 + k]; cstart[b0 + k] = run; run += c; }
            if (tid == NEIGH_T - 1) cstart[ncell] = run;
        }
        __syncthreads();
        for (int c = tid; c < n; c += NEIGH_T) order[cstart[fcell[c]] + fslot[c]] = (unsigned short)c;
        __syncthreads();
        for (int c = blockIdx.x * NEIGH_T + tid; c < n; c += gridDim.x * NEIGH_T) {
            const unsigned long long key = fkey[c];
            const int x = (int)(fxy[c] & 0xffffu), y = (int)(fxy[c] >> 16), xc = x / cell, yc = y / cell;
            const int y1 = yc > 0 ? yc - 1 : 0, y2 = yc + 1 < gh ? yc + 1 : gh - 1, x1 = xc > 0 ? xc - 1 : 0, x2 = xc + 1 < gw ? xc + 1 : gw - 1;
            int m = 0;
            int* list = d.nb + (size_t)c * DET_NBCAP;
            for (int yy = y1; yy <= y2; ++yy) {
                const int e0 = cstart[yy * gw + x1], e1 = cstart[yy * gw + x2 + 1];      // the cells of a grid row are contiguous
                for (int e = e0; e < e1; ++e) {
                    const int j = order[e];
                    if (!(fkey[j] > key)) continue;
                    const float ddx = (float)x - (float)(fxy[j] & 0xffffu), ddy = (float)y - (float)(fxy[j] >> 16);
                    if ((double)(ddx * ddx + ddy * ddy) < md2) { if (m < DET_NBCAP) list[m] = j; ++m; }
                }
            }
            d.nb_cnt[c] = m <= DET_NBCAP ? m : -1;
        }
        return;
    }
    const size_t cap = (size_t)cell * cell;
    for (int c = blockIdx.x * NEIGH_T + tid; c < n && c < d.n_cap; c += gridDim.x * NEIGH_T) {
        const unsigned long long key = d.cand[c];
        const int idx = (int)(key & 0xffffffffull);
        const int x = idx % W, y = idx / W, xc = x / cell, yc = y / cell;
        int cnt9[9], cid9[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int xx = xc + (q % 3) - 1, yy = yc + (q / 3) - 1;
            const bool in = xx >= 0 && yy >= 0 && xx < gw && yy < gh;
            cid9[q] = in ? yy * gw + xx : 0;
            cnt9[q] = in ? d.cell_cnt[cid9[q]] : 0;
        }
        int m = 0;
        int* list = d.nb + (size_t)c * DET_NBCAP;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const unsigned long long* ent = d.cell_ent + (size_t)cid9[q] * cap;
            const int* eci = d.cell_ci + (size_t)cid9[q] * cap;
            for (int e = 0; e < cnt9[q]; ++e) {
                const unsigned long long k2 = ent[e];
                if (!(k2 > key)) continue;
                const int i2 = (int)(k2 & 0xffffffffull);
                const float ddx = (float)x - (float)(i2 % W), ddy = (float)y - (float)(i2 / W);
                if ((double)(ddx * ddx + ddy * ddy) < md2) { if (m < DET_NBCAP) list[m] = eci[e]; ++m; }
            }
        }
        d.nb_cnt[c] = m <= DET_NBCAP ? m : -1;
    }
}

#define GREEDY_T 1024
#define DET_LDS_ST 32768      // candidate states kept in LDS (the rest in global memory)
// One candidate of one round on the general path; returns drop | wait << 1.  m >= 0: neighbour list in global memory;
// m < 0: list overflow / beyond n_cap -> scan the 3x3 cells of the global buckets.
__device__ __noinline__ int greedy_general_step(const DetDev& d, int c, int m, unsigned char* lst, int cell, int gw, int gh, double md2) {
    volatile unsigned char* ls = lst;
    volatile unsigned char* gs = d.state;
    const int W = d.W;
    bool drop = false, wait = false;
    auto st_get = [&](int ci) -> unsigned char { return ci < DET_LDS_ST ? ls[ci] : gs[ci]; };
    if (m >= 0) {
        const int* list = d.nb + (size_t)c * DET_NBCAP;
        for (int e = 0; e < m; ++e) { const unsigned char s2 = st_get(list[e]); drop |= (s2 == 2); wait |= (s2 == 1); }
    } else {
        const size_t cap = (size_t)cell * cell;
        const unsigned long long key = d.cand[c];
        const int idx = (int)(key & 0xffffffffull);
        const int x = idx % W, y = idx / W, xc = x / cell, yc = y / cell;
        const int x1 = xc > 0 ? xc - 1 : 0, y1 = yc > 0 ? yc - 1 : 0, x2 = xc + 1 < gw ? xc + 1 : gw - 1, y2 = yc + 1 < gh ? yc + 1 : gh - 1;
        for (int yy = y1; yy <= y2; ++yy)
            for (int xx = x1; xx <= x2; ++xx) {
                const int cc = yy * gw + xx, cnt = d.cell_cnt[cc];
                const unsigned long long* ent = d.cell_ent + (size_t)cc * cap;
                const int* eci = d.cell_ci + (size_t)cc * cap;
                for (int e = 0; e < cnt; ++e) {
                    const unsigned long long k2 = ent[e];
                    if (!(k2 > key)) continue;
                    const int i2 = (int)(k2 & 0xffffffffull);
                    const float ddx = (float)x - (float)(i2 % W), ddy = (float)y - (float)(i2 / W);
                    if ((double)(ddx * ddx + ddy * ddy) < md2) { const unsigned char s2 = st_get(eci[e]); drop |= (s2 == 2); wait |= (s2 == 1); }
                }
            }
    }
    return (drop ? 1 : 0) | (wait ? 2 : 0);
}
#define DET_LDS_TK 4096       // taken keys ranked from LDS
// Priority rounds over the neighbour lists until every candidate is decided, then rank-by-counting: one workgroup.
#define DET_CL_N 2048         // up to this many candidates the neighbour lists are packed into LDS as well
#define DET_CL_CAP 24576      // ... if they hold at most this many entries in total
#define GREEDY_LDS (DET_LDS_TK * 8 + DET_LDS_ST + DET_CL_CAP * 2 + (DET_CL_N + 4) * 4)
__global__ __launch_bounds__(GREEDY_T) void greedy_kernel(DetDev d, size_t bs) {
    det_shift(d, (size_t)blockIdx.z * bs);
    extern __shared__ __align__(16) unsigned char gdyn[];
    unsigned long long* tk = (unsigned long long*)gdyn;                         // [DET_LDS_TK]
    unsigned char* lst = gdyn + DET_LDS_TK * 8;                                 // [DET_LDS_ST]
    unsigned short* clist = (unsigned short*)(lst + DET_LDS_ST);                // [DET_CL_CAP]
    int* coff = (int*)(clist + DET_CL_CAP);                                     // [DET_CL_N + 1]
    __shared__ int s_w[16];
    __shared__ int s_flag;
    const int W = d.W, tid = threadIdx.x;
    float md; int cell, gw, gh;
    det_geometry(d, &md, &cell, &gw, &gh);
    const double md2 = (double)md * (double)md;
    const int n = d.counters[0];
    volatile unsigned char* ls = lst;
    volatile unsigned char* gs = d.state;
#define ST_GET(ci) ((ci) < DET_LDS_ST ? ls[(ci)] : gs[(ci)])
#define ST_SET(ci, v) do { if ((ci) < DET_LDS_ST) ls[(ci)] = (v); else gs[(ci)] = (v); } while (0)
    DBG_T(56);
    for (int c = tid; c < n; c += GREEDY_T) ST_SET(c, 1);
    // usual case: pack the (short) neighbour lists into LDS so that the rounds never leave the CU
    bool packed = n <= DET_CL_N && n <= d.n_cap;
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (packed) {
        const int c0 = 2 * tid, c1 = 2 * tid + 1, lane = tid & 63, wv = tid >> 6;
        const int m0 = c0 < n ? d.nb_cnt[c0] : 0, m1 = c1 < n ? d.nb_cnt[c1] : 0;
        if (m0 < 0 || m1 < 0) s_flag = 1;                                       // an overflowed list: general path
        const int sum = (m0 > 0 ? m0 : 0) + (m1 > 0 ? m1 : 0);
        const int inc = wave_incl_scan(sum);
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        int base = inc - sum;
        for (int w = 0; w < wv; ++w) base += s_w[w];
        if (tid == GREEDY_T - 1 && base + sum > DET_CL_CAP) s_flag = 1;
        __syncthreads();
        packed = (s_flag == 0);
        if (packed) {
            // 16 entries per trip, fetched as four 16-byte loads in flight together (a row holds DET_NBCAP = 64 ints, so reading
            // past the list's end stays inside the row)
            auto copy_list = [&](int c, int dst, int m) {
                const int4* l4 = (const int4*)(d.nb + (size_t)c * DET_NBCAP);
                for (int q = 0; q < m; q += 16) {
                    const int4 v0 = l4[q / 4], v1 = l4[q / 4 + 1], v2 = l4[q / 4 + 2], v3 = l4[q / 4 + 3];
                    const int v[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
                    for (int u = 0; u < 16; ++u) if (q + u < m) clist[dst + q + u] = (unsigned short)v[u];
                }
            };
            if (c0 < n) { coff[c0] = base; copy_list(c0, base, m0); }
            if (c1 < n) { coff[c1] = base + m0; copy_list(c1, base + m0, m1); }
            if (c0 == n - 1) coff[n] = base + m0;
            if (c1 == n - 1) coff[n] = base + m0 + m1;
        }
    }
    __threadfence_block();
    __syncthreads();
    DBG_T(57);
    int pending, rounds = 0;
    do {
        pending = 0; ++rounds;
        if (rounds == 1) DBG_T(10);
        for (int c = tid; c < n; c += GREEDY_T) {
            if (ST_GET(c) != 1) continue;
            bool drop = false, wait = false;
            const int m = packed ? -2 : (c < d.n_cap ? d.nb_cnt[c] : -1);
            if (m == -2) {
                const int o = coff[c], o1 = coff[c + 1];
                for (int e = o; e < o1 && !drop; e += 8) {       // 8 neighbours per trip: indices first, then their states
                    // unconditional loads (indices clamped to the list's last entry: re-reading it changes nothing), so that
                    // the eight index reads and then the eight state reads are each in flight together.  Plain loads: a
                    // stale state only delays a decision, and the barrier of the round forces a re-read.
                    int id8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) id8[u] = (int)clist[min(e + u, o1 - 1)];
                    unsigned char s8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s8[u] = lst[id8[u]];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { drop |= (s8[u] == 2); wait |= (s8[u] == 1); }
                }
            } else {   // dense candidate sets: lists / cell buckets in global memory (out of line: keeps the usual path compact)
                const int dw = greedy_general_step(d, c, m, lst, cell, gw, gh, md2);
                drop = (dw & 1) != 0; wait = (dw & 2) != 0;
            }
            if (rounds == 1) DBG_T(11);
            if (drop) ST_SET(c, 3);
            else if (!wait) ST_SET(c, 2);
            else pending = 1;
        }
        if (rounds == 1) DBG_T(12);
        __threadfence_block();
        pending = __syncthreads_or(pending);
        if (rounds == 1) DBG_T(13);
        if (rounds == 2) DBG_T(14);
    } while (pending);
    DBG_T(59);
#ifdef RVIO_DBG_CLOCKS
    if (tid == 0) { g_dbg[62] = n; g_dbg[63] = rounds; }
#endif
    // taken candidates -> list; rank by counting; the strongest F leave in descending order
    if (tid == 0) s_flag = 0;
    __syncthreads();
    const bool small = n <= DET_LDS_TK;          // then the taken keys go straight to LDS
    for (int c = tid; c < n; c += GREEDY_T)
        if (ST_GET(c) == 2) {
            const unsigned long long key = d.cand[c];
            if (small) tk[atomicAdd(&s_flag, 1)] = key;
            else d.acc[atomicAdd(&d.counters[1], 1)] = key;
        }
#undef ST_GET
#undef ST_SET
    __threadfence_block();
    __syncthreads();
    const int na = small ? s_flag : ((volatile int*)d.counters)[1];
    const bool in_lds = small || na <= DET_LDS_TK;
    if (!small && in_lds) for (int a = tid; a < na; a += GREEDY_T) tk[a] = d.acc[a];
    __syncthreads();
    for (int a = tid; a < na; a += GREEDY_T) {
        const unsigned long long key = in_lds ? tk[a] : d.acc[a];
        int r = 0;
        if (in_lds) for (int b = 0; b < na; ++b) r += (tk[b] > key) ? 1 : 0;
        else for (int b = 0; b < na; ++b) r += (d.acc[b] > key) ? 1 : 0;
        if (r < d.F) {
            const int idx = (int)(key & 0xffffffffull);
            d.raw_xy[2 * r] = (float)(idx % W); d.raw_xy[2 * r + 1] = (float)(idx / W);
        }
    }
    DBG_T(60);
    if (tid == 0) {
        *d.n_out = na < d.F ? na : d.F;
        *d.maxkey = (int)0x80000000;                       // consumed by the threshold pass; ready for the next image
        d.counters[3] = 0;                                 // ... and so is the provisional list
    }
}

#define SP_WIN 7
#define SP_WW (2 * SP_WIN + 1)
#define SP_PW (SP_WW + 2)
#define SP_MARG 12
#define SP_RS (SP_PW + 1 + 2 * SP_MARG)
#define SP_T 256
// subpix_kernel16's LDS row stride in bytes: 11 dwords, so the 16 window rows of a DPP row of lanes start in 16 different banks (43 bytes =
// 10.75 dwords put every third row into the same one) and a row is 11 whole dwords for the loads
#define SP_LS 44
// one workgroup of 4 waves per corner; thread t <-> window term (i, j) = (t / 16, t % 16) (row / column 15 are padding)
__global__ __launch_bounds__(SP_T) void subpix_kernel(const uint8_t* __restrict__ src, int stride, DetDev d, size_t src_bs, size_t bs, int dbg_tag) {
    DBG_I(blockIdx.x == 0 && blockIdx.z == 0, dbg_tag, 2);
    src = zoff(src, src_bs); det_shift(d, (size_t)blockIdx.z * bs);
    __shared__ unsigned char reg[SP_RS * SP_RS];
    __shared__ double s_part[2][4][5];        // by iteration parity: one barrier per iteration
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = *d.n_out;
    if (p >= n) return;
    const int W = d.W, H = d.H;
    const float tx = d.raw_xy[2 * p], ty = d.raw_xy[2 * p + 1];
    // the estimate may wander SP_MARG px from the start before a sample has to come from global memory again:
    // cache that neighbourhood (coordinates clamped at load time = the replicated border)
    const int rx0 = (int)tx - (SP_PW - 1) / 2 - SP_MARG, ry0 = (int)ty - (SP_PW - 1) / 2 - SP_MARG;
    for (int e = tid; e < SP_RS * SP_RS; e += SP_T) {
        const int j = e / SP_RS, i = e % SP_RS;
        reg[e] = src[(size_t)min(max(ry0 + j, 0), H - 1) * stride + min(max(rx0 + i, 0), W - 1)];
    }
    __syncthreads();
    auto pix = [&](int x, int y) -> float {
        const int i = x - rx0, j = y - ry0;
        if ((unsigned)i < (unsigned)SP_RS && (unsigned)j < (unsigned)SP_RS) return (float)reg[j * SP_RS + i];
        return (float)src[(size_t)min(max(y, 0), H - 1) * stride + min(max(x, 0), W - 1)];
    };
    const int wi = tid >> 4, wj = tid & 15;
    const bool live = wi < SP_WW && wj < SP_WW;
    const double wm = live ? (double)d.spmask[wi * SP_WW + wj] : 0.0, px = wj - SP_WIN, py = wi - SP_WIN;
    float cx = tx, cy = ty;
    const double eps = 1e-2 * 1e-2;
    int iter = 0;
    double err = 0;
    do {
        if (iter == 3) DBG_T(41);
        // getRectSubPix: element (pi, pj) of the 17x17 bilinear patch (replicated border).  Every thread evaluates the four
        // patch elements its window term needs itself (same expression as a stored patch would hold): no patch in LDS, no barrier.
        const float ox = cx - (float)(SP_PW - 1) * 0.5f, oy = cy - (float)(SP_PW - 1) * 0.5f;
        const int ix = (int)floorf(ox), iy = (int)floorf(oy);
        float fa = ox - (float)ix;
        const float fb = oy - (float)iy;
        fa = fmaxf(fa, 0.0001f);
        const float a11 = (1.f - fa) * (1.f - fb), a12 = fa * (1.f - fb), a21 = (1.f - fa) * fb, a22 = fa * fb;
        auto samp = [&](int pi, int pj) -> float {
            const int x = ix + pj, y = iy + pi;
            return ((pix(x, y) * a11 + pix(x + 1, y) * a12) + pix(x, y + 1) * a21) + pix(x + 1, y + 1) * a22;
        };
        if (iter == 3) DBG_T(42);
        double ra = 0, rb = 0, rc = 0, r1s = 0, r2s = 0;
        if (live) {
            float sE, sW, sS, sN;                       // patch elements (wi+1, wj+2), (wi+1, wj), (wi+2, wj+1), (wi, wj+1)
            const int bx = ix + wj - rx0, by = iy + wi - ry0;
            if ((unsigned)bx <= (unsigned)(SP_RS - 4) && (unsigned)by <= (unsigned)(SP_RS - 4)) {
                // the 4x4 pixel block behind the four elements lies in the cached neighbourhood: 12 plain LDS reads
                const unsigned char* q = reg + by * SP_RS + bx;
                const float p01 = q[1], p02 = q[2];
                const float p10 = q[SP_RS], p11 = q[SP_RS + 1], p12 = q[SP_RS + 2], p13 = q[SP_RS + 3];
                const float p20 = q[2 * SP_RS], p21 = q[2 * SP_RS + 1], p22 = q[2 * SP_RS + 2], p23 = q[2 * SP_RS + 3];
                const float p31 = q[3 * SP_RS + 1], p32 = q[3 * SP_RS + 2];
                sE = ((p12 * a11 + p13 * a12) + p22 * a21) + p23 * a22;
                sW = ((p10 * a11 + p11 * a12) + p20 * a21) + p21 * a22;
                sS = ((p21 * a11 + p22 * a12) + p31 * a21) + p32 * a22;
                sN = ((p01 * a11 + p02 * a12) + p11 * a21) + p12 * a22;
            } else { sE = samp(wi + 1, wj + 2); sW = samp(wi + 1, wj); sS = samp(wi + 2, wj + 1); sN = samp(wi, wj + 1); }
            const double tgx = sE - sW;
            const double tgy = sS - sN;
            const double gxx = tgx * tgx * wm, gxy = tgx * tgy * wm, gyy = tgy * tgy * wm;
            ra = gxx; rb = gxy; rc = gyy;
            r1s = gxx * px + gxy * py;
            r2s = gxy * px + gyy * py;
        }
        if (iter == 3) DBG_T(43);
        // canonical order (oracle/detector.cpp): per window row a balanced tree over the 16 columns (j, j+8), (.., +4), (.., +2),
        // (.., +1) = four DPP row rotations; per wave (rows 4w..4w+3) (R0 + R1) + (R2 + R3); then (W0 + W1) + (W2 + W3)
        ra += dpp_f64<0x128>(ra); rb += dpp_f64<0x128>(rb); rc += dpp_f64<0x128>(rc); r1s += dpp_f64<0x128>(r1s); r2s += dpp_f64<0x128>(r2s);
        ra += dpp_f64<0x124>(ra); rb += dpp_f64<0x124>(rb); rc += dpp_f64<0x124>(rc); r1s += dpp_f64<0x124>(r1s); r2s += dpp_f64<0x124>(r2s);
        ra += dpp_f64<0x122>(ra); rb += dpp_f64<0x122>(rb); rc += dpp_f64<0x122>(rc); r1s += dpp_f64<0x122>(r1s); r2s += dpp_f64<0x122>(r2s);
        ra += dpp_f64<0x121>(ra); rb += dpp_f64<0x121>(rb); rc += dpp_f64<0x121>(rc); r1s += dpp_f64<0x121>(r1s); r2s += dpp_f64<0x121>(r2s);
        {
            const double w0 = (readlane_f64(ra, 0) + readlane_f64(ra, 16)) + (readlane_f64(ra, 32) + readlane_f64(ra, 48));
            const double w1 = (readlane_f64(rb, 0) + readlane_f64(rb, 16)) + (readlane_f64(rb, 32) + readlane_f64(rb, 48));
            const double w2 = (readlane_f64(rc, 0) + readlane_f64(rc, 16)) + (readlane_f64(rc, 32) + readlane_f64(rc, 48));
            const double w3 = (readlane_f64(r1s, 0) + readlane_f64(r1s, 16)) + (readlane_f64(r1s, 32) + readlane_f64(r1s, 48));
            const double w4 = (readlane_f64(r2s, 0) + readlane_f64(r2s, 16)) + (readlane_f64(r2s, 32) + readlane_f64(r2s, 48));
            if (lane == 0) { double* sp = s_part[iter & 1][wv]; sp[0] = w0; sp[1] = w1; sp[2] = w2; sp[3] = w3; sp[4] = w4; }
        }
        if (iter == 3) DBG_T(44);
        __syncthreads();
        if (iter == 3) DBG_T(45);
        const double (*sq)[5] = s_part[iter & 1];
        const double a = (sq[0][0] + sq[1][0]) + (sq[2][0] + sq[3][0]);
        const double b = (sq[0][1] + sq[1][1]) + (sq[2][1] + sq[3][1]);
        const double c = (sq[0][2] + sq[1][2]) + (sq[2][2] + sq[3][2]);
        const double bb1 = (sq[0][3] + sq[1][3]) + (sq[2][3] + sq[3][3]);
        const double bb2 = (sq[0][4] + sq[1][4]) + (sq[2][4] + sq[3][4]);
        const double det = a * c - b * b;
        if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
        const double scale = 1.0 / det;
        const float nx = (float)(cx + c * scale * bb1 - b * scale * bb2);
        const float ny = (float)(cy - b * scale * bb1 + a * scale * bb2);
        const float ex = nx - cx, ey = ny - cy;
        err = (double)(ex * ex + ey * ey);
        cx = nx; cy = ny;
        if (cx < 0 || cx >= W || cy < 0 || cy >= H) break;
        if (iter == 3) { DBG_T(46); DBG_T(47); }
    } while (++iter < 30 && err > eps);
#ifdef RVIO_DBG_CLOCKS
    if (p == 0 && tid == 0) g_dbg[48] = iter;
#endif
    if (fabsf(cx - tx) > SP_WIN || fabsf(cy - ty) > SP_WIN) { cx = tx; cy = ty; }
    if (tid == 0) { d.xy[2 * p] = cx; d.xy[2 * p + 1] = cy; }
    DBG_I(p == n - 1 && blockIdx.z == 0, dbg_tag, 3);
}

// Throughput form (batched launches): FOUR corners per wave, one 16-lane DPP row per corner, lane i of the row = window
// row i (lane 15 idles).  A lane evaluates the 15 terms of its window row from a 4 x 18 pixel footprint (every pixel converted once,
// every bilinear sample formed once), folds them by the
// canonical column tree (j, j + 8), (.., + 4), (.., + 2), (.., + 1) inside the lane, and the rows combine by four DPP steps inside the
// row: quad permutes for (R0 + R1) + (R2 + R3), half-row mirror and row mirror for (W0 + W1) + (W2 + W3) — every lane of the row ends
// with the five sums, so the 2 x 2 solve that follows needs no broadcast.  Same additions as subpix_kernel / the oracle (operands of
// some swapped; IEEE addition commutes), hence identical results; per corner and iteration about a third of the instructions of
// round 2's one-wave-per-corner form (the reduction, the 2 x 2 solve and the per-iteration set-up are shared by four corners;
// measured at 128 streams: 81 k -> 97 k frames/s).  The four corners of a
// wave iterate until the last one has converged (a converged row is masked off).
#define SP_LD (SP_LS / 4)
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ub0(unsigned v) { return (float)(v & 0xffu); }
__device__ __forceinline__ float ub1(unsigned v) { return (float)((v >> 8) & 0xffu); }
__device__ __forceinline__ float ub2(unsigned v) { return (float)((v >> 16) & 0xffu); }
__device__ __forceinline__ float ub3(unsigned v) { return (float)(v >> 24); }
__device__ __forceinline__ float ubn(unsigned v, int b) { return b == 0 ? ub0(v) : b == 1 ? ub1(v) : b == 2 ? ub2(v) : ub3(v); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void subpix_kernel16(const uint8_t* __restrict__ src, int stride, DetDev d, size_t src_bs, size_t bs) {
    src = zoff(src, src_bs); det_shift(d, (size_t)blockIdx.z * bs);
    __shared__ unsigned regs[4][SP_RS * SP_LD + 1];          // (+ 1: the sixth dword of the last row may be the one behind the slice)
    const int lane = threadIdx.x, wk = lane >> 4, wi = lane & 15;
    const int p = blockIdx.x * 4 + wk;
    const int n = *d.n_out;
    const bool act = p < n;
    const int W = d.W, H = d.H;
    unsigned* regw = regs[wk];
    unsigned char* reg = (unsigned char*)regw;
    float tx = 0.f, ty = 0.f;
    int rx0 = 0, ry0 = 0;
    if (act) {
        tx = d.raw_xy[2 * p]; ty = d.raw_xy[2 * p + 1];
        rx0 = (int)tx - (SP_PW - 1) / 2 - SP_MARG; ry0 = (int)ty - (SP_PW - 1) / 2 - SP_MARG;
        if (rx0 >= 0 && ry0 >= 0 && rx0 + SP_LS <= W && ry0 + SP_RS <= H) {
            // the neighbourhood lies inside the image: SP_LD dword loads per row (any byte alignment), dword stores
            const uint8_t* base = src + (size_t)ry0 * stride + rx0;
            for (int e = wi; e < SP_RS * SP_LD; e += 16) {
                const int j = e / SP_LD, i = e % SP_LD;
                regw[e] = ((const U32u*)(base + (size_t)j * stride))[i].v;
            }
        } else
            for (int e = wi; e < SP_RS * SP_RS; e += 16) {
                const int j = e / SP_RS, i = e % SP_RS;
                reg[j * SP_LS + i] = src[(size_t)min(max(ry0 + j, 0), H - 1) * stride + min(max(rx0 + i, 0), W - 1)];
            }
    }
    __syncthreads();
    auto pix = [&](int x, int y) -> float {
        const int i = x - rx0, j = y - ry0;
        if ((unsigned)i < (unsigned)SP_RS && (unsigned)j < (unsigned)SP_RS) return (float)reg[j * SP_LS + i];
        return (float)src[(size_t)min(max(y, 0), H - 1) * stride + min(max(x, 0), W - 1)];
    };
    const int wr = wi < SP_WW ? wi : SP_WW - 1;     // (lane 15 reads row 14's pixels; its terms are zeroed below)
    double wm[SP_WW];
#pragma unroll
    for (int wj = 0; wj < SP_WW; ++wj) wm[wj] = (double)d.spmask[wr * SP_WW + wj];
    const double py = wr - SP_WIN;
    float cx = tx, cy = ty;
    const double eps = 1e-2 * 1e-2;
    int iter = 0;
    bool run = act;
    while (__builtin_amdgcn_ballot_w64(run)) {
        if (run) {
            const float ox = cx - (float)(SP_PW - 1) * 0.5f, oy = cy - (float)(SP_PW - 1) * 0.5f;
            const int ix = (int)floorf(ox), iy = (int)floorf(oy);
            float fa = ox - (float)ix;
            const float fb = oy - (float)iy;
            fa = fmaxf(fa, 0.0001f);
            const float a11 = (1.f - fa) * (1.f - fb), a12 = fa * (1.f - fb), a21 = (1.f - fa) * fb, a22 = fa * fb;
            // the lane's footprint: pixel (c, r) = image (ix + c, iy + wr + r), c = 0..17, r = 0..3, held as the operand pairs of the packed
            // bilinear forms:  PP[c] = (row 2, row 0), QQ[c] = (row 3, row 1)  ->  samples (S2[c], S0[c]);
            //                  TT[c] = row 1 at (c, c + 8), UU[c] = row 2 at (c, c + 8)  ->  samples (S1[c], S1[c + 8])
            f2v PP[18], QQ[18], TT[10], UU[10];
            const int ux = ix - rx0, uy = iy - ry0;
            if (ux >= 0 && uy >= 0 && ux <= SP_RS - 4 - (SP_WW - 1) && uy <= SP_RS - 4 - (SP_WW - 1)) {
                // inside the cached neighbourhood: 6 dwords per row, shifted to the footprint's first byte, bytes converted in place
                const int sh = ux & 3;
                const unsigned* rowp = regw + (uy + wr) * SP_LD + (ux >> 2);
                unsigned A[4][5];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    unsigned dw[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) dw[j] = rowp[r * SP_LD + j];
#pragma unroll
                    for (int j = 0; j < 5; ++j) A[r][j] = __builtin_amdgcn_alignbyte(dw[j + 1], dw[j], sh);
                }
#pragma unroll
                for (int c = 0; c < 18; ++c) {
                    PP[c] = f2v{ubn(A[2][c >> 2], c & 3), ubn(A[0][c >> 2], c & 3)};
                    QQ[c] = f2v{ubn(A[3][c >> 2], c & 3), ubn(A[1][c >> 2], c & 3)};
                }
#pragma unroll
                for (int c = 0; c < 10; ++c) {
                    TT[c] = f2v{ubn(A[1][c >> 2], c & 3), ubn(A[1][(c + 8) >> 2], (c + 8) & 3)};
                    UU[c] = f2v{ubn(A[2][c >> 2], c & 3), ubn(A[2][(c + 8) >> 2], (c + 8) & 3)};
                }
            } else {
                // the estimate has wandered off the cached neighbourhood (or sits at the image border): pixel by pixel
#pragma unroll 1
                for (int c = 0; c < 18; ++c) {
                    const float r0 = pix(ix + c, iy + wr), r1 = pix(ix + c, iy + wr + 1), r2 = pix(ix + c, iy + wr + 2), r3 = pix(ix + c, iy + wr + 3);
                    // (dynamic index into register arrays would spill: a switch over the unrolled copies)
#pragma unroll
                    for (int cc = 0; cc < 18; ++cc)
                        if (cc == c) {
                            PP[cc] = f2v{r2, r0}; QQ[cc] = f2v{r3, r1};
                            if (cc < 10) { TT[cc].x = r1; UU[cc].x = r2; }
                            if (cc >= 8) { TT[cc - 8].y = r1; UU[cc - 8].y = r2; }
                        }
                }
            }
            // bilinear samples (getRectSubPix), two per packed operation: same IEEE operations as the scalar form, contraction off
            f2v S20[SP_WW + 1], S1p[9];
#pragma unroll
            for (int c = 1; c <= SP_WW; ++c) S20[c] = ((PP[c] * a11 + PP[c + 1] * a12) + QQ[c] * a21) + QQ[c + 1] * a22;     // (S2[c], S0[c])
#pragma unroll
            for (int c = 0; c < 9; ++c) S1p[c] = ((TT[c] * a11 + TT[c + 1] * a12) + UU[c] * a21) + UU[c + 1] * a22;          // (S1[c], S1[c + 8])
            auto S1 = [&](int c) -> float { return c < 8 ? S1p[c].x : S1p[c - 8].y; };
            // term wj of the window row: the five products; the canonical column tree (j, j + 8), (.., + 4), (.., + 2), (.., + 1) is
            // walked depth first — j, j + 8 -> u_j;  u_j + u_(j+4) -> v_j;  v_j + v_(j+2) -> w_j;  w_0 + w_1 — so few partial sums are alive
            auto term = [&](int wj, double* o) {
                if (wj >= SP_WW) { o[0] = o[1] = o[2] = o[3] = o[4] = 0.0; return; }       // the padding column
                const double tgx = S1(wj + 2) - S1(wj);
                const double tgy = S20[wj + 1].x - S20[wj + 1].y;
                const double px = wj - SP_WIN;
                const double gxx = tgx * tgx * wm[wj], gxy = tgx * tgy * wm[wj], gyy = tgy * tgy * wm[wj];
                o[0] = gxx; o[1] = gxy; o[2] = gyy;
                o[3] = gxx * px + gxy * py;
                o[4] = gxy * px + gyy * py;
            };
            auto uj = [&](int j, double* o) { double x[5], y[5]; term(j, x); term(j + 8, y); for (int c = 0; c < 5; ++c) o[c] = x[c] + y[c]; };
            auto vj = [&](int j, double* o) { double x[5], y[5]; uj(j, x); uj(j + 4, y); for (int c = 0; c < 5; ++c) o[c] = x[c] + y[c]; };
            auto wj_ = [&](int j, double* o) { double x[5], y[5]; vj(j, x); vj(j + 2, y); for (int c = 0; c < 5; ++c) o[c] = x[c] + y[c]; };
            double w0[5], w1[5], sm[5];
            wj_(0, w0); wj_(1, w1);
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                double v = w0[c] + w1[c];
                if (wi >= SP_WW) v = 0.0;               // the padding row
                // rows: lane i of the DPP row = window row i
                v += dpp_f64<0xb1>(v);      // quad_perm [1,0,3,2]: R0 + R1 | R2 + R3
                v += dpp_f64<0x4e>(v);      // quad_perm [2,3,0,1]: (R0 + R1) + (R2 + R3) = W, in all four lanes
                v += dpp_f64<0x141>(v);     // row_half_mirror: W0 + W1 | W2 + W3
                v += dpp_f64<0x140>(v);     // row_mirror: (W0 + W1) + (W2 + W3), in all sixteen lanes
                sm[c] = v;
            }
            const double a = sm[0], b = sm[1], c = sm[2], bb1 = sm[3], bb2 = sm[4];
            const double det = a * c - b * b;
            if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) run = false;
            else {
                const double scale = 1.0 / det;
                const float nx = (float)(cx + c * scale * bb1 - b * scale * bb2);
                const float ny = (float)(cy - b * scale * bb1 + a * scale * bb2);
                const float ex = nx - cx, ey = ny - cy;
                const double err = (double)(ex * ex + ey * ey);
                cx = nx; cy = ny;
                if (cx < 0 || cx >= W || cy < 0 || cy >= H) run = false;
                else run = ++iter < 30 && err > eps;
            }
        }
    }
    if (fabsf(cx - tx) > SP_WIN || fabsf(cy - ty) > SP_WIN) { cx = tx; cy = ty; }
    if (act && wi == 0) { d.xy[2 * p] = cx; d.xy[2 * p + 1] = cy; }
}


// cornerSubPix for any half-window 1 <= win <= 15 other than the stock 7 (FeatureDetector.cc:68: floor(nMinDist / 2)): the plain form.
// One workgroup per corner; the (2 win + 1)^2 window terms go to a zero-padded G x G grid in LDS (G = 16 / 32) and are summed in the
// canonical order of oracle/detector.cpp — per row a balanced tree over the columns, rows in groups of four, the groups by a balanced tree —
// by 5 G threads (one per quantity and row), then one thread per quantity.  Correct for every window, tuned for none: the stock window
// runs subpix_kernel / subpix_kernel16.
#define SPG_T 256
#define SPG_G 32
__global__ __launch_bounds__(SPG_T) void subpix_generic_kernel(const uint8_t* __restrict__ src, int stride, DetDev d, size_t src_bs, size_t bs) {
    src = zoff(src, src_bs); det_shift(d, (size_t)blockIdx.z * bs);
    __shared__ double term[5][SPG_G][SPG_G + 1];
    __shared__ double rowsum[5][SPG_G];
    __shared__ double tot[5];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n = *d.n_out;
    if (p >= n) return;
    const int W = d.W, H = d.H, win = d.sp_win, ww = 2 * win + 1, pw = ww + 2, G = ww <= 16 ? 16 : 32;
    const float tx = d.raw_xy[2 * p], ty = d.raw_xy[2 * p + 1];
    auto pix = [&](int x, int y) -> float { return (float)src[(size_t)min(max(y, 0), H - 1) * stride + min(max(x, 0), W - 1)]; };
    float cx = tx, cy = ty;
    const double eps = 1e-2 * 1e-2;
    int iter = 0;
    double err = 0;
    for (int e = tid; e < 5 * SPG_G * (SPG_G + 1); e += SPG_T) (&term[0][0][0])[e] = 0.0;      // padding stays zero
    __syncthreads();
    do {
        const float ox = cx - (float)(pw - 1) * 0.5f, oy = cy - (float)(pw - 1) * 0.5f;   // getRectSubPix, samplers.cpp
        const int ix = (int)floorf(ox), iy = (int)floorf(oy);
        float fa = ox - (float)ix;
        const float fb = oy - (float)iy;
        fa = fmaxf(fa, 0.0001f);
        const float a11 = (1.f - fa) * (1.f - fb), a12 = fa * (1.f - fb), a21 = (1.f - fa) * fb, a22 = fa * fb;
        auto samp = [&](int pi, int pj) -> float {
            const int x = ix + pj, y = iy + pi;
            return ((pix(x, y) * a11 + pix(x + 1, y) * a12) + pix(x, y + 1) * a21) + pix(x + 1, y + 1) * a22;
        };
        for (int e = tid; e < ww * ww; e += SPG_T) {
            const int wi = e / ww, wj = e - wi * ww;
            const double wm = (double)d.spmask[e], px = wj - win, py = wi - win;
            const double tgx = samp(wi + 1, wj + 2) - samp(wi + 1, wj);
            const double tgy = samp(wi + 2, wj + 1) - samp(wi, wj + 1);
            const double gxx = tgx * tgx * wm, gxy = tgx * tgy * wm, gyy = tgy * tgy * wm;
            term[0][wi][wj] = gxx; term[1][wi][wj] = gxy; term[2][wi][wj] = gyy;
            term[3][wi][wj] = gxx * px + gxy * py;
            term[4][wi][wj] = gxy * px + gyy * py;
        }
        __syncthreads();
        if (tid < 5 * G) {      // the j = 0 leaf of the balanced column tree: level s combines (k, k + s) for k < s
            const int q = tid / G, i = tid - q * G;
            double v[SPG_G];
#pragma unroll
            for (int j = 0; j < SPG_G; ++j) v[j] = j < G ? term[q][i][j] : 0.0;
            for (int sft = G / 2; sft >= 1; sft >>= 1)
#pragma unroll
                for (int k = 0; k < SPG_G / 2; ++k) if (k < sft) v[k] = v[k] + v[k + sft];
            rowsum[q][i] = v[0];
        }
        __syncthreads();
        if (tid < 5) {
            const double* R = rowsum[tid];
            double Wg[8];
            for (int g = 0; g < G / 4; ++g) Wg[g] = (R[4 * g] + R[4 * g + 1]) + (R[4 * g + 2] + R[4 * g + 3]);
            double t = (Wg[0] + Wg[1]) + (Wg[2] + Wg[3]);
            if (G == 32) t = t + ((Wg[4] + Wg[5]) + (Wg[6] + Wg[7]));
            tot[tid] = t;
        }
        __syncthreads();
        const double a = tot[0], b = tot[1], c = tot[2], bb1 = tot[3], bb2 = tot[4];
        const double det = a * c - b * b;
        if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
        const double scale = 1.0 / det;
        const float nx = (float)(cx + c * scale * bb1 - b * scale * bb2);
        const float ny = (float)(cy - b * scale * bb1 + a * scale * bb2);
        const float ex = nx - cx, ey = ny - cy;
        err = (double)(ex * ex + ey * ey);
        cx = nx; cy = ny;
        if (cx < 0 || cx >= W || cy < 0 || cy >= H) break;
    } while (++iter < 30 && err > eps);
    if (fabsf(cx - tx) > win || fabsf(cy - ty) > win) { cx = tx; cy = ty; }
    if (tid == 0) { d.xy[2 * p] = cx; d.xy[2 * p + 1] = cy; }
}

// cornerSubPix for half-windows 16 .. 63 (32 <= Tracker.nMinDist < 128): the canonical G x G grid (G = 64 / 128) does not fit LDS as doubles,
// so the first levels of the balanced column tree are taken in registers: the thread of (row i, folded column jf < 16) forms the terms of the
// columns jf + 16 m (m < G / 16) and adds them in the tree's own order ((j, j + G/2), then + G/4, ... down to + 16) before anything is stored —
// the same additions as oracle/detector.cpp, in the same association.  LDS holds the 16 folded columns of every row; the rest as above.
#define SPW_C 16
__global__ __launch_bounds__(SPG_T) void subpix_wide_kernel(const uint8_t* __restrict__ src, int stride, DetDev d, size_t src_bs, size_t bs) {
    src = zoff(src, src_bs); det_shift(d, (size_t)blockIdx.z * bs);
    extern __shared__ __align__(16) double spw_dyn[];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n = *d.n_out;
    if (p >= n) return;
    const int W = d.W, H = d.H, win = d.sp_win, ww = 2 * win + 1, pw = ww + 2, G = ww <= 64 ? 64 : 128, M = G / SPW_C;
    double* term = spw_dyn;                                   // [5][G][SPW_C + 1]
    double* rowsum = term + (size_t)5 * G * (SPW_C + 1);      // [5][G]
    double* tot = rowsum + 5 * G;                             // [5]
    const float tx = d.raw_xy[2 * p], ty = d.raw_xy[2 * p + 1];
    auto pix = [&](int x, int y) -> float { return (float)src[(size_t)min(max(y, 0), H - 1) * stride + min(max(x, 0), W - 1)]; };
    float cx = tx, cy = ty;
    const double eps = 1e-2 * 1e-2;
    int iter = 0;
    double err = 0;
    do {
        const float ox = cx - (float)(pw - 1) * 0.5f, oy = cy - (float)(pw - 1) * 0.5f;   // getRectSubPix, samplers.cpp
        const int ix = (int)floorf(ox), iy = (int)floorf(oy);
        float fa = ox - (float)ix;
        const float fb = oy - (float)iy;
        fa = fmaxf(fa, 0.0001f);
        const float a11 = (1.f - fa) * (1.f - fb), a12 = fa * (1.f - fb), a21 = (1.f - fa) * fb, a22 = fa * fb;
        auto samp = [&](int pi, int pj) -> float {
            const int x = ix + pj, y = iy + pi;
            return ((pix(x, y) * a11 + pix(x + 1, y) * a12) + pix(x, y + 1) * a21) + pix(x + 1, y + 1) * a22;
        };
        for (int e = tid; e < G * SPW_C; e += SPG_T) {
            const int wi = e / SPW_C, jf = e - wi * SPW_C;
            double v[5][8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int wj = jf + SPW_C * m;
                double gxx = 0.0, gxy = 0.0, gyy = 0.0, t3 = 0.0, t4 = 0.0;
                if (m < M && wi < ww && wj < ww) {
                    const double wm = (double)d.spmask[wi * ww + wj], px = wj - win, py = wi - win;
                    const double tgx = samp(wi + 1, wj + 2) - samp(wi + 1, wj);
                    const double tgy = samp(wi + 2, wj + 1) - samp(wi, wj + 1);
                    gxx = tgx * tgx * wm; gxy = tgx * tgy * wm; gyy = tgy * tgy * wm;
                    t3 = gxx * px + gxy * py;
                    t4 = gxy * px + gyy * py;
                }
                v[0][m] = gxx; v[1][m] = gxy; v[2][m] = gyy; v[3][m] = t3; v[4][m] = t4;
            }
            // the tree's levels above 16 columns: (m, m + M/2), then + M/4, ...
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                if (M == 8) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[q][m] = v[q][m] + v[q][m + 4];
                }
                v[q][0] = v[q][0] + v[q][2]; v[q][1] = v[q][1] + v[q][3];
                term[((size_t)q * G + wi) * (SPW_C + 1) + jf] = v[q][0] + v[q][1];
            }
        }
        __syncthreads();
        for (int e = tid; e < 5 * G; e += SPG_T) {          // the tree's remaining levels: + 8, + 4, + 2, + 1
            const double* t = term + (size_t)e * (SPW_C + 1);
            double v[SPW_C];
#pragma unroll
            for (int j = 0; j < SPW_C; ++j) v[j] = t[j];
#pragma unroll
            for (int sft = SPW_C / 2; sft >= 1; sft >>= 1)
#pragma unroll
                for (int k = 0; k < SPW_C / 2; ++k) if (k < sft) v[k] = v[k] + v[k + sft];
            rowsum[e] = v[0];
        }
        __syncthreads();
        if (tid < 5) {
            const double* R = rowsum + tid * G;
            double Wg[32];
            for (int g = 0; g < G / 4; ++g) Wg[g] = (R[4 * g] + R[4 * g + 1]) + (R[4 * g + 2] + R[4 * g + 3]);
            for (int m = G / 4; m > 1; m >>= 1)
                for (int g = 0; g < m / 2; ++g) Wg[g] = Wg[2 * g] + Wg[2 * g + 1];
            tot[tid] = Wg[0];
        }
        __syncthreads();
        const double a = tot[0], b = tot[1], c = tot[2], bb1 = tot[3], bb2 = tot[4];
        const double det = a * c - b * b;
        if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
        const double scale = 1.0 / det;
        const float nx = (float)(cx + c * scale * bb1 - b * scale * bb2);
        const float ny = (float)(cy - b * scale * bb1 + a * scale * bb2);
        const float ex = nx - cx, ey = ny - cy;
        err = (double)(ex * ex + ey * ey);
        cx = nx; cy = ny;
        if (cx < 0 || cx >= W || cy < 0 || cy >= H) break;
    } while (++iter < 30 && err > eps);
    if (fabsf(cx - tx) > win || fabsf(cy - ty) > win) { cx = tx; cy = ty; }
    if (tid == 0) { d.xy[2 * p] = cx; d.xy[2 * p + 1] = cy; }
}
// dynamic LDS of subpix_wide_kernel for a half-window
static inline size_t subpix_wide_lds(int win) {
    const int G = 2 * win + 1 <= 64 ? 64 : 128;
    return sizeof(double) * ((size_t)5 * G * (SPW_C + 1) + 5 * G + 8);
}
