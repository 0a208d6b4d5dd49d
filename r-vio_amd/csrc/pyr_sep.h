// pyr_sep.h — the body of pyramid_kernel (frontend_kernels.hip), written as per-thread PHASES between workgroup barriers so that the
// very same code can be run on a host thread by thread (tests/test_pyramid_phases.py compiles this header with g++ and walks
// blocks x phases x threads in plain loops: the phase code is checked against the oracle's pyrDown on the CPU, before any GPU sees it).
//
// buildOpticalFlowPyramid of cv::calcOpticalFlowPyrLK (Tracker.cc:244) in ONE launch: a workgroup owns an 8x8 tile of level 3 and
// everything above it — it stages the 85x85 patch of level 0 that tile depends on, forms its 41x41 / 19x19 dependency patches of
// levels 1 / 2 in LDS (cv::pyrDown: [1 4 6 4 1]/16 separable, BORDER_REFLECT_101 at every level's own size, (v+128)>>8) and
// stores the tiles it owns: 64x64 of level 0 (the copy of the frame into the pyramid), 32x32 of level 1, 16x16 of level 2, 8x8 of
// level 3.
//
// Round 5: each pyrDown is SEPARABLE here — a vertical pass over the full-resolution columns of the patch (u16 column sums, <= 4080),
// then a horizontal pass over those — instead of a 25-tap gather per output: 2.7 x fewer LDS loads for level 1 (15.6 k against 42 k per
// workgroup), and the reflect-101 index arithmetic (ten while-loops per output before) moved into four small index tables a workgroup
// fills once.  The result is the same integer: sum_k sum_j wy[k] wx[j] p[ry[k]][rx[j]] whichever sum runs first (no rounding before
// the final (v + 128) >> 8).
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#define PYR_FN __host__ __device__ __forceinline__
#else
#define PYR_FN static inline
#endif
#define PYR_T 256

PYR_FN int pyr_reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
    return i;
}
PYR_FN int pyr_imin(int a, int b) { return a < b ? a : b; }
PYR_FN int pyr_imax(int a, int b) { return a > b ? a : b; }

struct PyrOut { uint8_t* img[4]; int w[4], h[4]; };

// LDS of one workgroup (19.5 KB: eight workgroups per CU)
struct PyrLds {
    uint8_t L0[85 * 88];      // level-0 patch, row stride 88
    uint16_t V1[41 * 88];     // vertical 5-tap sums of L0 at the 41 rows level 1 needs, one per level-0 column
    uint8_t L1[41 * 44];      // level-1 patch
    uint16_t V2[19 * 44];     // vertical sums of L1 at the 19 rows level 2 needs
    uint8_t L2[19 * 20];      // level-2 patch
    uint32_t ty1[41][2], tx1[41][2], ty2[19][2], tx2[19][2];   // the five reflected source rows / columns of every patch row / column (u8 each, patch-relative)
};

struct PyrGeom {
    int X3, Y3, nl, copy0;            // nl: how many levels this workgroup produces (0: nothing to do)
    int w0, h0, ax, ay, pw, ph;       // level-0 patch [ax, ax + pw) x [ay, ay + ph)
    int w1, h1, cx, cy, qw, qh;       // level-1 patch
    int w2, h2, ex, ey, rw, rh;       // level-2 patch
    int w3, h3;
};

PYR_FN PyrGeom pyr_geom(const PyrOut& p, int bx_, int by_, int levels, int copy0) {
    PyrGeom g;
    g.X3 = bx_ * 8; g.Y3 = by_ * 8; g.copy0 = copy0; g.nl = 0;
    g.w0 = p.w[0]; g.h0 = p.h[0];
    g.w1 = g.h1 = g.cx = g.cy = g.qw = g.qh = 0; g.w2 = g.h2 = g.ex = g.ey = g.rw = g.rh = 0; g.w3 = g.h3 = 0;
    // level-0 patch [8 X3 - 14, 8 X3 + 70] clipped to the image
    g.ax = pyr_imax(8 * g.X3 - 14, 0); g.ay = pyr_imax(8 * g.Y3 - 14, 0);
    g.pw = pyr_imin(8 * g.X3 + 70, g.w0 - 1) - g.ax + 1; g.ph = pyr_imin(8 * g.Y3 + 70, g.h0 - 1) - g.ay + 1;
    if (g.pw <= 0 || g.ph <= 0) return g;
    g.nl = 1;
    if (levels < 2) return g;
    g.w1 = p.w[1]; g.h1 = p.h[1];
    g.cx = pyr_imax(4 * g.X3 - 6, 0); g.cy = pyr_imax(4 * g.Y3 - 6, 0);
    g.qw = pyr_imin(4 * g.X3 + 34, g.w1 - 1) - g.cx + 1; g.qh = pyr_imin(4 * g.Y3 + 34, g.h1 - 1) - g.cy + 1;
    if (g.qw <= 0 || g.qh <= 0) return g;
    g.nl = 2;
    if (levels < 3) return g;
    g.w2 = p.w[2]; g.h2 = p.h[2];
    g.ex = pyr_imax(2 * g.X3 - 2, 0); g.ey = pyr_imax(2 * g.Y3 - 2, 0);
    g.rw = pyr_imin(2 * g.X3 + 16, g.w2 - 1) - g.ex + 1; g.rh = pyr_imin(2 * g.Y3 + 16, g.h2 - 1) - g.ey + 1;
    if (g.rw <= 0 || g.rh <= 0) return g;
    g.nl = 3;
    if (levels < 4) return g;
    g.w3 = p.w[3]; g.h3 = p.h[3];
    g.nl = 4;
    return g;
}

// the five source indices reflect101(2 v - 2 + k, n) - origin, k = 0..4, packed as bytes
PYR_FN void pyr_taps(uint32_t* t, int v, int n, int origin) {
    uint32_t lo = 0;
    for (int k = 0; k < 4; ++k) lo |= (uint32_t)(pyr_reflect101(2 * v - 2 + k, n) - origin) << (8 * k);
    t[0] = lo;
    t[1] = (uint32_t)(pyr_reflect101(2 * v + 2, n) - origin);
}
PYR_FN int pyr_comb(int a0, int a1, int a2, int a3, int a4) { return a2 * 6 + (a1 + a3) * 4 + a0 + a4; }

// phase 0: the level-0 patch into LDS (three rows of 85 columns per trip: consecutive lanes, consecutive bytes) + the index tables
PYR_FN void pyr_phase0(const PyrGeom& g, int tid, PyrLds& s, const uint8_t* src, int stride) {
    if (g.nl < 1) return;
    if (tid < 255) {
        const int r = tid / 85, c = tid - 85 * r;
        if (c < g.pw) {
            const uint8_t* sp = src + (size_t)g.ay * stride + g.ax + c;
            for (int yy = r; yy < g.ph; yy += 3) s.L0[yy * 88 + c] = sp[(size_t)yy * stride];
        }
    }
    if (g.nl >= 2) {
        if (tid < 41) { if (tid < g.qh) pyr_taps(s.ty1[tid], g.cy + tid, g.h0, g.ay); }
        else if (tid < 82) { const int t = tid - 41; if (t < g.qw) pyr_taps(s.tx1[t], g.cx + t, g.w0, g.ax); }
        else if (g.nl >= 3) {
            if (tid < 101) { const int t = tid - 82; if (t < g.rh) pyr_taps(s.ty2[t], g.ey + t, g.h1, g.cy); }
            else if (tid < 120) { const int t = tid - 101; if (t < g.rw) pyr_taps(s.tx2[t], g.ex + t, g.w1, g.cx); }
        }
    }
}

// phase 1: this workgroup's 64x64 tile of level 0 (when the frame is not the pyramid's level 0 already) + the vertical pass of level 1
PYR_FN void pyr_phase1(const PyrGeom& g, int tid, PyrLds& s, const PyrOut& p) {
    if (g.nl < 1) return;
    if (g.copy0) {
        uint8_t* d0 = p.img[0];
        for (int e = tid; e < 64 * 64; e += PYR_T) {
            const int x = 8 * g.X3 + (e & 63), y = 8 * g.Y3 + (e >> 6);
            if (x < g.w0 && y < g.h0) d0[(size_t)y * g.w0 + x] = s.L0[(y - g.ay) * 88 + (x - g.ax)];
        }
    }
    if (g.nl < 2 || tid >= 255) return;
    const int r = tid / 85, c = tid - 85 * r;
    if (c >= g.pw) return;
    for (int yy = r; yy < g.qh; yy += 3) {
        const uint32_t lo = s.ty1[yy][0], hi = s.ty1[yy][1];
        const int a0 = s.L0[(lo & 255) * 88 + c], a1 = s.L0[((lo >> 8) & 255) * 88 + c], a2 = s.L0[((lo >> 16) & 255) * 88 + c],
                  a3 = s.L0[(lo >> 24) * 88 + c], a4 = s.L0[hi * 88 + c];
        s.V1[yy * 88 + c] = (uint16_t)pyr_comb(a0, a1, a2, a3, a4);
    }
}

// phase 2: the horizontal pass of level 1 -> the level-1 patch in LDS, the owned 32x32 tile to HBM
PYR_FN void pyr_phase2(const PyrGeom& g, int tid, PyrLds& s, const PyrOut& p) {
    if (g.nl < 2 || tid >= 246) return;
    const int r = tid / 41, xx = tid - 41 * r;
    if (xx >= g.qw) return;
    const uint32_t lo = s.tx1[xx][0], hi = s.tx1[xx][1];
    const int x0 = lo & 255, x1 = (lo >> 8) & 255, x2 = (lo >> 16) & 255, x3 = lo >> 24, x4 = hi;
    const int x = g.cx + xx;
    const bool mine_x = x >= 4 * g.X3 && x < 4 * g.X3 + 32;
    uint8_t* d1 = p.img[1];
    for (int yy = r; yy < g.qh; yy += 6) {
        const uint16_t* v = s.V1 + yy * 88;
        const int o = (pyr_comb(v[x0], v[x1], v[x2], v[x3], v[x4]) + 128) >> 8;
        s.L1[yy * 44 + xx] = (uint8_t)o;
        const int y = g.cy + yy;
        if (mine_x && y >= 4 * g.Y3 && y < 4 * g.Y3 + 32) d1[(size_t)y * g.w1 + x] = (uint8_t)o;
    }
}

// phase 3: the vertical pass of level 2 over the level-1 patch
PYR_FN void pyr_phase3(const PyrGeom& g, int tid, PyrLds& s) {
    if (g.nl < 3 || tid >= 246) return;
    const int r = tid / 41, c = tid - 41 * r;
    if (c >= g.qw) return;
    for (int yy = r; yy < g.rh; yy += 6) {
        const uint32_t lo = s.ty2[yy][0], hi = s.ty2[yy][1];
        const int a0 = s.L1[(lo & 255) * 44 + c], a1 = s.L1[((lo >> 8) & 255) * 44 + c], a2 = s.L1[((lo >> 16) & 255) * 44 + c],
                  a3 = s.L1[(lo >> 24) * 44 + c], a4 = s.L1[hi * 44 + c];
        s.V2[yy * 44 + c] = (uint16_t)pyr_comb(a0, a1, a2, a3, a4);
    }
}

// phase 4: the horizontal pass of level 2 -> the level-2 patch in LDS, the owned 16x16 tile to HBM
PYR_FN void pyr_phase4(const PyrGeom& g, int tid, PyrLds& s, const PyrOut& p) {
    if (g.nl < 3 || tid >= 247) return;
    const int r = tid / 19, xx = tid - 19 * r;
    if (xx >= g.rw) return;
    const uint32_t lo = s.tx2[xx][0], hi = s.tx2[xx][1];
    const int x0 = lo & 255, x1 = (lo >> 8) & 255, x2 = (lo >> 16) & 255, x3 = lo >> 24, x4 = hi;
    const int x = g.ex + xx;
    const bool mine_x = x >= 2 * g.X3 && x < 2 * g.X3 + 16;
    uint8_t* d2 = p.img[2];
    for (int yy = r; yy < g.rh; yy += 13) {
        const uint16_t* v = s.V2 + yy * 44;
        const int o = (pyr_comb(v[x0], v[x1], v[x2], v[x3], v[x4]) + 128) >> 8;
        s.L2[yy * 20 + xx] = (uint8_t)o;
        const int y = g.ey + yy;
        if (mine_x && y >= 2 * g.Y3 && y < 2 * g.Y3 + 16) d2[(size_t)y * g.w2 + x] = (uint8_t)o;
    }
}

// phase 5: the owned 8x8 tile of level 3 straight from the level-2 patch (64 outputs: a 25-tap gather each)
PYR_FN void pyr_phase5(const PyrGeom& g, int tid, PyrLds& s, const PyrOut& p) {
    if (g.nl < 4 || tid >= 64) return;
    const int x = g.X3 + (tid & 7), y = g.Y3 + (tid >> 3);
    if (x >= g.w3 || y >= g.h3) return;
    int xs[5], rows[5];
    for (int k = 0; k < 5; ++k) xs[k] = pyr_reflect101(2 * x - 2 + k, g.w2) - g.ex;
    for (int k = 0; k < 5; ++k) {
        const uint8_t* q = s.L2 + (pyr_reflect101(2 * y - 2 + k, g.h2) - g.ey) * 20;
        rows[k] = pyr_comb(q[xs[0]], q[xs[1]], q[xs[2]], q[xs[3]], q[xs[4]]);
    }
    p.img[3][(size_t)y * g.w3 + x] = (uint8_t)((pyr_comb(rows[0], rows[1], rows[2], rows[3], rows[4]) + 128) >> 8);
}
