// klt16.hip — LKTrackerInvoker (cv::calcOpticalFlowPyrLK as called at Tracker.cc:237-244), throughput form for batched launches.
// Included inside the FP-contraction-off region of rvio_hip.hip: bit-identical to klt_kernel3 and oracle/frontend.cpp.
//
// FOUR features per wave, one 16-lane DPP row per feature; lane r of the row owns window row r (15 pixels; lane 15 has no window row
// and helps with the staging and the derivative rows only).  klt_kernel3 spends one wave on a feature: 225 window pixels over 64 lanes
// is 3.5 pixels per lane, so its wave reductions (five per level, two per iteration), its scalar float tail (weights, 2 x 2 solve,
// convergence tests) and its staging address arithmetic cost more instructions than the pixels do.  Here those are shared by four
// features, a lane reads its two rows of 16 bytes as dwords (one v_alignbyte per dword, bytes picked out of registers), the Scharr
// derivative rows never go to LDS (a lane computes derivative row r for its own window row and gets row r + 1 from its neighbour lane
// by DPP), and the sums of products are exact integers — 32 bits inside the lane (15 x 16320 x 4080 < 2^31), 64 bits across the row —
// so their order is free.  The float arithmetic (weights, matrix, step, tests) is klt_kernel3's expression for expression.
// Levels run one after another (template patch and search region of one level staged at a time: 1.5 KB of LDS per feature); the four
// features of a wave iterate a level until the last one has converged (a converged row is masked off).
#pragma once
#include "rvio_dev.h"
#include "frontend_dev.h"

#define K16_JS 36            // row stride of the staged 32 x 32 search region: 9 dwords — the 16 rows a row of lanes reads start in 16 different banks
#define K16_IS 20            // row stride of the staged 18 x 18 template patch: 5 dwords
struct __attribute__((packed)) K16U32 { unsigned v; };

// lane r of the row stages region rows 2r, 2r + 1 (32 bytes each) around (jx0, jy0), reflect-101 coordinates
__device__ __forceinline__ void k16_stage_j(uint8_t* Jw, const uint8_t* __restrict__ J, int w, int h, int jx0, int jy0, int r) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = 2 * r + rr;
        unsigned* o = (unsigned*)(Jw + row * K16_JS);
        if (jx0 >= 0 && jy0 >= 0 && jx0 + 32 <= w && jy0 + 32 <= h) {
            const K16U32* g = (const K16U32*)(J + (size_t)(jy0 + row) * w + jx0);
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = g[q].v;
        } else {
            const uint8_t* jrow = J + (size_t)reflect2(jy0 + row, h) * w;
#pragma unroll 1
            for (int q = 0; q < 8; ++q) {
                unsigned v = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) v |= (unsigned)jrow[reflect2(jx0 + 4 * q + bb, w)] << (8 * bb);
                o[q] = v;
            }
        }
    }
}
// lane r stages patch row r (and lanes 0, 1 rows 16, 17) of the 18 x 18 template source at (x0, y0) = (ipx - 1, ipy - 1)
__device__ __forceinline__ void k16_stage_i(uint8_t* Iw, const uint8_t* __restrict__ I, int w, int h, int x0, int y0, int r) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = r + 16 * pass;
        if (row < 18) {
            unsigned* o = (unsigned*)(Iw + row * K16_IS);
            if (x0 >= 0 && y0 >= 0 && x0 + K16_IS <= w && y0 + 18 <= h) {
                const K16U32* g = (const K16U32*)(I + (size_t)(y0 + row) * w + x0);
#pragma unroll
                for (int q = 0; q < 5; ++q) o[q] = g[q].v;
            } else {
                const uint8_t* irow = I + (size_t)reflect2(y0 + row, h) * w;
#pragma unroll 1
                for (int q = 0; q < 5; ++q) {
                    unsigned v = 0;
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) { const int c = 4 * q + bb; if (c < 18) v |= (unsigned)irow[reflect2(x0 + c, w)] << (8 * bb); }
                    o[q] = v;
                }
            }
        }
    }
}
__device__ __forceinline__ int k16_byte(const unsigned* dw, int c) { return (int)((dw[c >> 2] >> (8 * (c & 3))) & 0xffu); }
// exact sum over the 16 lanes of a DPP row, every lane ends with it
__device__ __forceinline__ long long k16_row_sum(int v) {
    long long x = v;
    x += dpp_i64<0xb1>(x);      // quad_perm [1,0,3,2]
    x += dpp_i64<0x4e>(x);      // quad_perm [2,3,0,1]
    x += dpp_i64<0x141>(x);     // row_half_mirror
    x += dpp_i64<0x140>(x);     // row_mirror
    return x;
}
// (float)(long long), |s| < 2^53: both conversions and the sum are exact in double, the one rounding is the double -> float one
__device__ __forceinline__ float k16_i64_to_f32(long long s) {
    const double hi = (double)(int)(s >> 32), lo = (double)(unsigned)(s & 0xffffffffLL);
    return (float)(hi * 4294967296.0 + lo);
}

__global__ __launch_bounds__(64) void klt_kernel16(PyrDev prev, PyrDev next, int levels, const int* __restrict__ n_pts_ptr,
                                                   const float* __restrict__ pts, float* __restrict__ out, unsigned char* __restrict__ status, size_t bs) {
    pyr_shift(prev, (size_t)blockIdx.z * bs); pyr_shift(next, (size_t)blockIdx.z * bs);
    n_pts_ptr = zoff(n_pts_ptr, bs); pts = zoff(pts, bs); out = zoff(out, bs); status = zoff(status, bs);
    __shared__ __align__(16) uint8_t Ipat[4][18 * K16_IS];
    __shared__ __align__(16) uint8_t Jreg[4][32 * K16_JS];
    const int lane = threadIdx.x, wk = lane >> 4, r = lane & 15;
    const int f = blockIdx.x * 4 + wk;
    const bool valid = f < *n_pts_ptr;
    float px = 0.f, py = 0.f;
    if (valid) { px = pts[2 * f]; py = pts[2 * f + 1]; }
    const float FLT_SCALE = 1.f / (1 << 20);
    const double eps2 = 0.01 * 0.01;
    uint8_t* Iw_ = Ipat[wk];
    uint8_t* Jw = Jreg[wk];
    const bool wrow = r < 15;            // the lane has a window row
    float nx = 0, ny = 0;
    int st = 1;
#pragma unroll 1
    for (int level = 3; level >= 0; --level) {
        if (level >= levels) continue;
        // (a run-time index into the by-value pyramid descriptors would put both of them into scratch: 200 bytes per lane written and read back
        //  by every wave — 96 MB per launch of 128 images by the write counter; constant indices + selects stay in scalar registers)
        const uint8_t* I = level == 0 ? prev.img[0] : level == 1 ? prev.img[1] : level == 2 ? prev.img[2] : prev.img[3];
        const uint8_t* J = level == 0 ? next.img[0] : level == 1 ? next.img[1] : level == 2 ? next.img[2] : next.img[3];
        const int w = level == 0 ? prev.w[0] : level == 1 ? prev.w[1] : level == 2 ? prev.w[2] : prev.w[3];
        const int h = level == 0 ? prev.h[0] : level == 1 ? prev.h[1] : level == 2 ? prev.h[2] : prev.h[3];
        const float sc = (float)(1. / (1 << level));
        float ppx = px * sc, ppy = py * sc;
        if (level == levels - 1) { nx = ppx; ny = ppy; } else { nx = nx * 2.f; ny = ny * 2.f; }
        ppx -= 7.f; ppy -= 7.f;
        const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
        bool lv = valid;
        if (ipx < -15 || ipx >= w || ipy < -15 || ipy >= h) { if (level == 0) st = 0; lv = false; }
        int jxl = ipx - 8, jyl = ipy - 8;
        __syncthreads();                 // (one wave: orders this level's staging behind the last level's reads for the compiler)
        if (lv) {
            k16_stage_i(Iw_, I, w, h, ipx - 1, ipy - 1, r);
            k16_stage_j(Jw, J, w, h, jxl, jyl, r);
        }
        __syncthreads();
        // calcSharrDeriv on the staged patch, derivative row r (all 16 lanes), packed (dx & 0xffff) | (dy << 16); outside the image the
        // derivative image is 0 (BORDER_CONSTANT); the template's bilinear window values and the spatial gradient matrix
        int Iw[15], Ixw[15], Iyw[15];
        float A11 = 0, A12 = 0, A22 = 0, D = 0;
        if (lv) {
            unsigned R0[5], R1[5], R2[5];          // patch rows r, r + 1, r + 2 (18 bytes each)
            const unsigned* pr = (const unsigned*)(Iw_ + r * K16_IS);
#pragma unroll
            for (int q = 0; q < 5; ++q) { R0[q] = pr[q]; R1[q] = pr[K16_IS / 4 + q]; R2[q] = pr[2 * (K16_IS / 4) + q]; }
            int dn[16], dq[16];
            const int Y = ipy + r;
#pragma unroll
            for (int xx = 0; xx < 16; ++xx) {
                const int X = ipx + xx;
                int g = 0;
                if (!(X < 0 || Y < 0 || X >= w || Y >= h)) {
                    const int a0 = k16_byte(R0, xx), a1 = k16_byte(R0, xx + 1), a2 = k16_byte(R0, xx + 2);
                    const int b0 = k16_byte(R1, xx), b2 = k16_byte(R1, xx + 2);
                    const int c0 = k16_byte(R2, xx), c1 = k16_byte(R2, xx + 1), c2 = k16_byte(R2, xx + 2);
                    const int t0m = (a0 + c0) * 3 + b0 * 10, t0p = (a2 + c2) * 3 + b2 * 10;
                    const int t1m = c0 - a0, t1c = c1 - a1, t1p = c2 - a2;
                    g = ((t0p - t0m) & 0xffff) | (((t1p + t1m) * 3 + t1c * 10) << 16);
                }
                dq[xx] = g;
            }
#pragma unroll
            for (int xx = 0; xx < 16; ++xx) dn[xx] = __builtin_amdgcn_update_dpp(0, dq[xx], 0x101, 0xf, 0xf, true);    // row_shl:1: derivative row r + 1
            const float a = ppx - ipx, b = ppy - ipy;
            const int iw00 = (int)rintf((1.f - a) * (1.f - b) * (1 << 14));
            const int iw01 = (int)rintf(a * (1.f - b) * (1 << 14));
            const int iw10 = (int)rintf((1.f - a) * b * (1 << 14));
            const int iw11 = (1 << 14) - iw00 - iw01 - iw10;
            int p11 = 0, p12 = 0, p22 = 0;
#pragma unroll
            for (int c = 0; c < 15; ++c) {
                // template pixel (c, r) of the window <-> patch (c + 1, r + 1)
                const int ival = descale(k16_byte(R1, c + 1) * iw00 + k16_byte(R1, c + 2) * iw01 + k16_byte(R2, c + 1) * iw10 + k16_byte(R2, c + 2) * iw11, 14 - 5);
                const int d00 = dq[c], d01 = dq[c + 1], d10 = dn[c], d11 = dn[c + 1];
                const int ixv = descale((short)(d00 & 0xffff) * iw00 + (short)(d01 & 0xffff) * iw01 + (short)(d10 & 0xffff) * iw10 + (short)(d11 & 0xffff) * iw11, 14);
                const int iyv = descale((d00 >> 16) * iw00 + (d01 >> 16) * iw01 + (d10 >> 16) * iw10 + (d11 >> 16) * iw11, 14);
                Iw[c] = wrow ? (short)ival : 0; Ixw[c] = wrow ? (short)ixv : 0; Iyw[c] = wrow ? (short)iyv : 0;
                p11 += Ixw[c] * Ixw[c]; p12 += Ixw[c] * Iyw[c]; p22 += Iyw[c] * Iyw[c];
            }
            const long long s11 = k16_row_sum(p11), s12 = k16_row_sum(p12), s22 = k16_row_sum(p22);
            A11 = k16_i64_to_f32(s11) * FLT_SCALE; A12 = k16_i64_to_f32(s12) * FLT_SCALE; A22 = k16_i64_to_f32(s22) * FLT_SCALE;
            D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * 15 * 15);
            if (minEig < 1e-3f || D < 1.1920929e-07f) { if (level == 0) st = 0; lv = false; }
        }
        D = 1.f / D;
        float npx = nx - 7.f, npy = ny - 7.f;
        float pdx = 0, pdy = 0;
        int j = 0;
        bool run = lv;
        while (__builtin_amdgcn_ballot_w64(run)) {
            if (run) {
                const int inx = (int)floorf(npx), iny = (int)floorf(npy);
                if (inx < -15 || inx >= w || iny < -15 || iny >= h) { if (level == 0) st = 0; run = false; }
                else {
                    int ox = inx - jxl, oy = iny - jyl;
                    if (ox < 0 || ox > 32 - 17 || oy < 0 || oy > 32 - 17) {   // window left the staged region: restage around it
                        jxl = inx - 8; jyl = iny - 8; ox = 8; oy = 8;
                        k16_stage_j(Jw, J, w, h, jxl, jyl, r);
                        // the lanes of this feature's 16-lane row now read region rows staged by OTHER lanes of the same wave: the LDS
                        // operations of a wave complete in order; the fences keep the compiler from moving those loads above the stores
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                    const float a = npx - inx, b = npy - iny;
                    const int iw00 = (int)rintf((1.f - a) * (1.f - b) * (1 << 14));
                    const int iw01 = (int)rintf(a * (1.f - b) * (1 << 14));
                    const int iw10 = (int)rintf((1.f - a) * b * (1 << 14));
                    const int iw11 = (1 << 14) - iw00 - iw01 - iw10;
                    // region rows oy + r, oy + r + 1, bytes ox .. ox + 15
                    const int sh = ox & 3;
                    const unsigned* jp = (const unsigned*)(Jw + (oy + r) * K16_JS) + (ox >> 2);
                    unsigned Ja[4], Jb[4];
                    {
                        unsigned t0[5], t1[5];
#pragma unroll
                        for (int q = 0; q < 5; ++q) { t0[q] = jp[q]; t1[q] = jp[K16_JS / 4 + q]; }
#pragma unroll
                        for (int q = 0; q < 4; ++q) { Ja[q] = __builtin_amdgcn_alignbyte(t0[q + 1], t0[q], sh); Jb[q] = __builtin_amdgcn_alignbyte(t1[q + 1], t1[q], sh); }
                    }
                    int pb1 = 0, pb2 = 0;
#pragma unroll
                    for (int c = 0; c < 15; ++c) {
                        // (24-bit multiply-adds through inline assembly, as klt_kernel3 has them, were measured here in round 6 and are SLOWER: 143.7 -> 139.5 k frames/s at 128
                        //  streams — the compiler folds the byte extraction into the multiply's operand select, which an asm operand forbids)
                        const int diff = descale(k16_byte(Ja, c) * iw00 + k16_byte(Ja, c + 1) * iw01 + k16_byte(Jb, c) * iw10 + k16_byte(Jb, c + 1) * iw11, 14 - 5) - Iw[c];
                        pb1 += diff * Ixw[c]; pb2 += diff * Iyw[c];
                    }
                    if (!wrow) { pb1 = 0; pb2 = 0; }
                    const long long sb1 = k16_row_sum(pb1), sb2 = k16_row_sum(pb2);
                    const float b1 = k16_i64_to_f32(sb1) * FLT_SCALE, b2 = k16_i64_to_f32(sb2) * FLT_SCALE;
                    const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
                    npx += dx; npy += dy;
                    nx = npx + 7.f; ny = npy + 7.f;
                    if ((double)dx * dx + (double)dy * dy <= eps2) run = false;
                    else if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) { nx -= dx * 0.5f; ny -= dy * 0.5f; run = false; }
                    else { pdx = dx; pdy = dy; run = ++j < 30; }
                }
            }
        }
        if (lv && st && level == 0) {
            const float fx = nx - 7.f, fy = ny - 7.f;
            const int rx = (int)rintf(fx), ry = (int)rintf(fy);
            if (rx < -15 || rx >= w || ry < -15 || ry >= h) st = 0;
        }
    }
    if (valid && r == 0) { out[2 * f] = nx; out[2 * f + 1] = ny; status[f] = (unsigned char)st; }
}
