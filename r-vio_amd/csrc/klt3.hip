// klt3.hip — LKTrackerInvoker (cv::calcOpticalFlowPyrLK as called at Tracker.cc:237-244), generation 3.
// Included inside the FP-contraction-off region of rvio_hip.hip: bit-identical to oracle/frontend.cpp.
//
// One wave per feature, lane l owns window pixels p = l + 64 q (q < 4, p < 225).  Changes over klt_kernel:
//   * ALL pyramid levels' template sources (an 18x18 u8 patch: the 16x16 the bilinear template reads plus the one-pixel ring its
//     Scharr derivatives need — calcSharrDeriv is applied to the staged patch, no derivative image exists) and 32x32 search regions are
//     fetched in ONE batch at kernel start (the template positions depend only on the input point; the search
//     regions are centred on the zero-motion guess and restaged only if the window leaves them) — one HBM/L2 round
//     trip instead of one per level (profiles/r01_b: ~3 us per level);
//   * the sums of products are reduced in 32-bit integers inside each 16-lane DPP row (|sum| < 2^31 by construction:
//     4 x 8160 x 4080 x 16 = 2.13e9) and only the four row totals are combined in 64 bits — exact, 4 DPP adds each.
#pragma once
#include "rvio_dev.h"
#include "frontend_dev.h"

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// exact sum over the wave of values whose 16-lane partial sums fit in int32
__device__ __forceinline__ long long wave_sum_i32rows(int v) {
    v += dpp_i32<0x128>(v);
    v += dpp_i32<0x124>(v);
    v += dpp_i32<0x122>(v);
    v += dpp_i32<0x121>(v);
    return ((long long)__builtin_amdgcn_readlane(v, 0) + (long long)__builtin_amdgcn_readlane(v, 16)) +
           ((long long)__builtin_amdgcn_readlane(v, 32) + (long long)__builtin_amdgcn_readlane(v, 48));
}

#define KLT3_JR 32
__device__ __forceinline__ void klt3_stage_j(uint8_t* Jr, const uint8_t* __restrict__ J, int w, int h, int jx0, int jy0, int lane) {
    const int r = lane >> 1, c0 = (lane & 1) * 16;
    const uint8_t* jrow = J + (size_t)reflect2(jy0 + r, h) * w;
    unsigned pk[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned v = 0;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) v |= (unsigned)jrow[reflect2(jx0 + c0 + 4 * g + bb, w)] << (8 * bb);
        pk[g] = v;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) ((unsigned*)Jr)[(r * KLT3_JR + c0) / 4 + g] = pk[g];
}

__global__ __launch_bounds__(64) void klt_kernel3(PyrDev prev, PyrDev next, int levels, const int* __restrict__ n_pts_ptr,
                                                  const float* __restrict__ pts, float* __restrict__ out, unsigned char* __restrict__ status, size_t bs) {
    DBG_S(blockIdx.x == 0 && blockIdx.z == 0, 0);
    pyr_shift(prev, (size_t)blockIdx.z * bs); pyr_shift(next, (size_t)blockIdx.z * bs);
    n_pts_ptr = zoff(n_pts_ptr, bs); pts = zoff(pts, bs); out = zoff(out, bs); status = zoff(status, bs);
    __shared__ uint8_t Ip[4][18 * 18 + 4];   // (Y0 - 1 .. Y0 + 16) x (X0 - 1 .. X0 + 16), reflect-101 coordinates
    __shared__ int dIp[4][16 * 16];
    __shared__ uint8_t Jr[4][KLT3_JR * KLT3_JR];
    const int f = blockIdx.x, lane = threadIdx.x;
    const float px = pts[2 * f], py = pts[2 * f + 1];
    if (f >= *n_pts_ptr) return;
    const float FLT_SCALE = 1.f / (1 << 20);
    const double eps2 = 0.01 * 0.01;
    // ---- prologue: one batch of global loads for every level
    int jx0[4], jy0[4];
#pragma unroll
    for (int level = 0; level < 4; ++level) {
        jx0[level] = 0; jy0[level] = 0;
        if (level < levels) {
            const int w = prev.w[level], h = prev.h[level];
            const float sc = (float)(1. / (1 << level));
            const float ppx = px * sc - 7.f, ppy = py * sc - 7.f;
            const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
            jx0[level] = ipx - 8; jy0[level] = ipy - 8;
            if (!(ipx < -15 || ipx >= w || ipy < -15 || ipy >= h)) {
                const uint8_t* I = prev.img[level];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int e = lane + 64 * q;
                    if (e < 18 * 18) {
                        const int X = ipx - 1 + (e % 18), Y = ipy - 1 + (e / 18);
                        Ip[level][e] = I[(size_t)reflect2(Y, h) * w + reflect2(X, w)];
                    }
                }
                klt3_stage_j(Jr[level], next.img[level], w, h, jx0[level], jy0[level], lane);
            }
        }
    }
    __syncthreads();
    // calcSharrDeriv on the staged patches: derivative at template pixel (X, Y); outside the image the derivative image is 0
    // (BORDER_CONSTANT), inside it the neighbours are the reflect-101 ones the patch already holds
#pragma unroll
    for (int level = 0; level < 4; ++level) {
        if (level < levels) {
            const int w = prev.w[level], h = prev.h[level];
            const float sc = (float)(1. / (1 << level));
            const int ipx = (int)floorf(px * sc - 7.f), ipy = (int)floorf(py * sc - 7.f);
            if (!(ipx < -15 || ipx >= w || ipy < -15 || ipy >= h)) {
                const uint8_t* P = Ip[level];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = lane + 64 * q, xx = e & 15, yy = e >> 4, X = ipx + xx, Y = ipy + yy;
                    const uint8_t* c = P + (yy + 1) * 18 + (xx + 1);
                    int g = 0;
                    if (!(X < 0 || Y < 0 || X >= w || Y >= h)) {
                        const int a0 = c[-19], a1 = c[-18], a2 = c[-17], b0 = c[-1], b2 = c[1], c0 = c[17], c1 = c[18], c2 = c[19];
                        const int t0m = (a0 + c0) * 3 + b0 * 10, t0p = (a2 + c2) * 3 + b2 * 10;
                        const int t1m = c0 - a0, t1c = c1 - a1, t1p = c2 - a2;
                        g = ((t0p - t0m) & 0xffff) | (((t1p + t1m) * 3 + t1c * 10) << 16);
                    }
                    dIp[level][e] = g;
                }
            }
        }
    }
    __syncthreads();
    float nx = 0, ny = 0;
    int st = 1;
    int wo16[4], wo18[4], woJ[4];      // this lane's window pixels as offsets into the 16-wide gradient patch / the 18-wide template patch / the 32-wide region
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int p = lane + 64 * q; wo16[q] = (p / 15) * 16 + (p % 15); wo18[q] = (p / 15 + 1) * 18 + (p % 15) + 1; woJ[q] = (p / 15) * KLT3_JR + (p % 15); }
#pragma unroll
    for (int level = 3; level >= 0; --level) {
        if (level >= levels) continue;
        const uint8_t* J = next.img[level];
        const int w = prev.w[level], h = prev.h[level];
        const float sc = (float)(1. / (1 << level));
        float ppx = px * sc, ppy = py * sc;
        if (level == levels - 1) { nx = ppx; ny = ppy; } else { nx = nx * 2.f; ny = ny * 2.f; }
        ppx -= 7.f; ppy -= 7.f;
        const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
        if (ipx < -15 || ipx >= w || ipy < -15 || ipy >= h) { if (level == 0) st = 0; continue; }
        float a = ppx - ipx, b = ppy - ipy;
        int iw00 = (int)rintf((1.f - a) * (1.f - b) * (1 << 14));
        int iw01 = (int)rintf(a * (1.f - b) * (1 << 14));
        int iw10 = (int)rintf((1.f - a) * b * (1 << 14));
        int iw11 = (1 << 14) - iw00 - iw01 - iw10;
        const uint8_t* Il = Ip[level]; const int* dIl = dIp[level]; uint8_t* Jl = Jr[level];
        int Iw[4], Ixw[4], Iyw[4];
        int p11 = 0, p12 = 0, p22 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Iw[q] = 0; Ixw[q] = 0; Iyw[q] = 0;
            if (lane + 64 * q < 225) {
                const int o = wo16[q], o8 = wo18[q];
                const int ival = descale(Il[o8] * iw00 + Il[o8 + 1] * iw01 + Il[o8 + 18] * iw10 + Il[o8 + 19] * iw11, 14 - 5);
                const int d00 = dIl[o], d01 = dIl[o + 1], d10 = dIl[o + 16], d11 = dIl[o + 17];
                const int ixv = descale((short)(d00 & 0xffff) * iw00 + (short)(d01 & 0xffff) * iw01 + (short)(d10 & 0xffff) * iw10 + (short)(d11 & 0xffff) * iw11, 14);
                const int iyv = descale((d00 >> 16) * iw00 + (d01 >> 16) * iw01 + (d10 >> 16) * iw10 + (d11 >> 16) * iw11, 14);
                Iw[q] = (short)ival; Ixw[q] = (short)ixv; Iyw[q] = (short)iyv;
                p11 += Ixw[q] * Ixw[q]; p12 += Ixw[q] * Iyw[q]; p22 += Iyw[q] * Iyw[q];
            }
        }
        const long long s11 = wave_sum_i32rows(p11), s12 = wave_sum_i32rows(p12), s22 = wave_sum_i32rows(p22);
        const float A11 = (float)s11 * FLT_SCALE, A12 = (float)s12 * FLT_SCALE, A22 = (float)s22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * 15 * 15);
        if (minEig < 1e-3f || D < 1.1920929e-07f) { if (level == 0) st = 0; continue; }
        D = 1.f / D;
        float npx = nx - 7.f, npy = ny - 7.f;
        float pdx = 0, pdy = 0;
        int jxl = jx0[level], jyl = jy0[level];
        for (int j = 0; j < 30; ++j) {
            const int inx = (int)floorf(npx), iny = (int)floorf(npy);
            if (inx < -15 || inx >= w || iny < -15 || iny >= h) { if (level == 0) st = 0; break; }
            int ox = inx - jxl, oy = iny - jyl;
            if (ox < 0 || ox > KLT3_JR - 17 || oy < 0 || oy > KLT3_JR - 17) {   // window left the staged region: restage around it
                jxl = inx - 8; jyl = iny - 8; ox = 8; oy = 8;
                __syncthreads();
                klt3_stage_j(Jl, J, w, h, jxl, jyl, lane);
                __syncthreads();
            }
            a = npx - inx; b = npy - iny;
            iw00 = (int)rintf((1.f - a) * (1.f - b) * (1 << 14));
            iw01 = (int)rintf(a * (1.f - b) * (1 << 14));
            iw10 = (int)rintf((1.f - a) * b * (1 << 14));
            iw11 = (1 << 14) - iw00 - iw01 - iw10;
            const int ob = oy * KLT3_JR + ox;
            int pb1 = 0, pb2 = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (lane + 64 * q < 225) {
                    const int o = ob + woJ[q];
                    const int diff = descale(Jl[o] * iw00 + Jl[o + 1] * iw01 + Jl[o + KLT3_JR] * iw10 + Jl[o + KLT3_JR + 1] * iw11, 14 - 5) - Iw[q];
                    pb1 += diff * Ixw[q]; pb2 += diff * Iyw[q];
                }
            }
            const long long sb1 = wave_sum_i32rows(pb1), sb2 = wave_sum_i32rows(pb2);
            const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
            npx += dx; npy += dy;
            nx = npx + 7.f; ny = npy + 7.f;
            if ((double)dx * dx + (double)dy * dy <= eps2) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) { nx -= dx * 0.5f; ny -= dy * 0.5f; break; }
            pdx = dx; pdy = dy;
        }
        if (st && level == 0) {
            const float fx = nx - 7.f, fy = ny - 7.f;
            const int rx = (int)rintf(fx), ry = (int)rintf(fy);
            if (rx < -15 || rx >= w || ry < -15 || ry >= h) st = 0;
        }
    }
    if (lane == 0) { out[2 * f] = nx; out[2 * f + 1] = ny; status[f] = (unsigned char)st; }
}
