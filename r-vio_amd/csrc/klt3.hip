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

// Round 6, what the phase stamps (tools/side_phase_clocks.py) and the disassembly said about this kernel:
//   * the iteration is an instruction-issue chain of ONE wave (~0.85 us per iteration, ~350 instructions of which the 32-bit integer multiplies
//     ran at quarter rate and the four pixel groups waited for their LDS loads one after the other behind exec masks).  Every product of the
//     window sums has operands of at most 24 significant bits — pixels < 2^8, bilinear weights <= 2^14, Scharr responses and interpolated values
//     are shorts, differences < 2^13 — so the full-rate 24-bit multiplies are exact; lanes beyond pixel 224 read a valid address and carry zero
//     gradient weights, so nothing is predicated and all loads of an iteration are in flight together;
//   * the FIXED part (staging 8 us, Scharr 2 us, 4 x 2.6 us of template set-up: 20 of the ~25 us of a frame at rest) was straight-line code run
//     once — 10 k instructions = 80 KB, unrolled over the four levels, fetched cold at ~100 cycles per 64-byte line: the levels are loops now
//     (one copy of the template / iteration / Scharr code, warm from the second level on), and a staged region that lies inside the image —
//     almost all do — is fetched with ONE unaligned 16-byte load per lane instead of 16 reflected byte loads.
#ifndef KLT_MUL24
#define KLT_MUL24 1
#endif
#if KLT_MUL24
__device__ __forceinline__ int mul24(int a, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int mad24(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
#else
__device__ __forceinline__ int mul24(int a, int b) { return a * b; }
__device__ __forceinline__ int mad24(int a, int b, int c) { return a * b + c; }
#endif
// exact sum over the wave of values whose 16-lane partial sums fit in int32, rounded ONCE to float like (float)(long long): the four row totals
// are added as doubles (exact below 2^53) and converted with round-to-nearest-even, which is what the int64 -> float conversion does
__device__ __forceinline__ float wave_sum_i32rows_f32(int v) {
    v += dpp_i32<0x128>(v);
    v += dpp_i32<0x124>(v);
    v += dpp_i32<0x122>(v);
    v += dpp_i32<0x121>(v);
    const double t = ((double)__builtin_amdgcn_readlane(v, 0) + (double)__builtin_amdgcn_readlane(v, 16)) +
                     ((double)__builtin_amdgcn_readlane(v, 32) + (double)__builtin_amdgcn_readlane(v, 48));
    return (float)t;
}
template <typename T>
__device__ __forceinline__ T sel4(const T* a, int l) { return l == 0 ? a[0] : (l == 1 ? a[1] : (l == 2 ? a[2] : a[3])); }
typedef unsigned klt_u4 __attribute__((ext_vector_type(4)));

#define KLT3_JR 32
#define KLT3_IS 20      // row stride of the staged template patch (18 bytes used: dword-aligned rows)
// a 32 x 32 region of J with its top-left corner at (jx0, jy0), reflect-101 outside the image: lane <-> (row, 16-byte half)
__device__ __forceinline__ bool klt3_inside(int w, int h, int x0, int y0, int n) { return x0 >= 0 && y0 >= 0 && x0 + n <= w && y0 + n <= h; }
__device__ __forceinline__ klt_u4 klt3_load_j_fast(const uint8_t* __restrict__ J, int w, int jx0, int jy0, int lane) {
    klt_u4 v;
    __builtin_memcpy(&v, J + (size_t)(jy0 + (lane >> 1)) * w + jx0 + (lane & 1) * 16, 16);   // (one unaligned global_load_dwordx4)
    return v;
}
__device__ __forceinline__ void klt3_store_j(uint8_t* Jr, klt_u4 v, int lane) { ((klt_u4*)Jr)[lane] = v; }   // byte offset (lane >> 1) * 32 + (lane & 1) * 16 = 16 lane
__device__ __forceinline__ void klt3_stage_j_slow(uint8_t* Jr, const uint8_t* __restrict__ J, int w, int h, int jx0, int jy0, int lane) {
    const int r = lane >> 1, c0 = (lane & 1) * 16;
    const uint8_t* jrow = J + (size_t)reflect2(jy0 + r, h) * w;
    for (int g = 0; g < 4; ++g) {
        unsigned v = 0;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) v |= (unsigned)jrow[reflect2(jx0 + c0 + 4 * g + bb, w)] << (8 * bb);
        ((unsigned*)Jr)[(r * KLT3_JR + c0) / 4 + g] = v;
    }
}
__device__ __forceinline__ void klt3_stage_j(uint8_t* Jr, const uint8_t* __restrict__ J, int w, int h, int jx0, int jy0, int lane) {
    if (klt3_inside(w, h, jx0, jy0, KLT3_JR)) klt3_store_j(Jr, klt3_load_j_fast(J, w, jx0, jy0, lane), lane);
    else klt3_stage_j_slow(Jr, J, w, h, jx0, jy0, lane);
}
// the 18 x 18 template source (Y0 - 1 .. Y0 + 16) x (X0 - 1 .. X0 + 16), reflect-101 coordinates, rows KLT3_IS bytes apart
__device__ __forceinline__ void klt3_stage_i_slow(uint8_t* Ipl, const uint8_t* __restrict__ I, int w, int h, int ipx, int ipy, int lane) {
    for (int q = 0; q < 6; ++q) {
        const int e = lane + 64 * q;
        if (e < 18 * 18) {
            const int X = ipx - 1 + (e % 18), Y = ipy - 1 + (e / 18);
            Ipl[(e / 18) * KLT3_IS + (e % 18)] = I[(size_t)reflect2(Y, h) * w + reflect2(X, w)];
        }
    }
}

__global__ __launch_bounds__(64) void klt_kernel3(PyrDev prev, PyrDev next, int levels, const int* __restrict__ n_pts_ptr,
                                                  const float* __restrict__ pts, float* __restrict__ out, unsigned char* __restrict__ status, size_t bs,
                                                  const unsigned long long* pyr_ready, unsigned long long pyr_target, FilterMeta* meta) {
    DBG_S(blockIdx.x == 0 && blockIdx.z == 0, 0);
    DBG_U(17);
    const size_t zo = (size_t)blockIdx.z * bs;
    n_pts_ptr = zoff(n_pts_ptr, bs); pts = zoff(pts, bs); out = zoff(out, bs); status = zoff(status, bs);
    __shared__ __align__(16) uint8_t Ip[4][18 * KLT3_IS];
    __shared__ int dIp[4][16 * 16];
    __shared__ __align__(16) uint8_t Jr[4][KLT3_JR * KLT3_JR];
    const int f = blockIdx.x, lane = threadIdx.x;
    const float px = pts[2 * f], py = pts[2 * f + 1];
    const int n_pts = *n_pts_ptr;
    // pyr_ready (round 6, one pipelined stream): the new image's pyramid comes from the image chain's queue — instead of a stream-level event wait in front of this
    // launch (a barrier packet on the side chain's serial path, ~3 us per frame) every workgroup polls the chain's counter here, under its first loads' latency.
    // A wait that times out: the sticky error bit is set, nothing is written (status and positions stay as book-keeping left them) and the sequence fails at the next sync.
    if (pyr_ready && !stage_wait_wave(pyr_ready, pyr_target, meta)) return;
    if (f >= n_pts) return;
    if (levels > 4) levels = 4;
    const float FLT_SCALE = 1.f / (1 << 20);
    const double eps2 = 0.01 * 0.01;
    // ---- prologue: one batch of global loads for every level.  A level whose regions lie inside the image (the 32 x 32 region of J holds the
    // 20 x 18 bytes fetched of I) takes 1 + 2 loads per lane, all levels' in flight together; the others go through the reflected byte loads.
    {
        klt_u4 jv[4], iva[4];
        unsigned ivb[4];
        unsigned fast = 0, slow = 0;
#pragma unroll
        for (int level = 0; level < 4; ++level) {
            jv[level] = klt_u4{0, 0, 0, 0}; iva[level] = klt_u4{0, 0, 0, 0}; ivb[level] = 0;
            if (level < levels) {
                const int w = prev.w[level], h = prev.h[level];
                const float sc = (float)(1. / (1 << level));
                const int ipx = (int)floorf(px * sc - 7.f), ipy = (int)floorf(py * sc - 7.f);
                if (!(ipx < -15 || ipx >= w || ipy < -15 || ipy >= h)) {
                    if (klt3_inside(w, h, ipx - 8, ipy - 8, KLT3_JR)) {
                        fast |= 1u << level;
                        jv[level] = klt3_load_j_fast(next.img[level] + zo, w, ipx - 8, ipy - 8, lane);
                        if (lane < 18) {
                            const uint8_t* ip = prev.img[level] + zo + (size_t)(ipy - 1 + lane) * w + ipx - 1;
                            __builtin_memcpy(&iva[level], ip, 16);
                            __builtin_memcpy(&ivb[level], ip + 16, 4);
                        }
                    } else slow |= 1u << level;
                }
            }
        }
#pragma unroll
        for (int level = 0; level < 4; ++level)
            if (fast & (1u << level)) {
                klt3_store_j(Jr[level], jv[level], lane);
                if (lane < 18) {
                    unsigned* d = (unsigned*)Ip[level] + lane * (KLT3_IS / 4);
                    d[0] = iva[level].x; d[1] = iva[level].y; d[2] = iva[level].z; d[3] = iva[level].w; d[4] = ivb[level];
                }
            }
        if (slow) {
#pragma unroll 1
            for (int level = 0; level < levels; ++level)
                if (slow & (1u << level)) {
                    const int w = sel4(prev.w, level), h = sel4(prev.h, level);
                    const float sc = (float)(1. / (1 << level));
                    const int ipx = (int)floorf(px * sc - 7.f), ipy = (int)floorf(py * sc - 7.f);
                    klt3_stage_i_slow(Ip[level], sel4(prev.img, level) + zo, w, h, ipx, ipy, lane);
                    klt3_stage_j_slow(Jr[level], sel4(next.img, level) + zo, w, h, ipx - 8, ipy - 8, lane);
                }
        }
    }
    __syncthreads();
    DBG_U(18);
    // calcSharrDeriv on the staged patches: derivative at template pixel (X, Y); outside the image the derivative image is 0
    // (BORDER_CONSTANT), inside it the neighbours are the reflect-101 ones the patch already holds
#pragma unroll 1
    for (int level = 0; level < levels; ++level) {
        const int w = sel4(prev.w, level), h = sel4(prev.h, level);
        const float sc = (float)(1. / (1 << level));
        const int ipx = (int)floorf(px * sc - 7.f), ipy = (int)floorf(py * sc - 7.f);
        if (!(ipx < -15 || ipx >= w || ipy < -15 || ipy >= h)) {
            const uint8_t* P = Ip[level];
            const bool all_in = ipx >= 0 && ipy >= 0 && ipx + 16 <= w && ipy + 16 <= h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = lane + 64 * q, xx = e & 15, yy = e >> 4, X = ipx + xx, Y = ipy + yy;
                const uint8_t* c = P + (yy + 1) * KLT3_IS + (xx + 1);
                const int a0 = c[-KLT3_IS - 1], a1 = c[-KLT3_IS], a2 = c[-KLT3_IS + 1], b0 = c[-1], b2 = c[1], c0 = c[KLT3_IS - 1], c1 = c[KLT3_IS], c2 = c[KLT3_IS + 1];
                const int t0m = (a0 + c0) * 3 + b0 * 10, t0p = (a2 + c2) * 3 + b2 * 10;
                const int t1m = c0 - a0, t1c = c1 - a1, t1p = c2 - a2;
                int g = ((t0p - t0m) & 0xffff) | (((t1p + t1m) * 3 + t1c * 10) << 16);
                if (!all_in && (X < 0 || Y < 0 || X >= w || Y >= h)) g = 0;
                dIp[level][e] = g;
            }
        }
    }
    __syncthreads();
    DBG_U(19);
    float nx = 0, ny = 0;
    int st = 1;
    int wo16[4], wo18[4], woJ[4];      // this lane's window pixels as offsets into the 16-wide gradient patch / the template patch / the 32-wide region
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // (lanes beyond the 225 window pixels: pixel 0's addresses, their gradient weights are zeroed below)
        const int p = (lane + 64 * q < 225) ? lane + 64 * q : 0;
        wo16[q] = (p / 15) * 16 + (p % 15); wo18[q] = (p / 15 + 1) * KLT3_IS + (p % 15) + 1; woJ[q] = (p / 15) * KLT3_JR + (p % 15);
    }
#pragma unroll 1
    for (int level = levels - 1; level >= 0; --level) {
        const uint8_t* J = sel4(next.img, level) + zo;
        const int w = sel4(prev.w, level), h = sel4(prev.h, level);
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
        {
            int tI[4][4], tD[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = wo16[q], o8 = wo18[q];
                tI[q][0] = Il[o8]; tI[q][1] = Il[o8 + 1]; tI[q][2] = Il[o8 + KLT3_IS]; tI[q][3] = Il[o8 + KLT3_IS + 1];
                tD[q][0] = dIl[o]; tD[q][1] = dIl[o + 1]; tD[q][2] = dIl[o + 16]; tD[q][3] = dIl[o + 17];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool on = lane + 64 * q < 225;
                const int ival = descale(mad24(tI[q][0], iw00, mad24(tI[q][1], iw01, mad24(tI[q][2], iw10, mul24(tI[q][3], iw11)))), 14 - 5);
                const int ixv = descale(mad24((short)(tD[q][0] & 0xffff), iw00, mad24((short)(tD[q][1] & 0xffff), iw01,
                                        mad24((short)(tD[q][2] & 0xffff), iw10, mul24((short)(tD[q][3] & 0xffff), iw11)))), 14);
                const int iyv = descale(mad24(tD[q][0] >> 16, iw00, mad24(tD[q][1] >> 16, iw01, mad24(tD[q][2] >> 16, iw10, mul24(tD[q][3] >> 16, iw11)))), 14);
                Iw[q] = on ? (int)(short)ival : 0; Ixw[q] = on ? (int)(short)ixv : 0; Iyw[q] = on ? (int)(short)iyv : 0;
                p11 = mad24(Ixw[q], Ixw[q], p11); p12 = mad24(Ixw[q], Iyw[q], p12); p22 = mad24(Iyw[q], Iyw[q], p22);
            }
        }
        const float A11 = wave_sum_i32rows_f32(p11) * FLT_SCALE, A12 = wave_sum_i32rows_f32(p12) * FLT_SCALE, A22 = wave_sum_i32rows_f32(p22) * FLT_SCALE;
        DBG_U(20 + 2 * (3 - level));
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * 15 * 15);
        if (minEig < 1e-3f || D < 1.1920929e-07f) { if (level == 0) st = 0; continue; }
        D = 1.f / D;
        float npx = nx - 7.f, npy = ny - 7.f;
        float pdx = 0, pdy = 0;
        int jxl = ipx - 8, jyl = ipy - 8;
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
            {
                int tJ[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = ob + woJ[q];
                    tJ[q][0] = Jl[o]; tJ[q][1] = Jl[o + 1]; tJ[q][2] = Jl[o + KLT3_JR]; tJ[q][3] = Jl[o + KLT3_JR + 1];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int diff = descale(mad24(tJ[q][0], iw00, mad24(tJ[q][1], iw01, mad24(tJ[q][2], iw10, mul24(tJ[q][3], iw11)))), 14 - 5) - Iw[q];
                    pb1 = mad24(diff, Ixw[q], pb1); pb2 = mad24(diff, Iyw[q], pb2);
                }
            }
            const float b1 = wave_sum_i32rows_f32(pb1) * FLT_SCALE, b2 = wave_sum_i32rows_f32(pb2) * FLT_SCALE;
            const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
            npx += dx; npy += dy;
            nx = npx + 7.f; ny = npy + 7.f;
            if ((double)dx * dx + (double)dy * dy <= eps2) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) { nx -= dx * 0.5f; ny -= dy * 0.5f; break; }
            pdx = dx; pdy = dy;
        }
        DBG_U(21 + 2 * (3 - level));
        if (st && level == 0) {
            const float fx = nx - 7.f, fy = ny - 7.f;
            const int rx = (int)rintf(fx), ry = (int)rintf(fy);
            if (rx < -15 || rx >= w || ry < -15 || ry >= h) st = 0;
        }
    }
    if (lane == 0) { out[2 * f] = nx; out[2 * f + 1] = ny; status[f] = (unsigned char)st; }
    DBG_U(28);
}
