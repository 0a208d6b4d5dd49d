// detector.hip — T7: FeatureDetector::DetectWithSubPix (FeatureDetector.cc:55-75) =
//   cv::goodFeaturesToTrack(im, corners, nFeatures, qualityLevel, s*minDistance)   + cv::cornerSubPix(win 7x7, 30 it, 0.01)
// restated for the device, bit-exact against oracle/detector.cpp (same float/double expression order, no contraction; this
// file is included inside the fp-contract(off) region).  s = 1 on the first image, 2 on refills (Tracker.cc:207,350): the
// first-image flag lives on the device, so the kernels pick s themselves.
//   mineig_kernel     Sobel 3x3 (scaled) -> products -> 3x3 box (double) -> lambda_min, + image maximum (atomic on an ordered key)
//   nms_kernel        threshold at max*quality, strict 3x3 local maxima away from the border -> candidate list + per-cell buckets
//   greedy_kernel     OpenCV's sequential min-distance selection in descending-strength order, computed as the
//                     lexicographically-first maximal independent set by priority rounds (a candidate is taken once no
//                     stronger undecided candidate lies within the distance; it is dropped once a taken one does): the
//                     fixpoint equals the sequential result; then rank-by-counting keeps the strongest nFeatures in order
//   subpix_kernel     one wave per corner: 17x17 bilinear patch in LDS, lane <-> row of the 15x15 window, double sums in
//                     the oracle's canonical order (row sums left->right, rows top->bottom), 2x2 solve, <= 30 iterations
#pragma once

struct DetDev {
    const int* first;        // Tracker's mbIsTheFirstImage (device)
    float* eig;              // W*H
    int* maxkey;             // ordered-int key of the image maximum
    int* counters;           // [0] n candidates, [1] n accepted, [2] n output corners
    int* cell_cnt;           // [cells at s=1]
    unsigned long long* cell_ent;   // bucketed candidate keys  [(W+64)*(H+64)]
    unsigned long long* cand;       // flat candidate keys      [W*H]
    unsigned long long* acc;        // accepted keys            [W*H]
    unsigned char* state;           // per pixel: 1 undecided, 2 taken, 3 dropped (only candidate pixels are ever read)
    float* raw_xy;           // goodFeaturesToTrack output [F][2]
    float* xy;               // after cornerSubPix         [F][2]
    const float* spmask;     // 15x15 Gaussian window of cornerSubPix (host-computed: expf is glibc's)
    int W, H, F;
    float min_dist;          // Tracker.nMinDist
    double quality;          // (double)(float)Tracker.nQualLvl
    int max_cells;
};

__device__ __forceinline__ int f2ord(float f) { const int b = __float_as_int(f); return b >= 0 ? b : (b ^ 0x7fffffff); }
__device__ __forceinline__ float ord2f(int k) { return __int_as_float(k >= 0 ? k : (k ^ 0x7fffffff)); }

#define DET_TW 64
#define DET_TH 4
__global__ __launch_bounds__(256) void mineig_kernel(const uint8_t* __restrict__ src, int stride, DetDev d) {
    __shared__ float sdx[DET_TH + 2][DET_TW + 2], sdy[DET_TH + 2][DET_TW + 2];
    __shared__ int s_max[4];
    const int W = d.W, H = d.H;
    const int tid = threadIdx.x, x0 = blockIdx.x * DET_TW, y0 = blockIdx.y * DET_TH;
    if (blockIdx.x == 0 && blockIdx.y == 0) {          // per-frame reset of the detector's counters (nothing in this kernel reads them)
        for (int i = tid; i < d.max_cells; i += 256) d.cell_cnt[i] = 0;
        if (tid < 3) d.counters[tid] = 0;
    }
    const double scale = 1.0 / (4.0 * 3.0 * 255.0);
    const float k1 = (float)scale, k0 = (float)(2.0 * scale);
    // gradients on the (TW+2) x (TH+2) halo; positions outside the image take the gradient AT the reflected position
    for (int e = tid; e < (DET_TW + 2) * (DET_TH + 2); e += 256) {
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
        for (int k = 1; k < 4; ++k) m = s_max[k] > m ? s_max[k] : m;
        atomicMax(d.maxkey, m);
    }
}

__device__ __forceinline__ void det_geometry(const DetDev& d, float* md, int* cell, int* gw, int* gh) {
    const float s = (*d.first) ? 1.f : 2.f;
    *md = s * d.min_dist;                                    // s*mnMinDistance (int * float)
    *cell = (int)rintf(*md);                                 // cvRound(minDistance)
    *gw = (d.W + *cell - 1) / *cell; *gh = (d.H + *cell - 1) / *cell;
}

__global__ __launch_bounds__(256) void nms_kernel(DetDev d) {
    const int W = d.W, H = d.H;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
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
    d.state[idx] = 1;
    d.cand[atomicAdd(&d.counters[0], 1)] = key;
    const int c = (y / cell) * gw + (x / cell);
    d.cell_ent[(size_t)c * cell * cell + atomicAdd(&d.cell_cnt[c], 1)] = key;
}

#define GREEDY_T 1024
__global__ __launch_bounds__(GREEDY_T) void greedy_kernel(DetDev d) {
    const int W = d.W, tid = threadIdx.x;
    float md; int cell, gw, gh;
    det_geometry(d, &md, &cell, &gw, &gh);
    const double md2 = (double)md * (double)md;
    const int n = d.counters[0];
    volatile unsigned char* st = d.state;
    const size_t cap = (size_t)cell * cell;
    int pending;
    do {
        pending = 0;
        for (int c = tid; c < n; c += GREEDY_T) {
            const unsigned long long key = d.cand[c];
            const int idx = (int)(key & 0xffffffffull);
            if (st[idx] != 1) continue;
            const int x = idx % W, y = idx / W, xc = x / cell, yc = y / cell;
            const int x1 = xc > 0 ? xc - 1 : 0, y1 = yc > 0 ? yc - 1 : 0, x2 = xc + 1 < gw ? xc + 1 : gw - 1, y2 = yc + 1 < gh ? yc + 1 : gh - 1;
            bool drop = false, wait = false;
            for (int yy = y1; yy <= y2; ++yy)
                for (int xx = x1; xx <= x2; ++xx) {
                    const int cc = yy * gw + xx, cnt = d.cell_cnt[cc];
                    const unsigned long long* ent = d.cell_ent + (size_t)cc * cap;
                    for (int e = 0; e < cnt; ++e) {
                        const unsigned long long k2 = ent[e];
                        if (k2 == key) continue;
                        const int i2 = (int)(k2 & 0xffffffffull);
                        const float ddx = (float)x - (float)(i2 % W), ddy = (float)y - (float)(i2 / W);
                        if ((double)(ddx * ddx + ddy * ddy) < md2) {
                            const unsigned char s2 = st[i2];
                            if (s2 == 2) drop = true;
                            else if (s2 == 1 && k2 > key) wait = true;
                        }
                    }
                }
            if (drop) st[idx] = 3;
            else if (!wait) st[idx] = 2;
            else pending = 1;
        }
        __threadfence_block();
        pending = __syncthreads_or(pending);
    } while (pending);
    // taken candidates -> list; rank by counting; the strongest F in descending order
    for (int c = tid; c < n; c += GREEDY_T) {
        const unsigned long long key = d.cand[c];
        if (st[(int)(key & 0xffffffffull)] == 2) d.acc[atomicAdd(&d.counters[1], 1)] = key;
    }
    __threadfence_block();
    __syncthreads();
    const int na = ((volatile int*)d.counters)[1];
    for (int a = tid; a < na; a += GREEDY_T) {
        const unsigned long long key = d.acc[a];
        int r = 0;
        for (int b = 0; b < na; ++b) r += (d.acc[b] > key) ? 1 : 0;
        if (r < d.F) {
            const int idx = (int)(key & 0xffffffffull);
            d.raw_xy[2 * r] = (float)(idx % W); d.raw_xy[2 * r + 1] = (float)(idx / W);
        }
    }
    if (tid == 0) {
        d.counters[2] = na < d.F ? na : d.F;
        *d.maxkey = (int)0x80000000;                       // consumed by nms_kernel; ready for the next image
    }
}

#define SP_WIN 7
#define SP_WW (2 * SP_WIN + 1)
#define SP_PW (SP_WW + 2)
__global__ __launch_bounds__(64) void subpix_kernel(const uint8_t* __restrict__ src, int stride, DetDev d) {
    __shared__ float patch[SP_PW * SP_PW];
    __shared__ float smask[SP_WW * SP_WW];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int n = d.counters[2];
    if (p >= n) return;
    const int W = d.W, H = d.H;
    for (int e = lane; e < SP_WW * SP_WW; e += 64) smask[e] = d.spmask[e];
    const float tx = d.raw_xy[2 * p], ty = d.raw_xy[2 * p + 1];
    float cx = tx, cy = ty;
    const double eps = 1e-2 * 1e-2;
    int iter = 0;
    double err = 0;
    do {
        // getRectSubPix: 17x17 bilinear patch, replicated border
        {
            const float ox = cx - (float)(SP_PW - 1) * 0.5f, oy = cy - (float)(SP_PW - 1) * 0.5f;
            const int ix = (int)floorf(ox), iy = (int)floorf(oy);
            float a = ox - (float)ix;
            const float b = oy - (float)iy;
            a = fmaxf(a, 0.0001f);
            const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
            for (int e = lane; e < SP_PW * SP_PW; e += 64) {
                const int i = e / SP_PW, j = e % SP_PW;
                const int ya = min(max(iy + i, 0), H - 1), yb = min(max(iy + i + 1, 0), H - 1);
                const int xa = min(max(ix + j, 0), W - 1), xb = min(max(ix + j + 1, 0), W - 1);
                const uint8_t* r0 = src + (size_t)ya * stride;
                const uint8_t* r1 = src + (size_t)yb * stride;
                patch[e] = (((float)r0[xa] * a11 + (float)r0[xb] * a12) + (float)r1[xa] * a21) + (float)r1[xb] * a22;
            }
        }
        __syncthreads();
        // lane <-> window row: the row's five sums, left to right
        double ra = 0, rb = 0, rc = 0, r1s = 0, r2s = 0;
        if (lane < SP_WW) {
            const float* sp = &patch[(lane + 1) * SP_PW + 1];
            const double py = lane - SP_WIN;
#pragma unroll
            for (int j = 0; j < SP_WW; ++j) {
                const double m = smask[lane * SP_WW + j];
                const double tgx = sp[j + 1] - sp[j - 1];
                const double tgy = sp[j + SP_PW] - sp[j - SP_PW];
                const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                const double px = j - SP_WIN;
                ra += gxx; rb += gxy; rc += gyy;
                r1s += gxx * px + gxy * py;
                r2s += gxy * px + gyy * py;
            }
        }
        // rows top to bottom (uniform result on every lane)
        double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
#pragma unroll
        for (int i = 0; i < SP_WW; ++i) {
            a += readlane_f64(ra, i); b += readlane_f64(rb, i); c += readlane_f64(rc, i);
            bb1 += readlane_f64(r1s, i); bb2 += readlane_f64(r2s, i);
        }
        __syncthreads();                                     // patch is rewritten by the next iteration
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
    if (fabsf(cx - tx) > SP_WIN || fabsf(cy - ty) > SP_WIN) { cx = tx; cy = ty; }
    if (lane == 0) { d.xy[2 * p] = cx; d.xy[2 * p + 1] = cy; }
}
