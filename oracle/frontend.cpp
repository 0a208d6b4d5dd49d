// oracle/frontend.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
// CPU restatement of the visual front end of the R-VIO hot path:
//   T3  cv::calcOpticalFlowPyrLK as called at Tracker.cc:237-244 (OpenCV 3.x
//       lkpyramid.cpp semantics restated from the published algorithm,
//       SURVEY.md appendix B.2 — OpenCV is NOT in /root/reference: unpinned)
//   T4  Tracker::UndistortAndNormalize Tracker.cc:100-132 (cv::undistortPoints, B.3)
//   T5  Ransac.cc:50-266, with glibc rand() restated (TYPE_3 additive feedback)
//   T6  Tracker::track book-keeping Tracker.cc:271-393
//   T7' FeatureDetector::FindNewer/ChessGrid FeatureDetector.cc:78-150 (grid
//       selection only; the corner detector's output is supplied by the caller)
#include "rvio_oracle.h"
#include "mat.hpp"
#include <chrono>
#include <deque>
#include <list>
#include <cstdio>

using namespace orc;

namespace {

// ------------------------------------------------------------ glibc rand()
// glibc random_r.c TYPE_3: r[i] = r[i-3] + r[i-31], output >> 1; srand(1) default.
// state: [0..30] words, [31] front index, [32] rear index, [33] initialised flag.
void rng_seed(int32_t* st, unsigned seed) {
    if (seed == 0) seed = 1;
    int32_t word = (int32_t)seed;
    st[0] = word;
    for (int i = 1; i < 31; ++i) {
        long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        st[i] = word;
    }
    st[31] = 3; st[32] = 0; st[33] = 1;
    for (int i = 0; i < 310; ++i) {
        uint32_t v = (uint32_t)st[st[31]] + (uint32_t)st[st[32]];
        st[st[31]] = (int32_t)v;
        st[31] = (st[31] + 1) % 31; st[32] = (st[32] + 1) % 31;
    }
}
int rng_next(int32_t* st) {
    if (!st[33]) rng_seed(st, 1);
    uint32_t v = (uint32_t)st[st[31]] + (uint32_t)st[st[32]];
    st[st[31]] = (int32_t)v;
    int out = (int)(v >> 1);
    st[31] = (st[31] + 1) % 31; st[32] = (st[32] + 1) % 31;
    return out;
}

M3 skew(const V3& w) {
    M3 S = m3_zero();
    S.m[0][1] = -w[2]; S.m[0][2] = w[1];
    S.m[1][0] = w[2];  S.m[1][2] = -w[0];
    S.m[2][0] = -w[1]; S.m[2][1] = w[0];
    return S;
}

inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
    return i;
}
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
inline int cv_round(float v) { return (int)std::nearbyintf(v); }  // round-half-even like cvRound
inline int cv_floor(float v) { return (int)std::floor(v); }

struct Level { int w = 0, h = 0; std::vector<uint8_t> img; std::vector<int16_t> dxy; };
struct Pyramid { std::vector<Level> lv; };

void pyr_down(const uint8_t* src, int w, int h, int stride, uint8_t* dst) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        std::vector<int> rows((size_t)5 * dw);
        for (int k = 0; k < 5; ++k) {
            const uint8_t* s = src + (size_t)reflect101(2 * y - 2 + k, h) * stride;
            int* r = &rows[(size_t)k * dw];
            for (int x = 0; x < dw; ++x) {
                int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = 2 * x, x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                r[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
            }
        }
        for (int x = 0; x < dw; ++x) {
            int v = rows[x] + rows[4 * dw + x] + (rows[dw + x] + rows[3 * dw + x]) * 4 + rows[2 * dw + x] * 6;
            dst[(size_t)y * dw + x] = (uint8_t)((v + 128) >> 8);
        }
    }
}

void scharr(const uint8_t* src, int w, int h, int stride, int16_t* dxy) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        std::vector<int> t0(w + 2), t1(w + 2);
        const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * stride;
        const uint8_t* r1 = src + (size_t)y * stride;
        const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * stride;
        for (int x = 0; x < w; ++x) { t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10; t1[x + 1] = r2[x] - r0[x]; }
        const int x0 = (w > 1 ? 1 : 0), x1 = (w > 1 ? w - 2 : 0);
        t0[0] = t0[x0 + 1]; t0[w + 1] = t0[x1 + 1];
        t1[0] = t1[x0 + 1]; t1[w + 1] = t1[x1 + 1];
        for (int x = 0; x < w; ++x) {
            dxy[((size_t)y * w + x) * 2] = (int16_t)(t0[x + 2] - t0[x]);
            dxy[((size_t)y * w + x) * 2 + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
}

// cv::CLAHE::apply for CV_8UC1 (OpenCV imgproc/src/clahe.cpp, 3.3+ / 4.x: CLAHE_CalcLut_Body + CLAHE_Interpolation_Body;
// third-party, not vendored in the reference: restated from the published algorithm, PARITY UNPINNED).
// The reference calls it as createCLAHE(3.0, Size(5,5))->apply(im, im)  (Tracker.cc:198-202).
//  * the image is extended to the right/bottom by (tiles - size % tiles) pixels, BORDER_REFLECT_101, unless BOTH
//    dimensions divide evenly (a dimension that divides evenly is still extended by a full `tiles` pixels);
//  * per tile: 256-bin histogram, clip at max(1, int(clip * area / 256)), excess redistributed evenly, the residual
//    one count every max(256/residual, 1) bins; lut[i] = saturate(cvRound(cumsum[i] * (255.f / area)));
//  * per pixel: bilinear blend (float) of the four neighbouring tile LUTs, cvRound.
void clahe_apply(const uint8_t* src, int w, int h, int stride, double clip, int tiles_x, int tiles_y, uint8_t* dst, int dstride) {
    int ew = w, eh = h;
    if (w % tiles_x != 0 || h % tiles_y != 0) { ew = w + (tiles_x - w % tiles_x); eh = h + (tiles_y - h % tiles_y); }
    const int tw = ew / tiles_x, th = eh / tiles_y, area = tw * th;
    const float lut_scale = 255.0f / (float)area;
    int clip_limit = 0;
    if (clip > 0.0) { clip_limit = (int)(clip * area / 256); clip_limit = std::max(clip_limit, 1); }
    std::vector<uint8_t> lut((size_t)tiles_x * tiles_y * 256);
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < tiles_x * tiles_y; ++t) {
        const int ty = t / tiles_x, tx = t % tiles_x;
        int hist[256] = {0};
        for (int y = ty * th; y < (ty + 1) * th; ++y) {
            const uint8_t* row = src + (size_t)reflect101(y, h) * stride;
            for (int x = tx * tw; x < (tx + 1) * tw; ++x) hist[row[reflect101(x, w)]]++;
        }
        if (clip_limit > 0) {
            int clipped = 0;
            for (int i = 0; i < 256; ++i)
                if (hist[i] > clip_limit) { clipped += hist[i] - clip_limit; hist[i] = clip_limit; }
            const int batch = clipped / 256;
            int residual = clipped - batch * 256;
            for (int i = 0; i < 256; ++i) hist[i] += batch;
            if (residual != 0) {
                const int step = std::max(256 / residual, 1);
                for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
            }
        }
        int sum = 0;
        for (int i = 0; i < 256; ++i) {
            sum += hist[i];
            const int v = cv_round((float)sum * lut_scale);
            lut[(size_t)t * 256 + i] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = cv_floor(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        ty1 = std::max(ty1, 0); ty2 = std::min(ty2, tiles_y - 1);
        const uint8_t* p1 = &lut[(size_t)ty1 * tiles_x * 256];
        const uint8_t* p2 = &lut[(size_t)ty2 * tiles_x * 256];
        for (int x = 0; x < w; ++x) {
            const float txf = (float)x * inv_tw - 0.5f;
            int tx1 = cv_floor(txf), tx2 = tx1 + 1;
            const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            tx1 = std::max(tx1, 0); tx2 = std::min(tx2, tiles_x - 1);
            const int v = src[(size_t)y * stride + x];
            const int i1 = tx1 * 256 + v, i2 = tx2 * 256 + v;
            const float res = ((float)p1[i1] * xa1 + (float)p1[i2] * xa) * ya1 + ((float)p2[i1] * xa1 + (float)p2[i2] * xa) * ya;
            const int r = cv_round(res);
            dst[(size_t)y * dstride + x] = (uint8_t)std::min(std::max(r, 0), 255);
        }
    }
}

const int kWin = 15, kMaxLevel = 3;

Pyramid build_pyramid(const uint8_t* img, int w, int h, int stride, bool with_deriv) {
    Pyramid p;
    Level l0; l0.w = w; l0.h = h; l0.img.resize((size_t)w * h);
    for (int y = 0; y < h; ++y) std::memcpy(&l0.img[(size_t)y * w], img + (size_t)y * stride, w);
    p.lv.push_back(std::move(l0));
    for (int l = 1; l <= kMaxLevel; ++l) {
        const Level& pr = p.lv.back();
        int nw = (pr.w + 1) / 2, nh = (pr.h + 1) / 2;
        if (nw <= kWin || nh <= kWin) break;  // buildOpticalFlowPyramid stops early
        Level nl; nl.w = nw; nl.h = nh; nl.img.resize((size_t)nw * nh);
        pyr_down(pr.img.data(), pr.w, pr.h, pr.w, nl.img.data());
        p.lv.push_back(std::move(nl));
    }
    if (with_deriv)
        for (auto& L : p.lv) { L.dxy.resize((size_t)L.w * L.h * 2); scharr(L.img.data(), L.w, L.h, L.w, L.dxy.data()); }
    return p;
}

// padded-image semantics: image border BORDER_REFLECT_101, derivative border constant 0
inline int pix(const Level& L, int x, int y) { return L.img[(size_t)reflect101(y, L.h) * L.w + reflect101(x, L.w)]; }
inline int der(const Level& L, int x, int y, int c) { return (x < 0 || y < 0 || x >= L.w || y >= L.h) ? 0 : L.dxy[((size_t)y * L.w + x) * 2 + c]; }

// LKTrackerInvoker for all levels of one point.  The float accumulators of
// OpenCV (whose order is build-dependent: scalar vs SSE/NEON lanes) are
// replaced by exact 64-bit integer sums converted once to float — order-free,
// so the HIP kernel can be bit-identical.
// float_acc = true (orc_klt_float, measurement only): the accumulators as OpenCV's SCALAR path holds them — float, every integer product
// converted and added in row-major window order (lkpyramid.cpp LKTrackerInvoker, acctype = itemtype = float without SIMD).  It is one of the
// build-dependent orders the header of this file speaks of; tests/test_opencv_distance.py reports how far the exact-sum definition sits from it.
// (measurement only: iteration counts of the calls so far — [points, iterations summed, largest per point, restages a 32 x 32 staged region would need])
static thread_local long long g_lk_stat[4] = {0, 0, 0, 0};   // (per thread: the multi-core build walks the features in parallel)
void lk_point(const Pyramid& P, const Pyramid& N, float px, float py, float* ox, float* oy, unsigned char* st, bool float_acc = false) {
    long long its_pt = 0;
    const int maxLevel = (int)std::min(P.lv.size(), N.lv.size()) - 1;
    const float FLT_SCALE = 1.f / (1 << 20);
    const double eps2 = 0.01 * 0.01;
    const int maxCount = 30;
    float nx = 0, ny = 0;
    *st = 1;
    for (int level = maxLevel; level >= 0; --level) {
        const Level& I = P.lv[level]; const Level& J = N.lv[level];
        float sc = (float)(1. / (1 << level));
        float ppx = px * sc, ppy = py * sc;
        if (level == maxLevel) { nx = ppx; ny = ppy; } else { nx = nx * 2.f; ny = ny * 2.f; }
        ppx -= 7.f; ppy -= 7.f;
        int ipx = cv_floor(ppx), ipy = cv_floor(ppy);
        if (ipx < -kWin || ipx >= I.w || ipy < -kWin || ipy >= I.h) { if (level == 0) *st = 0; continue; }
        float a = ppx - ipx, b = ppy - ipy;
        int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << 14));
        int iw01 = cv_round(a * (1.f - b) * (1 << 14));
        int iw10 = cv_round((1.f - a) * b * (1 << 14));
        int iw11 = (1 << 14) - iw00 - iw01 - iw10;
        short Iw[kWin * kWin], Ixw[kWin * kWin], Iyw[kWin * kWin];
        long long s11 = 0, s12 = 0, s22 = 0;
        float f11 = 0, f12 = 0, f22 = 0;
        for (int y = 0; y < kWin; ++y)
            for (int x = 0; x < kWin; ++x) {
                int X = ipx + x, Y = ipy + y;
                int ival = descale(pix(I, X, Y) * iw00 + pix(I, X + 1, Y) * iw01 + pix(I, X, Y + 1) * iw10 + pix(I, X + 1, Y + 1) * iw11, 14 - 5);
                int ixv = descale(der(I, X, Y, 0) * iw00 + der(I, X + 1, Y, 0) * iw01 + der(I, X, Y + 1, 0) * iw10 + der(I, X + 1, Y + 1, 0) * iw11, 14);
                int iyv = descale(der(I, X, Y, 1) * iw00 + der(I, X + 1, Y, 1) * iw01 + der(I, X, Y + 1, 1) * iw10 + der(I, X + 1, Y + 1, 1) * iw11, 14);
                Iw[y * kWin + x] = (short)ival; Ixw[y * kWin + x] = (short)ixv; Iyw[y * kWin + x] = (short)iyv;
                s11 += (long long)ixv * ixv; s12 += (long long)ixv * iyv; s22 += (long long)iyv * iyv;
                f11 += (float)(ixv * ixv); f12 += (float)(ixv * iyv); f22 += (float)(iyv * iyv);
            }
        float A11 = (float)s11 * FLT_SCALE, A12 = (float)s12 * FLT_SCALE, A22 = (float)s22 * FLT_SCALE;
        if (float_acc) { A11 = f11 * FLT_SCALE; A12 = f12 * FLT_SCALE; A22 = f22 * FLT_SCALE; }
        float D = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * kWin * kWin);
        if (minEig < 1e-3f || D < 1.1920929e-07f) { if (level == 0) *st = 0; continue; }
        D = 1.f / D;
        float npx = nx - 7.f, npy = ny - 7.f;
        float pdx = 0, pdy = 0;
        int jxl = ipx - 8, jyl = ipy - 8;
        for (int j = 0; j < maxCount; ++j) {
            int inx = cv_floor(npx), iny = cv_floor(npy);
            if (inx < -kWin || inx >= J.w || iny < -kWin || iny >= J.h) { if (level == 0) *st = 0; break; }
            ++its_pt;
            if (inx - jxl < 0 || inx - jxl > 15 || iny - jyl < 0 || iny - jyl > 15) { jxl = inx - 8; jyl = iny - 8; ++g_lk_stat[3]; }
            a = npx - inx; b = npy - iny;
            iw00 = cv_round((1.f - a) * (1.f - b) * (1 << 14));
            iw01 = cv_round(a * (1.f - b) * (1 << 14));
            iw10 = cv_round((1.f - a) * b * (1 << 14));
            iw11 = (1 << 14) - iw00 - iw01 - iw10;
            long long sb1 = 0, sb2 = 0;
            float fb1 = 0, fb2 = 0;
            for (int y = 0; y < kWin; ++y)
                for (int x = 0; x < kWin; ++x) {
                    int X = inx + x, Y = iny + y;
                    int diff = descale(pix(J, X, Y) * iw00 + pix(J, X + 1, Y) * iw01 + pix(J, X, Y + 1) * iw10 + pix(J, X + 1, Y + 1) * iw11, 14 - 5) - Iw[y * kWin + x];
                    sb1 += (long long)diff * Ixw[y * kWin + x]; sb2 += (long long)diff * Iyw[y * kWin + x];
                    fb1 += (float)(diff * Ixw[y * kWin + x]); fb2 += (float)(diff * Iyw[y * kWin + x]);
                }
            float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            if (float_acc) { b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE; }
            float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
            npx += dx; npy += dy;
            nx = npx + 7.f; ny = npy + 7.f;
            if ((double)dx * dx + (double)dy * dy <= eps2) break;
            if (j > 0 && std::fabs(dx + pdx) < 0.01 && std::fabs(dy + pdy) < 0.01) { nx -= dx * 0.5f; ny -= dy * 0.5f; break; }
            pdx = dx; pdy = dy;
        }
        if (*st && level == 0) {  // level-0 epilogue (err is requested by Tracker.cc:244)
            float fx = nx - 7.f, fy = ny - 7.f;
            int rx = cv_round(fx), ry = cv_round(fy);
            if (rx < -kWin || rx >= J.w || ry < -kWin || ry >= J.h) *st = 0;
        }
    }
    *ox = nx; *oy = ny;
    g_lk_stat[0]++; g_lk_stat[1] += its_pt; if (its_pt > g_lk_stat[2]) g_lk_stat[2] = its_pt;
}
extern "C" void orc_dbg_lk_stats(long long* out4) { for (int k = 0; k < 4; ++k) { out4[k] = g_lk_stat[k]; g_lk_stat[k] = 0; } }

// cv::undistortPoints without R/P (appendix B.3): 5 fixed-point iterations in double, float I/O
void undistort(const rvio_config* c, const float* in, int n, float* out) {
    const double fx = c->fx, fy = c->fy, cx = c->cx, cy = c->cy;
    const double k1 = c->k1, k2 = c->k2, p1 = c->p1, p2 = c->p2, k3 = c->k3;
    const double ifx = 1. / fx, ify = 1. / fy;
    if (c->fisheye) {
        // cv::fisheye::undistortPoints (Tracker.cc:118-119; OpenCV 3.3, the version the reference names, README.md:67): D = the four
        // coefficients the settings file calls k1, k2, p1, p2; ten fixed-point iterations theta <- theta_d / (1 + k theta^2 ...);
        // no R, no P.  Parity unpinned like the rest of the OpenCV side (OpenCV >= 3.4.2 iterates with Newton steps instead).
        const double k0 = c->k1, k1f = c->k2, k2f = c->p1, k3f = c->p2;
        for (int i = 0; i < n; ++i) {
            const double pwx = ((double)in[2 * i] - cx) / fx, pwy = ((double)in[2 * i + 1] - cy) / fy;
            double scale = 1.0;
            const double theta_d = std::sqrt(pwx * pwx + pwy * pwy);
            if (theta_d > 1e-8) {
                double theta = theta_d;
                for (int j = 0; j < 10; ++j) {
                    const double th2 = theta * theta, th4 = th2 * th2, th6 = th4 * th2, th8 = th6 * th2;
                    theta = theta_d / (1 + k0 * th2 + k1f * th4 + k2f * th6 + k3f * th8);
                }
                scale = std::tan(theta) / theta_d;
            }
            out[2 * i] = (float)(pwx * scale); out[2 * i + 1] = (float)(pwy * scale);
        }
        return;
    }
    for (int i = 0; i < n; ++i) {
        double x = in[2 * i], y = in[2 * i + 1];
        x = (x - cx) * ifx; y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {
            double r2 = x * x + y * y;
            double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            double dX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            double dY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - dX) * icdist; y = (y0 - dY) * icdist;
        }
        out[2 * i] = (float)x; out[2 * i + 1] = (float)y;
    }
}

// Ransac::GetRotation, Ransac.cc:120-155 (raw gyro, no bias removal — quirk D.3)
M3 gyro_rotation(const rvio_config* c, const rvio_imu* imu, int m) {
    M3 Ric; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ric.m[i][j] = c->T_bc[4 * i + j];
    M3 Rci = transpose(Ric);
    M3 R = m3_eye(); const M3 I = m3_eye();
    for (int s = 0; s < m; ++s) {
        V3 wm = v3(imu[s].w[0], imu[s].w[1], imu[s].w[2]);
        double dt = imu[s].dt, w1 = norm(wm), wdt = w1 * dt;
        M3 wx = skew(wm), wx2 = wx * wx, dR;
        if (w1 < c->small_angle) dR = (I - dt * wx) + (.5 * std::pow(dt, 2)) * wx2;
        else dR = (I - (std::sin(wdt) / w1) * wx) + ((1 - std::cos(wdt)) / std::pow(w1, 2)) * wx2;
        R = dR * R;
    }
    return (Rci * R) * Ric;
}

double sampson(const V3& p1, const V3& p2, const M3& E) {  // Ransac.cc:250-258
    V3 F1 = E * p1, F2 = transpose(E) * p2;
    V3 p2E = transpose(E) * p2;  // (pt2^T E) pt1 evaluated left to right
    double num = p2E[0] * p1[0] + p2E[1] * p1[1] + p2E[2] * p1[2];
    return std::pow(num, 2) / (std::pow(F1[0], 2) + std::pow(F1[1], 2) + std::pow(F2[0], 2) + std::pow(F2[1], 2));
}
double algebraic(const V3& p1, const V3& p2, const M3& E) {  // Ransac.cc:261-266
    V3 p2E = transpose(E) * p2;
    return std::fabs(p2E[0] * p1[0] + p2E[1] * p1[1] + p2E[2] * p1[2]);
}

// Ransac::FindInliers, Ransac.cc:180-247.  Quirk D.1: the reference spins forever for
// 17..31 candidates (16 disjoint pairs need 32 indices); here <32 candidates is treated
// like the reference's "too few" branch (return 0, flags untouched).
int ransac(const rvio_config* c, const double* P1, const double* P2, int n, const rvio_imu* imu, int m,
           unsigned char* flags, int32_t* rng, int* winner, int32_t* pairs_out) {
    const int nIter = 16;
    std::vector<int> cand;
    for (int i = 0; i < n; ++i) if (flags[i]) cand.push_back(i);
    const int nc = (int)cand.size();
    if (winner) *winner = 0;
    if (nc < 2 * nIter) return 0;
    int pairs[16][2];
    {   // SetPointPair, Ransac.cc:50-83
        std::vector<int> idx(nc);
        for (int i = 0; i < nc; ++i) idx[i] = i;
        for (int it = 0; it < nIter; ++it) {
            int a, b;
            do { a = rng_next(rng) % nc; } while (idx[a] == -1);
            do { b = rng_next(rng) % nc; } while (idx[b] == -1 || a == b);
            pairs[it][0] = cand[idx[a]]; pairs[it][1] = cand[idx[b]];
            idx[a] = -1; idx[b] = -1;
        }
    }
    if (pairs_out) for (int i = 0; i < 16; ++i) { pairs_out[2 * i] = pairs[i][0]; pairs_out[2 * i + 1] = pairs[i][1]; }
    M3 R = gyro_rotation(c, imu, m);
    auto col = [](const double* P, int i) { return v3(P[3 * i], P[3 * i + 1], P[3 * i + 2]); };
    M3 hyp[16]; int best = 0, bestIdx = 0;
    for (int it = 0; it < nIter; ++it) {  // SetRansacModel, Ransac.cc:86-117
        V3 A1 = col(P1, pairs[it][0]), A2 = col(P2, pairs[it][0]), B1 = col(P1, pairs[it][1]), B2 = col(P2, pairs[it][1]);
        V3 A0 = R * A1, B0 = R * B1;
        double c1 = A2[0] * A0[1] - A0[0] * A2[1], c2 = A0[1] * A2[2] - A2[1] * A0[2], c3 = A2[0] * A0[2] - A0[0] * A2[2];
        double c4 = B2[0] * B0[1] - B0[0] * B2[1], c5 = B0[1] * B2[2] - B2[1] * B0[2], c6 = B2[0] * B0[2] - B0[0] * B2[2];
        double alpha = std::atan2(c3 * c5 - c2 * c6, c1 * c6 - c3 * c4);
        double beta = std::atan2(-c3, c1 * std::sin(alpha) + c2 * std::cos(alpha));
        V3 t = v3(std::sin(beta) * std::cos(alpha), std::cos(beta), -std::sin(beta) * std::sin(alpha));
        hyp[it] = skew(t) * R;
        int cnt = 0;  // CountInliers, Ransac.cc:158-177
        for (int k = 0; k < nc; ++k) {
            double dist = c->use_sampson ? sampson(col(P1, cand[k]), col(P2, cand[k]), hyp[it]) : algebraic(col(P1, cand[k]), col(P2, cand[k]), hyp[it]);
            if (dist < c->inlier_thr) cnt++;
        }
        if (cnt > best) { best = cnt; bestIdx = it; }
    }
    if (winner) *winner = bestIdx;
    int newOut = 0;
    for (int k = 0; k < nc; ++k) {
        double dist = c->use_sampson ? sampson(col(P1, cand[k]), col(P2, cand[k]), hyp[bestIdx]) : algebraic(col(P1, cand[k]), col(P2, cand[k]), hyp[bestIdx]);
        if (dist > c->inlier_thr || std::isnan(dist)) { flags[cand[k]] = 0; newOut++; }
    }
    return nc - newOut;
}

struct Pt { float x, y; };

}  // namespace

// =================================================================== tracker
struct orc_tracker {
    rvio_config cfg;
    int F, Fu, maxLen, minLen;
    bool first = true;
    Pyramid last;                              // cached pyramid + Scharr of mLastImage (quirk D.12: allowed saving)
    std::vector<std::list<Pt>> hist;           // mvlTrackingHistory
    std::vector<int> inlierIdx;                // mvInlierIndices
    std::list<int> freeIdx;                    // mlFreeIndices
    std::vector<Pt> feats;                     // mvFeatsToTrack
    std::vector<double> pts1;                  // mPoints1ForRansac 3 x N
    int32_t rng[35] = {0};
    // outputs
    std::vector<unsigned char> types;          // mvFeatTypesForUpdate
    std::vector<std::list<Pt>> meas;           // mvlFeatMeasForUpdate
    // FeatureDetector grid params, FeatureDetector.cc:29-52
    int gridCols, gridRows, blocks, offX, offY, maxPerBlock;   // all int members upstream (FeatureDetector.h:66-77)
};

namespace {

// FeatureDetector::FindNewer + ChessGrid, FeatureDetector.cc:78-150
void find_newer(orc_tracker* T, const std::vector<Pt>& corners, const std::vector<Pt>& ref, std::deque<Pt>& out) {
    const rvio_config& c = T->cfg;
    const float W = (float)c.width, H = (float)c.height;
    std::vector<std::vector<Pt>> grid(T->blocks);
    auto outside = [&](const Pt& p) { return p.x <= T->offX || p.y <= T->offY || p.x >= (W - T->offX) || p.y >= (H - T->offY); };
    for (const Pt& p : ref) {
        if (outside(p)) continue;
        int col = (int)std::floor((p.x - T->offX) / c.block_x), row = (int)std::floor((p.y - T->offY) / c.block_y);
        grid.at(row * T->gridCols + col).push_back(p);
    }
    for (const Pt& p : corners) {
        if (outside(p)) continue;
        int col = (int)std::floor((p.x - T->offX) / c.block_x), row = (int)std::floor((p.y - T->offY) / c.block_y);
        float xl = col * c.block_x + T->offX, xr = xl + c.block_x, yt = row * c.block_y + T->offY, yb = yt + c.block_y;
        if (std::fabs(p.x - xl) < c.min_dist || std::fabs(p.x - xr) < c.min_dist || std::fabs(p.y - yt) < c.min_dist || std::fabs(p.y - yb) < c.min_dist) continue;
        auto& cell = grid.at(row * T->gridCols + col);
        if ((float)cell.size() < .75 * T->maxPerBlock) {
            bool ok = true;
            for (const Pt& q : cell) {
                float dx = p.x - q.x, dy = p.y - q.y;
                double dist = std::sqrt((double)dx * dx + (double)dy * dy);  // cv::norm(Point2f) -> double
                if (!(dist > 1 * c.min_dist)) { ok = false; break; }
            }
            if (ok) { out.push_back(p); cell.push_back(p); }
        }
    }
}

}  // namespace

extern "C" {

void orc_srand(int32_t state[35], unsigned seed) { rng_seed(state, seed); }
int orc_rand(int32_t state[35]) { return rng_next(state); }

void orc_undistort(const rvio_config* cfg, const float* px, int n, float* out) { undistort(cfg, px, n, out); }

int orc_ransac(const rvio_config* cfg, const double* p1, const double* p2, int n, const rvio_imu* imu, int m,
               unsigned char* flags, int32_t rng[35], int* winner, int32_t* pairs) {
    return ransac(cfg, p1, p2, n, imu, m, flags, rng, winner, pairs);
}

void orc_pyr_down(const uint8_t* src, int w, int h, int stride, uint8_t* dst) { pyr_down(src, w, h, stride, dst); }
void orc_scharr(const uint8_t* src, int w, int h, int stride, int16_t* dxy) { scharr(src, w, h, stride, dxy); }

void orc_klt(const uint8_t* prev, const uint8_t* next, int w, int h, int stride,
             const float* pts, int n, float* out, unsigned char* status) {
    Pyramid P = build_pyramid(prev, w, h, stride, true), N = build_pyramid(next, w, h, stride, false);
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n; ++i) lk_point(P, N, pts[2 * i], pts[2 * i + 1], &out[2 * i], &out[2 * i + 1], &status[i]);
}
// measurement only (tests/test_opencv_distance.py): the same tracker with OpenCV's scalar-path float accumulators
void orc_klt_float(const uint8_t* prev, const uint8_t* next, int w, int h, int stride,
             const float* pts, int n, float* out, unsigned char* status) {
    Pyramid P = build_pyramid(prev, w, h, stride, true), N = build_pyramid(next, w, h, stride, false);
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n; ++i) lk_point(P, N, pts[2 * i], pts[2 * i + 1], &out[2 * i], &out[2 * i + 1], &status[i], true);
}

orc_tracker* orc_tracker_create(const rvio_config* cfg) {
    orc_tracker* T = new orc_tracker();
    T->cfg = *cfg;
    T->F = cfg->n_features; T->Fu = (int)std::ceil(.5 * T->F);  // Tracker.cc:73-74
    T->maxLen = cfg->max_track_len; T->minLen = cfg->min_track_len;
    T->hist.resize(T->F);
    T->gridCols = (int)std::floor(cfg->width / cfg->block_x); T->gridRows = (int)std::floor(cfg->height / cfg->block_y);
    T->blocks = T->gridCols * T->gridRows;
    // int members upstream (FeatureDetector.h:69-77): the assignments truncate
    T->offX = (int)(.5 * (cfg->width - T->gridCols * cfg->block_x)); T->offY = (int)(.5 * (cfg->height - T->gridRows * cfg->block_y));
    T->maxPerBlock = (int)((float)T->F / T->blocks);
    return T;
}
void orc_tracker_destroy(orc_tracker* T) { delete T; }

void orc_clahe(const uint8_t* img, int w, int h, int stride, uint8_t* out) { clahe_apply(img, w, h, stride, 3.0, 5, 5, out, w); }

// Tracker::track, Tracker.cc:179-396 (image already mono8)
static void track_impl(orc_tracker* T, const uint8_t* img, int stride, const float* given_xy, const unsigned char* given_flag,
                       const rvio_imu* imu, int m, const float* cand, int n_cand, rvio_frame_info* info) {
    const rvio_config& c = T->cfg;
    const int w = c.width, h = c.height;
    rvio_frame_info fi; std::memset(&fi, 0, sizeof fi);
    Pyramid cur;
    std::vector<uint8_t> eq;
    if (img && c.enable_equalizer) {   // Tracker.cc:198-202
        eq.resize((size_t)w * h);
        clahe_apply(img, w, h, stride, 3.0, 5, 5, eq.data(), w);
        img = eq.data(); stride = w;
    }
    if (img) cur = build_pyramid(img, w, h, stride, true);
    T->types.clear(); T->meas.clear(); T->meas.resize(T->Fu);
    // corners: the caller's list, or (cand == NULL) FeatureDetector::DetectWithSubPix on the image the tracker sees
    std::vector<float> det;
    auto corners_for = [&](int s_factor) {
        if (cand || !img) return;
        det.assign((size_t)2 * T->F, 0.f);
        n_cand = orc_detect(&c, img, stride, s_factor, det.data());
        cand = det.data();
    };
    if (T->first) {  // :204-234
        corners_for(1);
        int n = std::min(n_cand, T->F);
        if (n > 0) {
            T->feats.clear();
            for (int i = 0; i < n; ++i) T->feats.push_back({cand[2 * i], cand[2 * i + 1]});
            std::vector<float> un(2 * n);
            undistort(&c, &T->feats[0].x, n, un.data());
            T->pts1.assign(3 * n, 0.0);
            for (int i = 0; i < n; ++i) {
                T->hist[i].push_back({un[2 * i], un[2 * i + 1]});
                T->pts1[3 * i] = un[2 * i]; T->pts1[3 * i + 1] = un[2 * i + 1]; T->pts1[3 * i + 2] = 1;
                T->inlierIdx.push_back(i);
            }
            for (int i = n; i < T->F; ++i) T->freeIdx.push_back(i);
            T->first = false;
        }
        fi.n_tracked_out = (int)T->feats.size();
    } else {
        const int N = (int)T->feats.size();
        fi.n_tracked_in = N;
        std::vector<float> tracked(2 * N), un(2 * N);
        std::vector<unsigned char> flag(N);
        if (img) {
#pragma omp parallel for schedule(dynamic, 4)
            for (int i = 0; i < N; ++i) lk_point(T->last, cur, T->feats[i].x, T->feats[i].y, &tracked[2 * i], &tracked[2 * i + 1], &flag[i]);
        } else for (int i = 0; i < N; ++i) { tracked[2 * i] = given_xy[2 * i]; tracked[2 * i + 1] = given_xy[2 * i + 1]; flag[i] = given_flag[i]; }
        for (int i = 0; i < N; ++i) fi.n_klt_ok += flag[i] ? 1 : 0;
        undistort(&c, tracked.data(), N, un.data());  // all points incl. status 0 (:252-253)
        std::vector<double> pts2(3 * N);
        for (int i = 0; i < N; ++i) { pts2[3 * i] = un[2 * i]; pts2[3 * i + 1] = un[2 * i + 1]; pts2[3 * i + 2] = 1; }
        int winner = 0;
        fi.n_ransac_inliers = ransac(&c, T->pts1.data(), pts2.data(), N, imu, m, flag.data(), T->rng, &winner, nullptr);
        fi.ransac_winner = winner;

        std::vector<Pt> newFeats; std::vector<int> newIdx; std::vector<double> newPts;
        int nMeas = 0;
        for (int i = 0; i < N; ++i) {  // lost tracks :279-303
            if (flag[i]) continue;
            int idx = T->inlierIdx[i];
            T->freeIdx.push_back(idx);
            if ((int)T->hist[idx].size() >= T->minLen && nMeas < T->Fu) {
                T->types.push_back('1'); T->meas[nMeas] = T->hist[idx]; nMeas++;
            }
            T->hist[idx].clear();
        }
        for (int i = 0; i < N; ++i) {  // tracked :305-342
            if (!flag[i]) continue;
            int idx = T->inlierIdx[i];
            newIdx.push_back(idx);
            newFeats.push_back({tracked[2 * i], tracked[2 * i + 1]});
            Pt ptUN = {un[2 * i], un[2 * i + 1]};
            if ((int)T->hist[idx].size() == T->maxLen) {
                if (nMeas < T->Fu) {
                    T->types.push_back('2'); T->meas[nMeas] = T->hist[idx];
                    while ((double)T->hist[idx].size() > T->maxLen - (std::ceil(.5 * T->maxLen) - 1)) T->hist[idx].pop_front();
                    nMeas++;
                } else T->hist[idx].pop_front();
            }
            T->hist[idx].push_back(ptUN);
            newPts.push_back(ptUN.x); newPts.push_back(ptUN.y); newPts.push_back(1);
        }
        if (!T->freeIdx.empty()) {  // refill :344-387
            std::vector<Pt> corners;
            corners_for(2);
            for (int i = 0; i < std::min(n_cand, T->F); ++i) corners.push_back({cand[2 * i], cand[2 * i + 1]});
            std::deque<Pt> fresh;
            find_newer(T, corners, newFeats, fresh);
            if (!fresh.empty()) {
                std::vector<float> fin, fun;
                for (const Pt& p : fresh) { fin.push_back(p.x); fin.push_back(p.y); }
                fun.resize(fin.size());
                undistort(&c, fin.data(), (int)fresh.size(), fun.data());
                size_t k = 0;
                for (;;) {
                    int idx = T->freeIdx.front();
                    newIdx.push_back(idx);
                    newFeats.push_back(fresh[k]);
                    T->hist[idx].push_back({fun[2 * k], fun[2 * k + 1]});
                    newPts.push_back(fun[2 * k]); newPts.push_back(fun[2 * k + 1]); newPts.push_back(1);
                    T->freeIdx.pop_front(); ++k;
                    if (T->freeIdx.empty() || k == fresh.size() || (int)newFeats.size() == T->F) break;
                }
            }
        }
        T->feats = newFeats; T->inlierIdx = newIdx; T->pts1 = newPts;
        fi.n_tracked_out = (int)newFeats.size();
        fi.n_feat_update = nMeas;
    }
    if (img) T->last = std::move(cur);
    if (info) *info = fi;
}

void orc_tracker_track(orc_tracker* T, const uint8_t* img, int stride, const rvio_imu* imu, int m,
                       const float* cand, int n_cand, rvio_frame_info* info) {
    track_impl(T, img, stride, nullptr, nullptr, imu, m, cand, n_cand, info);
}
// direct-track mode: the KLT result (vFeatsTracked, vInlierFlag of Tracker.cc:244)
// is supplied by the caller; everything after Tracker.cc:246 runs unchanged.
void orc_tracker_track_points(orc_tracker* T, const float* tracked_xy, const unsigned char* status,
                              const rvio_imu* imu, int m, const float* cand, int n_cand, rvio_frame_info* info) {
    track_impl(T, nullptr, 0, tracked_xy, status, imu, m, cand, n_cand, info);
}

void orc_tracker_get_tracks(orc_tracker* T, int32_t* n_feat, unsigned char* types, int32_t* len, float* meas) {
    *n_feat = (int)T->types.size();
    for (int f = 0; f < *n_feat; ++f) {
        types[f] = T->types[f]; len[f] = (int)T->meas[f].size();
        int k = 0;
        for (const Pt& p : T->meas[f]) { meas[((size_t)f * T->maxLen + k) * 2] = p.x; meas[((size_t)f * T->maxLen + k) * 2 + 1] = p.y; ++k; }
    }
}
void orc_tracker_get_points(orc_tracker* T, int32_t* n, float* xy, int32_t* hist_len) {
    *n = (int)T->feats.size();
    for (int i = 0; i < *n; ++i) { xy[2 * i] = T->feats[i].x; xy[2 * i + 1] = T->feats[i].y; hist_len[i] = (int)T->hist[T->inlierIdx[i]].size(); }
}

}  // extern "C"

// =================================================================== system
struct orc_system {
    rvio_config cfg;
    orc_tracker* trk;
    std::vector<double> x, P;
    int xdim = 26, d = 24, nClones = 0, nImg = 0;
    int info_form = 0;    // 1: update through orc_update_local/global (the device's formulation) instead of the literal path
    int last_rank = -1;   // nRank of the literal path's last tall compression (-1: fat / no update)
};

extern "C" {

void orc_system_set_information_form(orc_system* S, int on) { S->info_form = on; }
int orc_system_last_rank(orc_system* S) { return S->last_rank; }

orc_system* orc_system_create(const rvio_config* cfg) {
    orc_system* S = new orc_system();
    S->cfg = *cfg; S->trk = orc_tracker_create(cfg);
    int nmax = cfg->max_track_len - 1;
    S->x.assign(26 + 7 * (nmax + 1), 0.0); S->P.assign((size_t)(24 + 6 * (nmax + 1)) * (24 + 6 * (nmax + 1)), 0.0);
    return S;
}
void orc_system_destroy(orc_system* S) { orc_tracker_destroy(S->trk); delete S; }
void orc_system_set_state(orc_system* S, const double* x, int xdim, const double* P, int d) {
    std::memcpy(S->x.data(), x, sizeof(double) * xdim); std::memcpy(S->P.data(), P, sizeof(double) * d * d);
    S->xdim = xdim; S->d = d; S->nClones = (xdim - 26) / 7;
}
void orc_system_get_state(orc_system* S, double* x, int* xdim, double* P, int* d) {
    std::memcpy(x, S->x.data(), sizeof(double) * S->xdim); std::memcpy(P, S->P.data(), sizeof(double) * S->d * S->d);
    *xdim = S->xdim; *d = S->d;
}

// timed body of System::MonoVIO, System.cc:253-367
void orc_system_frame(orc_system* S, const uint8_t* img, int stride, const float* tracked_xy, const unsigned char* status,
                      const rvio_imu* imu, int m,
                      const float* cand, int n_cand, rvio_frame_info* info, double t_ms[4], double pose_p[3], double pose_q[4]) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    rvio_frame_info fi; std::memset(&fi, 0, sizeof fi);
    S->nImg++;
    auto t0 = clk::now();
    track_impl(S->trk, img, stride, tracked_xy, status, imu, m, cand, n_cand, &fi);
    auto t1 = clk::now();
    std::vector<double> xn(S->x.size());
    orc_propagate(&S->cfg, S->x.data(), S->xdim, S->P.data(), S->d, imu, m, xn.data());
    auto t2 = clk::now();
    if (S->nClones > S->cfg.min_track_len - 1) {  // System.cc:266
        const int Fu = S->trk->Fu, ML = S->cfg.max_track_len;
        std::vector<unsigned char> types(Fu); std::vector<int32_t> len(Fu); std::vector<float> meas((size_t)Fu * ML * 2);
        int32_t nf = 0;
        orc_tracker_get_tracks(S->trk, &nf, types.data(), len.data(), meas.data());
        rvio_tracks tr = {nf, ML, types.data(), len.data(), meas.data()};
        std::vector<double> xo(S->x.size()), Po(S->P.size());
        int32_t inf[4];
        if (S->info_form) {   // analysis mode: the information-form restatement of the same update (the device's formulation)
            const int nc6 = 6 * S->nClones;
            std::vector<double> blk((size_t)2 * nc6 * (nc6 + 1) + 8);
            orc_update_local(&S->cfg, xn.data(), S->xdim, S->P.data(), S->d, &tr, 0, 1, blk.data());
            orc_update_global(&S->cfg, xn.data(), S->xdim, S->P.data(), S->d, blk.data(), 1, xo.data(), Po.data(), inf);
            S->last_rank = -1;
        } else {
            orc_update(&S->cfg, xn.data(), S->xdim, S->P.data(), S->d, &tr, xo.data(), Po.data(), nullptr, nullptr, nullptr, nullptr, inf);
            S->last_rank = inf[2];
        }
        std::memcpy(S->x.data(), xo.data(), sizeof(double) * S->xdim);
        std::memcpy(S->P.data(), Po.data(), sizeof(double) * S->d * S->d);
        fi.n_feat_accepted = inf[0]; fi.n_rows = inf[1]; fi.updated = inf[3];
    } else {
        std::memcpy(S->x.data(), xn.data(), sizeof(double) * S->xdim);
    }
    auto t3 = clk::now();
    orc_augment_compose(&S->cfg, S->x.data(), &S->xdim, S->P.data(), &S->d, S->nImg > 1, pose_p, pose_q);
    S->nClones = (S->xdim - 26) / 7;
    auto t4 = clk::now();
    fi.n_clones = S->nClones;
    {
        rvio_frame_info keep = fi;
        (void)keep;
    }
    if (t_ms) { t_ms[0] = ms(t0, t1); t_ms[1] = ms(t1, t2); t_ms[2] = ms(t2, t3); t_ms[3] = ms(t3, t4); }
    if (info) *info = fi;
}

}  // extern "C"

extern "C" orc_tracker* orc_system_tracker(orc_system* S) { return S->trk; }
