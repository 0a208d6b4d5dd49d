// oracle/detector.cpp — CPU restatement of FeatureDetector::DetectWithSubPix (FeatureDetector.cc:55-75):
//   cv::goodFeaturesToTrack(im, corners, nCorners, qualityLevel, s*minDistance)        (blockSize 3, min-eigenvalue)
//   cv::cornerSubPix(im, corners, Size(7,7), Size(-1,-1), COUNT+EPS(30, 0.01))
// TEST INFRASTRUCTURE ONLY (see rvio_oracle.h).  OpenCV is a third-party dependency that is not vendored in the reference
// (CMakeLists.txt:44-50, "3.0 or >= 2.4.3", tested 3.3.1): the algorithms are restated from the published implementation
// (imgproc/src/featureselect.cpp, corner.cpp, cornersubpix.cpp, samplers.cpp) — PARITY UNPINNED.
// Where OpenCV's result depends on the build (SIMD/FMA) or on an accumulation order, a canonical order is fixed here and
// stated at the spot; the HIP kernels follow the same order so that GPU-vs-oracle parity is bit-exact.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rvio_oracle.h"

namespace {
inline int refl(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
    return i;
}
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cornerMinEigenVal(src u8, blockSize 3, ksize 3), corner.cpp: Sobel with scale 1/(2^(ksize-1) * blockSize * 255) folded into the
// SMOOTHING kernel of each separable pair (Sobel(): "if dx == 0 kx *= scale else ky *= scale"), float arithmetic, BORDER_REFLECT_101:
//   Dx = k0*R[y] + k1*(R[y-1] + R[y+1]),  R[y][x] = s[y][x+1] - s[y][x-1]
//   Dy = Q[y+1] - Q[y-1],                 Q[y][x] = k0*s[y][x] + k1*(s[y][x-1] + s[y][x+1])          k0 = float(2*scale), k1 = float(scale)
// covariance products in float; 3x3 unnormalised box sum (boxFilter uses double sums for float input): canonical order =
// vertical sums top->bottom, then the three columns left->right, rounded once to float;
// lambda_min = (a + c) - sqrt((a - c)^2 + b^2), a = .5 cxx, b = cxy, c = .5 cyy  (calcMinEigenVal, float).
// cv_order = true (orc_min_eig_cvorder, measurement only): the 3x3 box sums as cv::boxFilter forms them — RowSum: a running sum along the row
// (s += in[x+2] - in[x-1]), ColumnSum: a running sum down the column (add the newest row sum, emit, subtract the oldest), both in double.
void min_eig_map(const uint8_t* src, int w, int h, int stride, float* eig, bool cv_order = false) {
    const double scale = 1.0 / (4.0 * 3.0 * 255.0);
    const float k1 = (float)scale, k0 = (float)(2.0 * scale);
    // source with a 1-px reflected frame, as float
    const int pw = w + 2;
    std::vector<float> s((size_t)pw * (h + 2));
#pragma omp parallel for schedule(static)
    for (int y = -1; y <= h; ++y)
        for (int x = -1; x <= w; ++x) s[(size_t)(y + 1) * pw + (x + 1)] = (float)src[(size_t)refl(y, h) * stride + refl(x, w)];
    // the three product planes, with a 1-px reflected frame (boxFilter's border), as double
    std::vector<double> pr[3];
    for (auto& p : pr) p.assign((size_t)pw * (h + 2), 0.0);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float* r0 = &s[(size_t)y * pw + 1];
        const float* r1 = r0 + pw;
        const float* r2 = r1 + pw;
        for (int x = 0; x < w; ++x) {
            const float rr0 = r0[x + 1] - r0[x - 1], rr1 = r1[x + 1] - r1[x - 1], rr2 = r2[x + 1] - r2[x - 1];
            const float gx = k0 * rr1 + k1 * (rr0 + rr2);
            const float q0 = k0 * r0[x] + k1 * (r0[x - 1] + r0[x + 1]);
            const float q2 = k0 * r2[x] + k1 * (r2[x - 1] + r2[x + 1]);
            const float gy = q2 - q0;
            const size_t o = (size_t)(y + 1) * pw + (x + 1);
            pr[0][o] = (double)(gx * gx); pr[1][o] = (double)(gx * gy); pr[2][o] = (double)(gy * gy);
        }
    }
    for (auto& p : pr) {
        for (int y = 0; y < h; ++y) { p[(size_t)(y + 1) * pw] = p[(size_t)(y + 1) * pw + 1 + refl(-1, w)]; p[(size_t)(y + 1) * pw + w + 1] = p[(size_t)(y + 1) * pw + 1 + refl(w, w)]; }
        for (int x = 0; x < pw; ++x) { p[x] = p[(size_t)(1 + refl(-1, h)) * pw + x]; p[(size_t)(h + 1) * pw + x] = p[(size_t)(1 + refl(h, h)) * pw + x]; }
    }
    if (cv_order) {
        std::vector<float> cov[3];
        for (int k = 0; k < 3; ++k) {
            cov[k].assign((size_t)w * h, 0.f);
            std::vector<double> rs((size_t)w * (h + 2));            // row sums of the framed plane, rows -1 .. h
            for (int y = 0; y < h + 2; ++y) {
                const double* in = &pr[k][(size_t)y * pw];
                double sacc = (in[0] + in[1]) + in[2];
                rs[(size_t)y * w] = sacc;
                for (int x = 0; x + 1 < w; ++x) { sacc += in[x + 3] - in[x]; rs[(size_t)y * w + x + 1] = sacc; }
            }
            std::vector<double> SUM((size_t)w);
            for (int x = 0; x < w; ++x) SUM[x] = rs[x] + rs[(size_t)w + x];     // the first ksize - 1 rows
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const double s0 = SUM[x] + rs[(size_t)(y + 2) * w + x];
                    cov[k][(size_t)y * w + x] = (float)s0;
                    SUM[x] = s0 - rs[(size_t)y * w + x];
                }
        }
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            const float a = cov[0][i] * 0.5f, b = cov[1][i], c = cov[2][i] * 0.5f;
            eig[i] = (a + c) - std::sqrt((a - c) * (a - c) + b * b);
        }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float cov[3];
            for (int k = 0; k < 3; ++k) {
                const double* c0 = &pr[k][(size_t)y * pw + x];
                const double* c1 = c0 + pw;
                const double* c2 = c1 + pw;
                const double col0 = (c0[0] + c1[0]) + c2[0], col1 = (c0[1] + c1[1]) + c2[1], col2 = (c0[2] + c1[2]) + c2[2];
                cov[k] = (float)((col0 + col1) + col2);
            }
            const float a = cov[0] * 0.5f, b = cov[1], c = cov[2] * 0.5f;
            eig[(size_t)y * w + x] = (a + c) - std::sqrt((a - c) * (a - c) + b * b);
        }
}

// goodFeaturesToTrack, featureselect.cpp: threshold-to-zero at max*quality, strict 3x3 local maxima (val == dilate3x3) away from the
// 1-px border, sorted by value descending (ties: larger address first — the deterministic comparator of OpenCV >= 3.4.2), greedy
// minimum-distance selection on a grid with cell = cvRound(minDistance), at most max_corners.
int gftt(const uint8_t* src, int w, int h, int stride, int max_corners, double quality, double min_distance, float* out_xy) {
    std::vector<float> eig((size_t)w * h);
    min_eig_map(src, w, h, stride, eig.data());
    float mx = 0.f;
    bool any = false;
    for (float v : eig) { if (!any || v > mx) { mx = v; any = true; } }
    const float thr = (float)((double)mx * quality);
    for (float& v : eig) if (!(v > thr)) v = 0.f;
    std::vector<int> cand;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            const float v = eig[(size_t)y * w + x];
            if (v == 0.f) continue;
            float m = v;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) m = std::max(m, eig[(size_t)(y + dy) * w + x + dx]);
            if (v == m) cand.push_back(y * w + x);
        }
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return eig[a] > eig[b] ? true : (eig[a] < eig[b] ? false : a > b); });
    int n = 0;
    if (min_distance >= 1) {
        const int cell = (int)std::nearbyint(min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        std::vector<std::vector<int>> grid((size_t)gw * gh);
        const double md2 = min_distance * min_distance;
        for (int idx : cand) {
            const int y = idx / w, x = idx % w;
            const int xc = x / cell, yc = y / cell;
            const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            bool good = true;
            for (int yy = y1; yy <= y2 && good; ++yy)
                for (int xx = x1; xx <= x2 && good; ++xx)
                    for (int o : grid[(size_t)yy * gw + xx]) {
                        const float ddx = (float)x - (float)(o % w), ddy = (float)y - (float)(o / w);
                        if (ddx * ddx + ddy * ddy < md2) { good = false; break; }
                    }
            if (!good) continue;
            grid[(size_t)yc * gw + xc].push_back(idx);
            out_xy[2 * n] = (float)x; out_xy[2 * n + 1] = (float)y;
            if (++n == max_corners && max_corners > 0) break;
        }
    } else {
        for (int idx : cand) {
            out_xy[2 * n] = (float)(idx % w); out_xy[2 * n + 1] = (float)(idx / w);
            if (++n == max_corners && max_corners > 0) break;
        }
    }
    return n;
}

// getRectSubPix(u8 -> f32), samplers.cpp: bilinear patch of size pw x ph centred at c, replicated border.
// Canonical form (also for border patches): dst = ((s00*a11 + s01*a12) + s10*a21) + s11*a22 with clamped sample coordinates
// (OpenCV's border branch blends two samples instead of four where the column/row is replicated: equal up to float rounding).
void rect_subpix(const uint8_t* src, int w, int h, int stride, float cx, float cy, int pw, int ph, float* dst) {
    cx -= (float)(pw - 1) * 0.5f; cy -= (float)(ph - 1) * 0.5f;
    const int ix = (int)std::floor(cx), iy = (int)std::floor(cy);
    float a = cx - (float)ix;
    const float b = cy - (float)iy;
    a = std::max(a, 0.0001f);
    const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    for (int i = 0; i < ph; ++i) {
        const uint8_t* r0 = src + (size_t)clampi(iy + i, 0, h - 1) * stride;
        const uint8_t* r1 = src + (size_t)clampi(iy + i + 1, 0, h - 1) * stride;
        for (int j = 0; j < pw; ++j) {
            const int x0 = clampi(ix + j, 0, w - 1), x1 = clampi(ix + j + 1, 0, w - 1);
            dst[i * pw + j] = (((float)r0[x0] * a11 + (float)r0[x1] * a12) + (float)r1[x0] * a21) + (float)r1[x1] * a22;
        }
    }
}

// cornerSubPix, cornersubpix.cpp.  Accumulation order of the five double sums (OpenCV adds the (2 win + 1)^2 terms in one row-major
// chain; any fixed order differs from it by O(1e-16) relative): canonical = the terms on a zero-padded G x G grid, G = 16 for window
// half-sizes <= 7 (the stock 7: 15 rows / columns), 32 for half-sizes <= 15, 64 for <= 31, 128 for <= 63 (Tracker.nMinDist < 128); per row a
// balanced binary tree over the columns ((j, j + G/2), then + G/4, ... + 1), the rows in groups of four ((R0 + R1) + (R2 + R3)), the G/4
// groups by a balanced binary tree over neighbours (G = 16: (W0 + W1) + (W2 + W3); G = 32: that + ((W4 + W5) + (W6 + W7)); ...).
// row_major = true (orc_corner_subpix_rowmajor, measurement only): the five sums as one row-major chain each, OpenCV's own order.
void corner_subpix(const uint8_t* src, int w, int h, int stride, float* pts, int n, int win, int max_iter, double eps, bool row_major = false) {
    const int ww = 2 * win + 1, pw = ww + 2;
    const int G = ww <= 16 ? 16 : ww <= 32 ? 32 : ww <= 64 ? 64 : 128;
    if (win < 1 || ww > 128) return;
    std::vector<float> mask((size_t)ww * ww);
    for (int i = 0; i < ww; ++i) {
        const float y = (float)(i - win) / (float)win;
        const float vy = std::exp(-y * y);
        for (int j = 0; j < ww; ++j) {
            const float x = (float)(j - win) / (float)win;
            mask[(size_t)i * ww + j] = (float)(vy * std::exp(-x * x));
        }
    }
    eps *= eps;
    max_iter = std::min(std::max(max_iter, 1), 100);
    // (the points are independent: the multi-core build of the oracle, liborc_omp.so, runs them in parallel; results are identical)
#pragma omp parallel for schedule(dynamic, 4)
    for (int p = 0; p < n; ++p) {
        std::vector<float> patch((size_t)pw * pw);
        std::vector<double> term_((size_t)5 * G * G), tv_(G);
        const float tx = pts[2 * p], ty = pts[2 * p + 1];
        float cx = tx, cy = ty;
        int iter = 0;
        double err = 0;
        do {
            rect_subpix(src, w, h, stride, cx, cy, pw, pw, patch.data());
            // term[q][i][j] on a G x G grid (rows / columns >= ww are zero padding)
            std::fill(term_.begin(), term_.end(), 0.0);
            auto term = [&](int q, int i) { return &term_[((size_t)q * G + i) * G]; };
            for (int i = 0; i < ww; ++i) {
                const float* sp = &patch[(size_t)(i + 1) * pw + 1];
                const double py = i - win;
                for (int j = 0; j < ww; ++j) {
                    const double px = j - win;
                    const double m = mask[(size_t)i * ww + j];
                    const double tgx = sp[j + 1] - sp[j - 1];
                    const double tgy = sp[j + pw] - sp[j - pw];
                    const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    term(0, i)[j] = gxx; term(1, i)[j] = gxy; term(2, i)[j] = gyy;
                    term(3, i)[j] = gxx * px + gxy * py;
                    term(4, i)[j] = gxy * px + gyy * py;
                }
            }
            // per row a balanced tree over the G columns: (j, j+G/2), then +G/4, ... +1 -> R_i; groups of four rows
            // W_g = (R_4g + R_4g+1) + (R_4g+2 + R_4g+3); the G/4 groups by a balanced tree (G = 16: (W_0 + W_1) + (W_2 + W_3))
            double tot[5];
            for (int q = 0; q < 5; ++q) {
                double R[128];
                for (int i = 0; i < G; ++i) {
                    double* v = term(q, i);
                    double* t = tv_.data();
                    for (int s = G / 2; s >= 1; s >>= 1) {
                        for (int j = 0; j < G; ++j) t[j] = v[j] + v[(j + s) & (G - 1)];
                        for (int j = 0; j < G; ++j) v[j] = t[j];
                    }
                    R[i] = v[0];
                }
                double Wg[32];
                for (int g = 0; g < G / 4; ++g) Wg[g] = (R[4 * g] + R[4 * g + 1]) + (R[4 * g + 2] + R[4 * g + 3]);
                for (int m = G / 4; m > 1; m >>= 1)                 // neighbours pairwise, level by level
                    for (int g = 0; g < m / 2; ++g) Wg[g] = Wg[2 * g] + Wg[2 * g + 1];
                tot[q] = Wg[0];
            }
            if (row_major) {
                const float* patchp = patch.data();
                double ra = 0, rb = 0, rc = 0, r1 = 0, r2 = 0;
                for (int i = 0; i < ww; ++i) {
                    const float* sp = &patchp[(size_t)(i + 1) * pw + 1];
                    const double py = i - win;
                    for (int j = 0; j < ww; ++j) {
                        const double m = mask[(size_t)i * ww + j];
                        const double tgx = sp[j + 1] - sp[j - 1], tgy = sp[j + pw] - sp[j - pw];
                        const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                        const double px = j - win;
                        ra += gxx; rb += gxy; rc += gyy;
                        r1 += gxx * px + gxy * py;
                        r2 += gxy * px + gyy * py;
                    }
                }
                tot[0] = ra; tot[1] = rb; tot[2] = rc; tot[3] = r1; tot[4] = r2;
            }
            const double a = tot[0], b = tot[1], c = tot[2], bb1 = tot[3], bb2 = tot[4];
            const double det = a * c - b * b;
            if (std::fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            const double scale = 1.0 / det;
            const float nx = (float)(cx + c * scale * bb1 - b * scale * bb2);
            const float ny = (float)(cy - b * scale * bb1 + a * scale * bb2);
            const float ex = nx - cx, ey = ny - cy;
            err = (double)(ex * ex + ey * ey);                              // Point2f arithmetic: float
            cx = nx; cy = ny;
            if (cx < 0 || cx >= w || cy < 0 || cy >= h) break;
        } while (++iter < max_iter && err > eps);
        if (std::fabs(cx - tx) > win || std::fabs(cy - ty) > win) { cx = tx; cy = ty; }
        pts[2 * p] = cx; pts[2 * p + 1] = cy;
    }
}
}  // namespace

extern "C" {
void orc_min_eig(const uint8_t* img, int w, int h, int stride, float* eig) { min_eig_map(img, w, h, stride, eig); }
int orc_gftt(const uint8_t* img, int w, int h, int stride, int max_corners, double quality, double min_distance, float* out_xy) {
    return gftt(img, w, h, stride, max_corners, quality, min_distance, out_xy);
}
void orc_min_eig_cvorder(const uint8_t* img, int w, int h, int stride, float* eig) { min_eig_map(img, w, h, stride, eig, true); }
void orc_corner_subpix_rowmajor(const uint8_t* img, int w, int h, int stride, float* pts_xy, int n, int win) {
    corner_subpix(img, w, h, stride, pts_xy, n, win, 30, 1e-2, true);
}
void orc_corner_subpix(const uint8_t* img, int w, int h, int stride, float* pts_xy, int n, int win) {
    corner_subpix(img, w, h, stride, pts_xy, n, win, 30, 1e-2);
}
// FeatureDetector::DetectWithSubPix(im, nCorners = Tracker.nFeatures, s, corners), FeatureDetector.cc:55-75
int orc_detect(const rvio_config* cfg, const uint8_t* img, int stride, int s, float* out_xy) {
    const float min_dist = (float)s * cfg->min_dist;                       // s*mnMinDistance: int * float
    const int n = gftt(img, cfg->width, cfg->height, stride, cfg->n_features, (double)cfg->qual_lvl, (double)min_dist, out_xy);
    if (n > 0) {
        const int win = (int)std::floor(.5 * cfg->min_dist);               // subPixWinSize
        corner_subpix(img, cfg->width, cfg->height, stride, out_xy, n, win, 30, 1e-2);
    }
    return n;
}
}
