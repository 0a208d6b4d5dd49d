// frontend_kernels.hip — hand-written gfx950 kernels for the visual front end of the
// R-VIO hot path.  Compiled with FP contraction off: the
// KLT arithmetic (integer fixed-point + float32) is bit-identical to oracle/frontend.cpp.
//
//   pyramid_kernel    every level of the image pyramid in one launch (cv::pyrDown chain + the copy of the frame into level 0)
//   (klt_kernel3      LKTrackerInvoker, all levels, one wave per feature — klt3.hip; CLAHE — clahe.hip; detector — detector.hip)
//   ransac_kernel     UndistortAndNormalize + Ransac::FindInliers    (Tracker.cc:252-264, Ransac.cc:180-247); ransac_book_a_kernel = the same
//                     + bookkeep_a_kernel's body in one launch (run-ahead path)
//   bookkeep_a_kernel, bookkeep_b_kernel   track book-keeping: the Updater's hand-over / FindNewer + refill (Tracker.cc:271-393, FeatureDetector.cc:78-150)
//   stage_gate_kernel, stage_signal_kernel one-workgroup poll / bump of a device-side counter (StageSync): hand-over -> filter, corners -> refill
#include "rvio_dev.h"
#include "frontend_dev.h"
#include "pyr_sep.h"
#include "../../include/rvio_hip.h"

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
    return i;
}
// single reflection: valid (and identical to reflect101) whenever -n < i < 2n-1
__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }
// two branch-free reflections: identical to reflect101 for -2(n-1) <= i <= 3(n-1)  (the KLT search region stays within
// [-23, n+22] and every pyramid level has n >= 16)
__device__ __forceinline__ int reflect2(int i, int n) { return reflect1(reflect1(i, n), n); }

// ------------------------------------------------------------------ pyramid
// buildOpticalFlowPyramid of cv::calcOpticalFlowPyrLK (Tracker.cc:244) in ONE launch: a workgroup owns an 8x8 tile of level 3 and
// everything above it.  The body lives in pyr_sep.h as per-thread phases between barriers (the same code runs thread by thread on the
// host in tests/test_pyramid_phases.py): each cv::pyrDown as a vertical pass + a horizontal pass through LDS, the reflect-101 indices
// from tables the workgroup fills once.  The patches overlap between neighbours ((85/64)^2 = 1.8 x the level-0 reads, all L2 hits);
// nothing but the u8 levels is written: the Scharr derivatives of the template are formed by the KLT kernel from its staged patch.
__global__ __launch_bounds__(PYR_T) void pyramid_kernel(const uint8_t* __restrict__ src, int stride, PyrDev p, int levels, int copy0, size_t src_bs, size_t bs) {
    src = zoff(src, src_bs); pyr_shift(p, (size_t)blockIdx.z * bs);
    __shared__ PyrLds s;
    PyrOut o;
#pragma unroll
    for (int l = 0; l < 4; ++l) { o.img[l] = (uint8_t*)p.img[l]; o.w[l] = p.w[l]; o.h[l] = p.h[l]; }
    const PyrGeom g = pyr_geom(o, blockIdx.x, blockIdx.y, levels, copy0);   // (uniform over the workgroup: so is every early-out below)
    const int tid = threadIdx.x;
    if (g.nl < 1) return;
    pyr_phase0(g, tid, s, src, stride);
    __syncthreads();
    pyr_phase1(g, tid, s, o);
    if (g.nl < 2) return;
    __syncthreads();
    pyr_phase2(g, tid, s, o);
    if (g.nl < 3) return;
    __syncthreads();
    pyr_phase3(g, tid, s);
    __syncthreads();
    pyr_phase4(g, tid, s, o);
    if (g.nl < 4) return;
    __syncthreads();
    pyr_phase5(g, tid, s, o);
}

// The round-1..4 form (a 25-tap gather per output, reflect-101 per tap), kept for A/B timing in the instrumented build only (RVIO_PYR_V1=1)
#ifdef RVIO_DBG_CLOCKS
__device__ __forceinline__ int pyr_down_at(const uint8_t* __restrict__ src, int sw, int sx0, int sy0, int w, int h, int x, int y) {
    // pyrDown pixel (x, y) of the next level from the LDS patch `src` (row stride sw) that holds level pixels [sx0.., sy0..]
    int xs[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) xs[k] = reflect101(2 * x - 2 + k, w) - sx0;
    int rows[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint8_t* s = src + (reflect101(2 * y - 2 + k, h) - sy0) * sw;
        rows[k] = s[xs[2]] * 6 + (s[xs[1]] + s[xs[3]]) * 4 + s[xs[0]] + s[xs[4]];
    }
    return (rows[0] + rows[4] + (rows[1] + rows[3]) * 4 + rows[2] * 6 + 128) >> 8;
}
__global__ __launch_bounds__(PYR_T) void pyramid_kernel_v1(const uint8_t* __restrict__ src, int stride, PyrDev p, int levels, int copy0, size_t src_bs, size_t bs) {
    src = zoff(src, src_bs); pyr_shift(p, (size_t)blockIdx.z * bs);
    __shared__ uint8_t L0[85 * 88], L1[41 * 44], L2[19 * 20];
    const int tid = threadIdx.x;
    const int X3 = blockIdx.x * 8, Y3 = blockIdx.y * 8;
    const int w0 = p.w[0], h0 = p.h[0];
    // level-0 patch [8 X3 - 14, 8 X3 + 70] clipped to the image
    const int ax = max(8 * X3 - 14, 0), ay = max(8 * Y3 - 14, 0), bx = min(8 * X3 + 70, w0 - 1), by = min(8 * Y3 + 70, h0 - 1);
    const int pw = bx - ax + 1, ph = by - ay + 1;
    if (pw <= 0 || ph <= 0) return;
    for (int e = tid; e < pw * ph; e += PYR_T) { const int yy = e / pw, xx = e - yy * pw; L0[yy * 88 + xx] = src[(size_t)(ay + yy) * stride + ax + xx]; }
    __syncthreads();
    if (copy0) {   // this workgroup's 64x64 tile of level 0
        uint8_t* d0 = (uint8_t*)p.img[0];
        for (int e = tid; e < 64 * 64; e += PYR_T) {
            const int x = 8 * X3 + (e & 63), y = 8 * Y3 + (e >> 6);
            if (x < w0 && y < h0) d0[(size_t)y * w0 + x] = L0[(y - ay) * 88 + (x - ax)];
        }
    }
    if (levels < 2) return;
    const int w1 = p.w[1], h1 = p.h[1];
    const int cx = max(4 * X3 - 6, 0), cy = max(4 * Y3 - 6, 0), dx = min(4 * X3 + 34, w1 - 1), dy = min(4 * Y3 + 34, h1 - 1);
    const int qw = dx - cx + 1, qh = dy - cy + 1;
    if (qw <= 0 || qh <= 0) return;
    {
        uint8_t* d1 = (uint8_t*)p.img[1];
        for (int e = tid; e < qw * qh; e += PYR_T) {
            const int yy = e / qw, xx = e - yy * qw, x = cx + xx, y = cy + yy;
            const int v = pyr_down_at(L0, 88, ax, ay, w0, h0, x, y);
            L1[yy * 44 + xx] = (uint8_t)v;
            if (x >= 4 * X3 && x < 4 * X3 + 32 && y >= 4 * Y3 && y < 4 * Y3 + 32) d1[(size_t)y * w1 + x] = (uint8_t)v;
        }
    }
    if (levels < 3) return;
    __syncthreads();
    const int w2 = p.w[2], h2 = p.h[2];
    const int ex = max(2 * X3 - 2, 0), ey = max(2 * Y3 - 2, 0), fx = min(2 * X3 + 16, w2 - 1), fy = min(2 * Y3 + 16, h2 - 1);
    const int rw = fx - ex + 1, rh = fy - ey + 1;
    if (rw <= 0 || rh <= 0) return;
    {
        uint8_t* d2 = (uint8_t*)p.img[2];
        for (int e = tid; e < rw * rh; e += PYR_T) {
            const int yy = e / rw, xx = e - yy * rw, x = ex + xx, y = ey + yy;
            const int v = pyr_down_at(L1, 44, cx, cy, w1, h1, x, y);
            L2[yy * 20 + xx] = (uint8_t)v;
            if (x >= 2 * X3 && x < 2 * X3 + 16 && y >= 2 * Y3 && y < 2 * Y3 + 16) d2[(size_t)y * w2 + x] = (uint8_t)v;
        }
    }
    if (levels < 4) return;
    __syncthreads();
    const int w3 = p.w[3], h3 = p.h[3];
    if (tid < 64) {
        const int x = X3 + (tid & 7), y = Y3 + (tid >> 3);
        if (x < w3 && y < h3) ((uint8_t*)p.img[3])[(size_t)y * w3 + x] = (uint8_t)pyr_down_at(L2, 20, ex, ey, w2, h2, x, y);
    }
}
#endif

// calcSharrDeriv of one pyramid level, on demand (rvio_hip_debug_pyramid): un-normalised 3x3 Scharr with reflect-101 neighbours,
// int16 (dx,dy) packed — the same arithmetic the KLT kernel applies to its staged template patch
__device__ __forceinline__ int scharr_pack(int a0, int a1, int a2, int b0, int b2, int c0, int c1, int c2) {
    const int t0m = (a0 + c0) * 3 + b0 * 10, t0p = (a2 + c2) * 3 + b2 * 10;
    const int t1m = c0 - a0, t1c = c1 - a1, t1p = c2 - a2;
    return ((t0p - t0m) & 0xffff) | (((t1p + t1m) * 3 + t1c * 10) << 16);
}
__global__ __launch_bounds__(256) void scharr_debug_kernel(const uint8_t* __restrict__ src, int w, int h, int* __restrict__ dxy) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h), xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    const uint8_t *r0 = src + (size_t)ym * w, *r1 = src + (size_t)y * w, *r2 = src + (size_t)yp * w;
    dxy[(size_t)y * w + x] = scharr_pack(r0[xm], r0[x], r0[xp], r1[xm], r1[xp], r2[xm], r2[x], r2[xp]);
}

// ------------------------------------------------------------------ KLT
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// (the KLT kernel itself is klt_kernel3 in klt3.hip)

// ------------------------------------------------------------------ undistort (cv::undistortPoints, 5 fixed iterations)
__device__ __forceinline__ void undistort_pt(const DevCfg& c, float u, float v, float* ox, float* oy) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
    if (c.fisheye) {   // cv::fisheye::undistortPoints (Tracker.cc:118-119) with D = (k1, k2, p1, p2): ten fixed-point iterations on theta
        const double k0 = c.k1, k1 = c.k2, k2 = c.p1, k3 = c.p2;
        const double pwx = ((double)u - cx) / fx, pwy = ((double)v - cy) / fy;
        double scale = 1.0;
        const double theta_d = sqrt(pwx * pwx + pwy * pwy);
        if (theta_d > 1e-8) {
            double theta = theta_d;
            for (int j = 0; j < 10; ++j) {
                const double th2 = theta * theta, th4 = th2 * th2, th6 = th4 * th2, th8 = th6 * th2;
                theta = theta_d / (1 + k0 * th2 + k1 * th4 + k2 * th6 + k3 * th8);
            }
            scale = tan(theta) / theta_d;
        }
        *ox = (float)(pwx * scale); *oy = (float)(pwy * scale);
        return;
    }
    const double k1 = c.k1, k2 = c.k2, p1 = c.p1, p2 = c.p2, k3 = c.k3;
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = u, y = v;
    x = (x - cx) * ifx; y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        const double dX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const double dY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        x = (x0 - dX) * icdist; y = (y0 - dY) * icdist;
    }
    *ox = (float)x; *oy = (float)y;
}

// glibc random_r TYPE_3 (rand() as used by Ransac.cc:63-69); state layout as oracle/frontend.cpp
__device__ void rng_seed(int* st, unsigned seed) {
    if (seed == 0) seed = 1;
    int word = (int)seed;
    st[0] = word;
    for (int i = 1; i < 31; ++i) {
        long long hi = word / 127773, lo = word % 127773;
        long long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int)w; st[i] = word;
    }
    st[31] = 3; st[32] = 0; st[33] = 1;
    for (int i = 0; i < 310; ++i) {
        unsigned v = (unsigned)st[st[31]] + (unsigned)st[st[32]];
        st[st[31]] = (int)v;
        st[31] = (st[31] + 1) % 31; st[32] = (st[32] + 1) % 31;
    }
}
__device__ int rng_next(int* st) {
    if (!st[33]) rng_seed(st, 1);
    unsigned v = (unsigned)st[st[31]] + (unsigned)st[st[32]];
    st[st[31]] = (int)v;
    int out = (int)(v >> 1);
    st[31] = (st[31] + 1) % 31; st[32] = (st[32] + 1) % 31;
    return out;
}

// wave-inclusive scan of one int per lane with DPP row shifts + row broadcasts (no LDS round trips)
__device__ __forceinline__ int wave_incl_scan(int v) {
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); v += t;   // row_shr:1
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); v += t;   // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); v += t;   // row_shr:4
    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); v += t;   // row_shr:8
    // carry the totals of the previous rows (lane 15, 31, 47)
    const int lane = threadIdx.x & 63;
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    v += (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
    return v;
}
// block-wide exclusive scan of one int per thread (256 threads); returns exclusive prefix, *total = sum
__device__ int block_exscan(int v, int* total, int* s_w) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int inc = wave_incl_scan(v);
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int s = s_w[w]; if (w < wv) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// the same for a workgroup of blockDim.x / 64 <= 16 waves (s_w: 16 ints)
__device__ int block_exscan_n(int v, int* total, int* s_w) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int inc = wave_incl_scan(v);
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) { const int sv = s_w[w]; if (w < wv) base += sv; tot += sv; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
// The largest double T with !(sqrt(T) > m): for every s >= 0 (and NaN), (sqrt(s) > m) == (s > T) — sqrt is monotonic, so the comparison of the
// rounded root with m flips at ONE double, found here with the same sqrt by stepping up from m * m (exact for a float m; <= 2 steps: the
// roots of the doubles just above m^2 round back to m).  Lets a serial loop compare squared distances without changing a single decision.
__device__ __forceinline__ double sqrt_gt_threshold(double m) {
    if (!(m >= 0)) return -1.0;   // sqrt(s) > m holds for every s >= 0
    double T = m * m;
    for (int it = 0; it < 8; ++it) {
        const double nx = __longlong_as_double(__double_as_longlong(T) + 1);
        if (sqrt(nx) > m) break;
        T = nx;
    }
    return T;
}

__device__ __forceinline__ double sampson_err(d3 p1, d3 p2, const m33& E) {  // Ransac.cc:250-258
    const d3 F1 = mv33(E, p1), F2 = mv33(tr33(E), p2);
    const double num = F2.x * p1.x + F2.y * p1.y + F2.z * p1.z;
    return (num * num) / (F1.x * F1.x + F1.y * F1.y + F2.x * F2.x + F2.y * F2.y);
}
__device__ __forceinline__ double algebraic_err(d3 p1, d3 p2, const m33& E) {  // Ransac.cc:261-266
    const d3 F2 = mv33(tr33(E), p2);
    return fabs(F2.x * p1.x + F2.y * p1.y + F2.z * p1.z);
}

// UndistortAndNormalize of the tracked points + Ransac::FindInliers.  One workgroup, 256 threads.
// un1: previous-frame normalised coords (mPoints1ForRansac, z = 1), un2: output for this frame.
// Dynamic LDS: cand[F] ints + first[F] ints (SetPointPair).
__device__ __forceinline__ void ransac_body(const DevCfg& cfg, const int* n_pts_ptr, const float* tracked, const float* un1, float* un2,
                                            unsigned char* status, const rvio_imu* imu, int m, int* rng,
                                            rvio_frame_info* info, size_t bs, size_t imu_bs, unsigned char* dsh) {
    __shared__ int s_w[16];
    __shared__ int pairs[16][2];
    DBG_S(blockIdx.z == 0, 2);
    DBG_U(1);
    n_pts_ptr = zoff(n_pts_ptr, bs); tracked = zoff(tracked, bs); un1 = zoff(un1, bs); un2 = zoff(un2, bs); status = zoff(status, bs);
    imu = zoff(imu, imu_bs); rng = zoff(rng, bs); info = zoff(info, bs);
    __shared__ double hyp[16][9];
    __shared__ double dRs[RVIO_MAX_IMU][9];
    __shared__ double Rsh[9];
    __shared__ int cnt[16];
    __shared__ int s_rng[36];
    __shared__ int s_winner, s_newout;
    int* cand = (int*)dsh;
    int* first = cand + cfg.F;   // SetPointPair: index of the first draw that produced each candidate position
    const int tid = threadIdx.x, T = blockDim.x, N = *n_pts_ptr;   // T: 256 .. 1024 threads (a multiple of 256)
    // one batch of global reads: RNG state, IMU samples (-> per-sample delta rotations), the points
    if (tid < 35) s_rng[tid] = rng[tid];
    // GetRotation, Ransac.cc:120-155 (raw gyro, no bias removal): per-sample dR
    auto gyro_dR = [&](int s) {
        const m33 I = eye33();
        const d3 wm = mk3(imu[s].w[0], imu[s].w[1], imu[s].w[2]);
        const double dt = imu[s].dt, w1 = nrm3(wm), wdt = w1 * dt;
        const m33 wx = skew33(wm), wx2 = mul33(wx, wx);
        if (w1 < cfg.small_angle) return add33(sub33(I, scl33(dt, wx)), scl33(.5 * dt * dt, wx2));
        return add33(sub33(I, scl33(sin(wdt) / w1, wx)), scl33((1 - cos(wdt)) / (w1 * w1), wx2));
    };
    const int m_lds = m < RVIO_MAX_IMU ? m : RVIO_MAX_IMU;   // one delta rotation per thread for the first RVIO_MAX_IMU samples; a longer batch
                                                             // (dropped images) forms the remaining ones inside the product loop below
    if (tid >= 64 && tid < 64 + m_lds) {
        const int s = tid - 64;
        const m33 dR = gyro_dR(s);
        for (int k = 0; k < 9; ++k) dRs[s][k] = dR.m[k];
    }
    for (int i = tid; i < N; i += T) undistort_pt(cfg, tracked[2 * i], tracked[2 * i + 1], &un2[2 * i], &un2[2 * i + 1]);
    // ordered compaction of candidate indices (status != 0)
    int nc = 0;
    for (int base = 0; base < N; base += T) {
        const int i = base + tid;
        const int fl = (i < N && status[i]) ? 1 : 0;
        int tot;
        const int pos = block_exscan_n(fl, &tot, s_w);
        if (fl) cand[nc + pos] = i;
        nc += tot;
    }
    if (tid < 16) cnt[tid] = 0;
    for (int i = tid; i < nc; i += T) first[i] = 0x7fffffff;
    if (tid == 0) { s_winner = 0; s_newout = 0; info->n_tracked_in = N; info->n_klt_ok = nc; info->n_ransac_inliers = 0; info->ransac_winner = 0; }
    __syncthreads();
    DBG_U(2);
    if (nc < 32) return;   // Ransac.cc:201-205; 17..31 would spin forever in the reference
    // SetPointPair, Ransac.cc:50-83: 16 pairs of distinct candidates from the rand() stream —
    //     do a = rand() % nc while used[a];  do b = rand() % nc while used[b] || a == b;  pair = (a, b), both become used.
    // A draw is turned down exactly when its value has been drawn before (the first occurrence of a value is always taken: `used` holds
    // nothing but earlier first occurrences, and so does `a`), so the pairs are the first 32 FIRST OCCURRENCES of the stream in order and the
    // stream advances to the draw that delivered the 32nd.  glibc's rand() is the additive-feedback generator r[i] = r[i-3] + r[i-31]: one
    // turn of its 31-word ring is a stride-3 prefix sum, which a wave computes in four shuffle steps — 31 draws at a time instead of a
    // serial chain through LDS (round 6: 11 us of this one-workgroup kernel, which sits on the tracker's serial chain; now ~1 us).
    if (tid < 64) {
        const int lane = tid;
        if (lane == 0 && !s_rng[33]) rng_seed(s_rng, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        int f = s_rng[31], total = 0, kbase = 0;
        for (int turn = 0; turn < 4096; ++turn) {
            unsigned x = 0;
            int pos = 0;
            if (lane < 31) {
                pos = f + lane; if (pos >= 31) pos -= 31;          // draw `lane` of this turn rewrites ring word pos ...
                x = (unsigned)s_rng[pos];
                if (lane < 3) { int p2 = pos + 28; if (p2 >= 31) p2 -= 31; x += (unsigned)s_rng[p2]; }   // ... adding the word three behind it: an old one for the first three draws,
            }
#pragma unroll
            for (int d = 3; d < 31; d *= 2) { const unsigned y = (unsigned)__shfl_up((int)x, d, 64); if (lane >= d) x += y; }   // a new one (draw lane - 3) for the others
            int val = 0;
            if (lane < 31) { val = (int)((x >> 1) % (unsigned)nc); atomicMin(&first[val], kbase + lane); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            const bool acc = lane < 31 && first[val] == kbase + lane;
            const unsigned long long mk = __ballot(acc);
            const int rank = total + __popcll(mk & ((1ull << lane) - 1ull));
            if (acc && rank < 32) pairs[rank >> 1][rank & 1] = cand[val];
            const int na = __popcll(mk);
            const bool done = total + na >= 32;
            int c = 31;                                             // draws of this turn the stream advances by
            if (done) c = __ffsll((long long)__ballot(acc && rank == 31));
            if (lane < c && lane < 31) s_rng[pos] = (int)x;
            f = (f + c) % 31;
            if (done) break;
            total += na; kbase += 31;
        }
        if (lane == 0) { s_rng[31] = f; s_rng[32] = (f + 28) % 31; }   // (the rear index trails the front one by three)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (lane < 35) rng[lane] = s_rng[lane];
    }
    else if (tid < 128) {   // beside the pair selection: the gyro prior R = Rci (dR_m-1 ... dR_0) Ric, a serial product of the per-sample rotations (same for the 16 models)
        const m33 Ric = ldm33(cfg.Ric), Rci = ldm33(cfg.Rci);
        m33 R = eye33();
        for (int s = 0; s < m_lds; ++s) R = mul33(ldm33(dRs[s]), R);
        for (int s = m_lds; s < m; ++s) R = mul33(gyro_dR(s), R);
        R = mul33(mul33(Rci, R), Ric);
        if (tid == 64) for (int k = 0; k < 9; ++k) Rsh[k] = R.m[k];
    }
    __syncthreads();
    DBG_U(3);
    if (tid < 16) {
        const m33 R = ldm33(Rsh);
        // SetRansacModel, Ransac.cc:86-117
        const int ia = pairs[tid][0], ib = pairs[tid][1];
        const d3 A1 = mk3(un1[2 * ia], un1[2 * ia + 1], 1.0), A2 = mk3(un2[2 * ia], un2[2 * ia + 1], 1.0);
        const d3 B1 = mk3(un1[2 * ib], un1[2 * ib + 1], 1.0), B2 = mk3(un2[2 * ib], un2[2 * ib + 1], 1.0);
        const d3 A0 = mv33(R, A1), B0 = mv33(R, B1);
        const double c1 = A2.x * A0.y - A0.x * A2.y, c2 = A0.y * A2.z - A2.y * A0.z, c3 = A2.x * A0.z - A0.x * A2.z;
        const double c4 = B2.x * B0.y - B0.x * B2.y, c5 = B0.y * B2.z - B2.y * B0.z, c6 = B2.x * B0.z - B0.x * B2.z;
        const double alpha = atan2(c3 * c5 - c2 * c6, c1 * c6 - c3 * c4);
        const double beta = atan2(-c3, c1 * sin(alpha) + c2 * cos(alpha));
        const d3 t = mk3(sin(beta) * cos(alpha), cos(beta), -sin(beta) * sin(alpha));
        const m33 E = mul33(skew33(t), R);
        for (int k = 0; k < 9; ++k) hyp[tid][k] = E.m[k];
    }
    __syncthreads();
    DBG_U(4);
    // CountInliers, Ransac.cc:158-177: thread <-> candidate; the T / 256 groups of 256 threads split the 16 hypotheses among them
    {
        const int G = T >> 8, g = tid >> 8, kk = tid & 255, HG = 16 / G;   // (G = 1, 2 or 4)
        for (int base = 0; base < nc; base += 256) {
            const int k = base + kk;
            d3 p1 = mk3(0, 0, 1), p2 = mk3(0, 0, 1);
            if (k < nc) { const int idx = cand[k]; p1 = mk3(un1[2 * idx], un1[2 * idx + 1], 1.0); p2 = mk3(un2[2 * idx], un2[2 * idx + 1], 1.0); }
            for (int it = g * HG; it < (g + 1) * HG; ++it) {
                m33 E; for (int q = 0; q < 9; ++q) E.m[q] = hyp[it][q];
                const double dist = cfg.use_sampson ? sampson_err(p1, p2, E) : algebraic_err(p1, p2, E);
                const unsigned long long bal = __ballot((k < nc) && (dist < cfg.inlier_thr));
                if ((tid & 63) == 0 && bal) atomicAdd(&cnt[it], __popcll(bal));
            }
        }
    }
    __syncthreads();
    DBG_U(5);
    if (tid == 0) {
        int best = 0, bi = 0;
        for (int it = 0; it < 16; ++it) if (cnt[it] > best) { best = cnt[it]; bi = it; }
        s_winner = bi;
    }
    __syncthreads();
    {
        m33 E; for (int q = 0; q < 9; ++q) E.m[q] = hyp[s_winner][q];
        for (int k = tid; k < nc; k += T) {
            const int idx = cand[k];
            const d3 p1 = mk3(un1[2 * idx], un1[2 * idx + 1], 1.0), p2 = mk3(un2[2 * idx], un2[2 * idx + 1], 1.0);
            const double dist = cfg.use_sampson ? sampson_err(p1, p2, E) : algebraic_err(p1, p2, E);
            if (dist > cfg.inlier_thr || isnan(dist)) { status[idx] = 0; atomicAdd(&s_newout, 1); }
        }
    }
    __syncthreads();
    if (tid == 0) { info->n_ransac_inliers = nc - s_newout; info->ransac_winner = s_winner; }
    DBG_U(6);
}
__global__ __launch_bounds__(256) void ransac_kernel(DevCfg cfg, const int* n_pts_ptr, const float* tracked, const float* un1, float* un2,
                                                     unsigned char* status, const rvio_imu* imu, int m, int* rng,
                                                     rvio_frame_info* info, size_t bs, size_t imu_bs) {
    extern __shared__ __align__(16) unsigned char dsh_r[];
    ransac_body(cfg, n_pts_ptr, tracked, un1, un2, status, imu, m, rng, info, bs, imu_bs, dsh_r);
}

// ------------------------------------------------------------------ T6 book-keeping + refill
// One workgroup, 256 threads.  Track histories are per-slot arrays hist[F][max_len] (float2) with
// lengths hist_len[F]; which free slot a new feature takes is storage only and never reaches an output.
// Dynamic LDS: tfs[F] float2 (new feature order), cds[F] float2 (candidates), cid_t[F] / cid_c[F] short (grid cell
// of each tracked point / candidate, -1 = outside), cellp[4][F] float2 (per-wave ChessGrid cell).
// n_cand_dev != NULL: the corner count lives on the device (device detector), n_cand is ignored.
// Track book-keeping in two launches (Tracker.cc:271-393):
//   bookkeep_a_kernel   lost tracks -> type '1', tracks at the maximum length -> type '2', history roll, survivors in the new order:
//                       everything the UPDATER needs (mvFeatTypesForUpdate / mvlFeatMeasForUpdate).  Needs KLT + RANSAC only.
//   bookkeep_b_kernel   FindNewer / ChessGrid refill with the detector's corners, the list of features to track next (and the seeding of
//                       the first image).  Needs the detector — which is the long pole of the front end (CLAHE + GFTT + cornerSubPix
//                       ~160 us against ~90 us for pyramid + KLT + RANSAC): with the hand-over out before it finishes, the filter of a frame
//                       starts ~70 us earlier (pose latency), and in the pipelined path it no longer depends on the image chain at all.
// done / done_target (single instance, run-ahead mode): the device-side counter the filter of frame k-2 bumps when its last kernel has
// finished (rvio_dev.h StageSync) — the hand-over tables the first kernel rewrites are free then.  A stream-level event in its place costs the
// FILTER stream a marker packet per frame (~9 us of its serial chain); this costs one poll here.
// n <= max_len entries of a track history from src to dst, eight loads in flight per step: every load of a chunk is issued before its first
// store (a load-store pair per element is a dependent global round trip each: ~1 us per observation on a one-workgroup kernel of the serial
// chain).  dst may be src - shift (the in-place roll of a full track): a chunk reads [k0 + shift, k0 + 8 + shift) and then writes [k0, k0 + 8),
// which no later chunk reads.
__device__ __forceinline__ void hist_copy(float2* dst, const float2* src, int n) {
    for (int k0 = 0; k0 < n; k0 += 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(k0 + u < n) ? k0 + u : n - 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k0 + u < n) dst[k0 + u] = v[u];
    }
}
__device__ __forceinline__ void bookkeep_a_body(const DevCfg& cfg, TrackerDev t, size_t bs,
                                                const unsigned long long* done, unsigned long long done_target, FilterMeta* meta) {
    DBG_S(blockIdx.z == 0, 3);
    if (done && !stage_wait(done, done_target, meta)) return;   // (timed out: error bit 4 is set, nothing is rewritten)
    DBG_S(blockIdx.z == 0, 4);
    DBG_U(7);
    tracker_shift(t, (size_t)blockIdx.z * bs);
    __shared__ int s_w[16];
    const int tid = threadIdx.x, T = blockDim.x, Fu = cfg.Fu, ML = cfg.max_len;
    const int N = *t.n_pts;
    const int first = *t.first;
    float2* hist = (float2*)t.hist;
    float2* meas = (float2*)t.meas;
    const float2* tr2 = (const float2*)t.tracked;
    const float2* un2 = (const float2*)t.un2;
    float2* tfs = (float2*)t.tmp_feats;
    float2* tu = (float2*)t.tmp_un;
    int nMeas = 0;
    if (first) {   // first image (Tracker.cc:204-234): nothing to hand over, the refill half seeds the slots
        if (tid == 0) { t.mid[0] = 1; t.mid[1] = 0; t.mid[2] = 0; *t.n_feat = 0; t.info->n_feat_update = 0; }
        return;
    }
    // ---- lost tracks -> type '1' (Tracker.cc:279-303), in feature order
    for (int base = 0; base < N; base += T) {
        const int i = base + tid;
        // (flag and slot are fetched side by side, the history length right behind: two dependent round trips instead of three)
        const bool in = i < N;
        const unsigned char stt = in ? t.status[i] : (unsigned char)1;
        const int slot = in ? t.slot[i] : 0;
        const int hl = t.hist_len[slot];
        const bool lost = in && !stt;
        const int emit = (lost && hl >= cfg.min_len) ? 1 : 0;
        int tot;
        const int pos = nMeas + block_exscan_n(emit, &tot, s_w);
        if (emit && pos < Fu) {
            t.types[pos] = '1'; t.len[pos] = hl;
            hist_copy(meas + (size_t)pos * ML, hist + (size_t)slot * ML, hl);
        }
        if (lost) t.hist_len[slot] = 0;
        nMeas = (nMeas + tot < Fu) ? nMeas + tot : Fu;
    }
    DBG_U(8);
    // ---- tracked features (Tracker.cc:305-342): type '2' at max length, history roll, new order
    int nIn = 0;
    const int keep = ML - ((ML + 1) / 2 - 1);   // mnMaxTrackingLength-(ceil(.5*max)-1), Tracker.cc:326
    for (int base = 0; base < N; base += T) {
        const int i = base + tid;
        const bool in = i < N;
        const int ii = in ? i : 0;
        const unsigned char stt = in ? t.status[i] : (unsigned char)0;
        const int slot = t.slot[ii];
        const float2 pt = tr2[ii], pu = un2[ii];
        int hl = t.hist_len[slot];
        const bool trk = in && stt;
        const int full = (trk && hl == ML) ? 1 : 0;
        int tot2, totT;
        const int pos2 = nMeas + block_exscan_n(full, &tot2, s_w);
        const int posT = nIn + block_exscan_n(trk ? 1 : 0, &totT, s_w);
        if (trk) {
            float2* hs = hist + (size_t)slot * ML;
            if (full) {
                int shift = 1;
                if (pos2 < Fu) {
                    t.types[pos2] = '2'; t.len[pos2] = hl;
                    hist_copy(meas + (size_t)pos2 * ML, hs, hl);
                    shift = ML - keep;
                }
                hist_copy(hs, hs + shift, hl - shift);
                hl -= shift;
            }
            hs[hl] = pu;
            t.hist_len[slot] = hl + 1;
            tfs[posT] = pt; tu[posT] = pu; t.tmp_slot[posT] = slot;
        }
        nMeas = (nMeas + tot2 < Fu) ? nMeas + tot2 : Fu;
        nIn += totT;
    }
    if (tid == 0) { t.mid[0] = 0; t.mid[1] = nIn; t.mid[2] = nMeas; *t.n_feat = nMeas; t.info->n_feat_update = nMeas; }
    DBG_U(9);
}
// hand (single instance, run-ahead mode): the counter the gate in front of this frame's filter polls — the Updater's input is complete
__global__ __launch_bounds__(256) void bookkeep_a_kernel(DevCfg cfg, TrackerDev t, size_t bs,
                                                         const unsigned long long* done, unsigned long long done_target, FilterMeta* meta,
                                                         unsigned long long* hand) {
    bookkeep_a_body(cfg, t, bs, done, done_target, meta);
    if (hand) stage_signal(hand);
}
// The filter of a frame starts behind the hand-over half of that frame's book-keeping.  As a stream-level event that wait cost the filter
// stream ~19 us of its serial chain per frame in the pipelined run (an AQL barrier packet between augcomp and the per-feature launch, with
// four queues busy); this one-workgroup kernel polls the device-side counter instead (~5 us, launch gaps included).  One workgroup only:
// a poll inside the ~100 LDS-heavy workgroups of the per-feature launch is a priority inversion.
// A frame sequence broken by a timed-out counter (error bit 4: the hand-over half of book-keeping left WITHOUT rewriting its table, or this
// poll itself gave up) must not reach the filter as a stale hand-over table: the gate empties the table of its frame — race-free here, the
// table's previous reader (the filter of frame k - kHand) ran earlier on this very stream — and the update of the frame passes the state through.
__global__ __launch_bounds__(64) void stage_gate_kernel(const unsigned long long* ctr, unsigned long long target, FilterMeta* meta, int* n_feat, int dbg_tag) {
    DBG_I(true, dbg_tag, 4);
    const bool ok = stage_wait(ctr, target, meta);
    if (threadIdx.x == 0 && (!ok || (__hip_atomic_load(&meta->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4))) *n_feat = 0;
    DBG_I(true, dbg_tag, 5);
}
// ... and the other direction of the same idea: one workgroup behind cornerSubPix says "the corners of this frame are final"
__global__ __launch_bounds__(64) void stage_signal_kernel(unsigned long long* ctr) { stage_signal(ctr); }
// RANSAC and the hand-over half of book-keeping in ONE launch (both are one-workgroup stages of the side stream's serial chain, nothing
// separates them since the hand-over does not wait for the detector): one launch boundary less before the filter may start
__global__ __launch_bounds__(256) void ransac_book_a_kernel(DevCfg cfg, TrackerDev t, const rvio_imu* imu, int m, int* rng, size_t bs, size_t imu_bs,
                                                            const unsigned long long* done, unsigned long long done_target, FilterMeta* meta,
                                                            unsigned long long* hand) {
    extern __shared__ __align__(16) unsigned char dsh_ra[];
    ransac_body(cfg, t.n_pts, t.tracked, t.un1, t.un2, t.status, imu, m, rng, t.info, bs, imu_bs, dsh_ra);
    __threadfence_block();
    __syncthreads();
    bookkeep_a_body(cfg, t, bs, done, done_target, meta);
    if (hand) stage_signal(hand);
}

// corners / corners_target (single instance, run-ahead mode): the counter stage_signal_kernel bumps behind this frame's cornerSubPix
__device__ __forceinline__ void bookkeep_b_body(const DevCfg& cfg, TrackerDev t, const float* cand, int n_cand, const int* n_cand_dev, size_t bs,
                                                const unsigned long long* corners, unsigned long long corners_target, FilterMeta* meta, unsigned char* dsh) {
    if (corners && !stage_wait(corners, corners_target, meta)) return;   // (timed out: error bit 4 is set)
    DBG_S(blockIdx.z == 0, 6);
    DBG_U(10);
    tracker_shift(t, (size_t)blockIdx.z * bs); cand = zoff(cand, bs); if (n_cand_dev) n_cand_dev = zoff(n_cand_dev, bs);
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, F = cfg.F, ML = cfg.max_len;
    const int T = blockDim.x, nwv = T >> 6;   // 4 waves for batch handles, up to 16 for one stream: the cell walk below is one wave per grid cell
    float2* tfs = (float2*)dsh;
    float2* cds = tfs + F;
    short* cid_t = (short*)(cds + F);
    short* cid_c = cid_t + F;
    float2* cellp = (float2*)(dsh + (((size_t)20 * F + 7) & ~(size_t)7));        // 8-byte aligned
    const int first = t.mid[0], nIn = t.mid[1];
    float2* hist = (float2*)t.hist;
    float2* feats = (float2*)t.feats;
    float2* un1 = (float2*)t.un1;
    float2* tu = (float2*)t.tmp_un;
    if (n_cand_dev) n_cand = *n_cand_dev;
    const int nc = n_cand < F ? n_cand : F;
    for (int c = tid; c < nc; c += T) cds[c] = make_float2(cand[2 * c], cand[2 * c + 1]);
    if (first) {
        // first image, Tracker.cc:204-234: seed every slot with a detector corner
        __syncthreads();
        const int n0 = nc;
        for (int i = tid; i < F; i += T) {
            if (i < n0) {
                float ux, uy;
                undistort_pt(cfg, cds[i].x, cds[i].y, &ux, &uy);
                feats[i] = cds[i];
                un1[i] = make_float2(ux, uy);
                hist[(size_t)i * ML] = make_float2(ux, uy);
                t.hist_len[i] = 1; t.slot[i] = i;
            } else t.hist_len[i] = 0;
        }
        if (tid == 0) {
            *t.n_pts = n0;
            if (n0 > 0) { *t.first = 0; if (t.first_host) t.first_host[blockIdx.z] = 0; }
            t.info->n_tracked_in = 0; t.info->n_klt_ok = 0; t.info->n_ransac_inliers = 0; t.info->ransac_winner = 0;
            t.info->n_tracked_out = n0;
        }
        return;
    }
    for (int i = tid; i < nIn; i += T) tfs[i] = ((const float2*)t.tmp_feats)[i];   // the survivors, in the order the first half gave them
    __syncthreads();
    DBG_U(11);
    // ---- refill (Tracker.cc:344-387) through FindNewer/ChessGrid (FeatureDetector.cc:78-150)
    int nNew = 0;
    if (nIn < F && nc > 0) {
        const int cells = cfg.grid_cols * cfg.grid_rows;
        const float W = (float)cfg.W, H = (float)cfg.H, offX = cfg.off_x, offY = cfg.off_y;
        // grid cell of every tracked point and candidate (-1: outside / too close to a block edge)
        for (int i = tid; i < nIn; i += T) {
            const float2 p = tfs[i];
            int cell = -1;
            if (!(p.x <= offX || p.y <= offY || p.x >= (W - offX) || p.y >= (H - offY))) {
                const int col = (int)floorf((p.x - offX) / cfg.block_x), row = (int)floorf((p.y - offY) / cfg.block_y);
                cell = row * cfg.grid_cols + col;
            }
            cid_t[i] = (short)cell;
        }
        for (int c = tid; c < nc; c += T) {
            const float2 p = cds[c];
            int cell = -1;
            if (!(p.x <= offX || p.y <= offY || p.x >= (W - offX) || p.y >= (H - offY))) {
                const int col = (int)floorf((p.x - offX) / cfg.block_x), row = (int)floorf((p.y - offY) / cfg.block_y);
                const float xl = col * cfg.block_x + offX, xr = xl + cfg.block_x, yt = row * cfg.block_y + offY, yb = yt + cfg.block_y;
                if (!(fabsf(p.x - xl) < cfg.min_dist || fabsf(p.x - xr) < cfg.min_dist || fabsf(p.y - yt) < cfg.min_dist || fabsf(p.y - yb) < cfg.min_dist))
                    cell = row * cfg.grid_cols + col;
            }
            cid_c[c] = (short)cell;
            t.cand_acc[c] = 0;
        }
        __syncthreads();
        DBG_U(12);
        // one wave per grid cell: gather the cell's tracked points, then walk its candidates in detector order
        const double d2max = sqrt_gt_threshold((double)cfg.min_dist);
        float2* cp = cellp + (size_t)wv * F;
        for (int cell = wv; cell < cells; cell += nwv) {
            int cn = 0;
            for (int base = 0; base < nIn; base += 64) {
                const int i = base + lane;
                const bool in = (i < nIn) && (cid_t[i] == cell);
                const unsigned long long bal = __ballot(in);
                if (in) cp[cn + __popcll(bal & ((1ull << lane) - 1ull))] = tfs[i];
                cn += __popcll(bal);
            }
            for (int base = 0; base < nc; base += 64) {
                const int c = base + lane;
                unsigned long long bal = __ballot((c < nc) && (cid_c[c] == cell));
                while (bal) {
                    const int bit = __ffsll((long long)bal) - 1;
                    bal &= bal - 1ull;
                    const int cc = base + bit;
                    if (!((double)(float)cn < .75 * (double)cfg.max_per_block)) continue;
                    const float2 p = cds[cc];
                    bool close = false;
                    for (int q0 = 0; q0 < cn; q0 += 64) {
                        const int q = q0 + lane;
                        bool cl = false;
                        if (q < cn) {
                            const float dx = p.x - cp[q].x, dy = p.y - cp[q].y;
                            cl = !((double)dx * dx + (double)dy * dy > d2max);    // == !(sqrt(.) > min_dist), see sqrt_gt_threshold
                        }
                        if (__ballot(cl)) { close = true; break; }
                    }
                    if (!close) { if (lane == 0) { cp[cn] = p; t.cand_acc[cc] = 1; } cn++; }
                }
            }
        }
        __syncthreads();
        DBG_U(13);
        // accepted candidates keep detector order; the k-th accepted takes the k-th free slot
        const int room = F - nIn;
        for (int base = 0; base < nc; base += T) {
            const int c = base + tid;
            const int ac = (c < nc) ? t.cand_acc[c] : 0;
            int tot;
            const int k = nNew + block_exscan_n(ac, &tot, s_w);
            if (ac && k < room) tfs[nIn + k] = cds[c];
            nNew = (nNew + tot < room) ? nNew + tot : room;
        }
        __syncthreads();
        DBG_U(14);
        int nFree = 0;
        for (int base = 0; base < F; base += T) {
            const int s = base + tid;
            const int fr = (s < F && t.hist_len[s] == 0) ? 1 : 0;
            int tot;
            const int k = nFree + block_exscan_n(fr, &tot, s_w);
            if (fr && k < nNew) {
                float ux, uy;
                const float2 p = tfs[nIn + k];
                undistort_pt(cfg, p.x, p.y, &ux, &uy);
                tu[nIn + k] = make_float2(ux, uy);
                t.tmp_slot[nIn + k] = s;
                hist[(size_t)s * ML] = make_float2(ux, uy);
                t.hist_len[s] = 1;
            }
            nFree += tot;
        }
    }
    __syncthreads();
    DBG_U(15);
    const int nOut = nIn + nNew;
    for (int i = tid; i < nOut; i += T) { feats[i] = tfs[i]; un1[i] = tu[i]; t.slot[i] = t.tmp_slot[i]; }
    DBG_S(blockIdx.z == 0, 5);
    if (tid == 0) { *t.n_pts = nOut; t.info->n_tracked_out = nOut; }
    DBG_U(16);
}

__global__ __launch_bounds__(1024) void bookkeep_b_kernel(DevCfg cfg, TrackerDev t, const float* cand, int n_cand, const int* n_cand_dev, size_t bs,
                                                          const unsigned long long* corners, unsigned long long corners_target, FilterMeta* meta) {
    extern __shared__ __align__(16) unsigned char dsh[];
    bookkeep_b_body(cfg, t, cand, n_cand, n_cand_dev, bs, corners, corners_target, meta, dsh);
}
// (round 6) RANSAC and BOTH halves of book-keeping in one launch, for one stream in run-ahead mode: the hand-over counter is bumped between the
// halves exactly where the two-kernel form bumps it, the refill half polls the detector's counter as before — one launch boundary (~8 us) less
// on the tracker's serial chain, and up to 16 waves: the 16 RANSAC models are counted by four groups of threads, the ChessGrid walked a cell per wave.
// Dynamic LDS: max(RANSAC's 8 F + 16, book-keeping's) — the two stages use it one after the other.
__global__ __launch_bounds__(1024) void ransac_book_kernel(DevCfg cfg, TrackerDev t, const rvio_imu* imu, int m, int* rng, size_t bs, size_t imu_bs,
                                                           const unsigned long long* done, unsigned long long done_target, FilterMeta* meta,
                                                           unsigned long long* hand, const float* cand, const int* n_cand_dev,
                                                           const unsigned long long* corners, unsigned long long corners_target) {
    extern __shared__ __align__(16) unsigned char dsh_rb[];
    ransac_body(cfg, t.n_pts, t.tracked, t.un1, t.un2, t.status, imu, m, rng, t.info, bs, imu_bs, dsh_rb);
    __threadfence_block();
    __syncthreads();
    bookkeep_a_body(cfg, t, bs, done, done_target, meta);
    // the refill half reads what the hand-over half wrote through global memory (tmp_feats, tmp_un, tmp_slot, mid, hist_len): an agent-scope release by every
    // wave + a workgroup barrier — exactly what stage_signal does in front of its atomic, so with a hand-over counter nothing more is needed (round 6: a second
    // device-scope fence here — an L2 write-back on the multi-XCD part — sat on the side chain's serial path)
    if (hand) stage_signal(hand);
    else { __threadfence(); __syncthreads(); }
    bookkeep_b_body(cfg, t, cand, 0, n_cand_dev, bs, corners, corners_target, meta, dsh_rb);
}
// direct-track mode: the caller supplies vFeatsTracked / vInlierFlag (the KLT result)
__global__ void load_points_kernel(const int* n_pts_ptr, const float* in_xy, const unsigned char* in_st, float* tracked, unsigned char* status) {
    const int N = *n_pts_ptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        tracked[2 * i] = in_xy[2 * i]; tracked[2 * i + 1] = in_xy[2 * i + 1]; status[i] = in_st[i];
    }
}
