// rvio_dev.h — device-side data layout and small math helpers (gfx950 only).
//
// HBM layout of one filter instance (all owned by the rvio_hip handle):
//   FilterState[2]   double-buffered (x, P): every stage that is not a
//                    single-workgroup in-place update reads buffer `cur` and
//                    writes `cur^1`; the host toggles `cur` (the toggle count
//                    per entry point is data-independent, so no host sync).
//   x     : xdmax doubles  [qG pG g | qk pk v bg ba | n x (q p)]
//   P     : dmax x dmax doubles, COLUMN-major, ld = dmax; the active matrix is
//           the leading d x d block, d = 24 + 6 n.
//   n_clones lives in device memory (FilterMeta) so captured graphs stay valid
//   while the window fills.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RVIO_MAX_IMU 192     // IMU samples per call (= RVIO_HIP_MAX_IMU of the header): RANSAC forms one delta rotation per thread
#define RVIO_MAX_LEN 32      // max Tracker.nMaxTrackingLength supported (cfg E: 31)

struct DevCfg {
    double gravity, small_angle;
    double sg2, swg2, sa2, swa2;  // squared IMU sigmas (ImuNoiseMatrix diagonal, PreIntegrator.cc:40-44)
    double sigma_im;              // max(sigma_px, sigma_py) as float->double (Updater.cc:42-44)
    double inlier_thr;
    double Ric[9], Rci[9], tic[3], tci[3];  // row-major 3x3 (Updater.cc:46-53)
    float fx, fy, cx, cy, k1, k2, p1, p2, k3;
    float min_dist, off_x, off_y, max_per_block;  // FeatureDetector.cc:29-52; the last three hold the reference's INT members (FeatureDetector.h:69-77)
    float block_x, block_y;                       // float members upstream (FeatureDetector.h:72-73)
    int W, H;
    int F, Fu, max_len, min_len;
    int nmax, dmax, xdmax;
    int rho_max;   // max nullspace rows per feature = 2*max_len - 2
    int ldh;       // row stride (doubles) of stacked [Hx | r] rows = 6*nmax + 1
    int grid_cols, grid_rows;
    int use_sampson;
    int fisheye;   // Camera.Fisheye: cv::fisheye::undistortPoints instead of cv::undistortPoints (Tracker.cc:116-119)
    int levels;    // pyramid levels actually used (maxLevel+1)
};

struct FilterMeta {
    int n_clones;      // nCloneStates (System.cc:175)
    int img_count;     // nImageCountAfterInit (System.cc:176)
    int n_good;        // nGoodFeatCount of the last update
    int n_rows;        // nRowCount of the last update
    int updated;       // last update applied?
    int err;           // sticky device-side error flag (singular pivot etc.)
    int trunc_at;      // column at which the reference's rank scan stopped and dropped the type-'1' rows (Updater.cc:516-529), -1: no such truncation
    int pad;
};

// phase stamps for performance debugging (tools/dbg_clocks.py): DBG_T(i) records clock64() in slot i
// Batched launches: gridDim.z = filter instances; every per-instance buffer of instance z lies `bs`
// bytes behind instance 0's (one slab per instance, rvio_hip.hip).  bs = 0 and gridDim.z = 1 for a plain handle.
template <typename T>
__device__ __forceinline__ T* zoff(T* p, size_t bs) { return (T*)((char*)p + (size_t)blockIdx.z * bs); }
template <typename T>
__device__ __forceinline__ T* zoffi(T* p, size_t bs, int z) { return (T*)((char*)p + (size_t)z * bs); }
// XCD-aware (x, y, instance) of a workgroup of a batched launch.  The dispatcher is observed to place linear block b on XCD b % 8
// (a speed assumption only, MI355X_MICROARCH.md "Workgroup dispatch"): the remap gives every workgroup of one instance the same
// b % 8, so the strips / tiles / features of an instance share one XCD's L2 instead of fetching their common operands eight
// times.  Bijective when gridDim.z is a multiple of 8; identity otherwise (and for a plain handle, gridDim.z = 1).
struct BatchIdx { int x, y, z; };
__device__ __forceinline__ BatchIdx batch_remap() {
    const unsigned X = gridDim.x, Y = gridDim.y, B = gridDim.z;
    if (B & 7u) return {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const unsigned w = blockIdx.x + X * (blockIdx.y + Y * blockIdx.z);
    const unsigned xcd = w & 7u, slot = w >> 3, per = X * Y;
    const unsigned inner = slot % per, grp = slot / per;
    return {(int)(inner % X), (int)(inner / X), (int)(grp * 8u + xcd)};
}
// (measured, B = 2048: the remap pays for ug / final / gemm_T — few workgroups per instance sharing W, A, Pc, G, U — and costs
// 8-12 % for feat_build / gram_mfma, whose active workgroups are the first few of 100 / 13 slots: those keep the plain order)
__device__ __forceinline__ BatchIdx batch_plain() { return {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z}; }
// strides (bytes) of the per-instance inputs of a batched call: IMU samples and the Tracker -> Updater hand-over
struct BatchIn { size_t imu, n_feat, types, len, meas; };

// ---- wire format of a shard's share of the information block (the all-gather payload of the feature-sharded updater).
// [8 counters | S2 tiles | S1 tiles], 16 x 16 tiles of 256 doubles (row-major inside a tile), only what can be non-zero and is not a mirror image:
//   S1 (sum over the type-'1' features, any column range): the tiles (pt, qt) on and above the diagonal, pt < ntp = ceil(6n / 16), qt < ntq = tiles up to the
//      residual column 6n — row pt of the triangle holds ntq - pt tiles;
//   S2 (sum over the type-'2' features: full-length tracks, columns [0, 6 (ceil(max_len / 2) - 1)) of the OLDEST clones): the triangle of the tiles up
//      to t2 + the residual-column tile of each of its rows (x2 = 1 when that tile lies beyond t2).
// cfg E (6n = 180, 31 observations): 27 + 78 tiles = 215 KB instead of the 2 (6n + 1)^2 doubles = 524 KB of rounds 2-5; cfg B: 31 KB instead of 60 KB.
struct ShardLayout { int ntq, ntp, t2, x2, tiles2, tiles1; };
__host__ __device__ inline ShardLayout shard_layout(int c6, int max_len) {
    ShardLayout L;
    L.ntq = (c6 >> 4) + 1; L.ntp = (c6 + 15) >> 4;
    int hi2 = 6 * ((max_len + 1) / 2 - 1);
    if (hi2 > c6) hi2 = c6;
    L.t2 = hi2 > 0 ? (hi2 - 1) >> 4 : -1;
    if (L.t2 > L.ntp - 1) L.t2 = L.ntp - 1;
    L.x2 = (L.ntq - 1 > L.t2) ? 1 : 0;
    L.tiles2 = L.t2 < 0 ? 0 : (L.t2 + 1) * (L.t2 + 2) / 2 + L.x2 * (L.t2 + 1);
    L.tiles1 = L.ntp * L.ntq - L.ntp * (L.ntp - 1) / 2;
    return L;
}
__host__ __device__ inline int shard_payload_doubles(int c6, int max_len) { const ShardLayout L = shard_layout(c6, max_len); return 8 + 256 * (L.tiles2 + L.tiles1); }
// offset (doubles, from the start of the payload) of tile (pt, qt) of part 2 / part 1, -1 if the tile is not carried (it is zero)
__host__ __device__ inline int shard_tile2(const ShardLayout& L, int pt, int qt) {
    if (pt > L.t2 || qt < pt) return -1;
    const int row = pt * (L.t2 + 1 + L.x2) - pt * (pt - 1) / 2;
    if (qt <= L.t2) return 8 + 256 * (row + qt - pt);
    if (L.x2 && qt == L.ntq - 1) return 8 + 256 * (row + L.t2 + 1 - pt);
    return -1;
}
__host__ __device__ inline int shard_tile1(const ShardLayout& L, int pt, int qt) {
    if (pt >= L.ntp || qt < pt || qt >= L.ntq) return -1;
    return 8 + 256 * (L.tiles2 + pt * L.ntq - pt * (pt - 1) / 2 + qt - pt);
}

__device__ long long g_dbg[64];
__device__ long long g_dbg3[64];   // (round 6) feat_build_body's phases over all workgroups: sums [0..15], counts [32..47] (DBG_P)
__device__ long long g_dbg2[64];   // (round 6) phase stamps of the side chain's kernels: klt_kernel3 workgroup 0, RANSAC, both halves of book-keeping (tools/side_phase_clocks.py)
// (instrumented build only) start stamps of the filter chain's stages, one row of 8 per frame in a ring of 64 frames, on the constant
// 100 MHz clock all CUs share: tools/chain_clocks.py turns them into the in-situ timeline of the pipelined run
__device__ long long g_ring[64 * 8];
__device__ int g_ring_frame;
__device__ long long g_ring2[64 * 8];   // the same for the side stream's chain (pyramid, KLT, RANSAC, book-keeping)
__device__ int g_ring2_frame;
__device__ long long g_ring3[64 * 8];   // the image chain of frame `tag` (CLAHE ... cornerSubPix; the chains of consecutive frames overlap on two streams)
#ifdef RVIO_DBG_CLOCKS
#define DBG_I(cond, tag, id) do { if (threadIdx.x == 0 && (cond)) g_ring3[((tag) & 63) * 8 + (id)] = wall_clock64(); } while (0)
#else
#define DBG_I(cond, tag, id) do { } while (0)
#endif
#ifdef RVIO_DBG_CLOCKS
#define DBG_T(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_dbg[i] = clock64(); } while (0)
#define DBG_W(cond, i) do { if (cond) g_dbg[i] = wall_clock64(); } while (0)   /* constant 100 MHz clock: comparable across kernels */
#define DBG_U(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.z == 0) g_dbg2[i] = wall_clock64(); } while (0)
/* per-phase durations over ALL workgroups of a launch (10 ns ticks): g_dbg2[i] = the longest, g_dbg3[i] = the sum, g_dbg3[i + 32] = the count (i < 32 after the offset) */
#define DBG_P0() long long dbg_prev_ = wall_clock64()
#define DBG_P(i) do { if (threadIdx.x == 0 && (gridDim.z == 1 || ((blockIdx.z & 63) == 0 && blockIdx.x < 4))) {   /* (a batch: a sample of the workgroups — the atomics of 200 k workgroups would be the kernel) */ const long long t_ = wall_clock64(); atomicMax((unsigned long long*)&g_dbg2[i], (unsigned long long)(t_ - dbg_prev_)); \
                      atomicAdd((unsigned long long*)&g_dbg3[(i) - 30], (unsigned long long)(t_ - dbg_prev_)); atomicAdd((unsigned long long*)&g_dbg3[(i) + 2], 1ull); dbg_prev_ = t_; } } while (0)
#define DBG_R(cond, id) do { if (threadIdx.x == 0 && (cond)) { if ((id) == 0) g_ring_frame = g_ring_frame + 1; g_ring[(g_ring_frame & 63) * 8 + (id)] = wall_clock64(); } } while (0)
#define DBG_S(cond, id) do { if (threadIdx.x == 0 && (cond)) { if ((id) == 0) g_ring2_frame = g_ring2_frame + 1; g_ring2[(g_ring2_frame & 63) * 8 + (id)] = wall_clock64(); } } while (0)
#else
#define DBG_T(i) do { } while (0)
#define DBG_U(i) do { } while (0)
#define DBG_P0() do { } while (0)
#define DBG_P(i) do { } while (0)
#define DBG_W(cond, i) do { } while (0)
#define DBG_R(cond, id) do { } while (0)
#define DBG_S(cond, id) do { } while (0)
#endif

// ---------------------------------------------------------------- small math
struct d3 { double x, y, z; };
struct m33 { double m[9]; };  // row-major

__device__ __forceinline__ d3 mk3(double a, double b, double c) { d3 r; r.x = a; r.y = b; r.z = c; return r; }
__device__ __forceinline__ d3 add3(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 sub3(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 scl3(double s, d3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ double nrm3(d3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ d3 unit3(d3 a) { double n = nrm3(a); return mk3(a.x / n, a.y / n, a.z / n); }
__device__ __forceinline__ d3 ld3(const double* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(double* p, d3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

__device__ __forceinline__ m33 eye33() { m33 r; for (int i = 0; i < 9; ++i) r.m[i] = (i % 4 == 0) ? 1.0 : 0.0; return r; }
__device__ __forceinline__ m33 ldm33(const double* p) { m33 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
__device__ __forceinline__ m33 mul33(const m33& A, const m33& B) {
    m33 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
__device__ __forceinline__ d3 mv33(const m33& A, d3 v) {
    return mk3(A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z, A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z);
}
__device__ __forceinline__ m33 tr33(const m33& A) { m33 C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * j + i]; return C; }
__device__ __forceinline__ m33 add33(const m33& A, const m33& B) { m33 C; for (int i = 0; i < 9; ++i) C.m[i] = A.m[i] + B.m[i]; return C; }
__device__ __forceinline__ m33 sub33(const m33& A, const m33& B) { m33 C; for (int i = 0; i < 9; ++i) C.m[i] = A.m[i] - B.m[i]; return C; }
__device__ __forceinline__ m33 scl33(double s, const m33& A) { m33 C; for (int i = 0; i < 9; ++i) C.m[i] = s * A.m[i]; return C; }
// SkewSymm, util/Numerics.h:97-105
__device__ __forceinline__ m33 skew33(d3 w) {
    m33 S;
    S.m[0] = 0; S.m[1] = -w.z; S.m[2] = w.y;
    S.m[3] = w.z; S.m[4] = 0; S.m[5] = -w.x;
    S.m[6] = -w.y; S.m[7] = w.x; S.m[8] = 0;
    return S;
}

struct q4 { double x, y, z, w; };
__device__ __forceinline__ q4 ldq(const double* p) { q4 q; q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3]; return q; }
__device__ __forceinline__ void stq(double* p, q4 q) { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }
__device__ __forceinline__ q4 qnorm_pos(q4 q) {
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    return q;
}
// QuatMul, util/Numerics.h:30-63 (JPL; normalises, forces w>=0)
__device__ __forceinline__ q4 qmul(q4 a, q4 b) {
    q4 q;
    q.x = a.w * b.x + a.z * b.y - a.y * b.z + a.x * b.w;
    q.y = -a.z * b.x + a.w * b.y + a.x * b.z + a.y * b.w;
    q.z = a.y * b.x - a.x * b.y + a.w * b.z + a.z * b.w;
    q.w = -a.x * b.x - a.y * b.y - a.z * b.z + a.w * b.w;
    return qnorm_pos(q);
}
// QuatToRot, util/Numerics.h:111-120: I - 2 w [q]x + 2 [q]x^2
__device__ __forceinline__ m33 q2r(q4 q) {
    m33 qx = skew33(mk3(q.x, q.y, q.z));
    return add33(sub33(eye33(), scl33(2 * q.w, qx)), scl33(2.0, mul33(qx, qx)));
}
// RotToQuat, util/Numerics.h:126-167 (Breckenridge 4-branch)
__device__ __forceinline__ q4 r2q(const m33& R) {
    q4 q;
    const double r00 = R.m[0], r11 = R.m[4], r22 = R.m[8];
    const double T = r00 + r11 + r22;
    if (r00 > T && r00 > r11 && r00 > r22) {
        q.x = sqrt((1 + 2 * r00 - T) / 4); double k = 1 / (4 * q.x);
        q.y = k * (R.m[1] + R.m[3]); q.z = k * (R.m[2] + R.m[6]); q.w = k * (R.m[5] - R.m[7]);
    } else if (r11 > T && r11 > r00 && r11 > r22) {
        q.y = sqrt((1 + 2 * r11 - T) / 4); double k = 1 / (4 * q.y);
        q.x = k * (R.m[1] + R.m[3]); q.z = k * (R.m[5] + R.m[7]); q.w = k * (R.m[6] - R.m[2]);
    } else if (r22 > T && r22 > r00 && r22 > r11) {
        q.z = sqrt((1 + 2 * r22 - T) / 4); double k = 1 / (4 * q.z);
        q.x = k * (R.m[2] + R.m[6]); q.y = k * (R.m[5] + R.m[7]); q.w = k * (R.m[1] - R.m[3]);
    } else {
        q.w = sqrt((1 + T) / 4); double k = 1 / (4 * q.w);
        q.x = k * (R.m[5] - R.m[7]); q.y = k * (R.m[6] - R.m[2]); q.z = k * (R.m[1] - R.m[3]);
    }
    return qnorm_pos(q);
}
// dq from an error angle (Updater.cc:549-563)
__device__ __forceinline__ q4 small_q(double ex, double ey, double ez) {
    q4 q; q.x = .5 * ex; q.y = .5 * ey; q.z = .5 * ez;
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (n < 1) q.w = sqrt(1 - n * n);
    else { double k = 1 / sqrt(1 + n * n); q.x *= k; q.y *= k; q.z *= k; q.w = k; }
    return q;
}

// ---------------------------------------------------------------- wave helpers (wave64)
// All-reduce over the 64 lanes: 4 DPP row-rotate steps give every lane its 16-lane row total, then the
// four row totals are read with v_readlane and added in a fixed order.  ~25 VALU/SALU instructions instead of
// 6 ds_bpermute round trips; the result is uniform and identical in every lane.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0x128>(v);   // row_ror:8
    v += dpp_f64<0x124>(v);   // row_ror:4
    v += dpp_f64<0x122>(v);   // row_ror:2
    v += dpp_f64<0x121>(v);   // row_ror:1
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
// sum over lanes 0..15 (the first DPP row); identical to wave_sum when lanes >= 16 hold zeros
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_f64<0x128>(v);
    v += dpp_f64<0x124>(v);
    v += dpp_f64<0x122>(v);
    v += dpp_f64<0x121>(v);
    return readlane_f64(v, 0);
}
__device__ __forceinline__ double row16_allsum(double v) {   // the same four rotations in every 16-lane row; the first lane of a row holds row16_sum's bits
    v += dpp_f64<0x128>(v);
    v += dpp_f64<0x124>(v);
    v += dpp_f64<0x122>(v);
    v += dpp_f64<0x121>(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ long long dpp_i64(long long v) {
    int lo = (int)(v & 0xffffffffLL), hi = (int)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long readlane_i64(long long v, int l) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffLL), l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {   // exact: order-free
    v += dpp_i64<0x128>(v);
    v += dpp_i64<0x124>(v);
    v += dpp_i64<0x122>(v);
    v += dpp_i64<0x121>(v);
    return (readlane_i64(v, 0) + readlane_i64(v, 16)) + (readlane_i64(v, 32) + readlane_i64(v, 48));
}

// ---------------------------------------------------------------- device-side completion counter (augcomp_kernel2 -> bookkeep_kernel)
// device memory, zero at creation.  aug: bumped by every workgroup of the last kernel of a frame's filter chain (-> book-keeping of frame k+2);
// handover: bumped once by the hand-over half of book-keeping (-> the gate in front of the filter of the same frame);
// corners[c]: bumped once behind cornerSubPix of image chain c (-> the refill half of book-keeping).  ONE PRODUCER QUEUE PER COUNTER: the image
// chains of consecutive frames run on different queues and nothing orders them against each other, so each chain counts its own frames
// (round 3 had one counter for both chains: chain k+1 finishing first let the refill of frame k read a half-written corner list).
// One-workgroup consumers poll them; a stream-level event in their place costs the WAITING stream ~10-20 us of its serial chain in the
// pipelined run (measured in situ, profiles/r03_chain_clocks.txt).
#define RVIO_MAX_IC 3
struct StageSync { unsigned long long aug, handover, corners[RVIO_MAX_IC], pyr[RVIO_MAX_IC]; };     // pyr (round 6): the pyramid of an image chain's frame is complete (-> klt_kernel3)
// every thread of the workgroup calls these
__device__ __forceinline__ void stage_signal(unsigned long long* c) {
    // every wave: its stores performed in the XCD's L2 (workgroup scope) — then ONE wave writes that L2 back (agent scope: cumulative over what the barrier
    // ordered before it) and bumps the counter.  (Round 6: every wave ran the agent-scope release itself — up to sixteen L2 write-backs per signal.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x < 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Returns false when the counter has not arrived within STAGE_WAIT_TICKS of the constant 100 MHz clock (30 s: the producer's queue is
// dead or starved beyond anything a shared / preempted GPU does).  The caller then must NOT rewrite what the producer may still read:
// it leaves, and the sticky error bit 4 makes the next rvio_hip_sync / rvio_hip_get_frame_info fail (RVIO_ERR_STATE).
#define STAGE_WAIT_TICKS 3000000000ull
__device__ __forceinline__ bool stage_wait(const unsigned long long* c, unsigned long long target, FilterMeta* meta) {
    __shared__ int s_stage_ok;
    if (threadIdx.x == 0) {
        int ok = 1;
        if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(4);
                // once a wait of this handle has timed out (sticky bit 4) the frame sequence is broken: every later poll gives up at once instead of
                // spinning its own 30 s — a caller that keeps enqueueing frames without a sync must not queue N x 30 s of dead GPU time
                if (__hip_atomic_load(&meta->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4) { ok = 0; break; }
                if (wall_clock64() - t0 > STAGE_WAIT_TICKS) { atomicOr(&meta->err, 4); ok = 0; break; }
            }
        }
        s_stage_ok = ok;
    }
    // this CU's vector cache and the XCD's L2 drop what the producers have rewritten: ONE wave's agent-scope acquire (the caches are the CU's / the XCD's, not a
    // wave's), made the workgroup's by the barrier behind it (round 6: every wave ran the invalidate)
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    return s_stage_ok != 0;
}
// the same for a ONE-WAVE workgroup (every lane polls the same word: one request; no LDS, no barrier)
__device__ __forceinline__ bool stage_wait_wave(const unsigned long long* c, unsigned long long target, FilterMeta* meta) {
    bool ok = true;
    if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (__hip_atomic_load(&meta->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4) { ok = false; break; }
            if (wall_clock64() - t0 > STAGE_WAIT_TICKS) { if ((threadIdx.x & 63) == 0) atomicOr(&meta->err, 4); ok = false; break; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}


