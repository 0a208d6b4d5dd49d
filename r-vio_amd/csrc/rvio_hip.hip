// rvio_hip.hip — the C-ABI of include/rvio_hip.h: handle, HBM allocation, kernel sequencing.
// Single translation unit: the kernel files are included so that one hipcc call builds the whole library
// (the front end — frontend_kernels / klt3 / klt16 / clahe / detector — sits inside an FP-contraction-off region).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rvio_hip.h"
#include "rvio_dev.h"
#include "frontend_dev.h"
#include "filter_kernels.hip"
#include "filter_kernels2.hip"
#include "solve6.hip"
#include "solve7.hip"
#include "solve9.hip"   // round 5: the solve as blocked SPD factorisations on the matrix cores (one instance; every window up to 6n = 192)
#ifndef S9_BATCH
#define S9_BATCH (ab_env("RVIO_S9_BATCH") != nullptr)   // A/B timing: batch handles at 6n <= 64 through solve9_kernel<1, 4> instead of gemm_T + solve6
#endif
#pragma clang fp contract(off)
#include "frontend_kernels.hip"
#include "klt3.hip"
#include "klt16.hip"
#include "clahe.hip"
#include "detector.hip"
#pragma clang fp contract(fast)

struct rvio_hip {
    rvio_config cfg;
    DevCfg dc;
    int device = 0;
    hipStream_t stream = nullptr;     // filter stream (and the stream of every non-pipelined call)
    hipStream_t stream_t = nullptr;   // tracker stream of the pipelined whole-frame path
    hipStream_t ts = nullptr;         // stream the tracker kernels of the call in progress go to
    hipEvent_t evT[4] = {nullptr, nullptr, nullptr, nullptr};   // book-keeping(k) done: a ring by frame number (the image chain of frame k waits for frame k-3's)
    hipEvent_t evH[4] = {nullptr, nullptr, nullptr, nullptr};   // hand-over of frame k written (bookkeep_a_kernel): what the filter of frame k waits for
    // Tracker -> Updater hand-over tables in rotation: book-keeping(k) rewrites table k % kHand once filter(k - kHand) has read it.  Two
    // tables (rounds 1-2) let the side chain run at most two frames ahead of the filter; four were built to absorb the frames in which the
    // side chain is late (one in ten: the gate in front of the filter then waits ~100 us).  Measured: they do not — the side chain's own
    // period equals the filter's (134 against 136 us in situ), so it never builds the lead; what is late in those frames is the IMAGE
    // chain (greedy_kernel: 29-200 us, data dependent), which the refill half of book-keeping waits for.  Kept: it costs 30 KB per instance.
    static const int kHand = 4;
    hipEvent_t evF[kHand] = {nullptr, nullptr, nullptr, nullptr}, evIn[2] = {nullptr, nullptr};
    long frame_no = 0;
    bool piped = false, in_frame = false;
    struct TrackOut { int* n_feat; unsigned char* types; int* len; float* meas; } tout[kHand];
    std::string err;
    // filter state (double-buffered)
    FilterMeta* meta = nullptr;
    double* x[2] = {nullptr, nullptr};
    double* P[2] = {nullptr, nullptr};
    int cur = 0;
    int img_count = 0;      // host mirror of nImageCountAfterInit (data-independent)
    int n_clones_host = 0;  // host mirror of nCloneStates (data-independent)
    // update scratch
    double *partial = nullptr, *block = nullptr, *Ab = nullptr, *Tbuf = nullptr, *W = nullptr, *Mg = nullptr, *U = nullptr, *G = nullptr,
           *Pt1 = nullptr, *tm_global = nullptr, *gamma = nullptr, *pfinv = nullptr;
    int *nrows = nullptr, *acc = nullptr, *ndof = nullptr, *gram_cnt = nullptr;
    double* gpose = nullptr;   // batch handles (max_len <= 16): the pose chains geom4_kernel leaves for feat_build_kernel<4>, [Fu][(max_len-1) x 24]
    int* gvalid = nullptr;     // ... and the validity flag of each triangulation
    size_t trunc_lds = 0, gram_batch_lds = 0;   // gram_batch_lds != 0: batch handle whose [A|b] fits in LDS (gram_reduce_batch_kernel)
    // round 6 (literal.h): the rows of an update of <= LIT_FEATS features, exported by the per-feature kernel for the reference's literal sweep; the
    // state of the systolic array when it does not fit in the reduction's LDS (long windows, batch handles); nullptr: no literal path on this handle
    double *lit_rows = nullptr, *lit_state = nullptr;
    bool lit_state_global = false;
    size_t lit_batch_lds = 0;                   // dynamic LDS of lit_batch_kernel
    const double* last_Ab = nullptr;            // the [A|b] block of the last update (its meta row: frame_info)
    int feat_threads = 64;
    size_t feat_lds = 0, fprop_lds = 0, ug_lds = 0, book_lds = 0, jb_lds = 0;
    int book_waves = 4;
    int solve5_variant = 0;      // solve6_kernel (the LDS-tableau solve behind gemm_T_kernel: batch handles): 0 none, 1: <1,8,8>  2: <2,12,8>  3: <2,16,8>
    StageSync* stage_sync = nullptr;   // device-side completion counter of the filter chain (aug) and the value it reaches after the launches so far
    StageSync stage_tgt = {};
    const unsigned long long* klt_wait = nullptr; unsigned long long klt_target = 0;
    unsigned long long* pyr_signal = nullptr;   // pending: the next detector launch on the image chain's queue bumps it   // this frame's klt_kernel3 polls the image chain's pyramid counter
    int solve7_variant = 0;      // register-tableau solve with the T prologue (solve7.hip): 1: 6n <= 64, 2: <= 96, 3: <= 128, 4: <= 192
    int solve9_nt = 0;           // solve9_kernel (solve9.hip): tiles per side of the padded clone block (4, 6, 8, 12), 0: not used (batch handles, RVIO_SOLVE7=1)
    double* S9scr = nullptr;     // its slab of tiles in L2: 5 NT^2 x 256 doubles (+ the verdict of the Cholesky role)
    bool chol_ready = false;     // the slab holds L, G of the clone block the next solve will see (written by the role workgroup of the per-feature / propagate launch)
    float* eig_map = nullptr;    // W x H min-eigenvalue map of rvio_hip_get_corners(eig): allocated on first use
    size_t solve5_lds = 0, cholt_lds = 0;
    // staging
    rvio_imu* d_imu = nullptr;
    double* gathered = nullptr;   // rvio_hip_frame_sharded_dev: world x [S2 | S1] as the all-gather delivers them (allocated on first use)
    int gathered_world = 0;
    int imu_cap = RVIO_MAX_IMU;   // samples the host-side staging (d_imu, hb_imu, pinned ring) holds; grows on demand (ensure_imu_capacity)
    float* d_cand = nullptr;
    uint8_t* d_img = nullptr;
    static const int kIC = 3;                     // image chains in flight in run-ahead mode (streams, detector scratch sets, CLAHE LUT sets)
    DetDev dets[kIC] = {};                        // device detector (T7), allocated on first use: kIC sets of scratch — in run-ahead mode the
                                                  // detectors of consecutive frames run on two streams, by frame parity
    int det_set_last = 0;                         // the set the last call used (rvio_hip_get_corners)
    bool det_ready = false, use_det = false;
    hipStream_t stream_l = nullptr;               // long windows (6n > 96): = stream_c (one image chain in flight); the Cholesky factor of the clone block runs here, beside propagate / the per-feature stage of the frame it serves
    hipEvent_t evA = nullptr, evL = nullptr;      // augment/compose done (filter stream) -> stream_l;  factor in the slab (stream_l) -> the solve
    bool dx_pending = false;                      // split solve: dx = Pc y and the state injection ride in the Joseph stage's first launch (launch_ug_final)
    bool chol_async = false;                      // a factor of the CURRENT clone block is in flight on (or has left) stream_l
    hipStream_t stream_d = nullptr;               // side stream of the front end: forks from / joins the tracker stream (see build_pyramid_dev)
    hipStream_t stream_c = nullptr;               // CLAHE stream of the run-ahead mode (frame k+1 is equalised while frame k is still being detected)
    hipEvent_t evC[kIC] = {nullptr, nullptr, nullptr};   // equalised image + pyramid of the frame ready, by image chain
    hipStream_t stream_e = nullptr;               // third image-chain stream (the chain is ~200 us long in situ: two in flight made it a co-bottleneck of the 130 us period)
    int n_ic = 2;                                 // image chains in flight (<= kIC)
    int ic = 0;                                   // image chain (stream / detector scratch / LUT set) of the call in progress: frame_no % kIC in run-ahead mode, else the parity
    hipStream_t side = nullptr;                   // stream of pyramid / KLT / RANSAC of the call in progress (stream_d beside the detector, else ts)
    hipEvent_t evD0 = nullptr, evD1 = nullptr;
    uint8_t* hb_img[2] = {nullptr, nullptr};      // staging of rvio_hip_frame (host buffers), by frame parity
    rvio_imu* hb_imu[kHand + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // (run-ahead mode rotates kHand + 1 slots: see rvio_hip_frame)
    hipEvent_t book_wait = nullptr;   // run-ahead: the event book-keeping of the frame in flight has to wait for (filter k-2)
    // how "the filter of the frame with parity b has finished" is known: 0 = evF[b] was recorded behind it, 1 = the device-side counter
    // stage_sync->aug reaches fin_target[b] (single instance, run-ahead mode: no marker packet on the filter stream)
    int fin_mode[kHand] = {0, 0, 0, 0};
    unsigned long long fin_target[kHand] = {0, 0, 0, 0}, book_target = 0;
    bool book_dev = false;
    bool handover_evt = false;        // frame in flight: evH was recorded behind the hand-over half of book-keeping
    bool last_ra = false;             // the previous rvio_hip_frame call ran in run-ahead mode
    float* hb_cand[2] = {nullptr, nullptr};
    // pinned host ring of rvio_hip_frame: the caller's (pageable) buffers are packed into it on the host, the H2D copies then run
    // asynchronously from pinned memory.  Slot s may be refilled once evPin[s] (recorded behind its copies) has completed.
    static const int kPin = 3;
    uint8_t* pin[kPin] = {nullptr, nullptr, nullptr};
    hipEvent_t evPin[kPin] = {nullptr, nullptr, nullptr}, evPin2[kPin] = {nullptr, nullptr, nullptr};
    size_t pin_img = 0, pin_imu = 0, pin_bytes = 0;
    uint8_t *d_eq = nullptr, *d_lut2[kIC] = {nullptr, nullptr, nullptr};   // CLAHE output image and tile LUTs (enable_equalizer), the LUTs by image chain
    // Buffers the front end of frame k+1 would otherwise overwrite while book-keeping of frame k still reads them (run-ahead of the
    // image chain on the pipelined path, see track_dev_impl): equalised image, detector corner list and its count, by frame parity
    uint8_t* d_eq2[4] = {nullptr, nullptr, nullptr, nullptr};   // four, in rotation: the equalised image IS level 0 of its pyramid, which the KLT of the NEXT frame still reads
    int eq_slot = 0;
    float* det_xy2[3] = {nullptr, nullptr, nullptr};   // corner lists: by parity, in run-ahead mode three in rotation (dslot)
    int dslot = 0;
    int* det_nout = nullptr;
    int par = 0;                                  // parity of the call in progress / of the last call (getters)
    hipStream_t tail = nullptr;                   // stream that ran book-keeping in the call in progress (the hand-over event is recorded there)
    bool runahead = false;                        // call in progress: pipelined whole-frame path with the device detector
    bool private_queues = false;                  // this handle's streams own hardware queues (make_stream)
    bool queues_shared = false;                   // ... unless one of them had to come from the shared pool after all
    bool extra_queues = false;                    // a collective's queues run beside this handle's (rvio_hip_frame_sharded_dev with a communicator)
    bool dev_sync = false;                        // ... of ONE instance: hand-over -> filter and corners -> refill go through device-side counters (StageSync)
    bool gate_pending = false;                    // the filter of the frame in flight starts behind stage_gate_kernel (target: gate_target)
    unsigned long long gate_target = 0;
    int cl_tx = 0, cl_ty = 0, cl_tw = 0, cl_th = 0, cl_clip = 0;
    float cl_scale = 0.f;
    float* d_in_xy = nullptr;
    unsigned char* d_in_st = nullptr;
    // tracker
    TrackerDev t;
    PyrDev pyr[4];   // four in rotation: pyramid(k) (image stream, run-ahead) may be built while KLT(k-2), KLT(k-1) still match the others
    int pyr_cur = 0;
    std::vector<void*> allocs;
    // filter slab: the filter state, the update scratch and the Tracker -> Updater hand-over of ONE instance are carved from one
    // slab; a batch handle (rvio_hip_create_batch) owns `batch` slabs back to back and launches every filter kernel with
    // gridDim.z = batch (rvio_dev.h zoff)
    char* slab = nullptr;
    size_t slab_off = 0, slab_bytes = 0;
    bool slab_mode = false;
    int batch = 1;
    const rvio_imu* fuse_imu = nullptr;   // whole-frame path: propagate of this frame rides in the per-feature launch (feat_prop_kernel)
    const rvio_imu* time_imu = nullptr; int time_m = 0;   // the IMU batch of the last fused frame (rvio_hip_debug_time_kernel(8) only: the caller's buffer)
    int fuse_m = -1;                      // >= 0 while such a propagate is pending
    bool fuse_ok = false;
    bool one_stream = false;
    bool wide_px = false;            // throughput forms of the image kernels (several pixels per thread): batch handles of >= 8 instances
    bool front_end = true;           // a batch handle may carry the filter only
    bool det_in_slab = false;        // batch handle with front end: the detector's buffers are slab members too
    size_t img_bs = 0, imu_bs = 0;   // instance strides (bytes) of the image / IMU batch of the call in progress
    BatchIn bin = {0, 0, 0, 0, 0};   // strides of the hand-over read by feat_build (slab_bytes for the handle's own buffers)
    int* rng = nullptr;
    int* first_mirror = nullptr;     // pinned, device-mapped: mbIsTheFirstImage of every instance as book-keeping leaves it
    bool first_cleared = false;      // (cached) every instance has seen its first image: nms(k+1) need not wait for book-keeping(k)
    int* cand_scratch = nullptr;
    rvio_frame_info* d_info = nullptr;
    double* d_pose = nullptr;
};

// ---------------------------------------------------------------- environment surface of the SHIPPING library: two variables.
//   RVIO_PARANOID     every cross-queue hand-off in its most conservative form (A/B against the default, and the thing to set when a
//                     runtime / firmware is suspected).  "1" = all of it; a larger value is a bit mask for bisection:
//                       2  events with the default flags (system-scope release at every record) instead of hipEventDisableSystemFence
//                       4  no device-side polls: every hand-off is a stream-level event (no StageSync counters, no gate / signal kernels)
//                       8  plain hipStreamCreateWithFlags(hipStreamNonBlocking) streams instead of CU-mask streams with private queues
//                      16  rvio_hip_frame waits on the host for its H2D staging copies before it enqueues the frame
//                      32  every whole-frame call drains all streams of the handle before it returns
//   RVIO_NO_RUNAHEAD  the pipelined path without the run-ahead image chains (book-keeping back on the tracker stream)
// Every other RVIO_* switch (kernel forms, stream layouts, unsafe timing experiments) exists in the instrumented build only
// (-DRVIO_DBG_CLOCKS, tools/chain_clocks.py): a stray environment variable cannot change what the shipping pipeline launches.
enum { PAR_SYSFENCE = 2, PAR_NO_DEVPOLL = 4, PAR_PLAIN_STREAMS = 8, PAR_SYNC_COPIES = 16, PAR_DRAIN = 32, PAR_ALL = 62 };
static int paranoid_bits() {
    static const int v = [] {
        const char* e = getenv("RVIO_PARANOID");
        if (!e || !*e || !std::strcmp(e, "0")) return 0;
        const int b = atoi(e);
        return b > 1 ? (b & PAR_ALL) : (int)PAR_ALL;
    }();
    return v;
}
#ifdef RVIO_DBG_CLOCKS
static const char* ab_env(const char* name) { return getenv(name); }
#else
static const char* ab_env(const char*) { return nullptr; }
#endif
// The handle's events only order kernels of ONE device across its streams; the host only ever WAITS for them (hipEventSynchronize on the
// pinned ring's events, hipStreamSynchronize elsewhere): no timing, and no system-scope fence when they are recorded — that fence writes the
// dirty L2 lines of the recording queue back before the NEXT kernel of that queue may start (measured: a 35 us hole in the tracker stream per frame)
static unsigned ev_flags() { return (paranoid_bits() & PAR_SYSFENCE) || ab_env("RVIO_EVENT_SYSFENCE") ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence); }
#define kEvFlags ev_flags()
// Timing experiments that DROP correctness-critical stream waits exist only in an instrumented build (-DRVIO_DBG_CLOCKS, tools/chain_clocks.py):
// a stray environment variable must not be able to turn the shipping pipeline racy.
#ifdef RVIO_DBG_CLOCKS
static const int kDbgSkip = getenv("RVIO_DBG_SKIP") ? atoi(getenv("RVIO_DBG_SKIP")) : 0;
static const int kRaDepth = getenv("RVIO_RA_DEPTH") ? std::max(1, std::min(3, atoi(getenv("RVIO_RA_DEPTH")))) : 3;   // 2 = the image chain waits for book-keeping(k-2)
#else
static constexpr int kDbgSkip = 0;
static constexpr int kRaDepth = 3;
#endif
#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return RVIO_ERR_NO_DEVICE;                                                           \
        }                                                                                        \
    } while (0)

template <typename T>
static int dalloc(rvio_hip* h, T** p, size_t n) {
    void* q = nullptr;
    size_t bytes = n * sizeof(T);
    if (bytes == 0) bytes = 16;
    if (h->slab_mode) {   // bump allocation inside the filter slab (first pass, slab == nullptr: sizes only)
        *p = h->slab ? (T*)(h->slab + h->slab_off) : nullptr;
        h->slab_off += (bytes + 255) & ~(size_t)255;
        return RVIO_OK;
    }
    HIPCHK(h, hipMalloc(&q, bytes));
    HIPCHK(h, hipMemsetAsync(q, 0, bytes, h->stream));
    h->allocs.push_back(q);
    *p = (T*)q;
    return RVIO_OK;
}
// entry points that address ONE instance's front end (a batch handle is driven by rvio_hip_frame_batch_dev / _frame_tracks_dev)
#define FRONT_END_ONLY(h)                                                                                              \
    do {                                                                                                               \
        if ((h)->batch > 1) { (h)->err = "single-instance entry point called on a batch handle"; return RVIO_ERR_UNSUPPORTED; } \
    } while (0)
#define SYNC_FRONT(h)                                                                     \
    do {                                                                                  \
        if ((h)->stream_c) HIPCHK(h, hipStreamSynchronize((h)->stream_c));                \
        if ((h)->stream_e) HIPCHK(h, hipStreamSynchronize((h)->stream_e));                \
        if ((h)->stream_d) HIPCHK(h, hipStreamSynchronize((h)->stream_d));                \
        HIPCHK(h, hipStreamSynchronize((h)->stream_t));                                   \
    } while (0)
#define DALLOC(h, p, n)                               \
    do {                                              \
        int rc_ = dalloc((h), &(p), (n));             \
        if (rc_ != RVIO_OK) return rc_;               \
    } while (0)

extern "C" {

int rvio_hip_abi_version(void) { return RVIO_HIP_ABI_VERSION; }

// config/rvio_euroc.yaml:8-111
void rvio_config_euroc(rvio_config* c) {
    std::memset(c, 0, sizeof *c);
    c->imu_rate = 200;
    c->sigma_g = 1.6968e-04; c->sigma_wg = 1.9393e-05; c->sigma_a = 2.0e-3; c->sigma_wa = 3.0e-3;
    c->gravity = 9.8082; c->small_angle = 0.001745329;
    c->width = 752; c->height = 480;
    c->fx = 458.654f; c->fy = 457.296f; c->cx = 367.215f; c->cy = 248.375f;
    c->k1 = -0.28340811f; c->k2 = 0.07395907f; c->p1 = 0.00019359f; c->p2 = 1.76187114e-05f; c->k3 = 0.f;
    c->sigma_px = 0.002180293f; c->sigma_py = 0.002186767f;
    const double T[16] = {0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975,
                          0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768,
                          -0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949,
                          0.0, 0.0, 0.0, 1.0};
    std::memcpy(c->T_bc, T, sizeof T);
    c->fisheye = 0;
    c->n_features = 200; c->max_track_len = 15; c->min_track_len = 3;
    c->min_dist = 15; c->qual_lvl = 0.01f; c->block_x = 150; c->block_y = 120;
    c->enable_equalizer = 1; c->use_sampson = 1; c->inlier_thr = 1e-5;
    c->ini_thr_angle = 0.005; c->ini_thr_displ = 0.01; c->ini_enable_alignment = 1;
}

static void fill_devcfg(const rvio_config* c, DevCfg* d) {
    std::memset(d, 0, sizeof *d);
    d->gravity = c->gravity; d->small_angle = c->small_angle;
    d->sg2 = c->sigma_g * c->sigma_g; d->swg2 = c->sigma_wg * c->sigma_wg;
    d->sa2 = c->sigma_a * c->sigma_a; d->swa2 = c->sigma_wa * c->sigma_wa;
    d->sigma_im = (double)std::max(c->sigma_px, c->sigma_py);   // float max, widened (Updater.cc:42-44)
    d->inlier_thr = c->inlier_thr;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { d->Ric[3 * i + j] = c->T_bc[4 * i + j]; d->Rci[3 * j + i] = c->T_bc[4 * i + j]; }
        d->tic[i] = c->T_bc[4 * i + 3];
    }
    for (int i = 0; i < 3; ++i) d->tci[i] = -(d->Rci[3 * i] * d->tic[0] + d->Rci[3 * i + 1] * d->tic[1] + d->Rci[3 * i + 2] * d->tic[2]);
    d->fx = c->fx; d->fy = c->fy; d->cx = c->cx; d->cy = c->cy;
    d->k1 = c->k1; d->k2 = c->k2; d->p1 = c->p1; d->p2 = c->p2; d->k3 = c->k3;
    d->W = c->width; d->H = c->height;
    d->F = c->n_features; d->Fu = (int)std::ceil(.5 * c->n_features);
    d->max_len = c->max_track_len; d->min_len = c->min_track_len;
    d->nmax = c->max_track_len - 1; d->dmax = 24 + 6 * d->nmax; d->xdmax = 26 + 7 * d->nmax;
    d->rho_max = 2 * c->max_track_len - 2;
    d->ldh = 6 * d->nmax + 1;
    // FeatureDetector ctor, FeatureDetector.cc:29-52
    d->min_dist = c->min_dist;
    d->block_x = c->block_x; d->block_y = c->block_y;
    // (mnGridCols/Rows, mnOffsetX/Y and mnMaxFeatsPerBlock are `int` members upstream: the assignments truncate, FeatureDetector.h:66-77)
    d->grid_cols = (int)std::floor(c->width / c->block_x);
    d->grid_rows = (int)std::floor(c->height / c->block_y);
    d->off_x = (float)(int)(.5 * (c->width - d->grid_cols * c->block_x));
    d->off_y = (float)(int)(.5 * (c->height - d->grid_rows * c->block_y));
    d->max_per_block = (d->grid_cols * d->grid_rows > 0) ? (float)(int)((float)c->n_features / (d->grid_cols * d->grid_rows)) : 0.f;
    d->use_sampson = c->use_sampson;
    d->fisheye = c->fisheye ? 1 : 0;
    // buildOpticalFlowPyramid: stop when a level is not larger than the window
    int w = c->width, hgt = c->height, lv = 1;
    for (int l = 1; l <= 3; ++l) { w = (w + 1) / 2; hgt = (hgt + 1) / 2; if (w <= 15 || hgt <= 15) break; lv++; }
    d->levels = lv;
}

// The per-instance filter buffers (state, update scratch, Tracker -> Updater hand-over, IMU staging): called twice — sizes, then pointers
static int alloc_filter_slab(rvio_hip* h, bool need_tm_global) {
    const DevCfg& d = h->dc;
    const size_t dm = d.dmax, PP = dm * dm, ldh = d.ldh;
    TrackerDev& t = h->t;
    DALLOC(h, h->meta, 1);
    for (int b = 0; b < 2; ++b) { DALLOC(h, h->x[b], (size_t)d.xdmax + 8); DALLOC(h, h->P[b], PP); }
    DALLOC(h, h->partial, (size_t)d.Fu * ldh * ldh);   // per-feature shares G_f = Hn^T [Hn | r] of the information block
    DALLOC(h, h->block, 2 * ldh * ldh);   // [S2 | S1]: the type-'2' and type-'1' sums of the information block (gram_reduce_kernel)
    DALLOC(h, h->Ab, 2 * ldh * ldh);
    DALLOC(h, h->gram_cnt, 8);
    DALLOC(h, h->stage_sync, 1);
    DALLOC(h, h->Tbuf, ldh * ldh); DALLOC(h, h->W, ldh * ldh);
    DALLOC(h, h->U, dm * ldh); DALLOC(h, h->G, dm * ldh);
    DALLOC(h, h->Pt1, PP);
    DALLOC(h, h->gamma, d.Fu); DALLOC(h, h->pfinv, (size_t)3 * d.Fu);
    DALLOC(h, h->nrows, d.Fu); DALLOC(h, h->acc, d.Fu); DALLOC(h, h->ndof, d.Fu);
    DALLOC(h, h->d_imu, RVIO_MAX_IMU);
    DALLOC(h, h->d_info, 1); DALLOC(h, h->d_pose, 8);
    DALLOC(h, t.n_feat, 1); DALLOC(h, t.types, d.Fu); DALLOC(h, t.len, d.Fu); DALLOC(h, t.meas, (size_t)2 * d.Fu * d.max_len);
    if (need_tm_global) DALLOC(h, h->tm_global, (size_t)d.Fu * d.rho_max * ldh);
    static const bool no_lit = getenv("RVIO_NO_LITERAL") != nullptr;   // A/B: the structural rank rule alone, as up to round 5
    if (!no_lit && d.ldh <= 190 && d.nmax + 1 <= 40 && d.rho_max < 254 && lit_slab_doubles(d.ldh, d.rho_max) * sizeof(double) <= 144 * 1024) {
        DALLOC(h, h->lit_rows, lit_rows_doubles(d.ldh, d.rho_max));
        if (h->lit_state_global) DALLOC(h, h->lit_state, lit_state_doubles(d.ldh - 1));
    }
    static const bool no_geom4 = ab_env("RVIO_NO_GEOM4") != nullptr;   // A/B timing
    if (h->batch > 1 && d.max_len <= GEOM4_ML && !no_geom4) { DALLOC(h, h->gpose, (size_t)d.Fu * (d.max_len - 1) * 24); DALLOC(h, h->gvalid, d.Fu); }
    return RVIO_OK;
}

static int detector_alloc(rvio_hip* h);
static int detector_check(rvio_hip* h);
// The per-instance front-end buffers (staging, CLAHE, tracker tables, two pyramids; the detector's for a batch handle)
static int alloc_frontend_slab(rvio_hip* h) {
    const DevCfg& d = h->dc;
    TrackerDev& t = h->t;
    DALLOC(h, h->d_cand, (size_t)2 * d.F);
    DALLOC(h, h->d_img, (size_t)d.W * d.H);
    if (h->cfg.enable_equalizer) {
        DALLOC(h, h->d_eq2[0], (size_t)d.W * d.H); DALLOC(h, h->d_eq2[1], (size_t)d.W * d.H); DALLOC(h, h->d_eq2[2], (size_t)d.W * d.H);
        DALLOC(h, h->d_eq2[3], (size_t)d.W * d.H);
        h->d_eq = h->d_eq2[0];
        for (int k = 0; k < std::max(2, h->n_ic); ++k) DALLOC(h, h->d_lut2[k], (size_t)h->cl_tx * h->cl_ty * 256);
    }
    DALLOC(h, h->d_in_xy, (size_t)2 * d.F); DALLOC(h, h->d_in_st, d.F);
    DALLOC(h, h->rng, 40); DALLOC(h, h->cand_scratch, (size_t)2 * d.F + 8);
    DALLOC(h, t.first, 1); DALLOC(h, t.n_pts, 1);
    DALLOC(h, t.feats, (size_t)2 * d.F); DALLOC(h, t.un1, (size_t)2 * d.F); DALLOC(h, t.slot, d.F);
    DALLOC(h, t.hist, (size_t)2 * d.F * d.max_len); DALLOC(h, t.hist_len, d.F);
    DALLOC(h, t.tracked, (size_t)2 * d.F); DALLOC(h, t.un2, (size_t)2 * d.F); DALLOC(h, t.status, d.F);
    DALLOC(h, t.tmp_feats, (size_t)2 * d.F); DALLOC(h, t.tmp_un, (size_t)2 * d.F); DALLOC(h, t.tmp_slot, d.F);
    DALLOC(h, t.cand_acc, d.F); DALLOC(h, t.mid, 4);
    DALLOC(h, t.cell_pts, (size_t)d.grid_cols * d.grid_rows * 2 * d.F * 2);
    for (int k = 1; k < rvio_hip::kHand; ++k) {
        DALLOC(h, h->tout[k].n_feat, 1); DALLOC(h, h->tout[k].types, d.Fu); DALLOC(h, h->tout[k].len, d.Fu);
        DALLOC(h, h->tout[k].meas, (size_t)2 * d.Fu * d.max_len);
    }
    for (int b = 0; b < 4; ++b) {
        int w = d.W, hg = d.H;
        for (int l = 0; l < 4; ++l) {
            uint8_t* im = nullptr; short* dx = nullptr;   // (no derivative images: the KLT kernel forms them from its staged patch)
            if (l < d.levels) DALLOC(h, im, (size_t)w * hg);
            h->pyr[b].img[l] = im; h->pyr[b].dxy[l] = dx; h->pyr[b].w[l] = w; h->pyr[b].h[l] = hg;
            w = (w + 1) / 2; hg = (hg + 1) / 2;
        }
    }
    if (h->det_in_slab) return detector_alloc(h);
    return RVIO_OK;
}

// The four streams of a handle each need a hardware queue of their own: the run-ahead pipeline is four concurrent chains, and two of them
// on one queue serialise (measured: 4.5 k instead of 6.6 k frames/s).  hipStreamCreate deals streams onto a pool of GPU_MAX_HW_QUEUES (4)
// shared queues by reference count, so whether a handle gets four distinct ones depends on every stream the process created before it
// (torch's, another library's).  A stream created with a CU mask owns a private queue; the mask here enables every CU.
// ... for ONE handle: beyond four busy queues the command processor time-slices (eight handles with four private queues each ran at a quarter
// of the rate of eight handles on the shared pool), so only the first live handle of a process takes private queues; the others share the pool
// as before (many streams per GPU are what batch handles are for).
static std::atomic<int> g_private_queue_handles{0};
static hipError_t make_stream(rvio_hip* h, hipStream_t* s, bool front_end = false) {
    static const int mode = (paranoid_bits() & PAR_PLAIN_STREAMS) ? 0 : (ab_env("RVIO_STREAM_MODE") ? atoi(ab_env("RVIO_STREAM_MODE")) : 1);
    if (mode == 0 || !h->private_queues) { h->queues_shared = true; return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, h->device);
    if (e != hipSuccess) return e;
    const int words = (prop.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> mask((size_t)std::max(words, 1), 0xffffffffu);
    if (prop.multiProcessorCount % 32) mask.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
    // experiment: the front-end streams leave every `fe_skip`-th CU to the filter stream (RVIO_FE_SKIP=4: three quarters of the chip)
    static const int fe_skip = ab_env("RVIO_FE_SKIP") ? atoi(ab_env("RVIO_FE_SKIP")) : 0;
    if (front_end && fe_skip > 1) for (int c = 0; c < prop.multiProcessorCount; ++c) if (c % fe_skip == 0) mask[c / 32] &= ~(1u << (c % 32));
    e = hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { (void)hipGetLastError(); h->queues_shared = true; return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
    return e;
}

static bool profiler_serialises() {
    const char* e = getenv("ROCPROF_COUNTER_COLLECTION");
    return e && *e && std::strcmp(e, "0") != 0 && std::strcmp(e, "False") != 0 && std::strcmp(e, "false") != 0;
}

// The dynamic-LDS limit of a kernel is a property of the PROCESS, not of a handle: a later handle with a smaller need (a shorter window, a
// smaller cornerSubPix half-window) must not lower it under an earlier live handle's launches — the limit only ever goes up.
static hipError_t lds_attr(const void* fn, int bytes) {
    static std::mutex mu;
    static std::map<const void*, int> limit;
    std::lock_guard<std::mutex> lk(mu);
    int& cur = limit[fn];
    if (bytes <= cur) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}
static int create_impl(const rvio_config* cfg, int device, int batch, bool front_end, rvio_hip** out) {
    if (!cfg || !out) return RVIO_ERR_INVALID;
    *out = nullptr;
    if (cfg->max_track_len < 3 || cfg->max_track_len > RVIO_MAX_LEN || cfg->n_features < 2 || cfg->min_track_len < 2) return RVIO_ERR_INVALID;
    if (batch < 1) return RVIO_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0) return RVIO_ERR_NO_DEVICE;
    rvio_hip* h = new rvio_hip();
    h->cfg = *cfg; h->device = device; h->batch = batch;
    h->front_end = front_end; h->det_in_slab = front_end && batch > 1;
    h->wide_px = batch >= 8;
    if (const char* e = ab_env("RVIO_WIDE_PX")) h->wide_px = atoi(e) != 0;   // A/B timing and tests (the two forms must agree bit for bit)
    fill_devcfg(cfg, &h->dc);
    const DevCfg& d = h->dc;
    if (d.grid_cols * d.grid_rows < 1) { delete h; return RVIO_ERR_INVALID; }
    *out = h;   // returned even on allocation failure so last_error is readable
    HIPCHK(h, hipSetDevice(device));
    h->private_queues = g_private_queue_handles.fetch_add(1) == 0;
    if (!h->private_queues) g_private_queue_handles.fetch_sub(1);
    HIPCHK(h, make_stream(h, &h->stream));
    h->one_stream = ab_env("RVIO_ONE_STREAM") != nullptr;   // profiling only: every kernel on the filter stream (clean per-kernel times)
    if (h->one_stream) h->stream_t = h->stream_d = h->stream_c = h->stream_e = h->stream;
    else {
        HIPCHK(h, make_stream(h, &h->stream_t, true));
        HIPCHK(h, make_stream(h, &h->stream_d, true));
        HIPCHK(h, make_stream(h, &h->stream_c, true));
        // Image chains in flight.  Two (default): with the filter, tracker and side streams that makes FOUR busy queues.  A third chain on a
        // fifth queue was measured (RVIO_IC=3): the frame period goes from 131 to 180-250 us whatever CUs the front end is kept off — beyond
        // four busy queues the command processor time-slices them.
        // Long windows (96 < 6n <= 192: the solve in its split form, filter chain >= 250 us): ONE image chain in flight is enough (the chain is ~180 us),
        // and the queue that frees runs the Cholesky factor of the clone block beside the filter chain (augment_compose_dev).  A FIFTH queue for it was
        // measured: cfg C 3.2 k frames/s instead of 3.9 k — the command processor time-slices beyond four busy queues.
        const int c6m_ = 6 * (h->cfg.max_track_len - 1);
        if (batch == 1 && c6m_ > 96 && c6m_ <= 192) h->n_ic = 1;
        if (const char* e = ab_env("RVIO_IC")) h->n_ic = std::max(1, std::min((int)rvio_hip::kIC, atoi(e)));
        if (h->n_ic > 2) HIPCHK(h, make_stream(h, &h->stream_e, true));
        if (batch == 1 && c6m_ > 96 && c6m_ <= 192 && h->n_ic == 1) {
            h->stream_l = h->stream_c;
            HIPCHK(h, hipEventCreateWithFlags(&h->evA, kEvFlags));
            HIPCHK(h, hipEventCreateWithFlags(&h->evL, kEvFlags));
        }
    }
    HIPCHK(h, hipEventCreateWithFlags(&h->evD0, kEvFlags));
    HIPCHK(h, hipEventCreateWithFlags(&h->evD1, kEvFlags));
    for (int b = 0; b < rvio_hip::kIC; ++b) HIPCHK(h, hipEventCreateWithFlags(&h->evC[b], kEvFlags));
    h->ts = h->stream;
    for (int b = 0; b < 2; ++b) {
        HIPCHK(h, hipEventCreateWithFlags(&h->evT[b], kEvFlags));
        HIPCHK(h, hipEventCreateWithFlags(&h->evT[b + 2], kEvFlags));
        HIPCHK(h, hipEventCreateWithFlags(&h->evH[b], kEvFlags));
        HIPCHK(h, hipEventCreateWithFlags(&h->evH[b + 2], kEvFlags));
        HIPCHK(h, hipEventCreateWithFlags(&h->evF[b], kEvFlags));
        HIPCHK(h, hipEventCreateWithFlags(&h->evF[b + 2], kEvFlags));
        HIPCHK(h, hipEventCreateWithFlags(&h->evIn[b], kEvFlags));
    }
    const size_t ldh = d.ldh;
    // launch geometry (decides two optional slab members)
    if (d.Fu > GRAM_MAX_FEATS) { h->err = "Tracker.nFeatures too large for the Gram stage (ceil(F/2) <= 2048)"; return RVIO_ERR_UNSUPPORTED; }
#ifndef FEAT_T_SMALL
#define FEAT_T_SMALL 128     // threads of a per-feature workgroup at 6n <= 127 (same-box A/B of 64 against 128 at B = 2048: profiles/r06_feat_threads_ab.txt)
#endif
    h->feat_threads = (d.ldh <= 128) ? FEAT_T_SMALL : 256;
    if (const char* ft = ab_env("RVIO_FEAT_THREADS")) h->feat_threads = atoi(ft);   // A/B timing only (64, 128 or 256)
    h->trunc_lds = trunc_lds_doubles(d.max_len) * sizeof(double);
    {   // the literal sweep runs in the workgroup that finishes the reduction (literal.h): its ring / rotation tables always in that launch's LDS, the
        // array's state too when it fits (gram_reduce_kernel holds ~11 KB of static LDS) — else, and for batch handles (occupancy), in the slab
        const size_t aux = lit_aux_doubles(d.ldh, d.rho_max) * sizeof(double), st = lit_state_doubles(d.ldh - 1) * sizeof(double);
        const size_t slab = lit_slab_doubles(d.ldh, d.rho_max) * sizeof(double);   // (a feature's raw block for the nullspace sweep: at least one must fit)
        h->lit_state_global = batch > 1 || aux + st > 144 * 1024;
        h->lit_batch_lds = std::max(aux, slab);
        h->trunc_lds = std::max(h->trunc_lds, std::max(h->lit_state_global ? aux : aux + st, std::min((size_t)4, (size_t)(144 * 1024) / slab) * slab));
    }
    h->feat_lds = feat_lds_doubles(d.max_len, d.ldh, true) * sizeof(double);
    bool need_tm_global = false;
    // (a batch handle keeps T in global memory as well: a third less LDS per feature workgroup = 8 instead of 5 resident per CU)
    if (h->feat_lds > 150 * 1024 || batch > 1) {
        h->feat_lds = feat_lds_doubles(d.max_len, d.ldh, false) * sizeof(double);
        need_tm_global = true;
    }
    if (h->feat_lds > 160 * 1024) { h->err = "per-feature LDS footprint exceeds 160 KiB"; return RVIO_ERR_UNSUPPORTED; }
    const size_t c6m = ldh - 1;
    if (front_end && cfg->enable_equalizer) {   // CLAHE(3.0, 5x5), Tracker.cc:198-202
        h->cl_tx = 5; h->cl_ty = 5;
        int ew = d.W, eh = d.H;
        if (d.W % h->cl_tx != 0 || d.H % h->cl_ty != 0) { ew = d.W + (h->cl_tx - d.W % h->cl_tx); eh = d.H + (h->cl_ty - d.H % h->cl_ty); }
        h->cl_tw = ew / h->cl_tx; h->cl_th = eh / h->cl_ty;
        const int area = h->cl_tw * h->cl_th;
        h->cl_clip = std::max((int)(3.0 * area / 256), 1);
        h->cl_scale = 255.0f / (float)area;
    }
    if (h->det_in_slab) { int rc = detector_check(h); if (rc != RVIO_OK) return rc; }
    // instance slab(s)
    h->slab_mode = true; h->slab = nullptr; h->slab_off = 0;
    { int rc = alloc_filter_slab(h, need_tm_global); if (rc != RVIO_OK) return rc; }
    if (front_end) { int rc = alloc_frontend_slab(h); if (rc != RVIO_OK) return rc; }
    h->slab_bytes = h->slab_off;
    {
        void* q = nullptr;
        HIPCHK(h, hipMalloc(&q, h->slab_bytes * (size_t)batch));
        HIPCHK(h, hipMemsetAsync(q, 0, h->slab_bytes * (size_t)batch, h->stream));
        h->allocs.push_back(q);
        h->slab = (char*)q; h->slab_off = 0;
    }
    { int rc = alloc_filter_slab(h, need_tm_global); if (rc != RVIO_OK) return rc; }
    if (front_end) { int rc = alloc_frontend_slab(h); if (rc != RVIO_OK) return rc; }
    h->slab_mode = false;
    h->bin = {0, h->slab_bytes, h->slab_bytes, h->slab_bytes, h->slab_bytes};
    TrackerDev& t = h->t;
    t.info = h->d_info;
    t.first_host = nullptr;
    if (front_end) {
        HIPCHK(h, hipHostMalloc((void**)&h->first_mirror, sizeof(int) * (size_t)batch, hipHostMallocMapped));
        for (int i = 0; i < batch; ++i) h->first_mirror[i] = 1;
        void* dp_ = nullptr;
        HIPCHK(h, hipHostGetDevicePointer(&dp_, h->first_mirror, 0));
        t.first_host = (int*)dp_;
    }
    h->tout[0] = {t.n_feat, t.types, t.len, t.meas};   // Tracker -> Updater hand-over, double-buffered for the pipelined path
    if (front_end) {   // mbIsTheFirstImage = true in every instance
        std::vector<int> ones((size_t)batch, 1);
        HIPCHK(h, hipMemcpy2DAsync(t.first, h->slab_bytes, ones.data(), sizeof(int), sizeof(int), (size_t)batch, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    HIPCHK(h, lds_attr((const void*)feat_build_kernel<16>, (int)h->feat_lds));
    HIPCHK(h, lds_attr((const void*)feat_build_kernel<4>, (int)h->feat_lds));
    HIPCHK(h, lds_attr((const void*)gram_reduce_kernel, (int)h->trunc_lds));
    HIPCHK(h, lds_attr((const void*)block_sum_kernel, (int)h->trunc_lds));
    HIPCHK(h, lds_attr((const void*)lit_batch_kernel, (int)h->lit_batch_lds));
    if (batch > 1 && gram_batch_lds_doubles(d.max_len, d.ldh) * sizeof(double) <= 64 * 1024) {
        h->gram_batch_lds = gram_batch_lds_doubles(d.max_len, d.ldh) * sizeof(double);
        HIPCHK(h, lds_attr((const void*)gemm_T_lds_kernel, 2 * 64 * 65 * (int)sizeof(double)));
        HIPCHK(h, lds_attr((const void*)gram_reduce_batch_kernel<4>, (int)h->gram_batch_lds));
        HIPCHK(h, lds_attr((const void*)gram_reduce_batch_kernel<6>, (int)h->gram_batch_lds));
    }
    // propagate rides in the per-feature launch: its workgroup builds no feature, so its buffers (Prop3Lds<16>, 86 KB) and the per-feature footprint
    // share the launch's dynamic LDS — max of the two, which fits one CU for every window (rounds 2-4: static + dynamic, the SUM: long windows
    // fell back to 8-sample chunks, cfg E to a propagate launch of its own on the chain)
    h->fprop_lds = h->feat_lds;
    // (round 5: + the Cholesky role of solve9 at 6n <= 96 — one more workgroup whose buffers live in the launch's dynamic LDS too)
    if (batch == 1 && c6m <= 96) h->fprop_lds = std::max(h->fprop_lds, c6m <= 64 ? sizeof(S9CholLds<4, 4>) : sizeof(S9CholLds<6, 4>));
    h->fprop_lds = std::max(h->fprop_lds, sizeof(Prop3Lds<16>));
    h->fuse_ok = batch == 1 && !ab_env("RVIO_NO_FUSED_PROPAGATE") && h->fprop_lds <= 160 * 1024;
    if (h->fuse_ok) HIPCHK(h, lds_attr((const void*)feat_prop_kernel, (int)h->fprop_lds));
    // the refill half of book-keeping walks the ChessGrid one wave per cell with a per-wave list of the cell's points (F float2 each): as many
    // waves as the 160 KB of LDS hold for one stream (16 at F <= 800: 20 cells -> two rounds instead of five), 4 for batch handles (occupancy)
    h->book_waves = 4;
    if (h->batch == 1) for (int nwv = 16; nwv > 4; nwv /= 2) if ((((size_t)20 * d.F + 7) & ~(size_t)7) + (size_t)nwv * d.F * 8 + 16 <= (size_t)150 * 1024) { h->book_waves = nwv; break; }
    h->book_lds = (((size_t)20 * d.F + 7) & ~(size_t)7) + (size_t)h->book_waves * d.F * 8 + 16;
    HIPCHK(h, lds_attr((const void*)bookkeep_b_kernel, (int)h->book_lds));
    HIPCHK(h, lds_attr((const void*)ransac_book_kernel, (int)h->book_lds));
    {
        {   // fully unrolled solve kernel: variants <column chunks, rows per wave> for c6 <= 126
            int rpw = 0, nch = 0, nw = 8;
            if (c6m <= 60) { h->solve5_variant = 1; nch = 1; rpw = 8; }
            else if (c6m <= 96) { h->solve5_variant = 2; nch = 2; rpw = 12; }
            else if (c6m <= 126) { h->solve5_variant = 3; nch = 2; rpw = 16; }
            h->solve7_variant = (c6m <= 64) ? 1 : (c6m <= 96) ? 2 : (c6m <= 128) ? 3 : (c6m <= 192) ? 4 : 0;
            if (ab_env("RVIO_SOLVE6") && h->solve5_variant) h->solve7_variant = 0;   // A/B timing: the LDS-tableau kernel behind gemm_T_kernel
            // one instance: the blocked SPD solve (solve9.hip).  Measured on full-load updates (tools/solve9_probe.py, profiles/r05_solve9_probe.txt), solve kernel alone:
            // 6n = 84: 94.0 us against solve7's 102.6; 120: 173 against 212; 180: 511 against 797.  At 6n <= 96 the Cholesky of the clone block — the part that does
            // not depend on the measurements — rides as one more workgroup in the per-feature launch (pipelined frame) or in propagate's launch (staged entry
            // points), off the chain; the solve kernel then starts at Q = A L.  RVIO_SOLVE7=1 (instrumented build) keeps the register-tableau elimination: A/B timing.
            if ((batch == 1 || (c6m <= 64 && S9_BATCH)) && c6m <= 192 && !ab_env("RVIO_SOLVE7")) {
                h->solve9_nt = (c6m <= 64) ? 4 : (c6m <= 96) ? 6 : (c6m <= 128) ? 8 : 12;
                DALLOC(h, h->S9scr, S9_SLAB_DOUBLES(h->solve9_nt) * (size_t)batch);
                if (h->solve9_nt == 4) HIPCHK(h, lds_attr((const void*)solve9_small_kernel, (int)sizeof(S9SmallLds)));
            }
            // batch handles: throughput, not latency — solve6 keeps four instances resident per CU (33 KB of LDS against 112 KB) and the
            // multi-workgroup gemm_T_kernel costs nothing there (measured at B = 2048: 2.67 ms per batched frame against 3.09)
            // (round 3, measured and NOT adopted: solve7 with T through the L2 scratch instead of LDS — 11 KB of LDS, eight workgroups per CU, no gemm_T
            // launch — as the batch form at 6n <= 64, RVIO_BATCH_SOLVE7: 2.62 ms per batched frame at B = 2048 against 2.29 with solve6 behind gemm_T)
            if (batch > 1 && h->solve5_variant && !ab_env("RVIO_SOLVE7") && !(h->solve7_variant == 1 && ab_env("RVIO_BATCH_SOLVE7"))) h->solve7_variant = 0;
            if (batch > 1 && h->solve7_variant == 1) h->solve7_variant = 5;
#ifdef RVIO_DBG_CLOCKS
            if (h->solve7_variant == 1)
            {
                HIPCHK(h, lds_attr((const void*)solve7_kernel<1, 16, 4>, (3 * 64 * 65 + 24 * 64) * (int)sizeof(double)));
                HIPCHK(h, lds_attr((const void*)solve7_kernel<1, 8, 8>, (3 * 64 * 65 + 24 * 64) * (int)sizeof(double)));
                HIPCHK(h, lds_attr((const void*)solve7_kernel<1, 4, 16>, (3 * 64 * 65 + 24 * 64) * (int)sizeof(double)));
            }
#else
            // shipping library: the register-tableau solve survives for batch handles beyond solve6's windows only (6n > 126: solve7_kernel<3, 16, 12>)
            if (h->solve7_variant != 4 || h->solve9_nt) h->solve7_variant = 0;
#endif
            if (h->solve5_variant) {
                h->solve5_lds = (size_t)(nw * rpw) * (64 * nch + 1) * sizeof(double);
                const int lds = (int)std::max(h->solve5_lds, (size_t)1024);
                HIPCHK(h, lds_attr((const void*)solve6_kernel<1, 8, 8>, lds));
                HIPCHK(h, lds_attr((const void*)solve6_kernel<2, 12, 8>, lds));
                HIPCHK(h, lds_attr((const void*)solve6_kernel<2, 16, 8>, lds));
            }
        }
        const size_t c6t = (c6m + 15) / 16;
        if (batch >= 128 && c6m <= 60 && !ab_env("RVIO_NO_JOSEPH_FUSED")) {   // the Joseph form of a batch handle in one kernel, one workgroup per instance
            const size_t ls = c6m + 1, dmx = 24 + c6m;
            h->jb_lds = (3 * dmx * ls + std::max((size_t)c6m * ls, (size_t)JB_TL_DOUBLES)) * sizeof(double);
            if (h->jb_lds > 160 * 1024) h->jb_lds = 0;
            else HIPCHK(h, lds_attr((const void*)joseph_batch_kernel, (int)h->jb_lds));
        }
        h->ug_lds = 2 * 16 * (c6t * 16 + 1) * sizeof(double);
        HIPCHK(h, lds_attr((const void*)ug_kernel, (int)h->ug_lds));
        HIPCHK(h, lds_attr((const void*)ug_lds_kernel, (int)(UGL_LDS_DOUBLES * sizeof(double))));
        HIPCHK(h, lds_attr((const void*)final_lds_kernel, (int)(FNL_LDS_DOUBLES * sizeof(double))));
        HIPCHK(h, lds_attr((const void*)joseph_lds_kernel, (int)(JL_LDS_DOUBLES * sizeof(double))));
    }
    if (batch > 1 && !h->solve5_variant && !h->solve7_variant) { h->err = "batched filter: clone window too long for the unrolled solve kernel (6n <= 126)"; return RVIO_ERR_UNSUPPORTED; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RVIO_OK;
}

int rvio_hip_create(const rvio_config* cfg, int device, rvio_hip** out) { return create_impl(cfg, device, 1, true, out); }
// B independent instances behind one handle: every stage is ONE launch with gridDim.z = B.
// front_end = 0: filter only (rvio_hip_frame_tracks_dev); 1: with CLAHE / detector / KLT / RANSAC / book-keeping (rvio_hip_frame_batch_dev)
int rvio_hip_create_batch(const rvio_config* cfg, int device, int n_instances, int front_end, rvio_hip** out) {
    return create_impl(cfg, device, n_instances, front_end != 0, out);
}
int rvio_hip_batch_size(const rvio_hip* h) { return h ? h->batch : 0; }

void rvio_hip_destroy(rvio_hip* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream_c) hipStreamSynchronize(h->stream_c);
    if (h->stream_e) hipStreamSynchronize(h->stream_e);
    if (h->stream_d) hipStreamSynchronize(h->stream_d);
    if (h->stream_t) hipStreamSynchronize(h->stream_t);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->private_queues) g_private_queue_handles.fetch_sub(1);
    for (void* p : h->allocs) hipFree(p);
    for (int k = 0; k < rvio_hip::kPin; ++k) { if (h->pin[k]) hipHostFree(h->pin[k]); if (h->evPin[k]) hipEventDestroy(h->evPin[k]); if (h->evPin2[k]) hipEventDestroy(h->evPin2[k]); }
    if (h->first_mirror) hipHostFree(h->first_mirror);
    if (h->evA) hipEventDestroy(h->evA);
    if (h->evL) hipEventDestroy(h->evL);
    if (h->evD0) hipEventDestroy(h->evD0);
    if (h->evD1) hipEventDestroy(h->evD1);
    if (h->stream_d && !h->one_stream) hipStreamDestroy(h->stream_d);
    if (h->stream_c && !h->one_stream) hipStreamDestroy(h->stream_c);
    if (h->stream_e && !h->one_stream) hipStreamDestroy(h->stream_e);
    for (int b = 0; b < rvio_hip::kIC; ++b) if (h->evC[b]) hipEventDestroy(h->evC[b]);
    for (int b = 0; b < 4; ++b) { if (h->evT[b]) hipEventDestroy(h->evT[b]); if (h->evH[b]) hipEventDestroy(h->evH[b]); }
    for (int b = 0; b < rvio_hip::kHand; ++b) if (h->evF[b]) hipEventDestroy(h->evF[b]);
    for (int b = 0; b < 2; ++b) if (h->evIn[b]) hipEventDestroy(h->evIn[b]);
    if (h->stream_t && !h->one_stream) hipStreamDestroy(h->stream_t);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}
const char* rvio_hip_last_error(const rvio_hip* h) { return h ? h->err.c_str() : "null handle"; }
void* rvio_hip_stream(rvio_hip* h) { return h ? (void*)h->stream : nullptr; }
// every stream of the handle, nothing else (no error check: the recovery paths use it)
static int drain_all(rvio_hip* h) {
    HIPCHK(h, hipSetDevice(h->device));
    SYNC_FRONT(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RVIO_OK;
}
// (Read-backs go through the handle's own stream: the CU-mask streams of the first live handle are created by
// hipExtStreamCreateWithCUMask as BLOCKING streams, a later handle's as non-blocking ones — a NULL-stream hipMemcpy would synchronise with
// the former only, and with whatever else the process has on its NULL stream.)
int rvio_hip_sync(rvio_hip* h) {
    if (!h) return RVIO_ERR_INVALID;
    { const int rc = drain_all(h); if (rc != RVIO_OK) return rc; }
    // a device-side stage counter that timed out (stage_wait, rvio_dev.h) left the frame sequence broken: a hard error, not a flag to poll.
    // rvio_hip_initialize is the way out of it (it resets the counters and clears the flag).
    int e = 0;
    HIPCHK(h, hipMemcpyAsync(&e, &h->meta->err, sizeof e, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (e & 4) { h->err = "a device-side stage counter timed out (filter -> book-keeping): the frame sequence is invalid, re-initialise"; return RVIO_ERR_STATE; }
    return RVIO_OK;
}

// ------------------------------------------------------------------ state
static int set_state_range(rvio_hip* h, int lo, int hi, const double* x, int xdim, const double* P, int d) {
    if (!h || !x || !P) return RVIO_ERR_INVALID;
    const int n = (xdim - 26) / 7;
    if (xdim != 26 + 7 * n || d != 24 + 6 * n || n < 0 || n > h->dc.nmax) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    FilterMeta m; std::memset(&m, 0, sizeof m);
    m.n_clones = n; m.img_count = h->img_count;
    if (h->chol_async) { HIPCHK(h, hipStreamSynchronize(h->stream_l)); h->chol_async = false; }   // (a factor in flight reads the covariance that is about to be replaced)
    for (int i = lo; i < hi; ++i) {
        const size_t o = (size_t)i * h->slab_bytes;
        double* Pi = (double*)((char*)h->P[h->cur] + o);
        HIPCHK(h, hipMemsetAsync(Pi, 0, sizeof(double) * h->dc.dmax * h->dc.dmax, h->stream));
        HIPCHK(h, hipMemcpyAsync((char*)h->x[h->cur] + o, x, sizeof(double) * xdim, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpy2DAsync(Pi, sizeof(double) * h->dc.dmax, P, sizeof(double) * d, sizeof(double) * d, d, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync((char*)h->meta + o, &m, sizeof m, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->n_clones_host = n;
    h->chol_ready = false;   // (the slab's factor belongs to the covariance that was just replaced)
    return RVIO_OK;
}
// (a batch handle: every instance receives the same state)
int rvio_hip_set_state(rvio_hip* h, const double* x, int xdim, const double* P, int d) { return h ? set_state_range(h, 0, h->batch, x, xdim, P, d) : RVIO_ERR_INVALID; }
// one instance of a batch handle; the window length is common to all instances (it depends on the frame count only)
int rvio_hip_set_state_at(rvio_hip* h, int instance, const double* x, int xdim, const double* P, int d) {
    if (!h || instance < 0 || instance >= h->batch || xdim != 26 + 7 * h->n_clones_host) return RVIO_ERR_INVALID;
    return set_state_range(h, instance, instance + 1, x, xdim, P, d);
}

int rvio_hip_get_state_at(rvio_hip* h, int instance, double* x, int* xdim, double* P, int* d) {
    if (!h || instance < 0 || instance >= h->batch) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t o = (size_t)instance * h->slab_bytes;
    const int n = h->n_clones_host, dd = 24 + 6 * n, xd = 26 + 7 * n;
    if (xdim) *xdim = xd;
    if (d) *d = dd;
    if (x) HIPCHK(h, hipMemcpyAsync(x, (char*)h->x[h->cur] + o, sizeof(double) * xd, hipMemcpyDeviceToHost, h->stream));
    if (P) HIPCHK(h, hipMemcpy2DAsync(P, sizeof(double) * dd, (char*)h->P[h->cur] + o, sizeof(double) * h->dc.dmax, sizeof(double) * dd, dd,
                                      hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RVIO_OK;
}
int rvio_hip_get_state(rvio_hip* h, double* x, int* xdim, double* P, int* d) { return rvio_hip_get_state_at(h, 0, x, xdim, P, d); }

// System::initialize (System.cc:115-170): runs once, on the host; result uploaded.
int rvio_hip_initialize(rvio_hip* h, const double w[3], const double a[3], int n_imu) {
    if (!h || !w || !a) return RVIO_ERR_INVALID;
    const rvio_config& c = h->cfg;
    double an = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    double g[3] = {a[0] / an, a[1] / an, a[2] / an};
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (c.ini_enable_alignment) {
        double xv[3], yv[3];
        const double ex[3] = {1, 0, 0};
        for (int i = 0; i < 3; ++i) xv[i] = ex[i] - (g[i] * g[0] * ex[0] + g[i] * g[1] * ex[1] + g[i] * g[2] * ex[2]);
        double xn = std::sqrt(xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2]);
        for (int i = 0; i < 3; ++i) xv[i] /= xn;
        yv[0] = -g[2] * xv[1] + g[1] * xv[2]; yv[1] = g[2] * xv[0] - g[0] * xv[2]; yv[2] = -g[1] * xv[0] + g[0] * xv[1];
        double yn = std::sqrt(yv[0] * yv[0] + yv[1] * yv[1] + yv[2] * yv[2]);
        for (int i = 0; i < 3; ++i) yv[i] /= yn;
        for (int i = 0; i < 3; ++i) { R[3 * i] = xv[i]; R[3 * i + 1] = yv[i]; R[3 * i + 2] = g[i]; }
    }
    // RotToQuat (Numerics.h:126-167), host copy
    double q[4]; const double T = R[0] + R[4] + R[8];
    if (R[0] > T && R[0] > R[4] && R[0] > R[8]) { q[0] = std::sqrt((1 + 2 * R[0] - T) / 4); double k = 1 / (4 * q[0]); q[1] = k * (R[1] + R[3]); q[2] = k * (R[2] + R[6]); q[3] = k * (R[5] - R[7]); }
    else if (R[4] > T && R[4] > R[0] && R[4] > R[8]) { q[1] = std::sqrt((1 + 2 * R[4] - T) / 4); double k = 1 / (4 * q[1]); q[0] = k * (R[1] + R[3]); q[2] = k * (R[5] + R[7]); q[3] = k * (R[6] - R[2]); }
    else if (R[8] > T && R[8] > R[0] && R[8] > R[4]) { q[2] = std::sqrt((1 + 2 * R[8] - T) / 4); double k = 1 / (4 * q[2]); q[0] = k * (R[2] + R[6]); q[1] = k * (R[5] + R[7]); q[3] = k * (R[1] - R[3]); }
    else { q[3] = std::sqrt((1 + T) / 4); double k = 1 / (4 * q[3]); q[0] = k * (R[5] - R[7]); q[1] = k * (R[6] - R[2]); q[2] = k * (R[1] - R[3]); }
    double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= qn;
    if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
    double x[26] = {0}; double P[576] = {0};
    for (int i = 0; i < 4; ++i) x[i] = q[i];
    for (int i = 0; i < 3; ++i) x[7 + i] = g[i];
    if (n_imu > 1) for (int i = 0; i < 3; ++i) { x[20 + i] = w[i]; x[23 + i] = a[i] - c.gravity * g[i]; }
    const double dt = 1. / c.imu_rate;
    auto D = [&](int i, double v) { P[i * 24 + i] = v; };
    for (int i = 0; i < 6; ++i) D(i, std::pow(1e-3, 2));
    for (int i = 6; i < 9; ++i) D(i, n_imu * dt * std::pow(c.sigma_a, 2));
    for (int i = 18; i < 21; ++i) D(i, n_imu * dt * std::pow(c.sigma_wg, 2));
    for (int i = 21; i < 24; ++i) D(i, n_imu * dt * std::pow(c.sigma_wa, 2));
    h->img_count = 0;
    // a (re-)initialised filter starts with an empty window: the tracker starts over too (mbIsTheFirstImage, Tracker.cc:88), or its
    // histories would be longer than the window they refer to
    if (h->front_end) {
        // (plain drains, not rvio_hip_sync: re-initialising is the recovery path after a stage counter timed out — RVIO_ERR_STATE —, so the
        // sticky flag must not keep the handle from getting here; set_state below rewrites FilterMeta, flag included)
        int rc0 = drain_all(h);
        if (rc0 != RVIO_OK) return rc0;
        // the device-side counters start over together with their host-side targets: after a time-out they no longer agree
        HIPCHK(h, hipMemsetAsync(h->stage_sync, 0, sizeof(StageSync), h->stream));
        h->stage_tgt = StageSync{};
        for (int b = 0; b < rvio_hip::kHand; ++b) { h->fin_mode[b] = 0; h->fin_target[b] = 0; }
        h->book_wait = nullptr; h->book_dev = false; h->book_target = 0; h->gate_pending = false; h->gate_target = 0; h->last_ra = false;
        std::vector<int> ones((size_t)h->batch, 1);
        HIPCHK(h, hipMemcpy2DAsync(h->t.first, h->slab_bytes, ones.data(), sizeof(int), sizeof(int), (size_t)h->batch, hipMemcpyHostToDevice, h->stream));
        for (int i = 0; i < h->batch; ++i) {
            const size_t o = (size_t)i * h->slab_bytes;
            HIPCHK(h, hipMemsetAsync((char*)h->t.n_pts + o, 0, sizeof(int), h->stream));
            HIPCHK(h, hipMemsetAsync((char*)h->t.hist_len + o, 0, sizeof(int) * h->dc.F, h->stream));
            for (int b = 0; b < rvio_hip::kHand; ++b) HIPCHK(h, hipMemsetAsync((char*)h->tout[b].n_feat + o, 0, sizeof(int), h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->frame_no = 0; h->piped = false; h->in_frame = false; h->fuse_m = -1;
        for (int i = 0; i < h->batch; ++i) h->first_mirror[i] = 1;
        h->first_cleared = false;
    }
    return rvio_hip_set_state(h, x, 26, P, 24);
}

// ------------------------------------------------------------------ P1
// PreIntegrator::propagate iterates whatever list it is handed (PreIntegrator.cc:96-97) — a dropped image or a stalled camera driver makes
// that list long.  The kernels take any m (propagate and RANSAC's gyro prior walk the samples in chunks); what is sized is the staging of
// the HOST-buffer entry points: RVIO_HIP_MAX_IMU samples are allocated up front, a longer batch grows it once (the host waits for the
// handle's streams, allocates, goes on) instead of being refused.
static int ensure_imu_capacity(rvio_hip* h, int m) {
    if (m <= h->imu_cap) return RVIO_OK;
    int rc = drain_all(h);   // (an earlier device-side error stays visible through rvio_hip_sync / rvio_hip_get_frame_info)
    if (rc != RVIO_OK) return rc;
    const int cap = std::max(2 * h->imu_cap, (m + 63) & ~63);
    auto grow = [&](rvio_imu** p) -> int {
        void* q = nullptr;
        HIPCHK(h, hipMalloc(&q, sizeof(rvio_imu) * (size_t)cap));
        h->allocs.push_back(q);      // (the old block stays where it was — a slab member, or a block freed with the handle)
        *p = (rvio_imu*)q;
        return RVIO_OK;
    };
    if ((rc = grow(&h->d_imu)) != RVIO_OK) return rc;
    for (int k = 0; k <= rvio_hip::kHand; ++k) if (h->hb_imu[k] && (rc = grow(&h->hb_imu[k])) != RVIO_OK) return rc;
    for (int k = 0; k < rvio_hip::kPin; ++k) if (h->pin[k]) { hipHostFree(h->pin[k]); h->pin[k] = nullptr; }   // rvio_hip_frame lays the ring out again
    h->imu_cap = cap;
    return RVIO_OK;
}
static int propagate_dev(rvio_hip* h, const rvio_imu* d_imu, int m, size_t imu_bs = 0, hipStream_t st = nullptr) {   // imu_bs = 0: every instance integrates the same samples
    static const bool prop_b = ab_env("RVIO_NO_PROP_B") == nullptr;   // A/B timing
    if (!st) st = h->stream;
    h->time_imu = d_imu; h->time_m = m;   // (rvio_hip_debug_time_kernel(8))
    if (h->batch > 8 && prop_b)
        hipLaunchKernelGGL(propagate_kernel3b, dim3(1, 1, h->batch), dim3(256), 0, st, h->dc, h->meta, h->n_clones_host, h->x[h->cur], h->P[h->cur], d_imu, m,
                           h->slab_bytes, imu_bs);
    else if (h->batch == 1 && h->solve9_nt && h->solve9_nt <= 6 && h->n_clones_host >= 1 && !imu_bs) {
        // plain handle, 6n <= 96: the Cholesky role of the solve (solve9.hip) as a second workgroup — the clone block it factors is the one the update
        // behind this propagate will see (propagation does not touch it)
        const int nc = h->n_clones_host;
        if (h->solve9_nt == 4) hipLaunchKernelGGL(propagate_chol_kernel<2>, dim3(2), dim3(256), 0, st, h->dc, h->meta, nc, h->x[h->cur], h->P[h->cur], d_imu, m, h->S9scr);
        else hipLaunchKernelGGL(propagate_chol_kernel<3>, dim3(2), dim3(256), 0, st, h->dc, h->meta, nc, h->x[h->cur], h->P[h->cur], d_imu, m, h->S9scr);
        h->chol_ready = true;
    }
    else
    hipLaunchKernelGGL(propagate_kernel3, dim3(1, 1, h->batch), dim3(256), 0, st, h->dc, h->meta, h->n_clones_host, h->x[h->cur], h->P[h->cur], d_imu, m,
                       h->slab_bytes, imu_bs);
    HIPCHK(h, hipGetLastError());
    return RVIO_OK;
}
int rvio_hip_propagate(rvio_hip* h, const rvio_imu* imu, int m) {
    if (!h || (!imu && m > 0) || m < 0) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    { const int rcg = ensure_imu_capacity(h, m); if (rcg != RVIO_OK) return rcg; }
    if (m > 0) HIPCHK(h, hipMemcpyAsync(h->d_imu, imu, sizeof(rvio_imu) * m, hipMemcpyHostToDevice, h->stream));
    return propagate_dev(h, h->d_imu, m);
}

// ------------------------------------------------------------------ U1..U10
static int upload_tracks(rvio_hip* h, const rvio_tracks* tr) {
    const DevCfg& d = h->dc;
    if (!tr || tr->n_feat < 0 || tr->n_feat > d.Fu) return RVIO_ERR_INVALID;
    std::vector<float> meas((size_t)d.Fu * d.max_len * 2, 0.f);
    for (int f = 0; f < tr->n_feat; ++f) {
        if (tr->len[f] < 2 || tr->len[f] > d.max_len || tr->len[f] > tr->max_len) return RVIO_ERR_INVALID;
        if (tr->len[f] - 1 > h->n_clones_host) return RVIO_ERR_INVALID;
        std::memcpy(&meas[(size_t)f * d.max_len * 2], tr->meas + (size_t)f * tr->max_len * 2, sizeof(float) * 2 * tr->len[f]);
    }
    int nf = tr->n_feat;
    HIPCHK(h, hipMemcpyAsync(h->t.n_feat, &nf, sizeof nf, hipMemcpyHostToDevice, h->stream));
    if (nf > 0) {
        HIPCHK(h, hipMemcpyAsync(h->t.types, tr->types, nf, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->t.len, tr->len, sizeof(int) * nf, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->t.meas, meas.data(), sizeof(float) * meas.size(), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));   // `meas` is a stack-lifetime staging buffer
    return RVIO_OK;
}

static LitArgs lit_args(const rvio_hip* h, size_t lds_bytes) { return LitArgs{h->lit_rows, h->lit_rows ? h->lit_state : nullptr, h->t.n_feat, lds_bytes / sizeof(double)}; }
// the share reduction of a batch handle whose [A|b] fits in LDS (6n <= 90): tiles of 16, 4 x 4 up to 6n = 63, 6 x 6 beyond
static void launch_gram_batch(rvio_hip* h, int n) {
    const dim3 g(1, 1, h->batch), b(256);
    if (h->dc.ldh - 1 <= 63)
        hipLaunchKernelGGL(gram_reduce_batch_kernel<4>, g, b, h->gram_batch_lds, h->stream, h->dc, n, h->partial, h->nrows, h->t.types, h->t.len, h->block, h->slab_bytes, h->bin);
    else
        hipLaunchKernelGGL(gram_reduce_batch_kernel<6>, g, b, h->gram_batch_lds, h->stream, h->dc, n, h->partial, h->nrows, h->t.types, h->t.len, h->block, h->slab_bytes, h->bin);
    if (h->lit_rows)   // the literal sweep for the instances whose small stacks need it (literal.h)
        hipLaunchKernelGGL(lit_batch_kernel, g, b, h->lit_batch_lds, h->stream, h->dc, n, (const int*)h->nrows,
                           (const unsigned char*)h->t.types, (const int*)h->t.len, h->block, h->slab_bytes, h->bin, lit_args(h, h->lit_batch_lds));
}

static int update_local_dev(rvio_hip* h, int rank, int world, bool combine) {
    const DevCfg& d = h->dc;
    const int n = h->n_clones_host;
    const size_t bs = h->slab_bytes;
    const int B = h->batch;
    if (h->fuse_m >= 0) {   // propagate + U1..U5 in one launch (independent: see feat_prop_kernel); single instance (sharded or not: every rank propagates, builds its features)
        // solve9 at 6n <= 96: the Cholesky of the clone block as one more workgroup of this launch (the solve of this very update follows on the stream)
        const bool chol = h->solve9_nt && h->solve9_nt <= 6 && n >= 1;
        double* cs = chol ? h->S9scr : (double*)nullptr;
        const int extra = 1 + (chol ? 1 : 0);
        hipLaunchKernelGGL(feat_prop_kernel, dim3(d.Fu + extra), dim3(256), h->fprop_lds, h->stream, d, n, h->x[h->cur], h->P[h->cur],
                           h->t.n_feat, h->t.types, h->t.len, h->t.meas, h->partial, h->nrows, h->acc, h->ndof, h->gamma, h->pfinv, h->tm_global, h->bin,
                           h->meta, h->fuse_imu, h->fuse_m, cs, h->solve9_nt, rank, world, h->lit_rows);
        h->chol_ready = chol;
        h->fuse_m = -1;
    } else
    if (B == 1)   // one stream: the latency form (every operand load of a gate tile in flight at once)
    hipLaunchKernelGGL(feat_build_kernel<16>, dim3(d.Fu, 1, B), dim3(h->feat_threads), h->feat_lds, h->stream, d, n, h->x[h->cur], h->P[h->cur],
                       h->t.n_feat, h->t.types, h->t.len, h->t.meas, rank, world, h->partial, h->nrows, h->acc, h->ndof, h->gamma, h->pfinv,
                       h->tm_global, bs, h->bin, h->meta, (const double*)nullptr, (const int*)nullptr, h->lit_rows);
    else {
    if (h->gpose)   // U1 + U2 four features per wave, ahead of the per-feature kernel (which then only fetches the pose chain and the triple)
        hipLaunchKernelGGL(geom4_kernel, dim3((d.Fu + 3) / 4, 1, B), dim3(64), 0, h->stream, d, n, h->x[h->cur], h->t.n_feat, h->t.types, h->t.len, h->t.meas,
                           h->gpose, h->pfinv, h->gvalid, bs, h->bin);
    hipLaunchKernelGGL(feat_build_kernel<4>, dim3(d.Fu, 1, B), dim3(h->feat_threads), h->feat_lds, h->stream, d, n, h->x[h->cur], h->P[h->cur],
                       h->t.n_feat, h->t.types, h->t.len, h->t.meas, rank, world, h->partial, h->nrows, h->acc, h->ndof, h->gamma, h->pfinv,
                       h->tm_global, bs, h->bin, h->meta, (const double*)h->gpose, (const int*)h->gvalid, h->lit_rows);
    }
    // unsharded: the last workgroup turns the block into [A|b] in place (rank truncation included); sharded: the block is the payload
    // one stream: 64 elements per workgroup (the shares are remote reads: spread them over many CUs); batch handles: 256 (fewer, fuller workgroups)
    const int gram_chunk = (B == 1) ? 64 : 256;
    static const bool no_gram_batch = ab_env("RVIO_NO_GRAM_BATCH") != nullptr;   // A/B timing
    if (B >= 128 && world == 1 && combine && h->gram_batch_lds && !no_gram_batch)   // batch handle, [A|b] fits in LDS: one workgroup per instance, stored tiles only
        launch_gram_batch(h, n);
    else
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(std::max(1, std::min(1024, (6 * n * d.ldh + gram_chunk - 1) / gram_chunk)), 1, B), dim3(256), h->trunc_lds, h->stream, d, n,
                       h->partial, h->nrows, h->t.types, h->t.len, h->block, h->gram_cnt, (world == 1 && combine) ? 1 : 0, (B == 1) ? 1 : 0, bs, h->bin, lit_args(h, h->trunc_lds));
    HIPCHK(h, hipGetLastError());
    return RVIO_OK;
}

static void launch_solve(rvio_hip* h, int n, const double* Ab, bool defer_dx = false) {   // defer_dx: the caller launches the Joseph stage right behind (update_global_dev)
    const DevCfg& d = h->dc;
    double *xin = h->x[h->cur], *xout = h->x[h->cur ^ 1], *Pc = h->P[h->cur];
    const dim3 gb(1, 1, h->batch);
    if (h->solve9_nt) {   // blocked SPD factorisations on the matrix cores (solve9.hip): one workgroup, one launch
        if (h->chol_async && hipStreamWaitEvent(h->stream, h->evL, 0) != hipSuccess) { h->chol_async = false; h->chol_ready = false; }   // the factor that started behind the last augment / compose (a failed wait: the solve factors Pcc itself)
        const bool pre = h->chol_ready || h->chol_async;   // L, G of the clone block are in the slab already (role workgroup of this update's per-feature / propagate launch; stream_l)
        h->chol_ready = false; h->chol_async = false;
        switch (h->solve9_nt) {
        case 4:
            static const bool s9_generic = ab_env("RVIO_S9_GENERIC") != nullptr;   // A/B timing: the generic kernel (tiles through the L2 slab) at 6n <= 64
            if (pre && !s9_generic) {
                // in the frame's update the Joseph stage follows on the stream: dx = Pc y and the state injection become role workgroups of that launch (launch_ug_final)
                static const bool no_dx_small = ab_env("RVIO_S9_NO_DX_ROLE") != nullptr;   // A/B timing
                const bool role = defer_dx && h->batch == 1 && 6 * n <= 60 && !no_dx_small && !ab_env("RVIO_NO_JOSEPH_LDS");
                hipLaunchKernelGGL(solve9_small_kernel, dim3(1), dim3(1024), sizeof(S9SmallLds), h->stream, d, h->meta, n, Ab, xin, Pc, h->S9scr, h->W, xout,
                                   role ? h->S9scr + S9_YP_OFF(4) : (double*)nullptr);
                h->dx_pending = role;
            }
#ifdef RVIO_DBG_CLOCKS   // (A/B form RVIO_S9_GENERIC: compiled into the instrumented build only — round 6, kernel forms no handle of the shipping library launches)
            else if (pre) hipLaunchKernelGGL((solve9_kernel<1, 4, true>), gb, dim3(1024), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->S9scr, h->W, xout, h->slab_bytes, (size_t)0);
#endif
            else hipLaunchKernelGGL((solve9_kernel<1, 4>), gb, dim3(1024), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->S9scr, h->W, xout, h->slab_bytes,
                                    h->batch > 1 ? S9_SLAB_DOUBLES(4) * sizeof(double) : (size_t)0);
            return;
        case 6:
        {
            // the frame's update at 64 < 6n_max <= 96: dx = Pc y and the state injection ride in the Joseph stage's launch (joseph_lds_kernel while the window still
            // holds <= 10 clones, ug_tile_kernel<0> beyond 64 columns; the two-launch LDS form in between has no roles)
            static const bool no_dx_mid = ab_env("RVIO_S9_NO_DX_ROLE") != nullptr;   // A/B timing
            const bool role = defer_dx && h->batch == 1 && !no_dx_mid && (6 * n <= 60 ? !ab_env("RVIO_NO_JOSEPH_LDS") : (6 * n > 64 && !ab_env("RVIO_NO_UG_TILE")));
            if (pre) hipLaunchKernelGGL((solve9_kernel<2, 3, true>), gb, dim3(576), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->S9scr, h->W, xout, h->slab_bytes, (size_t)0, role ? 1 : 0);
            else hipLaunchKernelGGL((solve9_kernel<2, 3>), gb, dim3(576), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->S9scr, h->W, xout, h->slab_bytes, (size_t)0, role ? 1 : 0);
            h->dx_pending = role;
        }
            return;
        default: break;
        }
        // 6n > 96: the split form — the four product phases as launches that fill the chip, the two factorisations as one workgroup each (solve9.hip)
        static const bool s9_one = ab_env("RVIO_S9_ONE") != nullptr;   // A/B timing: everything in ONE workgroup
        const int NT = h->solve9_nt, nwg = (NT * NT + 3) / 4;
        if (s9_one) {
            if (NT == 8) hipLaunchKernelGGL((solve9_kernel<2, 4>), gb, dim3(1024), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->S9scr, h->W, xout, h->slab_bytes, (size_t)0);
            else hipLaunchKernelGGL((solve9_kernel<3, 4>), gb, dim3(1024), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->S9scr, h->W, xout, h->slab_bytes, (size_t)0);
            return;
        }
        if (!pre) {
            if (NT == 8) hipLaunchKernelGGL((solve9_chol_kernel<2, 4>), dim3(1), dim3(1024), 0, h->stream, d, n, Pc, h->S9scr);
            else hipLaunchKernelGGL((solve9_chol_kernel<3, 4>), dim3(1), dim3(1024), 0, h->stream, d, n, Pc, h->S9scr);
        }
        hipLaunchKernelGGL(solve9_prod_kernel<0>, dim3(nwg), dim3(256), 0, h->stream, d, n, Ab, h->S9scr, h->W, NT);
        hipLaunchKernelGGL(solve9_prod_kernel<1>, dim3(nwg), dim3(256), 0, h->stream, d, n, Ab, h->S9scr, h->W, NT);
        static const bool sweep4 = ab_env("RVIO_S9_SWEEP4") != nullptr;   // A/B timing: the sweep on 2 x 2 waves with 16 / 36 tiles each (no spills, one wave per matrix pipe)
#ifdef RVIO_DBG_CLOCKS
        if (sweep4) {
            if (NT == 8) hipLaunchKernelGGL((solve9_sweep_kernel<4, 2>), dim3(1), dim3(256), 0, h->stream, d, Ab, h->S9scr);
            else hipLaunchKernelGGL((solve9_sweep_kernel<6, 2>), dim3(1), dim3(256), 0, h->stream, d, Ab, h->S9scr);
        } else
#else
        (void)sweep4;
#endif
        if (NT == 8) hipLaunchKernelGGL((solve9_sweep_kernel<2, 4>), dim3(1), dim3(1024), 0, h->stream, d, Ab, h->S9scr);
        else hipLaunchKernelGGL((solve9_sweep_kernel<3, 4>), dim3(1), dim3(1024), 0, h->stream, d, Ab, h->S9scr);
        hipLaunchKernelGGL(solve9_prod_kernel<2>, dim3(nwg), dim3(256), 0, h->stream, d, n, Ab, h->S9scr, h->W, NT);
        hipLaunchKernelGGL(solve9_prod_kernel<3>, dim3(nwg), dim3(256), 0, h->stream, d, n, Ab, h->S9scr, h->W, NT);
        static const bool no_dx_role = ab_env("RVIO_S9_NO_DX_ROLE") != nullptr;   // A/B timing
        if (defer_dx && 6 * n > 64 && !no_dx_role && !ab_env("RVIO_NO_UG_TILE")) h->dx_pending = true;   // (6n <= 64 while the window fills: launch_ug_final takes its short-window kernels)
        else hipLaunchKernelGGL(solve9_dx_kernel, dim3(1), dim3(1024), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->S9scr, h->W, xout, NT);
        return;
    }
    switch (h->solve7_variant) {   // T = s2 I + A Pcc is formed by the kernel itself
#ifdef RVIO_DBG_CLOCKS   // (the register-tableau solve of rounds 3-4 at 6n <= 128: plain handles run solve9, batch handles solve6 there — A/B forms of the instrumented build)
    case 1: {
        static const int nw = ab_env("RVIO_S7_NW") ? atoi(ab_env("RVIO_S7_NW")) : 4;
        const size_t lds = (size_t)(3 * 64 * 65 + 24 * 64) * sizeof(double);
        if (nw == 8) hipLaunchKernelGGL((solve7_kernel<1, 8, 8>), gb, dim3(512), lds, h->stream, d, h->meta, n, Ab, xin, Pc, h->Tbuf, h->W, xout, h->slab_bytes);
        else if (nw == 16) hipLaunchKernelGGL((solve7_kernel<1, 4, 16>), gb, dim3(1024), lds, h->stream, d, h->meta, n, Ab, xin, Pc, h->Tbuf, h->W, xout, h->slab_bytes);
        else hipLaunchKernelGGL((solve7_kernel<1, 16, 4>), gb, dim3(256), lds, h->stream, d, h->meta, n, Ab, xin, Pc, h->Tbuf, h->W, xout, h->slab_bytes);
        return;
    }
    case 2: {
        static const int nw2 = ab_env("RVIO_S7_V2") ? atoi(ab_env("RVIO_S7_V2")) : 12;   // waves of the 6n <= 96 form (measured at 6n = 84: 8 -> 109 us, 12 -> 102, 16 -> 119)
        if (nw2 == 16) hipLaunchKernelGGL((solve7_kernel<2, 6, 16>), gb, dim3(1024), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->Tbuf, h->W, xout, h->slab_bytes);
        else if (nw2 == 12) hipLaunchKernelGGL((solve7_kernel<2, 8, 12>), gb, dim3(768), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->Tbuf, h->W, xout, h->slab_bytes);
        else hipLaunchKernelGGL((solve7_kernel<2, 12, 8>), gb, dim3(512), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->Tbuf, h->W, xout, h->slab_bytes);
        return;
    }
    case 3: hipLaunchKernelGGL((solve7_kernel<2, 16, 8>), gb, dim3(512), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->Tbuf, h->W, xout, h->slab_bytes); return;
    case 5: hipLaunchKernelGGL((solve7_kernel<1, 16, 4, false, 8>), gb, dim3(256), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->Tbuf, h->W, xout, h->slab_bytes); return;
#endif
    case 4: hipLaunchKernelGGL((solve7_kernel<3, 16, 12>), gb, dim3(768), 0, h->stream, d, h->meta, n, Ab, xin, Pc, h->Tbuf, h->W, xout, h->slab_bytes); return;
    default: break;
    }
    if (h->solve5_variant == 1)
        hipLaunchKernelGGL((solve6_kernel<1, 8, 8>), dim3(1, 1, h->batch), dim3(512), h->solve5_lds, h->stream, d, h->meta, n, h->Tbuf, Ab, xin, Pc, h->W, xout, h->slab_bytes);
    else if (h->solve5_variant == 2)
        hipLaunchKernelGGL((solve6_kernel<2, 12, 8>), dim3(1, 1, h->batch), dim3(512), h->solve5_lds, h->stream, d, h->meta, n, h->Tbuf, Ab, xin, Pc, h->W, xout, h->slab_bytes);
    else if (h->solve5_variant == 3)
        hipLaunchKernelGGL((solve6_kernel<2, 16, 8>), dim3(1, 1, h->batch), dim3(512), h->solve5_lds, h->stream, d, h->meta, n, h->Tbuf, Ab, xin, Pc, h->W, xout, h->slab_bytes);
}

// U = Pc W, G = U A  (K H = [0 | G]);  Joseph form (Updater.cc:615-619): P1 = (I-KH) P,  P+ = sym(P1 - P1c G^T + s2 G U^T)
static void launch_ug_final(rvio_hip* h, int n, const double* Ab, double* Pn, bool ug, bool fin) {
    const DevCfg& d = h->dc;
    const int c6 = 6 * n, dd = 24 + c6, B = h->batch;
    const size_t bs = h->slab_bytes;
    double* Pc = h->P[h->cur];
    const int nt = (dd + 15) / 16, npair = nt * (nt + 1) / 2;
    static const bool no_ugl = ab_env("RVIO_NO_UGL") != nullptr;   // A/B timing
    static const bool no_jl = ab_env("RVIO_NO_JOSEPH_LDS") != nullptr;   // A/B timing
    if (h->jb_lds && ug && fin) {   // batch handle, 6n <= 60: P -> P+ in one kernel (U, G, P1 never leave the CU)
        hipLaunchKernelGGL(joseph_batch_kernel, dim3(1, 1, B), dim3(JB_THREADS), h->jb_lds, h->stream, d, n, Pc, h->W, Ab, Pn, bs);
    } else if (B == 1 && c6 <= 60 && ug && fin && !no_jl) {   // one instance, short window: both stages in ONE launch, a workgroup per tile pair of P+
        const bool dxr = h->dx_pending;   // the all-LDS solve left dx = Pc y and the state injection to role workgroups of this launch
        h->dx_pending = false;
        hipLaunchKernelGGL(joseph_lds_kernel, dim3(npair + (dxr ? (dd + 23) / 24 : 0)), dim3(256), JL_LDS_DOUBLES * sizeof(double), h->stream, d, n, Pc, h->W, Ab, Pn,
                           h->meta, (const double*)h->x[h->cur], h->x[h->cur ^ 1], dxr ? (const double*)h->S9scr : (const double*)nullptr, npair, h->solve9_nt);
    } else if (B == 1 && c6 <= 64 && !no_ugl) {   // one instance, short window: every operand of a workgroup staged in LDS with one batch of loads
        if (ug) hipLaunchKernelGGL(ug_lds_kernel, dim3(nt), dim3(256), UGL_LDS_DOUBLES * sizeof(double), h->stream, d, n, Pc, h->W, Ab, h->U, h->G, h->Pt1);
        if (fin) hipLaunchKernelGGL(final_lds_kernel, dim3((npair + 3) / 4), dim3(256), FNL_LDS_DOUBLES * sizeof(double), h->stream, d, n, h->Pt1, h->G, h->U, Pn);
    } else if (B == 1 && !ab_env("RVIO_NO_UG_TILE")) {   // one instance, 6n > 64: one wave per output tile, the chip is this instance's alone
        const int c6t = (c6 + 15) / 16;
        if (ug) {
            const bool dxr = h->dx_pending;   // the split solve left dx = Pc y and the state injection to role workgroups of this launch
            h->dx_pending = false;
            const double* nod = nullptr;
            hipLaunchKernelGGL(ug_tile_kernel<0>, dim3((nt * c6t + 3) / 4 + (dxr ? (dd + 23) / 24 : 0)), dim3(256), 0, h->stream, d, n, Pc, h->W, Ab, h->U, h->G, h->Pt1,
                               h->meta, (const double*)h->x[h->cur], h->x[h->cur ^ 1], dxr ? (const double*)h->S9scr : nod, h->solve9_nt);
            hipLaunchKernelGGL(ug_tile_kernel<1>, dim3((nt * c6t + 3) / 4), dim3(256), 0, h->stream, d, n, Pc, h->W, Ab, h->U, h->G, h->Pt1, h->meta, nod, (double*)nullptr, nod, 0);
            hipLaunchKernelGGL(ug_tile_kernel<2>, dim3((nt * nt + 3) / 4), dim3(256), 0, h->stream, d, n, Pc, h->W, Ab, h->U, h->G, h->Pt1, h->meta, nod, (double*)nullptr, nod, 0);
        }
        if (fin) hipLaunchKernelGGL(final_tile_kernel, dim3(npair), dim3(256), 0, h->stream, d, n, h->Pt1, h->G, h->U, Pn);
    } else {
        if (ug) hipLaunchKernelGGL(ug_kernel, dim3((dd + 15) / 16, 1, B), dim3(256), h->ug_lds, h->stream, d, n, Pc, h->W, Ab, h->U, h->G, h->Pt1, bs);
        if (fin) hipLaunchKernelGGL(final_kernel, dim3((npair + 3) / 4, 1, B), dim3(256), 0, h->stream, d, n, h->Pt1, h->G, h->U, Pn, bs);
    }
}

// combined: d_blocks is the handle's own block, already turned into [A|b] by gram_reduce_kernel (unsharded update)
static int update_global_dev(rvio_hip* h, const double* d_blocks, int world, bool combined) {
    const DevCfg& d = h->dc;
    const int n = h->n_clones_host, c6 = 6 * n, dd = 24 + c6;
    const long ldh = d.ldh;
    double* Pc = h->P[h->cur];
    double* Pn = h->P[h->cur ^ 1];
    const double* Ab = d_blocks;
    const size_t bs = h->slab_bytes;
    const int B = h->batch;
    if (B > 1 && !combined) { h->err = "a batch handle runs the unsharded updater only"; return RVIO_ERR_UNSUPPORTED; }
    if (!combined) {   // gathered shards [S2 | S1]: sum both parts in rank order, then the rank truncation -> Ab = [A|b]
        const int eg = std::max(1, std::min(64, (int)((c6 * ldh + 255) / 256)));
        hipLaunchKernelGGL(block_sum_kernel, dim3(eg), dim3(256), h->trunc_lds, h->stream, d, n, d_blocks, world, (size_t)shard_payload_doubles(c6, d.max_len), h->Ab, h->gram_cnt,
                           (const int*)h->nrows, (const unsigned char*)h->t.types, (const int*)h->t.len, lit_args(h, h->trunc_lds));
        Ab = h->Ab;
    }
    const int tt = (c6 + 31) / 32;
    static const bool no_gtl = ab_env("RVIO_NO_GEMM_T_LDS") != nullptr;   // A/B timing
    if (!h->solve7_variant && !h->solve9_nt) {
        if (B >= 128 && d.ldh - 1 <= 64 && !no_gtl)
            hipLaunchKernelGGL(gemm_T_lds_kernel, dim3(1, 1, B), dim3(256), (size_t)2 * c6 * (c6 + 1) * sizeof(double), h->stream, d, n, Ab, Pc, h->Tbuf, bs);
        else
            hipLaunchKernelGGL(gemm_T_kernel, dim3(tt, tt, B), dim3(256), 0, h->stream, d, n, Ab, Pc, h->Tbuf, bs);
    }
    h->last_Ab = Ab;
    launch_solve(h, n, Ab, /*defer_dx=*/true);
    launch_ug_final(h, n, Ab, Pn, true, true);
    HIPCHK(h, hipGetLastError());
    h->cur ^= 1;
    return RVIO_OK;
}

int rvio_hip_update_tracked(rvio_hip* h) {
    if (!h) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = update_local_dev(h, 0, 1, true);
    if (rc != RVIO_OK) return rc;
    return update_global_dev(h, h->block, 1, true);
}
int rvio_hip_update(rvio_hip* h, const rvio_tracks* tracks) {
    if (!h) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = upload_tracks(h, tracks);
    if (rc != RVIO_OK) return rc;
    return rvio_hip_update_tracked(h);
}
int rvio_hip_update_local(rvio_hip* h, const rvio_tracks* tracks, int rank, int world, double** d_block, int* n_doubles) {
    if (!h || world < 1 || rank < 0 || rank >= world) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    if (tracks) { int rc = upload_tracks(h, tracks); if (rc != RVIO_OK) return rc; }
    int rc = update_local_dev(h, rank, world, false);
    if (d_block) *d_block = h->block;
    if (n_doubles) *n_doubles = shard_payload_doubles(6 * h->n_clones_host, h->dc.max_len);
    return rc;
}
int rvio_hip_update_global(rvio_hip* h, const double* d_blocks, int world) {
    if (!h || !d_blocks || world < 1) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    return update_global_dev(h, d_blocks, world, false);
}

int rvio_hip_get_update_diag(rvio_hip* h, int32_t* n_feat, int32_t* accepted, double* gamma, int32_t* ndof, double* pfinv) {
    if (!h) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    SYNC_FRONT(h);   // image / side / tracker streams first
    int nf = 0;
    HIPCHK(h, hipMemcpyAsync(&nf, h->t.n_feat, sizeof nf, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n_feat) *n_feat = nf;
    if (nf > 0) {
        if (accepted) HIPCHK(h, hipMemcpyAsync(accepted, h->acc, sizeof(int) * nf, hipMemcpyDeviceToHost, h->stream));
        if (gamma) HIPCHK(h, hipMemcpyAsync(gamma, h->gamma, sizeof(double) * nf, hipMemcpyDeviceToHost, h->stream));
        if (ndof) HIPCHK(h, hipMemcpyAsync(ndof, h->ndof, sizeof(int) * nf, hipMemcpyDeviceToHost, h->stream));
        if (pfinv) HIPCHK(h, hipMemcpyAsync(pfinv, h->pfinv, sizeof(double) * 3 * nf, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RVIO_OK;
}

// ------------------------------------------------------------------ S1 + S2
static int augment_compose_dev(rvio_hip* h, int do_augment) {
    const DevCfg& d = h->dc;
    const int c = h->cur, o = c ^ 1;
    // one stream: width (one entry per thread, ~29 workgroups); a batch: every workgroup builds Vk first (a serial section of one thread),
    // so few fat workgroups per instance (the chip is full anyway)
    static const int aug_wgs = ab_env("RVIO_AUG_WGS") ? atoi(ab_env("RVIO_AUG_WGS")) : 4;   // A/B timing
    const int cg = 1 + (h->batch >= 128 ? std::max(1, aug_wgs) : std::max(1, std::min(64, (d.dmax * d.dmax + 255) / 256)));
    unsigned long long* done = nullptr;
    if (h->batch == 1) { done = &h->stage_sync->aug; h->stage_tgt.aug += (unsigned long long)cg; }
    if (h->chol_async) { HIPCHK(h, hipStreamWaitEvent(h->stream, h->evL, 0)); h->chol_async = false; }   // (a frame without an update: its factor was never consumed)
    hipLaunchKernelGGL(augcomp_kernel2, dim3(cg, 1, h->batch), dim3(256), 0, h->stream, d, h->n_clones_host, do_augment, h->x[c], h->P[c], h->x[o], h->P[o], h->d_pose,
                       h->slab_bytes, done);
    HIPCHK(h, hipGetLastError());
    h->cur = o;
    h->chol_ready = false;   // the clone block has changed (window slide / new clone)
    if (do_augment && h->n_clones_host < d.nmax) h->n_clones_host++;
    // Long windows (6n > 96, solve in its split form): Pcc = L L^T of the NEXT update is known from here on — propagation leaves the clone block
    // alone — so the factor starts now on a stream of its own and runs beside propagate, the wait for the tracker and the per-feature stage; the
    // solve's first product waits for it (launch_solve).  ~35 us at 6n = 120, ~95 us at 6n = 180 off the filter chain.
    static const bool no_async = ab_env("RVIO_S9_NO_ASYNC") != nullptr;   // A/B timing
    if (h->solve9_nt >= 8 && h->stream_l && !no_async && !profiler_serialises() && h->n_clones_host >= 1) {
        HIPCHK(h, hipEventRecord(h->evA, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream_l, h->evA, 0));
        if (h->solve9_nt == 8) hipLaunchKernelGGL((solve9_chol_kernel<2, 4>), dim3(1), dim3(1024), 0, h->stream_l, d, h->n_clones_host, h->P[h->cur], h->S9scr);
        else hipLaunchKernelGGL((solve9_chol_kernel<3, 4>), dim3(1), dim3(1024), 0, h->stream_l, d, h->n_clones_host, h->P[h->cur], h->S9scr);
        HIPCHK(h, hipEventRecord(h->evL, h->stream_l));
        h->chol_async = true;
    }
    return RVIO_OK;
}
int rvio_hip_augment_compose(rvio_hip* h, int do_augment) {
    if (!h) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    return augment_compose_dev(h, do_augment);
}

// ------------------------------------------------------------------ T7 detector (device), allocated on first use
static int detector_check(rvio_hip* h) {
    const int cell1 = (int)std::nearbyint((double)h->cfg.min_dist);
    if (cell1 < 1) { h->err = "Tracker.nMinDist < 1 is not supported by the device detector"; return RVIO_ERR_UNSUPPORTED; }
    const int spw = (int)std::floor(.5 * h->cfg.min_dist);   // cornerSubPix half-window, FeatureDetector.cc:68
    if (spw < 1 || spw > 63) { h->err = "device cornerSubPix takes half-windows 1..63 (2 <= Tracker.nMinDist < 128)"; return RVIO_ERR_UNSUPPORTED; }
    return RVIO_OK;
}
static int detector_alloc_set(rvio_hip* h, DetDev& q) {   // the scratch of ONE detector in flight
    const DevCfg& d = h->dc;
    const int cell1 = (int)std::nearbyint((double)h->cfg.min_dist), cell2 = (int)std::nearbyint((double)(2.f * h->cfg.min_dist));
    const size_t npx = (size_t)d.W * d.H;
    q.W = d.W; q.H = d.H; q.F = d.F; q.min_dist = h->cfg.min_dist; q.quality = (double)h->cfg.qual_lvl;
    q.max_cells = ((d.W + cell1 - 1) / cell1) * ((d.H + cell1 - 1) / cell1);
    q.first = h->t.first;
#ifdef RVIO_DBG_CLOCKS
    DALLOC(h, q.eig, npx);   // the two-pass A/B form (RVIO_DET_TWO_PASS) sends the map through HBM
#else
    q.eig = nullptr;   // (the pipeline keeps the min-eigenvalue map in LDS; rvio_hip_get_corners(eig) allocates ONE map per handle on first use)
#endif
    DALLOC(h, q.maxkey, 1); DALLOC(h, q.counters, 4); DALLOC(h, q.cell_cnt, (size_t)q.max_cells);
    DALLOC(h, q.cell_ent, (size_t)(d.W + cell2) * (d.H + cell2)); DALLOC(h, q.cell_ci, (size_t)(d.W + cell2) * (d.H + cell2));
    q.n_cap = (int)std::min(npx, (size_t)16384);
    DALLOC(h, q.nb, (size_t)q.n_cap * DET_NBCAP); DALLOC(h, q.nb_cnt, (size_t)q.n_cap);
    DALLOC(h, q.prov, npx); DALLOC(h, q.cand, npx); DALLOC(h, q.acc, npx); DALLOC(h, q.state, npx);
    DALLOC(h, q.raw_xy, (size_t)2 * d.F);
    return RVIO_OK;
}
static int detector_alloc(rvio_hip* h) {   // DALLOCs only (runs twice for a slab)
    const DevCfg& d = h->dc;
    int rc = RVIO_OK;
    for (int k = 0; k < h->n_ic; ++k) if ((rc = detector_alloc_set(h, h->dets[k])) != RVIO_OK) return rc;
    for (int k = h->n_ic; k < rvio_hip::kIC; ++k) h->dets[k] = h->dets[0];
    DALLOC(h, h->det_xy2[0], (size_t)2 * d.F); DALLOC(h, h->det_xy2[1], (size_t)2 * d.F); DALLOC(h, h->det_xy2[2], (size_t)2 * d.F);
    DALLOC(h, h->det_nout, 3);
    float* mask = nullptr;
    const size_t mside = (size_t)std::max(31, 2 * (int)std::floor(.5 * h->cfg.min_dist) + 1);
    DALLOC(h, mask, mside * mside);
    for (DetDev* q : {&h->dets[0], &h->dets[1], &h->dets[2]}) { q->xy = h->det_xy2[0]; q->n_out = h->det_nout; q->spmask = mask; q->sp_win = (int)std::floor(.5 * h->cfg.min_dist); }
    return RVIO_OK;
}
static int detector_init(rvio_hip* h) {
    if (h->det_ready) return RVIO_OK;
    int rc = detector_check(h);
    if (rc != RVIO_OK) return rc;
    if (!h->det_in_slab && (rc = detector_alloc(h)) != RVIO_OK) return rc;
    DetDev& q = h->dets[0];
    // cornerSubPix window (cornersubpix.cpp): float expf on the host, so that device and oracle share glibc's values
    const int spw = q.sp_win, spww = 2 * spw + 1;
    std::vector<float> hm((size_t)spww * spww);
    for (int i = 0; i < spww; ++i) {
        const float y = (float)(i - spw) / (float)spw;
        const float vy = std::exp(-y * y);
        for (int j = 0; j < spww; ++j) { const float x = (float)(j - spw) / (float)spw; hm[(size_t)i * spww + j] = (float)(vy * std::exp(-x * x)); }
    }
    HIPCHK(h, hipMemcpyAsync(const_cast<float*>(q.spmask), hm.data(), sizeof(float) * hm.size(), hipMemcpyHostToDevice, h->stream));   // one copy, shared by all instances
    HIPCHK(h, hipStreamSynchronize(h->stream));   // (hm is a local)
    if (spw > 15) HIPCHK(h, lds_attr((const void*)subpix_wide_kernel, (int)subpix_wide_lds(spw)));
    std::vector<int> minkey((size_t)h->batch, (int)0x80000000);
    for (DetDev* qq : {&h->dets[0], &h->dets[1], &h->dets[2]})
        HIPCHK(h, hipMemcpy2DAsync(qq->maxkey, h->slab_bytes ? h->slab_bytes : sizeof(int), minkey.data(), sizeof(int), sizeof(int), (size_t)(h->det_in_slab ? h->batch : 1),
                                   hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, lds_attr((const void*)neigh_kernel, (int)NEIGH_LDS));
    HIPCHK(h, lds_attr((const void*)greedy_kernel, (int)GREEDY_LDS));
    h->det_ready = true;
    return RVIO_OK;
}
// Front-end sequencing.  Two streams: the IMAGE stream h->ts (CLAHE and FeatureDetector::DetectWithSubPix: the longest chain, ~170 us)
// and the side stream h->side (pyramid, KLT, RANSAC: ~90 us).  Two modes:
//  * plain (per-stage calls, or a caller-side corner list): the side stream joins back and book-keeping runs on h->ts;
//  * run-ahead (pipelined whole-frame path with the device detector): book-keeping runs on the SIDE stream, so the image stream is
//    free for CLAHE + detector of frame k+1 as soon as the detector of frame k is done — the image chain never reads tracker state,
//    except mbIsTheFirstImage (the detector's distance factor), hence one wait on book-keeping(k-1) in front of nms(k).  What
//    book-keeping(k) still reads while frame k+1 is being detected is double-buffered by frame parity (equalised image, corner list).
static DetDev det_view(const rvio_hip* h) {
    DetDev q = h->dets[h->runahead ? h->ic : 0];
    q.xy = h->det_xy2[h->dslot]; q.n_out = h->det_nout + h->dslot;
    return q;
}
// the stream CLAHE and the detector of the call in progress run on: in run-ahead mode the image chains of consecutive frames
// alternate between two streams (each with its own detector scratch and CLAHE LUTs), so that two of them are in flight — the chain is
// ~150 us long, the longest of the frame, and with one stream it WAS the frame period
static hipStream_t image_stream_of(const rvio_hip* h, int ic) { return ic == 0 ? h->stream_t : (ic == 1 ? h->stream_c : h->stream_e); }
static hipStream_t image_stream(const rvio_hip* h) { return h->runahead ? image_stream_of(h, h->ic) : h->ts; }
static int detect_dev(rvio_hip* h, const uint8_t* img, int stride, size_t src_bs, hipEvent_t first_flag_ready) {
    const DevCfg& d = h->dc;
    const size_t bs = h->slab_bytes;
    const unsigned B = (unsigned)h->batch;
    const DetDev q = det_view(h);
    const hipStream_t ds = image_stream(h);
    h->det_set_last = h->runahead ? h->ic : 0;
    static const bool two_pass = ab_env("RVIO_DET_TWO_PASS") != nullptr;   // A/B timing
    if (h->pyr_signal && (h->wide_px || two_pass)) { hipLaunchKernelGGL(stage_signal_kernel, dim3(1), dim3(64), 0, ds, h->pyr_signal); h->pyr_signal = nullptr; }   // (forms without the folded signal)
    if (h->wide_px && !two_pass) {
        // batch handles of >= 8 instances: the fused pass in its throughput form (one wave per strip, rows walked with the state in registers)
        hipLaunchKernelGGL(mineig_nms_strip_kernel, dim3((d.W + DET_SW - 1) / DET_SW, (d.H + DET_SH - 1) / DET_SH, B), dim3(64), 0, ds, img, stride, q, src_bs, bs);
        if (first_flag_ready && !(kDbgSkip & 2)) HIPCHK(h, hipStreamWaitEvent(ds, first_flag_ready, 0));
        hipLaunchKernelGGL(nms_threshold_kernel, dim3(16, 1, B), dim3(NMS_T), 0, ds, q, bs);
    }
#ifdef RVIO_DBG_CLOCKS
    else if (two_pass) {   // rounds 1-3: the map through HBM
        const dim3 g((d.W + DET_TW - 1) / DET_TW, (d.H + DET_TH - 1) / DET_TH, B);
        hipLaunchKernelGGL(mineig_kernel, g, dim3(DET_T), 0, ds, img, stride, q, src_bs, bs, (int)h->frame_no);
        if (first_flag_ready && !(kDbgSkip & 2)) HIPCHK(h, hipStreamWaitEvent(ds, first_flag_ready, 0));   // nms reads mbIsTheFirstImage as book-keeping(k-1) left it
        hipLaunchKernelGGL(nms_kernel, g, dim3(DET_T), 0, ds, q, bs);
    }
#endif
    else {
        // one stream: min-eigenvalue map + strict 3x3 local maxima in one pass (the map stays in LDS), then the image-wide threshold on the provisional list
        hipLaunchKernelGGL(mineig_nms_kernel, dim3((d.W + DET_TW - 1) / DET_TW, (d.H + DET_FH - 1) / DET_FH, B), dim3(DET_T), 0, ds, img, stride, q, src_bs, bs, (int)h->frame_no, h->pyr_signal);
        h->pyr_signal = nullptr;
        if (first_flag_ready && !(kDbgSkip & 2)) HIPCHK(h, hipStreamWaitEvent(ds, first_flag_ready, 0));   // the threshold pass reads mbIsTheFirstImage (cell size) as book-keeping(k-1) left it
        hipLaunchKernelGGL(nms_threshold_kernel, dim3(16, 1, B), dim3(NMS_T), 0, ds, q, bs);
    }
    // (every workgroup rebuilds the candidate buckets in its LDS before it walks its share of the candidates: 8 of them for the latency of one
    //  stream, fewer for batch handles, whose width comes from the streams)
    static const int nb_env = ab_env("RVIO_NEIGH_BLOCKS") ? atoi(ab_env("RVIO_NEIGH_BLOCKS")) : 0;   // A/B timing
    const unsigned neigh_blocks = nb_env > 0 ? (unsigned)nb_env : (h->wide_px ? NEIGH_BLOCKS_WIDE : NEIGH_BLOCKS);
    hipLaunchKernelGGL(neigh_kernel, dim3(neigh_blocks, 1, B), dim3(NEIGH_T), NEIGH_LDS, ds, q, bs);
    hipLaunchKernelGGL(greedy_kernel, dim3(1, 1, B), dim3(GREEDY_T), GREEDY_LDS, ds, q, bs);
    if (q.sp_win > 15)        // Tracker.nMinDist >= 32: the summation grid no longer fits LDS whole
        hipLaunchKernelGGL(subpix_wide_kernel, dim3(d.F, 1, B), dim3(SPG_T), subpix_wide_lds(q.sp_win), ds, img, stride, q, src_bs, bs);
    else if (q.sp_win != SP_WIN)   // a cornerSubPix window other than the stock 7: the plain form
        hipLaunchKernelGGL(subpix_generic_kernel, dim3(d.F, 1, B), dim3(SPG_T), 0, ds, img, stride, q, src_bs, bs);
    else if (h->wide_px)
        hipLaunchKernelGGL(subpix_kernel16, dim3((d.F + 3) / 4, 1, B), dim3(64), 0, ds, img, stride, q, src_bs, bs);
    else
        hipLaunchKernelGGL(subpix_kernel, dim3(d.F, 1, B), dim3(SP_T), 0, ds, img, stride, q, src_bs, bs, (int)h->frame_no);
    HIPCHK(h, hipGetLastError());
    return RVIO_OK;
}

// ------------------------------------------------------------------ T1..T6
static int build_pyramid_dev(rvio_hip* h, const uint8_t* d_img, int stride, int b) {
    const DevCfg& d = h->dc;
    PyrDev& p = h->pyr[b];
    const size_t bs = h->slab_bytes;
    const unsigned B = (unsigned)h->batch;
    size_t src_bs = h->img_bs;       // the caller's images: instance stride of the call in progress
    bool forked = false, pyramid_done = false;
    // the whole pyramid in one launch (pyrDown chain + the copy of the frame into level 0); one workgroup per 8x8 tile of level 3
    auto launch_pyramid = [&](hipStream_t st) {
        const int w3 = (((d.W + 1) / 2 + 1) / 2 + 1) / 2, h3 = (((d.H + 1) / 2 + 1) / 2 + 1) / 2;
        PyrDev pv = p;
        const bool own = h->cfg.enable_equalizer != 0;   // d_img is the handle's equalised image: level 0 without a copy
        if (own) { h->pyr[b].img[0] = d_img; pv.img[0] = d_img; }
#ifdef RVIO_DBG_CLOCKS
        static const bool pyr_v1 = ab_env("RVIO_PYR_V1") != nullptr;   // A/B timing: the 25-tap gather form
        if (pyr_v1) hipLaunchKernelGGL(pyramid_kernel_v1, dim3((w3 + 7) / 8, (h3 + 7) / 8, B), dim3(PYR_T), 0, st, d_img, stride, pv, d.levels, own ? 0 : 1, src_bs, bs);
        else
#endif
        hipLaunchKernelGGL(pyramid_kernel, dim3((w3 + 7) / 8, (h3 + 7) / 8, B), dim3(PYR_T), 0, st, d_img, stride, pv, d.levels, own ? 0 : 1, src_bs, bs);
    };
    // Run-ahead mode: the image chain of frame k (CLAHE, detector; with the equaliser also the pyramid) rewrites buffers that book-keeping /
    // KLT of earlier frames read — equalised image k % 4, corner list and count k % 3 — so it starts behind book-keeping(k-3), with or
    // without the equaliser (the detector alone rewrites det_xy2[k % 3] / det_nout[k % 3], which bookkeep_b(k-3) reads).
    if (h->runahead && h->frame_no >= (long)kRaDepth && !(kDbgSkip & 1))
        HIPCHK(h, hipStreamWaitEvent(image_stream(h), h->evT[(h->frame_no - kRaDepth) & 3], 0));
    if (h->cfg.enable_equalizer) {   // clahe->apply(im, im), Tracker.cc:198-202
        // The equalised image of frame k doubles as level 0 of frame k's pyramid (no copy), so it lives until the KLT of frame k+1 has
        // matched against it: four buffers in rotation (like the pyramids, and three corner lists).  Slot k % 4 was last read by KLT(k-3)
        // (as the previous image) and by the detector / pyramid of frame k-4; in run-ahead mode CLAHE(k) waits for book-keeping(k-3), which
        // followed KLT(k-3) on the side stream and was the last reader of corner list k % 3.  (With three / three / two buffers the wait
        // was for book-keeping(k-2): an image chain is ~190 us long, so it started late enough to hold book-keeping(k) up.)
        h->eq_slot = (h->eq_slot + 1) % 4;
        uint8_t* eq = h->d_eq2[h->eq_slot];
        hipStream_t cs = image_stream(h);
        uint8_t* lut = h->d_lut2[h->runahead ? h->ic : h->par];
        // (lane-private 16-bit histogram columns, no LDS-atomic conflicts: every handle.  A counter sees the pixels of ONE lane column of the tile,
        // ceil(tw / 64) th of them — 1296 at 1080p —, so 16 bits hold for any image a camera delivers; the 32-bit per-wave form stays as the fall-back)
        static const bool lut1 = ab_env("RVIO_CLAHE_LUT1") != nullptr;   // A/B timing
        if (((h->cl_tw + 63) / 64) * h->cl_th <= 65535 && !lut1)
            if (h->wide_px) {
                // (eight rows of byte loads in flight per thread instead of four: the histogram of a batch is load-latency bound; 135.5 -> 136.7 k frames/s at 128 streams)
                hipLaunchKernelGGL((clahe_lut_kernel2<256, 8>), dim3(h->cl_tx * h->cl_ty, 1, B), dim3(256), 0, cs, d_img, d.W, d.H, stride, h->cl_tx, h->cl_tw, h->cl_th,
                                   h->cl_clip, h->cl_scale, lut, src_bs, bs, (int)h->frame_no);
            }
            else hipLaunchKernelGGL(clahe_lut_kernel2<1024>, dim3(h->cl_tx * h->cl_ty, 1, B), dim3(1024), 0, cs, d_img, d.W, d.H, stride, h->cl_tx, h->cl_tw, h->cl_th,
                                    h->cl_clip, h->cl_scale, lut, src_bs, bs, (int)h->frame_no);
        else
        hipLaunchKernelGGL(clahe_lut_kernel, dim3(h->cl_tx * h->cl_ty, 1, B), dim3(CLAHE_LUT_T), 0, cs, d_img, d.W, d.H, stride, h->cl_tx, h->cl_tw, h->cl_th,
                           h->cl_clip, h->cl_scale, lut, src_bs, bs, (int)h->frame_no);
        if (h->wide_px && d.W % 4 == 0 && stride % 4 == 0 && ((uintptr_t)d_img & 3) == 0 && src_bs % 4 == 0)
            hipLaunchKernelGGL(clahe_interp_kernel4, dim3((d.W / 4 + 63) / 64, (d.H + 15) / 16, B), dim3(256), 0, cs, d_img, d.W, d.H, stride, h->cl_tx, h->cl_ty,
                               1.0f / (float)h->cl_tw, 1.0f / (float)h->cl_th, lut, eq, src_bs, bs);
        else
            hipLaunchKernelGGL(clahe_interp_kernel, dim3((d.W + 63) / 64, (d.H + 3) / 4, B), dim3(256), 0, cs, d_img, d.W, d.H, stride, h->cl_tx, h->cl_ty,
                               1.0f / (float)h->cl_tw, 1.0f / (float)h->cl_th, lut, eq, src_bs, bs);
        d_img = eq; stride = d.W; src_bs = bs;
        if (h->runahead) {   // the side stream (KLT) waits for the pyramid of the equalised image; the detector follows on the image stream itself
            // the pyramid rides on the image stream: it needs nothing from the side stream's chain (KLT(k-1), RANSAC, book-keeping), which is
            // the longest serial chain of the front end — 19 us less of it; the image chain has the slack
            launch_pyramid(cs);
            pyramid_done = true;
            static const bool no_pyr_poll = ab_env("RVIO_NO_PYR_POLL") != nullptr;   // A/B timing
            // klt_kernel3 polls the chain's counter itself (no barrier packet on the side stream) — on a handle whose four streams own their hardware queues only (the first
            // live handle of the process, make_stream): 200 polling workgroups per frame in front of kernels of OTHER handles on a shared queue timed the eight-handle
            // leg of the bench out (a consumer may only spin where everything it waits for was submitted earlier to queues nobody else feeds)
            // ... and not on a handle that runs the sharded frame over a real collective, nor at long windows (6n > 96: the Cholesky factor's launches share the
            // copy queue): with the forced-sharded cfg E run of the bench two runs in six stalled for the poll's full 30 s (none in six without it) — more busy
            // queues than the command processor keeps resident, and a queue of spinning workgroups in front of the one that would release them
            if (h->dev_sync && h->private_queues && !h->queues_shared && !h->extra_queues && 6 * h->dc.nmax <= 96 && !h->wide_px && !no_pyr_poll) {
                h->pyr_signal = &h->stage_sync->pyr[h->ic];      // bumped by the detector's first launch on this queue (detect_dev), right behind the pyramid
                h->stage_tgt.pyr[h->ic]++;
                h->klt_wait = &h->stage_sync->pyr[h->ic]; h->klt_target = h->stage_tgt.pyr[h->ic];
            } else {
                HIPCHK(h, hipEventRecord(h->evC[h->ic], cs));
                HIPCHK(h, hipStreamWaitEvent(h->stream_d, h->evC[h->ic], 0));
            }
            forked = true;
        }
    }
    h->side = h->ts;
    if (h->use_det) {   // FeatureDetector::DetectWithSubPix on the image the tracker sees (Tracker.cc:207,350)
        // fork: pyramid / KLT / RANSAC go to the side stream (the image is complete on ts here), the detector stays on ts
        if (!forked) {
            HIPCHK(h, hipEventRecord(h->evD0, image_stream(h)));
            HIPCHK(h, hipStreamWaitEvent(h->stream_d, h->evD0, 0));
        }
        h->side = h->stream_d;
        // run-ahead: book-keeping(k-1) ran on the side stream; its hand-over event also says that mbIsTheFirstImage is final
        // ... until the flag has gone to 0 in every instance (it never comes back: Tracker.cc:233): the host sees that in the mirror
        // book-keeping writes (a stale 1 only keeps the wait one frame longer) and the detector chain then paces itself
        if (!h->first_cleared) {
            bool all0 = true;
            for (int i = 0; i < h->batch && all0; ++i) all0 = ((volatile int*)h->first_mirror)[i] == 0;
            h->first_cleared = all0;
        }
        const hipEvent_t flag = (h->runahead && h->frame_no >= 1 && !h->first_cleared) ? h->evT[(h->frame_no - 1) & 3] : nullptr;
        const int rc = detect_dev(h, d_img, stride, src_bs, flag);
        if (rc != RVIO_OK) return rc;
        // corners of frame k ready (the refill half of book-keeping on the side stream waits for it): a one-workgroup signal behind
        // cornerSubPix that book-keeping polls, or a stream-level event
        // (one counter per image chain: each has ONE producer queue, so "count >= the frames this chain has been handed" means THIS frame's corners)
        // (instrumented build, RVIO_DBG_ONE_CORNERS: round 3's single counter for both chains — what tests/test_gpu_flatout.py was measured against)
        static const bool one_corners = ab_env("RVIO_DBG_ONE_CORNERS") != nullptr;
        const int cix = one_corners ? 0 : h->ic;
        if (h->dev_sync) { hipLaunchKernelGGL(stage_signal_kernel, dim3(1), dim3(64), 0, image_stream(h), &h->stage_sync->corners[cix]); h->stage_tgt.corners[cix]++; }
        else if (h->runahead) HIPCHK(h, hipEventRecord(h->evD1, image_stream(h)));
    }
    if (!pyramid_done) launch_pyramid(h->side);
    HIPCHK(h, hipGetLastError());
    return RVIO_OK;
}

// everything after Tracker.cc:246; status/tracked already on the device
static int post_klt_dev(rvio_hip* h, const rvio_imu* d_imu, int m, const float* d_cand, int n_cand) {
    const size_t bs = h->slab_bytes;
    const unsigned B = (unsigned)h->batch;
    // run-ahead mode: RANSAC rides in the launch of book-keeping's hand-over half (both one workgroup, back to back on the side stream)
    static const bool no_ra_fuse = ab_env("RVIO_NO_FUSED_RANSAC") != nullptr;
    const bool fused = h->use_det && h->runahead && !no_ra_fuse;
    if (!fused)
    hipLaunchKernelGGL(ransac_kernel, dim3(1, 1, B), dim3(256), (size_t)8 * h->dc.F + 16, h->side, h->dc, h->t.n_pts, h->t.tracked, h->t.un1, h->t.un2,
                       h->t.status, d_imu, m, h->rng, h->d_info, bs, h->imu_bs);
    h->tail = h->ts;
    h->handover_evt = false;
    const unsigned long long* done = nullptr; unsigned long long done_target = 0;
    const unsigned long long* corners = nullptr; unsigned long long corners_target = 0;
    if (h->use_det) {   // the detector's corner list replaces the caller's
        const float* xy = h->det_xy2[h->dslot];
        const int* nout = h->det_nout + h->dslot;
        if (h->runahead) {   // book-keeping on the side stream, behind RANSAC: the hand-over half once filter(k-2) has let go of the tables, the refill half once the corners are there
            if (h->book_wait && !(kDbgSkip & 4)) HIPCHK(h, hipStreamWaitEvent(h->side, h->book_wait, 0));
            h->book_wait = nullptr;
            if (h->book_dev && !(kDbgSkip & 4)) { done = &h->stage_sync->aug; done_target = h->book_target; }
            h->book_dev = false;
            h->tail = h->side;
            unsigned long long* hand = nullptr;
            if (h->dev_sync) { hand = &h->stage_sync->handover; h->stage_tgt.handover++; }
            // one stream, device-side counters: RANSAC and both halves of book-keeping are ONE launch (the refill half polls the detector's counter inside it)
            static const bool no_book_fuse = ab_env("RVIO_NO_FUSED_BOOK") != nullptr;   // A/B timing
            if (fused && h->dev_sync && !no_book_fuse) {
                const int cix = ab_env("RVIO_DBG_ONE_CORNERS") ? 0 : h->ic;
                h->gate_pending = true; h->gate_target = h->stage_tgt.handover;
                hipLaunchKernelGGL(ransac_book_kernel, dim3(1, 1, B), dim3(64 * h->book_waves), h->book_lds, h->tail, h->dc, h->t, d_imu, m, h->rng, bs, h->imu_bs,
                                   done, done_target, h->meta, hand, xy, nout, &h->stage_sync->corners[cix], h->stage_tgt.corners[cix]);
                HIPCHK(h, hipGetLastError());
                return RVIO_OK;
            }
            if (fused)
                hipLaunchKernelGGL(ransac_book_a_kernel, dim3(1, 1, B), dim3(256), (size_t)8 * h->dc.F + 16, h->tail, h->dc, h->t, d_imu, m, h->rng, bs, h->imu_bs,
                                   done, done_target, h->meta, hand);
            else
            hipLaunchKernelGGL(bookkeep_a_kernel, dim3(1, 1, B), dim3(256), 0, h->tail, h->dc, h->t, bs, done, done_target, h->meta, hand);
            // the Updater's input is complete: the filter of this frame waits for THIS — the gate kernel on the filter stream polls the
            // counter the launch above bumps, and the refill half below polls the detector's; or two stream-level events
            if (h->dev_sync) { h->gate_pending = true; h->gate_target = h->stage_tgt.handover; const int cix = ab_env("RVIO_DBG_ONE_CORNERS") ? 0 : h->ic; corners = &h->stage_sync->corners[cix]; corners_target = h->stage_tgt.corners[cix]; }
            else {
                HIPCHK(h, hipEventRecord(h->evH[h->frame_no & 3], h->tail));
                h->handover_evt = true;
                HIPCHK(h, hipStreamWaitEvent(h->side, h->evD1, 0));
            }
        } else {             // join the side stream (long finished when the detector is)
            HIPCHK(h, hipEventRecord(h->evD1, h->side));
            HIPCHK(h, hipStreamWaitEvent(h->ts, h->evD1, 0));
            hipLaunchKernelGGL(bookkeep_a_kernel, dim3(1, 1, B), dim3(256), 0, h->tail, h->dc, h->t, bs, done, done_target, h->meta, (unsigned long long*)nullptr);
        }
        hipLaunchKernelGGL(bookkeep_b_kernel, dim3(1, 1, B), dim3(64 * h->book_waves), h->book_lds, h->tail, h->dc, h->t, xy, 0, nout, bs, corners, corners_target, h->meta);
    } else {
        hipLaunchKernelGGL(bookkeep_a_kernel, dim3(1), dim3(256), 0, h->ts, h->dc, h->t, (size_t)0, (const unsigned long long*)nullptr, 0ull, h->meta, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(bookkeep_b_kernel, dim3(1), dim3(64 * h->book_waves), h->book_lds, h->ts, h->dc, h->t, d_cand, n_cand, (const int*)nullptr, (size_t)0,
                           (const unsigned long long*)nullptr, 0ull, h->meta);
    }
    HIPCHK(h, hipGetLastError());
    return RVIO_OK;
}

static int track_dev_impl(rvio_hip* h, const uint8_t* d_img, int stride, const rvio_imu* d_imu, int m, const float* d_cand, int n_cand) {
    HIPCHK(h, hipSetDevice(h->device));
    if (h->piped && h->ts == h->stream) SYNC_FRONT(h);
    int rc;
    h->use_det = (d_cand == nullptr);   // no corner list from the caller: run FeatureDetector::DetectWithSubPix on the device
    if (h->use_det && (rc = detector_init(h)) != RVIO_OK) return rc;
    const bool piped_call = h->ts == h->stream_t && !h->one_stream;
    h->par = piped_call ? (int)(h->frame_no & 1) : 0;
    static const bool no_runahead = getenv("RVIO_NO_RUNAHEAD") != nullptr;   // A/B timing only
    h->runahead = piped_call && h->use_det && !no_runahead;
    // A/B timing: stream-level events instead.  Also when a counter-collecting profiler is attached (rocprofv3 --pmc exports
    // ROCPROF_COUNTER_COLLECTION): it serialises kernels across queues, and a kernel that polls a counter another queue's kernel bumps would sit
    // there until its 30 s time-out.
    static const bool no_devsync = (paranoid_bits() & PAR_NO_DEVPOLL) || ab_env("RVIO_NO_DEVFLAG") != nullptr || ab_env("RVIO_NO_DEVSYNC") != nullptr || profiler_serialises();
    h->dev_sync = h->runahead && h->batch == 1 && !no_devsync;
    h->gate_pending = false;
    h->dslot = h->runahead ? (int)(h->frame_no % 3) : h->par;
    h->ic = h->runahead ? (int)(h->frame_no % h->n_ic) : h->par;
    const int nb = (h->pyr_cur + 1) % 4;   // pyramid of the new image; pyr_cur holds mLastImage's (slot nb was last read by KLT(k-3))
    rc = build_pyramid_dev(h, d_img, stride, nb);
    if (rc != RVIO_OK) return rc;
    static const bool klt3 = ab_env("RVIO_KLT3") != nullptr;   // A/B timing
    if (h->wide_px && !klt3)    // batch handles of >= 8 instances: the throughput form, four features per wave
        hipLaunchKernelGGL(klt_kernel16, dim3((h->dc.F + 3) / 4, 1, h->batch), dim3(64), 0, h->side, h->pyr[h->pyr_cur], h->pyr[nb], h->dc.levels, h->t.n_pts, h->t.feats,
                           h->t.tracked, h->t.status, h->slab_bytes);
    else
        hipLaunchKernelGGL(klt_kernel3, dim3(h->dc.F, 1, h->batch), dim3(64), 0, h->side, h->pyr[h->pyr_cur], h->pyr[nb], h->dc.levels, h->t.n_pts, h->t.feats,
                           h->t.tracked, h->t.status, h->slab_bytes, h->klt_wait, h->klt_target, h->meta);
    h->klt_wait = nullptr;
    rc = post_klt_dev(h, d_imu, m, d_cand, std::min(n_cand, h->dc.F));
    h->pyr_cur = nb;   // im.copyTo(mLastImage), Tracker.cc:395
    return rc;
}
int rvio_hip_track_dev(rvio_hip* h, const uint8_t* d_img, int stride, const rvio_imu* d_imu, int m, const float* d_cand, int n_cand) {
    if (!h || !d_img || m < 0 || n_cand < 0) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    return track_dev_impl(h, d_img, stride, d_imu, m, d_cand, n_cand);
}

int rvio_hip_track(rvio_hip* h, const uint8_t* img, int stride, const rvio_imu* imu, int m, const float* cand_xy, int n_cand) {
    if (!h || !img || (!imu && m > 0) || m < 0 || n_cand < 0 || stride < h->dc.W) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    HIPCHK(h, hipSetDevice(h->device));
    { const int rcg = ensure_imu_capacity(h, m); if (rcg != RVIO_OK) return rcg; }
    const int nc = std::min(n_cand, h->dc.F);
    HIPCHK(h, hipMemcpy2DAsync(h->d_img, h->dc.W, img, stride, h->dc.W, h->dc.H, hipMemcpyHostToDevice, h->stream));
    if (m > 0) HIPCHK(h, hipMemcpyAsync(h->d_imu, imu, sizeof(rvio_imu) * m, hipMemcpyHostToDevice, h->stream));
    if (nc > 0 && cand_xy) HIPCHK(h, hipMemcpyAsync(h->d_cand, cand_xy, sizeof(float) * 2 * nc, hipMemcpyHostToDevice, h->stream));
    return rvio_hip_track_dev(h, h->d_img, h->dc.W, h->d_imu, m, cand_xy ? h->d_cand : nullptr, cand_xy ? nc : 0);
}

// direct-track mode: the caller supplies the KLT result
int rvio_hip_track_points(rvio_hip* h, const float* tracked_xy, const unsigned char* status, int n_pts,
                          const rvio_imu* imu, int m, const float* cand_xy, int n_cand) {
    if (!h || (!imu && m > 0) || m < 0 || n_cand < 0 || n_pts < 0 || n_pts > h->dc.F) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    HIPCHK(h, hipSetDevice(h->device));
    { const int rcg = ensure_imu_capacity(h, m); if (rcg != RVIO_OK) return rcg; }
    const int nc = std::min(n_cand, h->dc.F);
    if (n_pts > 0) {
        HIPCHK(h, hipMemcpyAsync(h->d_in_xy, tracked_xy, sizeof(float) * 2 * n_pts, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_in_st, status, n_pts, hipMemcpyHostToDevice, h->stream));
    }
    if (m > 0) HIPCHK(h, hipMemcpyAsync(h->d_imu, imu, sizeof(rvio_imu) * m, hipMemcpyHostToDevice, h->stream));
    if (nc > 0) HIPCHK(h, hipMemcpyAsync(h->d_cand, cand_xy, sizeof(float) * 2 * nc, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(load_points_kernel, dim3(8), dim3(256), 0, h->stream, h->t.n_pts, h->d_in_xy, h->d_in_st, h->t.tracked, h->t.status);   // (single instance only)
    h->use_det = false;   // no image in this mode
    h->side = h->ts;
    return post_klt_dev(h, h->d_imu, m, h->d_cand, nc);
}

int rvio_hip_get_tracks(rvio_hip* h, int32_t* n_feat, unsigned char* types, int32_t* len, float* meas) {
    if (!h) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    HIPCHK(h, hipSetDevice(h->device));
    SYNC_FRONT(h);   // image / side / tracker streams first
    const DevCfg& d = h->dc;
    int nf = 0;
    HIPCHK(h, hipMemcpyAsync(&nf, h->t.n_feat, sizeof nf, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n_feat) *n_feat = nf;
    if (nf > 0) {
        if (types) HIPCHK(h, hipMemcpyAsync(types, h->t.types, nf, hipMemcpyDeviceToHost, h->stream));
        if (len) HIPCHK(h, hipMemcpyAsync(len, h->t.len, sizeof(int) * nf, hipMemcpyDeviceToHost, h->stream));
        if (meas) HIPCHK(h, hipMemcpyAsync(meas, h->t.meas, sizeof(float) * 2 * d.max_len * nf, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RVIO_OK;
}

// mvFeatsToTrack and the length of each feature's tracking history, of one instance
int rvio_hip_get_tracker_points_at(rvio_hip* h, int instance, int32_t* n, float* xy, int32_t* hist_len) {
    if (!h || instance < 0 || instance >= h->batch) return RVIO_ERR_INVALID;
    if (!h->front_end) { h->err = "this batch handle was created without its front end"; return RVIO_ERR_UNSUPPORTED; }
    HIPCHK(h, hipSetDevice(h->device));
    SYNC_FRONT(h);   // image / side / tracker streams first
    const DevCfg& d = h->dc;
    const size_t o = (size_t)instance * h->slab_bytes;
    int np = 0;
    HIPCHK(h, hipMemcpyAsync(&np, (char*)h->t.n_pts + o, sizeof np, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n) *n = np;
    if (np > 0) {
        if (xy) HIPCHK(h, hipMemcpyAsync(xy, (char*)h->t.feats + o, sizeof(float) * 2 * np, hipMemcpyDeviceToHost, h->stream));
        if (hist_len) {
            std::vector<int> slot(np), hl(d.F);
            HIPCHK(h, hipMemcpyAsync(slot.data(), (char*)h->t.slot + o, sizeof(int) * np, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipMemcpyAsync(hl.data(), (char*)h->t.hist_len + o, sizeof(int) * d.F, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            for (int i = 0; i < np; ++i) hist_len[i] = hl[slot[i]];
        }
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RVIO_OK;
}
int rvio_hip_get_tracker_points(rvio_hip* h, int32_t* n, float* xy, int32_t* hist_len) {
    if (!h) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    return rvio_hip_get_tracker_points_at(h, 0, n, xy, hist_len);
}

// ------------------------------------------------------------------ whole frame (System.cc:253-367)
// Pieces for callers that sequence the frame themselves (staged timing, sharded updater):
// frame_plan advances nImageCountAfterInit and reports MonoVIO's two data-independent branches.
int rvio_hip_frame_plan(rvio_hip* h, int* do_update, int* do_augment) {
    if (!h) return RVIO_ERR_INVALID;
    h->img_count++;
    if (do_update) *do_update = (h->n_clones_host > h->cfg.min_track_len - 1) ? 1 : 0;   // System.cc:266
    if (do_augment) *do_augment = (h->img_count > 1) ? 1 : 0;                            // System.cc:280
    return RVIO_OK;
}
int rvio_hip_propagate_dev(rvio_hip* h, const rvio_imu* d_imu, int m) {
    if (!h || (!d_imu && m > 0) || m < 0) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    return propagate_dev(h, d_imu, m);
}

// `stream` goes on once the filter of the last frame with parity b has finished (see fin_mode)
static int wait_filter_done(rvio_hip* h, int b, hipStream_t stream) {
    if (h->fin_mode[b] == 0) HIPCHK(h, hipStreamWaitEvent(stream, h->evF[b], 0));
    else HIPCHK(h, hipStreamSynchronize(h->stream));   // (a call that left run-ahead mode: rare, the host waits)
    return RVIO_OK;
}
static int frame_tail_dev(rvio_hip* h, const rvio_imu* d_imu, int m, bool propagated = false) {
    h->img_count++;
    int rc = propagated ? RVIO_OK : propagate_dev(h, d_imu, m);
    if (rc != RVIO_OK) return rc;
    if (h->n_clones_host > h->cfg.min_track_len - 1) {   // System.cc:266
        rc = rvio_hip_update_tracked(h);
        if (rc != RVIO_OK) return rc;
    }
    return augment_compose_dev(h, h->img_count > 1);      // System.cc:280
}
// The body of System::MonoVIO after Tracker::track (System.cc:263-365) on device-resident hand-over tables, for every instance
// of the handle in ONE launch per stage: d_n_feat[B], d_types[B][Fu], d_len[B][Fu], d_meas[B][Fu][max_track_len][2] (the layout
// of rvio_tracks with max_len = max_track_len), d_imu[B][imu_stride] (imu_stride = 0: one IMU batch shared by all instances).
int rvio_hip_frame_tracks_dev(rvio_hip* h, const rvio_imu* d_imu, int imu_stride, int m, const int32_t* d_n_feat, const unsigned char* d_types,
                              const int32_t* d_len, const float* d_meas) {
    if (!h || (!d_imu && m > 0) || m < 0 || imu_stride < 0 || (imu_stride > 0 && imu_stride < m)) return RVIO_ERR_INVALID;
    if (!d_n_feat || !d_types || !d_len || !d_meas) return RVIO_ERR_INVALID;
    if (h->piped || h->in_frame) { h->err = "rvio_hip_frame_tracks_dev on a handle that runs the pipelined image path"; return RVIO_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    const DevCfg& d = h->dc;
    h->img_count++;
    const bool upd = h->n_clones_host > h->cfg.min_track_len - 1;   // System.cc:266
    // A filter-only batch: PreIntegrator::propagate (one latency-bound workgroup per instance, two per CU) runs on the handle's second stream
    // BESIDE the per-feature stage of the update — U1-U5 and the share reduction read only the clone states and P[24:,24:], which propagation does
    // not touch (the reason feat_prop_kernel may fuse them for one stream) — and joins in front of the solve, which needs the propagated rows.
    static const bool no_overlap = ab_env("RVIO_NO_PROP_OVERLAP") != nullptr;   // A/B timing
    const bool overlap = upd && h->batch > 1 && !h->front_end && !h->one_stream && !no_overlap;
    int rc;
    if (overlap) {
        HIPCHK(h, hipEventRecord(h->evD0, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream_t, h->evD0, 0));
        rc = propagate_dev(h, d_imu, m, (size_t)imu_stride * sizeof(rvio_imu), h->stream_t);
        HIPCHK(h, hipEventRecord(h->evD1, h->stream_t));
    } else rc = propagate_dev(h, d_imu, m, (size_t)imu_stride * sizeof(rvio_imu));
    if (rc != RVIO_OK) return rc;
    if (upd) {
        const TrackerDev t0 = h->t;
        const BatchIn b0 = h->bin;
        h->t.n_feat = const_cast<int*>(d_n_feat); h->t.types = const_cast<unsigned char*>(d_types);
        h->t.len = const_cast<int*>(d_len); h->t.meas = const_cast<float*>(d_meas);
        h->bin = {0, sizeof(int32_t), (size_t)d.Fu, sizeof(int32_t) * (size_t)d.Fu, sizeof(float) * 2 * (size_t)d.Fu * d.max_len};
        rc = update_local_dev(h, 0, 1, true);
        if (overlap) HIPCHK(h, hipStreamWaitEvent(h->stream, h->evD1, 0));
        if (rc == RVIO_OK) rc = update_global_dev(h, h->block, 1, true);
        h->t = t0; h->bin = b0;
        if (rc != RVIO_OK) return rc;
    }
    return augment_compose_dev(h, h->img_count > 1);      // System.cc:280
}

// Pipelined: the tracker (pyramid, KLT, RANSAC, book-keeping) never reads the filter state, so frame k's front end runs
// on its own stream while frame k-1's propagate/update/augment still occupy the filter stream.  The Tracker -> Updater
// hand-over is double-buffered; two events per buffer order (a) update(k) after track(k), (b) track(k+2) after update(k).
// PreIntegrator::propagate needs nothing from the tracker either: it is enqueued first and runs beside the front end.
static int frame_dev_impl(rvio_hip* h, const uint8_t* d_img, int stride, const rvio_imu* d_imu, int m, const float* d_cand, int n_cand, bool staged,
                          bool begin_only = false, bool defer_propagate = false) {   // defer_propagate: the caller runs update_local_dev itself right behind (the sharded frame)
    if (h->in_frame) { h->err = "rvio_hip_frame_begin_dev without rvio_hip_frame_end"; return RVIO_ERR_INVALID; }
    const int b = (int)(h->frame_no & 1);                   // parity: the staging of rvio_hip_frame
    const int hb = (int)(h->frame_no % rvio_hip::kHand);    // hand-over table of this frame
    h->t.n_feat = h->tout[hb].n_feat; h->t.types = h->tout[hb].types; h->t.len = h->tout[hb].len; h->t.meas = h->tout[hb].meas;
    static const bool no_ra = getenv("RVIO_NO_RUNAHEAD") != nullptr;
    const bool ra = !d_cand && !h->one_stream && !no_ra;   // run-ahead mode (track_dev_impl): book-keeping runs on the side stream
    if (h->frame_no >= rvio_hip::kHand) {   // the filter of frame k - kHand has consumed this hand-over buffer: only the stream that runs book-keeping has to know.
        // (In run-ahead mode the image chains — CLAHE, detector — never touch the hand-over: making them wait here tied image(k) to
        // filter(k-2) and with it the frame period to image chain + filter chain over two frames.)
        // Run-ahead: of the side stream's chain (pyramid, KLT, RANSAC, book-keeping) only book-keeping writes the hand-over, so the wait
        // goes right in front of it (post_klt_dev) — at the head of the frame it tied KLT(k) to filter(k-2) and made
        // [side chain + filter chain] the period of two frames.
        if (!ra) { int rcw = wait_filter_done(h, hb, h->stream_t); if (rcw == RVIO_OK) rcw = wait_filter_done(h, hb, h->stream_d); if (rcw != RVIO_OK) return rcw; }
        else if (h->fin_mode[hb] == 0) h->book_wait = h->evF[hb];
        else { h->book_dev = true; h->book_target = h->fin_target[hb]; }
    } else if (!h->piped) HIPCHK(h, hipStreamSynchronize(h->stream));   // first pipelined frame: everything enqueued so far is done
    h->piped = true;
    if (m < 0) return RVIO_ERR_INVALID;
    static const bool dbg_host = ab_env("RVIO_DBG_HOST") != nullptr;
    static double acc[5] = {0, 0, 0, 0, 0}; static long nacc = 0;
    auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = dbg_host ? now() : 0;
    if (staged) {   // the IMU batch was copied on the tracker stream (run-ahead: the side stream): propagate (filter stream) and RANSAC need it
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->evIn[b], 0));
        HIPCHK(h, hipStreamWaitEvent(h->stream_d, h->evIn[b], 0));
    }
    // propagate: with an update in this frame (and nobody sequencing the update from outside) it rides in the per-feature launch,
    // otherwise it goes to the filter stream right behind augment/compose(k-1)
    const bool fuse = h->fuse_ok && (!begin_only || defer_propagate) && h->n_clones_host > h->cfg.min_track_len - 1;
    int rc = RVIO_OK;
    h->fuse_m = -1;
    if (!fuse) rc = propagate_dev(h, d_imu, m, h->imu_bs);
    if (rc != RVIO_OK) return rc;
    const double t1 = dbg_host ? now() : 0;
    h->ts = h->stream_t;
    rc = track_dev_impl(h, d_img, stride, d_imu, m, d_cand, n_cand);
    h->ts = h->stream;
    if (rc != RVIO_OK) return rc;
    const double t2 = dbg_host ? now() : 0;
    HIPCHK(h, hipEventRecord(h->evT[h->frame_no & 3], h->tail));      // behind book-keeping, on the stream that ran it
    // the filter needs the hand-over, not the refill: in run-ahead mode it waits for the first half of book-keeping only
    if (h->gate_pending) {
        if (!(kDbgSkip & 16)) hipLaunchKernelGGL(stage_gate_kernel, dim3(1), dim3(64), 0, h->stream, &h->stage_sync->handover, h->gate_target, h->meta, h->t.n_feat, (int)h->frame_no);
        h->gate_pending = false;
    } else if (!(kDbgSkip & 16)) HIPCHK(h, hipStreamWaitEvent(h->stream, h->handover_evt ? h->evH[h->frame_no & 3] : h->evT[h->frame_no & 3], 0));
    const double t3 = dbg_host ? now() : 0;
    if (begin_only) {   // the caller sequences update / augment itself, then rvio_hip_frame_end
        h->in_frame = true;
        if (fuse) { h->fuse_imu = d_imu; h->fuse_m = m; }   // (the sharded frame: consumed by its update_local_dev, same condition)
        return RVIO_OK;
    }
    if (fuse) { h->fuse_imu = d_imu; h->fuse_m = m; }   // consumed by the per-feature launch of this frame's update (same condition: it runs)
    if (fuse) { h->time_imu = d_imu; h->time_m = m; }
    rc = frame_tail_dev(h, d_imu, m, /*propagated=*/true);
    h->fuse_m = -1;
    const double t4 = dbg_host ? now() : 0;
    // the filter of this frame is finished when ... single instance in run-ahead mode: its last kernel has bumped the device-side counter
    // (book-keeping of frame k+2 polls it); otherwise an event behind it
    static const bool no_devflag = (paranoid_bits() & PAR_NO_DEVPOLL) || ab_env("RVIO_NO_DEVFLAG") != nullptr || profiler_serialises();
    if (ra && h->batch == 1 && !no_devflag) { h->fin_mode[hb] = 1; h->fin_target[hb] = h->stage_tgt.aug; }
    else { if (!(kDbgSkip & 8)) HIPCHK(h, hipEventRecord(h->evF[hb], h->stream)); h->fin_mode[hb] = 0; }
    if (dbg_host) {
        const double t5 = now();
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4;
        if (++nacc % 100 == 0) { std::fprintf(stderr, "host us/frame: propagate %.1f track %.1f evT %.1f tail %.1f evF %.1f\n", acc[0] / 100, acc[1] / 100, acc[2] / 100, acc[3] / 100, acc[4] / 100); for (double& a : acc) a = 0; }
    }
    h->frame_no++;
    if (rc == RVIO_OK && (paranoid_bits() & PAR_DRAIN)) rc = drain_all(h);
    return rc;
}
int rvio_hip_frame_dev(rvio_hip* h, const uint8_t* d_img, int stride, const rvio_imu* d_imu, int m, const float* d_cand, int n_cand) {
    if (!h || !d_img) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    HIPCHK(h, hipSetDevice(h->device));
    return frame_dev_impl(h, d_img, stride, d_imu, m, d_cand, n_cand, false);
}
// One camera frame of EVERY instance of a batch handle created with its front end: d_imgs[B] (instance stride img_stride bytes,
// row stride `stride`), d_imu[B][imu_stride] (0: shared).  Same pipelined body as rvio_hip_frame_dev, every launch with gridDim.z = B;
// corners always come from the device detector.
int rvio_hip_frame_batch_dev(rvio_hip* h, const uint8_t* d_imgs, int stride, size_t img_stride, const rvio_imu* d_imu, int imu_stride, int m) {
    if (!h || !d_imgs || (!d_imu && m > 0) || imu_stride < 0 || (imu_stride > 0 && imu_stride < m) || stride < h->dc.W) return RVIO_ERR_INVALID;
    if (!h->front_end) { h->err = "this batch handle was created without its front end"; return RVIO_ERR_UNSUPPORTED; }
    if (h->batch > 1 && img_stride < (size_t)stride * h->dc.H) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    h->img_bs = img_stride; h->imu_bs = (size_t)imu_stride * sizeof(rvio_imu);
    const int rc = frame_dev_impl(h, d_imgs, stride, d_imu, m, nullptr, 0, false);
    h->img_bs = 0; h->imu_bs = 0;
    return rc;
}
// The pipelined frame split open for callers that sequence the update themselves (the feature-sharded updater):
//   frame_begin_dev   propagate on the filter stream, the front end on its streams, filter stream ordered after the front end
//   ... rvio_hip_frame_plan, rvio_hip_update_local / collective / rvio_hip_update_global (or update_tracked), rvio_hip_augment_compose,
//       all on the filter stream (rvio_hip_stream) ...
//   frame_end         closes the frame (hand-over buffer released for frame k+2)
int rvio_hip_frame_begin_dev(rvio_hip* h, const uint8_t* d_img, int stride, const rvio_imu* d_imu, int m, const float* d_cand, int n_cand) {
    if (!h || !d_img) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    HIPCHK(h, hipSetDevice(h->device));
    return frame_dev_impl(h, d_img, stride, d_imu, m, d_cand, n_cand, false, /*begin_only=*/true);
}
int rvio_hip_frame_end(rvio_hip* h) {
    if (!h || !h->in_frame) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventRecord(h->evF[h->frame_no % rvio_hip::kHand], h->stream));
    h->fin_mode[h->frame_no % rvio_hip::kHand] = 0;
    h->frame_no++;
    h->in_frame = false;
    if (paranoid_bits() & PAR_DRAIN) return drain_all(h);
    return RVIO_OK;
}
// RCCL's ncclAllGather, resolved at run time from the RCCL instance the process has ALREADY loaded (the communicator the caller hands over
// belongs to it: torch ships its own librccl.so) and only then from the system's.  No link-time dependency: a single-GPU user never loads RCCL.
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static nccl_allgather_fn resolve_allgather(std::string* why) {
    static nccl_allgather_fn fn = nullptr;
    if (fn) return fn;
    void* lib = nullptr;
    for (const char* name : {"librccl.so", "librccl.so.1"}) if ((lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!lib) {   // torch's copy is loaded by path, not by soname lookup: look for a loaded object whose name ends in librccl.so*
        fn = (nccl_allgather_fn)dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (fn) return fn;
    }
    if (!lib) for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (lib) fn = (nccl_allgather_fn)dlsym(lib, "ncclAllGather");
    if (!fn && why) *why = "ncclAllGather not found (no librccl.so loaded or loadable)";
    return fn;
}
// One pipelined frame with the feature-sharded updater behind ONE call: the front end and propagate replicated, U1-U5 + the
// share reduction on the features f % world == rank, ONE ncclAllGather of the [S2 | S1] blocks enqueued on the filter stream between the two
// kernels it separates (plain stream order: no helper stream, no event, no host synchronisation), the replicated global stage, augmentation /
// composition.  comm: the caller's ncclComm_t; NULL only with world == 1 (the collective is skipped).  allgather: NULL = resolve RCCL's
// ncclAllGather from the loaded process image; a caller may hand in the entry point itself (same signature).
int rvio_hip_frame_sharded_dev(rvio_hip* h, const uint8_t* d_img, int stride, const rvio_imu* d_imu, int m, const float* d_cand, int n_cand,
                               int rank, int world, void* comm, void* allgather) {
    if (!h || !d_img || world < 1 || rank < 0 || rank >= world || (!comm && world > 1)) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    HIPCHK(h, hipSetDevice(h->device));
    if (comm) h->extra_queues = true;   // the collective brings queues of its own: more than the four this handle's chains own (see the pyramid poll, build_pyramid_dev)
    const size_t nblk_max = (size_t)shard_payload_doubles(6 * h->dc.nmax, h->dc.max_len);    // the full window's payload: what the receive buffer is sized for
    nccl_allgather_fn ag = (nccl_allgather_fn)allgather;
    if (comm && !ag && !(ag = resolve_allgather(&h->err))) return RVIO_ERR_UNSUPPORTED;
    if (comm && h->gathered_world < world) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        void* q = nullptr;
        HIPCHK(h, hipMalloc(&q, sizeof(double) * nblk_max * (size_t)world));
        h->allocs.push_back(q);
        h->gathered = (double*)q; h->gathered_world = world;
    }
    int rc = frame_dev_impl(h, d_img, stride, d_imu, m, d_cand, n_cand, false, /*begin_only=*/true, /*defer_propagate=*/true);
    if (rc != RVIO_OK) return rc;
    h->img_count++;
    if (h->n_clones_host > h->cfg.min_track_len - 1) {   // System.cc:266
        rc = update_local_dev(h, rank, world, false);
        const double* blocks = h->block;
        if (rc == RVIO_OK && comm) {
            const size_t nblk = (size_t)shard_payload_doubles(6 * h->n_clones_host, h->dc.max_len);     // (grows with the window: 8 + 256 doubles per carried tile)
            const int nrc = ag(h->block, h->gathered, nblk, /*ncclFloat64*/ 8, comm, h->stream);
            if (nrc != 0) { h->err = "ncclAllGather failed (ncclResult_t " + std::to_string(nrc) + ")"; rc = RVIO_ERR_NO_DEVICE; }
            blocks = h->gathered;
        }
        if (rc == RVIO_OK) rc = update_global_dev(h, blocks, world, false);
    }
    if (rc == RVIO_OK) rc = augment_compose_dev(h, h->img_count > 1);      // System.cc:280
    const int rce = rvio_hip_frame_end(h);
    return rc != RVIO_OK ? rc : rce;
}
// The same body fed from HOST buffers — what System::MonoVIO holds at System.cc:253 (a cv::Mat and the IMU list).
// The three H2D copies go to the tracker stream into staging buffers double-buffered by frame parity, so they overlap
// the previous frame's filter work like the tracker kernels do.
int rvio_hip_frame(rvio_hip* h, const uint8_t* img, int stride, const rvio_imu* imu, int m, const float* cand_xy, int n_cand) {
    if (!h || !img || (!imu && m > 0) || m < 0 || n_cand < 0 || stride < h->dc.W) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    HIPCHK(h, hipSetDevice(h->device));
    { const int rcg = ensure_imu_capacity(h, m); if (rcg != RVIO_OK) return rcg; }
    const int nc = cand_xy ? std::min(n_cand, h->dc.F) : 0;   // cand_xy == NULL: device detector
    if (!h->hb_img[0])
        for (int k = 0; k < 2; ++k) {
            DALLOC(h, h->hb_img[k], (size_t)h->dc.W * h->dc.H);
            DALLOC(h, h->hb_imu[k], (size_t)h->imu_cap);
            if (k == 1) for (int q = 2; q <= rvio_hip::kHand; ++q) DALLOC(h, h->hb_imu[q], (size_t)h->imu_cap);
            DALLOC(h, h->hb_cand[k], (size_t)2 * h->dc.F);
            HIPCHK(h, hipStreamSynchronize(h->stream));   // DALLOC clears on the filter stream
        }
    const size_t npx = (size_t)h->dc.W * h->dc.H;
    if (!h->pin[0]) {
        h->pin_img = 0; h->pin_imu = (npx + 255) & ~(size_t)255;
        const size_t pin_cand = h->pin_imu + ((sizeof(rvio_imu) * (size_t)h->imu_cap + 255) & ~(size_t)255);
        h->pin_bytes = pin_cand + sizeof(float) * 2 * h->dc.F;
        for (int k = 0; k < rvio_hip::kPin; ++k) {
            HIPCHK(h, hipHostMalloc((void**)&h->pin[k], h->pin_bytes, hipHostMallocDefault));
            if (!h->evPin[k]) HIPCHK(h, hipEventCreateWithFlags(&h->evPin[k], kEvFlags));     // (the ring is laid out again after ensure_imu_capacity)
            if (!h->evPin2[k]) HIPCHK(h, hipEventCreateWithFlags(&h->evPin2[k], kEvFlags));
        }
    }
    const int b = (int)(h->frame_no & 1);
    const int ps = (int)(h->frame_no % rvio_hip::kPin);
    uint8_t* pp = h->pin[ps];
    const size_t pin_cand = h->pin_imu + ((sizeof(rvio_imu) * (size_t)h->imu_cap + 255) & ~(size_t)255);
    HIPCHK(h, hipEventSynchronize(h->evPin[ps]));   // the copies issued from this slot three frames ago are done (no-op before its first use)
    HIPCHK(h, hipEventSynchronize(h->evPin2[ps]));
    if (stride == h->dc.W) std::memcpy(pp, img, npx);   // (a continuous cv::Mat: one copy)
    else for (int y = 0; y < h->dc.H; ++y) std::memcpy(pp + (size_t)y * h->dc.W, img + (size_t)y * stride, (size_t)h->dc.W);
    if (m > 0) std::memcpy(pp + h->pin_imu, imu, sizeof(rvio_imu) * m);
    if (nc > 0) std::memcpy(pp + pin_cand, cand_xy, sizeof(float) * 2 * nc);
    static const bool no_runahead = getenv("RVIO_NO_RUNAHEAD") != nullptr;
    const bool ra = !cand_xy && !h->one_stream && !no_runahead;
    int imu_slot = b;
    if (ra) {
        // run-ahead mode.  The IMU batch goes to the SIDE stream (RANSAC runs there; propagate on the filter stream waits for evIn): it
        // has to wait for the filter of frame k-2 (the last reader of hb_imu[b]), and that wait must not sit in front of an image chain.
        // kHand + 1 IMU slots in rotation: slot k % (kHand + 1) was last read by filter(k - kHand - 1) / RANSAC(k - kHand - 1), and this copy follows
        // book-keeping(k-1) on the side stream, which waited for filter(k - 1 - kHand) — no wait of its own (one here would again put a filter in front of KLT(k)).
        imu_slot = (int)(h->frame_no % (rvio_hip::kHand + 1));
        if (!h->last_ra) for (int i = 0; i < rvio_hip::kHand && i < h->frame_no; ++i) { const int rcw = wait_filter_done(h, i, h->stream_d); if (rcw != RVIO_OK) return rcw; }
        if (h->frame_no < 2 && !h->piped) HIPCHK(h, hipStreamSynchronize(h->stream));
        if (m > 0) HIPCHK(h, hipMemcpyAsync(h->hb_imu[imu_slot], pp + h->pin_imu, sizeof(rvio_imu) * m, hipMemcpyHostToDevice, h->stream_d));
        HIPCHK(h, hipEventRecord(h->evIn[b], h->stream_d));
        HIPCHK(h, hipEventRecord(h->evPin[ps], h->stream_d));
        // The image goes to the stream of this frame's image chain (image_stream: tracker stream / fourth stream by parity).  hb_img[b]
        // was last read by frame k-2: its CLAHE / detector (same stream, earlier) and — without the equaliser — its pyramid on the side
        // stream, which book-keeping(k-2) followed.
        hipStream_t is = image_stream_of(h, (int)(h->frame_no % h->n_ic));
        if (h->frame_no >= 2) HIPCHK(h, hipStreamWaitEvent(is, h->evT[(h->frame_no - 2) & 3], 0));
        HIPCHK(h, hipMemcpyAsync(h->hb_img[b], pp, npx, hipMemcpyHostToDevice, is));
        HIPCHK(h, hipEventRecord(h->evPin2[ps], is));
    } else {
        if (h->frame_no >= 2) { const int rcw = wait_filter_done(h, (int)((h->frame_no - 2) % rvio_hip::kHand), h->stream_t); if (rcw != RVIO_OK) return rcw; }   // filter(k-2) has consumed hb_imu[b]
        else if (!h->piped) HIPCHK(h, hipStreamSynchronize(h->stream));
        if (m > 0) HIPCHK(h, hipMemcpyAsync(h->hb_imu[b], pp + h->pin_imu, sizeof(rvio_imu) * m, hipMemcpyHostToDevice, h->stream_t));
        HIPCHK(h, hipEventRecord(h->evIn[b], h->stream_t));                                  // propagate (filter stream) only needs the IMU batch
        HIPCHK(h, hipMemcpyAsync(h->hb_img[b], pp, npx, hipMemcpyHostToDevice, h->stream_t));
        if (nc > 0) HIPCHK(h, hipMemcpyAsync(h->hb_cand[b], pp + pin_cand, sizeof(float) * 2 * nc, hipMemcpyHostToDevice, h->stream_t));
        HIPCHK(h, hipEventRecord(h->evPin[ps], h->stream_t));
    }
    h->last_ra = ra;
    if (paranoid_bits() & PAR_SYNC_COPIES) {   // the staging copies have landed before anything that reads them is enqueued
        HIPCHK(h, hipStreamSynchronize(h->stream_d));
        HIPCHK(h, hipStreamSynchronize(h->stream_t));
        if (h->stream_c) HIPCHK(h, hipStreamSynchronize(h->stream_c));
        if (h->stream_e) HIPCHK(h, hipStreamSynchronize(h->stream_e));
    }
    return frame_dev_impl(h, h->hb_img[b], h->dc.W, h->hb_imu[imu_slot], m, cand_xy ? h->hb_cand[b] : nullptr, nc, true);
}
// direct-track variant of the whole frame (host inputs)
int rvio_hip_frame_points(rvio_hip* h, const float* tracked_xy, const unsigned char* status, int n_pts,
                          const rvio_imu* imu, int m, const float* cand_xy, int n_cand) {
    int rc = rvio_hip_track_points(h, tracked_xy, status, n_pts, imu, m, cand_xy, n_cand);
    if (rc != RVIO_OK) return rc;
    return frame_tail_dev(h, h->d_imu, m);
}

int rvio_hip_get_frame_info(rvio_hip* h, rvio_frame_info* info) {
    if (!h || !info) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    SYNC_FRONT(h);   // image / side / tracker streams first
    FilterMeta m;
    double lit_rank = -1.0;
    HIPCHK(h, hipMemcpyAsync(info, h->d_info, sizeof *info, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&m, h->meta, sizeof m, hipMemcpyDeviceToHost, h->stream));
    if (h->last_Ab) HIPCHK(h, hipMemcpyAsync(&lit_rank, h->last_Ab + (size_t)h->dc.ldh * (h->dc.ldh - 1) + 5, sizeof lit_rank, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    info->n_clones = h->n_clones_host; info->n_feat_accepted = m.n_good; info->n_rows = m.n_rows; info->updated = m.updated;
    info->reserved[0] = m.err;
    info->reserved[1] = (m.updated && lit_rank >= 0) ? (int)lit_rank : -1;
    info->rank_truncated_at = m.updated ? m.trunc_at : -1;
    if (m.err & 4) { h->err = "a device-side stage counter timed out (filter -> book-keeping): the frame sequence is invalid, re-initialise"; return RVIO_ERR_STATE; }
    return RVIO_OK;
}
int rvio_hip_get_pose(rvio_hip* h, double p[3], double q[4]) {
    if (!h) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    double buf[8];
    HIPCHK(h, hipMemcpyAsync(buf, h->d_pose, sizeof buf, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (p) for (int i = 0; i < 3; ++i) p[i] = buf[i];
    if (q) for (int i = 0; i < 4; ++i) q[i] = buf[3 + i];
    return RVIO_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ diagnostics for parity tests
extern "C" {
// output of the device detector for the most recent image: corner count, refined corners, goodFeaturesToTrack corners
// before cornerSubPix, and the min-eigenvalue map (each pointer may be NULL)
int rvio_hip_get_corners(rvio_hip* h, int32_t* n, float* xy, float* raw_xy, float* eig) {
    if (!h) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    if (!h->det_ready) { h->err = "the device detector has not run (pass a NULL corner list to track/frame)"; return RVIO_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    SYNC_FRONT(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int cnt = 0;
    HIPCHK(h, hipMemcpyAsync(&cnt, h->det_nout + h->dslot, sizeof cnt, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n) *n = cnt;
    if (xy && cnt > 0) HIPCHK(h, hipMemcpyAsync(xy, h->det_xy2[h->dslot], sizeof(float) * 2 * cnt, hipMemcpyDeviceToHost, h->stream));
    const DetDev& ds_ = h->dets[h->det_set_last];
    if (raw_xy && cnt > 0) HIPCHK(h, hipMemcpyAsync(raw_xy, ds_.raw_xy, sizeof(float) * 2 * cnt, hipMemcpyDeviceToHost, h->stream));
    if (eig) {
        // the pipeline no longer stores the min-eigenvalue map (mineig_nms_kernel keeps it in LDS): recomputed on demand from the image the
        // detector saw — level 0 of the current pyramid — by the map-only kernel (same arithmetic); its side effects on the detector's
        // per-frame scratch are undone (the image maximum returns to its rest value)
        const DevCfg& d = h->dc;
        if (!h->eig_map) { const bool sm = h->slab_mode; h->slab_mode = false; const int rc = dalloc(h, &h->eig_map, (size_t)d.W * d.H); h->slab_mode = sm; if (rc != RVIO_OK) return rc; }
        DetDev dm = ds_;
        dm.eig = h->eig_map;
        hipLaunchKernelGGL(mineig_kernel, dim3((d.W + DET_TW - 1) / DET_TW, (d.H + DET_TH - 1) / DET_TH, 1), dim3(DET_T), 0, h->stream, h->pyr[h->pyr_cur].img[0], d.W, dm, (size_t)0,
                           (size_t)0, 0);
        const int rest = (int)0x80000000;
        HIPCHK(h, hipMemcpyAsync(ds_.maxkey, &rest, sizeof rest, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(eig, h->eig_map, sizeof(float) * d.W * d.H, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RVIO_OK;
}
// pyramid level `level` of the most recent image: u8 image (w*h) and int16 (dx,dy) derivative (w*h*2)
int rvio_hip_debug_pyramid(rvio_hip* h, int level, int32_t* w, int32_t* hgt, uint8_t* img, int16_t* dxy) {
    if (!h || level < 0 || level >= h->dc.levels) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    HIPCHK(h, hipSetDevice(h->device));
    SYNC_FRONT(h);   // image / side / tracker streams first
    const PyrDev& p = h->pyr[h->pyr_cur];
    if (w) *w = p.w[level];
    if (hgt) *hgt = p.h[level];
    const size_t n = (size_t)p.w[level] * p.h[level];
    if (img) HIPCHK(h, hipMemcpyAsync(img, p.img[level], n, hipMemcpyDeviceToHost, h->stream));
    if (dxy) {   // calcSharrDeriv of that level, computed on demand (the pipeline keeps no derivative image)
        int* tmp = nullptr;
        HIPCHK(h, hipMalloc((void**)&tmp, n * sizeof(int)));
        hipLaunchKernelGGL(scharr_debug_kernel, dim3((p.w[level] + 63) / 64, (p.h[level] + 3) / 4), dim3(256), 0, h->stream, p.img[level], p.w[level], p.h[level], tmp);
        HIPCHK(h, hipMemcpyAsync(dxy, tmp, n * 2 * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        hipFree(tmp);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RVIO_OK;
}
// raw vFeatsTracked of the last track() call (n entries = mnFeatsToTrack that entered the call)
int rvio_hip_debug_tracked(rvio_hip* h, int n, float* xy, float* un_xy) {
    if (!h || n < 0 || n > h->dc.F) return RVIO_ERR_INVALID;
    FRONT_END_ONLY(h);
    HIPCHK(h, hipSetDevice(h->device));
    SYNC_FRONT(h);   // image / side / tracker streams first
    if (n > 0 && xy) HIPCHK(h, hipMemcpyAsync(xy, h->t.tracked, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, h->stream));
    if (n > 0 && un_xy) HIPCHK(h, hipMemcpyAsync(un_xy, h->t.un2, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RVIO_OK;
}
// Average device time (microseconds, HIP events on the handle's stream) of `iters` back-to-back launches of one
// hot kernel on the operands left behind by the last frame: which = 0 -> solve kernel (W = T^-1 + injection),
// 1 -> klt_kernel3 on the two resident pyramids, 2 -> feat_build_kernel.  Outputs land in scratch / are idempotent.
int rvio_hip_debug_time_kernel(rvio_hip* h, int which, int iters, float* avg_us) {
    if (!h || !avg_us || iters < 1) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    { const int rcd = drain_all(h); if (rcd != RVIO_OK) return rcd; }   // (every queue of the handle: the run-ahead image chains too)
    const DevCfg& d = h->dc;
    const int n = h->n_clones_host;
    hipEvent_t e0, e1;
    double *bx = nullptr, *bP = nullptr;
    if (which == 8) {   // (the fused launch propagates in place: the state is put aside and restored behind the timed launches)
        if (h->batch > 1 || !h->fuse_ok || !h->time_imu || n < 1) return RVIO_ERR_UNSUPPORTED;
        HIPCHK(h, hipMalloc(&bx, sizeof(double) * d.xdmax)); HIPCHK(h, hipMalloc(&bP, sizeof(double) * d.dmax * d.dmax));
        HIPCHK(h, hipMemcpyAsync(bx, h->x[h->cur], sizeof(double) * d.xdmax, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(bP, h->P[h->cur], sizeof(double) * d.dmax * d.dmax, hipMemcpyDeviceToDevice, h->stream));
    }
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    HIPCHK(h, hipEventRecord(e0, h->stream));
    for (int it = 0; it < iters; ++it) {
        if (which == 8) {
            // feat_prop_kernel exactly as the pipelined frame launches it: the per-feature workgroups of the last hand-over table + PreIntegrator::propagate on the
            // IMU batch of the last frame (the caller's device buffer must still be alive) + at 6n <= 96 the Cholesky role
            const bool chol = h->solve9_nt && h->solve9_nt <= 6;
            hipLaunchKernelGGL(feat_prop_kernel, dim3(d.Fu + 1 + (chol ? 1 : 0)), dim3(256), h->fprop_lds, h->stream, d, n, h->x[h->cur], h->P[h->cur],
                               h->t.n_feat, h->t.types, h->t.len, h->t.meas, h->partial, h->nrows, h->acc, h->ndof, h->gamma, h->pfinv, h->tm_global, h->bin,
                               h->meta, h->time_imu, h->time_m, chol ? h->S9scr : (double*)nullptr, h->solve9_nt, 0, 1, h->lit_rows);
        } else if (which == 9) {   // the detector's selection kernel (one workgroup: priority-ordered maximal independent set) on the candidates of the last detector call
            if (h->batch > 1 || !h->det_ready) return RVIO_ERR_UNSUPPORTED;
            const DetDev q = [&] { DetDev v = h->dets[h->det_set_last]; v.xy = h->det_xy2[h->dslot]; v.n_out = h->det_nout + h->dslot; return v; }();
            hipLaunchKernelGGL(greedy_kernel, dim3(1, 1, 1), dim3(GREEDY_T), GREEDY_LDS, h->stream, q, h->slab_bytes);
        } else
        if (which == 0) {
            h->chol_ready = h->solve9_nt && (h->solve9_nt <= 6 || h->stream_l) && !ab_env("RVIO_S9_FULL");   // (time what the chain sees: the Cholesky factor rides in the per-feature launch / runs on its own queue; the slab holds the factor of the last update)
            launch_solve(h, n, h->block, /*defer_dx=*/true);   // as the frame's update launches it: dx = Pc y and the state injection are roles of the Joseph launch behind it
            h->dx_pending = false;
        } else if (which == 1) {
            // KLT as the frame ran it cannot be repeated (book-keeping has moved the features to where they were tracked): match the CURRENT
            // image back onto the PREVIOUS one from the current feature positions instead — the same displacement magnitudes, reversed, the
            // refilled corners included (they exist in the previous image too).  Outputs land in t.tracked / t.status (scratch between frames).
            if (h->batch > 1) return RVIO_ERR_UNSUPPORTED;
            hipLaunchKernelGGL(klt_kernel3, dim3(d.F), dim3(64), 0, h->stream, h->pyr[h->pyr_cur], h->pyr[(h->pyr_cur + 3) % 4], d.levels, h->t.n_pts, h->t.feats,
                               h->t.tracked, h->t.status, (size_t)0, (const unsigned long long*)nullptr, 0ull, h->meta);
        } else if (which == 2) {
            if (h->batch == 1)
            hipLaunchKernelGGL(feat_build_kernel<16>, dim3(d.Fu, 1, h->batch), dim3(h->feat_threads), h->feat_lds, h->stream, d, n, h->x[h->cur], h->P[h->cur],
                               h->t.n_feat, h->t.types, h->t.len, h->t.meas, 0, 1, h->partial, h->nrows, h->acc, h->ndof, h->gamma, h->pfinv, h->tm_global,
                               h->slab_bytes, h->bin, h->meta, (const double*)nullptr, (const int*)nullptr, h->lit_rows);
            else
            hipLaunchKernelGGL(feat_build_kernel<4>, dim3(d.Fu, 1, h->batch), dim3(h->feat_threads), h->feat_lds, h->stream, d, n, h->x[h->cur], h->P[h->cur],
                               h->t.n_feat, h->t.types, h->t.len, h->t.meas, 0, 1, h->partial, h->nrows, h->acc, h->ndof, h->gamma, h->pfinv, h->tm_global,
                               h->slab_bytes, h->bin, h->meta, (const double*)h->gpose, (const int*)h->gvalid, h->lit_rows);
        } else if (which == 3 && h->batch >= 128 && h->gram_batch_lds) {
            launch_gram_batch(h, n);
        } else if (which == 3) {   // reduction of the per-feature shares + rank truncation (reads `partial`, rewrites `block`: idempotent)
            hipLaunchKernelGGL(gram_reduce_kernel, dim3(std::max(1, std::min(1024, (6 * n * d.ldh + (h->batch == 1 ? 63 : 255)) / (h->batch == 1 ? 64 : 256))), 1, h->batch), dim3(256), h->trunc_lds, h->stream, d, n,
                               h->partial, h->nrows, h->t.types, h->t.len, h->block, h->gram_cnt, 1, (h->batch == 1) ? 1 : 0, h->slab_bytes, h->bin, lit_args(h, h->trunc_lds));
        } else if (which == 4 || which == 5) {   // U, G, P1 strips / the Joseph form on the operands of the last update, in the form the handle launches
            launch_ug_final(h, n, h->block, h->P[h->cur ^ 1], which == 4, which == 5);   // (outputs: scratch / the spare covariance buffer, overwritten by the next stage anyway)
        } else if (which == 7) {   // U, G, P1 + the Joseph form as the handle launches them for a whole update (one instance, 6n <= 60: ONE kernel)
            launch_ug_final(h, n, h->block, h->P[h->cur ^ 1], true, true);
        } else if (which == 6) {   // cornerSubPix on the corners of the last detector call (reads raw_xy, rewrites xy with the same values)
            if (h->batch > 1 || !h->det_ready) return RVIO_ERR_UNSUPPORTED;
            const DetDev q = [&] { DetDev v = h->dets[h->det_set_last]; v.xy = h->det_xy2[h->dslot]; v.n_out = h->det_nout + h->dslot; return v; }();
            const uint8_t* im = h->pyr[h->pyr_cur].img[0];   // level 0 of the current pyramid = the image the detector saw
            if (q.sp_win > 15) hipLaunchKernelGGL(subpix_wide_kernel, dim3(d.F, 1, 1), dim3(SPG_T), subpix_wide_lds(q.sp_win), h->stream, im, d.W, q, (size_t)0, h->slab_bytes);
            else if (q.sp_win != SP_WIN) hipLaunchKernelGGL(subpix_generic_kernel, dim3(d.F, 1, 1), dim3(SPG_T), 0, h->stream, im, d.W, q, (size_t)0, h->slab_bytes);
            else if (h->wide_px) hipLaunchKernelGGL(subpix_kernel16, dim3((d.F + 3) / 4, 1, 1), dim3(64), 0, h->stream, im, d.W, q, (size_t)0, h->slab_bytes);
            else hipLaunchKernelGGL(subpix_kernel, dim3(d.F, 1, 1), dim3(SP_T), 0, h->stream, im, d.W, q, (size_t)0, h->slab_bytes, 0);
        } else return RVIO_ERR_INVALID;
    }
    HIPCHK(h, hipEventRecord(e1, h->stream));
    if (which == 8) {
        HIPCHK(h, hipMemcpyAsync(h->x[h->cur], bx, sizeof(double) * d.xdmax, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->P[h->cur], bP, sizeof(double) * d.dmax * d.dmax, hipMemcpyDeviceToDevice, h->stream));
        h->chol_ready = false;
    }
    HIPCHK(h, hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (bx) { HIPCHK(h, hipStreamSynchronize(h->stream)); (void)hipFree(bx); (void)hipFree(bP); }
    *avg_us = ms * 1e3f / iters;
    HIPCHK(h, hipGetLastError());
    return RVIO_OK;
}

// Test hook against results that depend on LEFT-OVER state: every stream is drained, then
//   what & 1   the filter's scratch — per-feature shares, information block, T, W, U, G, the d x d temporary, the gate's diagnostics, and the
//              SPARE state / covariance buffer (every stage writes its output in full) — is filled with 0xff bytes (NaN doubles, -1 ints)
//   what & 2   a kernel of 160 KB workgroups rewrites the LDS of the whole chip with signalling-NaN patterns
//   what & 4   the Tracker -> Updater hand-over tables (types / len / meas of all kHand tables; the counts stay) and the tracker's per-frame
//              scratch (vFeatsTracked, vFeatsUndistNorm, the next-frame order being built, FindNewer flags, ChessGrid cells) likewise
//   what & 8   sets the device-side error bit 4 ("a stage counter timed out"): the recovery path through rvio_hip_initialize can be tested
// A frame sequence must give the same results with any of 1 | 2 | 4 between its frames (tests/test_gpu_leftover.py).
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned long long pattern, int* sink) {
    extern __shared__ __align__(16) unsigned long long lp[];
    const int nw = 160 * 1024 / 8;
    for (int i = threadIdx.x; i < nw; i += 256) lp[i] = pattern ^ (unsigned long long)i;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8);   // 20 us: long enough for every CU to be handed a workgroup
    if (lp[(threadIdx.x * 977) % nw] == 1ull && sink) *sink = 1;      // (keeps the stores alive)
}
int rvio_hip_debug_poison(rvio_hip* h, int what) {
    if (!h) return RVIO_ERR_INVALID;
    { const int rc = drain_all(h); if (rc != RVIO_OK) return rc; }
    const DevCfg& d = h->dc;
    const size_t dm = d.dmax, PP = dm * dm, ldh = d.ldh;
    auto fill = [&](void* p, size_t bytes) -> hipError_t {
        if (!p || !bytes) return hipSuccess;
        hipError_t e = hipSuccess;
        for (int z = 0; z < h->batch && e == hipSuccess; ++z) e = hipMemsetAsync((char*)p + (size_t)z * h->slab_bytes, 0xff, bytes, h->stream);
        return e;
    };
    if (what & 1) {
        HIPCHK(h, fill(h->partial, sizeof(double) * d.Fu * ldh * ldh));
        HIPCHK(h, fill(h->block, sizeof(double) * 2 * ldh * ldh)); HIPCHK(h, fill(h->Ab, sizeof(double) * 2 * ldh * ldh));
        HIPCHK(h, fill(h->Tbuf, sizeof(double) * ldh * ldh)); HIPCHK(h, fill(h->W, sizeof(double) * ldh * ldh));
        if (h->S9scr) {   // the slab holds the Cholesky factor a PRE solve would read (role workgroup / stream_l): it goes with the slab — the next solve factors Pcc itself
            HIPCHK(h, hipMemsetAsync(h->S9scr, 0xff, sizeof(double) * S9_SLAB_DOUBLES(h->solve9_nt), h->stream));
            h->chol_ready = false; h->chol_async = false;   // (drain_all above has waited for stream_l)
        }
        HIPCHK(h, fill(h->U, sizeof(double) * dm * ldh)); HIPCHK(h, fill(h->G, sizeof(double) * dm * ldh));
        HIPCHK(h, fill(h->Pt1, sizeof(double) * PP));
        HIPCHK(h, fill(h->gamma, sizeof(double) * d.Fu)); HIPCHK(h, fill(h->pfinv, sizeof(double) * 3 * d.Fu));
        HIPCHK(h, fill(h->nrows, sizeof(int) * d.Fu)); HIPCHK(h, fill(h->acc, sizeof(int) * d.Fu)); HIPCHK(h, fill(h->ndof, sizeof(int) * d.Fu));
        if (h->tm_global) HIPCHK(h, fill(h->tm_global, sizeof(double) * d.Fu * d.rho_max * ldh));
        if (h->gpose) { HIPCHK(h, fill(h->gpose, sizeof(double) * d.Fu * (d.max_len - 1) * 24)); HIPCHK(h, fill(h->gvalid, sizeof(int) * d.Fu)); }
        HIPCHK(h, fill(h->x[h->cur ^ 1], sizeof(double) * d.xdmax)); HIPCHK(h, fill(h->P[h->cur ^ 1], sizeof(double) * PP));
    }
    if ((what & 4) && h->front_end) {
        const TrackerDev& t = h->t;
        for (int k = 0; k < rvio_hip::kHand; ++k) {
            HIPCHK(h, fill(h->tout[k].types, d.Fu)); HIPCHK(h, fill(h->tout[k].len, sizeof(int) * d.Fu));
            HIPCHK(h, fill(h->tout[k].meas, sizeof(float) * 2 * d.Fu * d.max_len));
        }
        HIPCHK(h, fill(t.tracked, sizeof(float) * 2 * d.F)); HIPCHK(h, fill(t.un2, sizeof(float) * 2 * d.F));
        HIPCHK(h, fill(t.tmp_feats, sizeof(float) * 2 * d.F)); HIPCHK(h, fill(t.tmp_un, sizeof(float) * 2 * d.F)); HIPCHK(h, fill(t.tmp_slot, sizeof(int) * d.F));
        HIPCHK(h, fill(t.cand_acc, sizeof(int) * d.F));
        HIPCHK(h, fill(t.cell_pts, sizeof(float) * (size_t)d.grid_cols * d.grid_rows * 2 * d.F * 2));
    }
    if (what & 2) {
        static bool attr = false;
        if (!attr) { HIPCHK(h, lds_attr((const void*)lds_poison_kernel, 160 * 1024)); attr = true; }
        hipLaunchKernelGGL(lds_poison_kernel, dim3(1024), dim3(256), 160 * 1024, h->stream, 0x7ff4dead00000000ull, (int*)nullptr);
        HIPCHK(h, hipGetLastError());
    }
    if (what & 8) {
        const int four = 4;   // (|= 4 on a drained handle: read-modify-write through the host)
        int e = 0;
        HIPCHK(h, hipMemcpyAsync(&e, &h->meta->err, sizeof e, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        e |= four;
        HIPCHK(h, hipMemcpyAsync(&h->meta->err, &e, sizeof e, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RVIO_OK;
}

// Test hook against ordering holes between the handle's streams: a one-wave kernel that occupies `which` (0 filter stream, 1 tracker /
// image chain 0, 2 side stream: KLT, RANSAC, book-keeping, 3 image chain 1) for `usec` microseconds, enqueued where the call is made.
// Every dependency of the pipeline has to hold under ANY pacing of its queues, so a frame sequence with stalls sprinkled over its streams
// must give the results of the synchronised run bit for bit (tests/test_gpu_flatout.py) — independent of how fast the box happens to be.
__global__ __launch_bounds__(64) void stall_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
int rvio_hip_debug_stall(rvio_hip* h, int which, int usec) {
    if (!h || which < 0 || which > 3 || usec < 0 || usec > 1000000) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = which == 0 ? h->stream : which == 1 ? h->stream_t : which == 2 ? h->stream_d : h->stream_c;
    if (!st) return RVIO_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)usec * 100ull);   // the constant 100 MHz clock
    HIPCHK(h, hipGetLastError());
    return RVIO_OK;
}

// Test hook against timing dependence INSIDE kernels (cross-wave hand-overs through LDS flags, last-block patterns, atomics): `wgs`
// workgroups on a stream of their own hammer HBM, L2 and LDS for `usec` microseconds beside whatever the handle has in flight — a box
// under load.  Waves of the pipeline's kernels then share SIMDs, LDS ports and L2 slices with the noise and run at a different relative
// pace; not one bit of any result may move (tests/test_gpu_flatout.py).
__global__ __launch_bounds__(256) void noise_kernel(unsigned long long ticks, double* scratch, size_t n_per_wg) {
    __shared__ double sh[4096];
    double* g = scratch + (size_t)blockIdx.x * n_per_wg;
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = (double)i;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    double acc = 0;
    unsigned it = 0;
    while (wall_clock64() - t0 < ticks) {
        for (size_t i = threadIdx.x; i < n_per_wg; i += 256) { const double v = g[i] + sh[(i + it) & 4095]; g[i] = v * 0.999; acc += v; }
        sh[(threadIdx.x * 17 + it) & 4095] = acc;
        __syncthreads();
        ++it;
    }
    if (acc == 1.2345e300) g[0] = acc;
}
int rvio_hip_debug_kernel_forms(rvio_hip* h, int throughput) {
    if (!h) return RVIO_ERR_INVALID;
    h->wide_px = throughput != 0;
    return RVIO_OK;
}
int rvio_hip_debug_noise(rvio_hip* h, int wgs, int usec) {
    if (!h || wgs < 1 || wgs > 4096 || usec < 0 || usec > 1000000) return RVIO_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    static hipStream_t ns = nullptr;
    static double* scratch = nullptr;
    const size_t per = 8192;   // 64 KB per workgroup: L2-resident for a few hundred workgroups, HBM beyond
    if (!ns) {
        HIPCHK(h, hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
        HIPCHK(h, hipMalloc((void**)&scratch, sizeof(double) * per * 4096));
        HIPCHK(h, hipMemset(scratch, 0, sizeof(double) * per * 4096));
    }
    hipLaunchKernelGGL(noise_kernel, dim3(wgs), dim3(256), 0, ns, (unsigned long long)usec * 100ull, scratch, per);
    HIPCHK(h, hipGetLastError());
    return RVIO_OK;
}

int rvio_hip_debug_ring(rvio_hip* h, long long* out512, int* frame) {
    if (!h || !out512 || !frame) return RVIO_ERR_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_ring), sizeof(long long) * 512));
    HIPCHK(h, hipMemcpyFromSymbol(frame, HIP_SYMBOL(g_ring_frame), sizeof(int)));
    return RVIO_OK;
}
int rvio_hip_debug_ring2(rvio_hip* h, long long* out512, int* frame) {
    if (!h || !out512 || !frame) return RVIO_ERR_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_ring2), sizeof(long long) * 512));
    HIPCHK(h, hipMemcpyFromSymbol(frame, HIP_SYMBOL(g_ring2_frame), sizeof(int)));
    return RVIO_OK;
}
int rvio_hip_debug_ring3(rvio_hip* h, long long* out512) {
    if (!h || !out512) return RVIO_ERR_INVALID;
    int rc = rvio_hip_sync(h);
    if (rc != RVIO_OK) return rc;
    HIPCHK(h, hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_ring3), sizeof(long long) * 512));
    return RVIO_OK;
}
// experiments (tools/at_rest_literal.py): every update the literal sweep can take takes it (process-wide switch)
int rvio_hip_debug_literal_force(rvio_hip* h, int on) {
    if (!h) return RVIO_ERR_INVALID;
    const int rc = drain_all(h);
    if (rc != RVIO_OK) return rc;
    const int v = on ? 1 : 0;
    HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(g_lit_force), &v, sizeof v));
    return RVIO_OK;
}
int rvio_hip_debug_clocks2(rvio_hip* h, long long* out64) {
    if (!h || !out64) return RVIO_ERR_INVALID;
    const int rc = drain_all(h);
    if (rc != RVIO_OK) return rc;
    HIPCHK(h, hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dbg2), sizeof(long long) * 64));
    return RVIO_OK;
}
// the per-phase sums / counts of feat_build_body over all workgroups (DBG_P) into out64, the longest phases (g_dbg2[30..40]) into max64; both cleared
int rvio_hip_debug_phases(rvio_hip* h, long long* out64, long long* max64) {
    if (!h || !out64 || !max64) return RVIO_ERR_INVALID;
    const int rc = drain_all(h);
    if (rc != RVIO_OK) return rc;
    static const long long zero[64] = {0};
    HIPCHK(h, hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dbg3), sizeof(long long) * 64));
    HIPCHK(h, hipMemcpyFromSymbol(max64, HIP_SYMBOL(g_dbg2), sizeof(long long) * 64));
    HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(g_dbg3), zero, sizeof zero));
    HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(g_dbg2), zero, sizeof zero));
    return RVIO_OK;
}
int rvio_hip_debug_clocks(rvio_hip* h, long long* out64) {
    if (!h || !out64) return RVIO_ERR_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dbg), sizeof(long long) * 64));
    return RVIO_OK;
}
}
