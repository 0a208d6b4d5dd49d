/*
 * rvio_hip.h — C-ABI of the MI355X-native robocentric MSCKF hot path.
 *
 * The reference (rpng/R-VIO) has no FFI/plugin layer: its per-frame hot path
 * sits behind three C++ member calls made from System::MonoVIO
 * (src/rvio/System.cc:258,263,268).  Each entry point below replaces one of
 * those calls (or a slice of MonoVIO itself) and cites it.  Everything is
 * POD + plain pointers; no C++/torch types cross this boundary.
 *
 * Conventions
 *   - state vector x  : [qG(4) pG(3) g(3) | qk(4) pk(3) v(3) bg(3) ba(3) | n x (q(4) p(3))]
 *                       JPL quaternions [x y z w]   (System.cc:142-149,326-331)
 *   - covariance  P   : (24+6n)^2 doubles, COLUMN-major (Eigen::MatrixXd layout),
 *                       error state [thG pG g | thk pk v bg ba | n x (th p)]
 *   - all functions return RVIO_OK (0) or a negative rvio_status; the silent
 *     early-outs of the reference (too few features etc.) are reported through
 *     rvio_frame_info, not through the return code.
 *   - a handle owns one HIP stream and all device memory of one filter
 *     instance; a handle is not thread-safe, distinct handles are independent.
 *   - functions ending in _dev take DEVICE pointers, all others HOST pointers.
 *   - every entry point is asynchronous on the handle's stream unless it
 *     returns data to host memory (get_* / *_sync), which synchronise.
 */
#ifndef RVIO_HIP_H
#define RVIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVIO_HIP_ABI_VERSION 5   /* 3: rvio_hip_frame_sharded_dev, IMU batches of any length, error bits 4 (hard) / 8;  4: the rvio_hip_debug_poison / _stall / _noise / _kernel_forms test hooks are part of the exported surface;  5: the payload of rvio_hip_update_local / _global is the packed wire format of csrc/rvio_dev.h shard_layout */
/* IMU samples the staging of the host-buffer entry points is allocated for (0.96 s at 200 Hz).  PreIntegrator::propagate iterates any
 * list (PreIntegrator.cc:96-97), and so does every entry point here: a longer batch (dropped images) is accepted — the staging grows once,
 * at the price of one host synchronisation; the _dev entry points take any m. */
#define RVIO_HIP_MAX_IMU 192

typedef enum rvio_status {
    RVIO_OK = 0,
    RVIO_ERR_INVALID = -1,     /* bad argument / size                                   */
    RVIO_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime error (see last_error)    */
    RVIO_ERR_UNSUPPORTED = -3, /* e.g. Tracker.nMinDist >= 128 with the device detector (cornerSubPix half-windows 1..63) */
    RVIO_ERR_STATE = -4        /* call out of sequence (e.g. update before set_state)    */
} rvio_status;

/* All parameters of config/rvio_euroc.yaml that the hot path consumes
 * (key -> consumer table: SURVEY.md appendix A).  Types mirror what the
 * reference stores them as (float32 intrinsics: Tracker.cc:39-62; float32
 * sigma_im: Updater.cc:42-44). */
typedef struct rvio_config {
    /* IMU.*  (PreIntegrator.cc:32-44, System.cc:60-67) */
    double imu_rate;      /* IMU.dps          */
    double sigma_g;       /* IMU.sigma_g      */
    double sigma_wg;      /* IMU.sigma_wg     */
    double sigma_a;       /* IMU.sigma_a      */
    double sigma_wa;      /* IMU.sigma_wa     */
    double gravity;       /* IMU.nG           */
    double small_angle;   /* IMU.nSmallAngle  */
    /* Camera.* */
    int32_t width, height;            /* FeatureDetector.cc:37-38               */
    float fx, fy, cx, cy;             /* Tracker.cc:39-42                       */
    float k1, k2, p1, p2, k3;         /* Tracker.cc:51-61                       */
    float sigma_px, sigma_py;         /* Updater.cc:42-44 (sigma_im = max)      */
    double T_bc[16];                  /* Camera.T_BC0, ROW-major 4x4 (Updater.cc:46-53) */
    int32_t fisheye;                  /* Camera.Fisheye: cv::fisheye::undistortPoints with D = (k1,k2,p1,p2) (Tracker.cc:116-119) */
    /* Tracker.* */
    int32_t n_features;               /* Tracker.nFeatures          (Tracker.cc:73)  */
    int32_t max_track_len;            /* Tracker.nMaxTrackingLength (Tracker.cc:78)  */
    int32_t min_track_len;            /* Tracker.nMinTrackingLength (Tracker.cc:79)  */
    float   min_dist;                 /* Tracker.nMinDist           (FeatureDetector.cc:31) */
    float   qual_lvl;                 /* Tracker.nQualLvl           */
    float   block_x, block_y;         /* Tracker.nBlockSizeX/Y, stored as float like upstream (FeatureDetector.h:72-73) */
    int32_t enable_equalizer;         /* Tracker.EnableEqualizer    (Tracker.cc:70-71) */
    int32_t use_sampson;              /* Tracker.UseSampson         (Ransac.cc:34-35) */
    double  inlier_thr;               /* Tracker.nInlierThrd        (Ransac.cc:37)    */
    /* INI.* (System.cc:77-91) */
    double  ini_thr_angle;            /* INI.nThresholdAngle */
    double  ini_thr_displ;            /* INI.nThresholdDispl */
    int32_t ini_enable_alignment;     /* INI.EnableAlignment */
    int32_t reserved0;
} rvio_config;

/* Fill *cfg with the stock values of config/rvio_euroc.yaml:8-111. */
void rvio_config_euroc(rvio_config* cfg);

/* One IMU sample: struct ImuData (InputBuffer.h:35-51). */
typedef struct rvio_imu {
    double w[3];   /* AngularVel   */
    double a[3];   /* LinearAccel  */
    double t;      /* Timestamp    */
    double dt;     /* TimeInterval */
} rvio_imu;

/* The Tracker -> Updater hand-over: mvFeatTypesForUpdate + mvlFeatMeasForUpdate
 * (Tracker.h:67-74), flattened.  meas[f][k] is the k-th (oldest first)
 * undistorted-normalised observation (cv::Point2f) of feature f. */
typedef struct rvio_tracks {
    int32_t n_feat;              /* mvFeatTypesForUpdate.size()                 */
    int32_t max_len;             /* row stride of meas, >= every len[f]         */
    const unsigned char* types;  /* n_feat x '1' (lost) | '2' (max length)      */
    const int32_t* len;          /* n_feat track lengths                        */
    const float* meas;           /* n_feat x max_len x 2 float32                */
} rvio_tracks;

/* What the reference only logs through ROS_DEBUG, made observable. */
typedef struct rvio_frame_info {
    int32_t n_clones;          /* nCloneStates after the frame                  */
    int32_t n_tracked_in;      /* mnFeatsToTrack entering track()               */
    int32_t n_klt_ok;          /* status!=0 after KLT                           */
    int32_t n_ransac_inliers;  /* FindInliers return value                      */
    int32_t n_feat_update;     /* features handed to update()                   */
    int32_t n_feat_accepted;   /* nGoodFeatCount                                */
    int32_t n_rows;            /* stacked rows nRowCount                        */
    int32_t updated;           /* 1 if the EKF update was applied               */
    int32_t n_tracked_out;     /* mnFeatsToTrack leaving track() (after refill) */
    int32_t ransac_winner;     /* nWinnerHypothesisIdx                          */
    int32_t reserved[5];       /* [0]: sticky device-side error flag: 1 = singular pivot in the solve, 2 = a track the window cannot hold was
                                * dropped, 4 = a device-side stage counter (filter done -> book-keeping) timed out,
                                * 8 = a non-positive pivot in a feature's gate matrix S_f (indefinite covariance handed in): that feature was rejected
                                * [1]: nRank of the reference's literal Givens sweep + leading-row scan (Updater.cc:493-523) when this update ran it on the
                                * device (an update of a handful of features whose rank decision the structure of the stack does not settle), -1 otherwise */
    int32_t rank_truncated_at; /* Updater.cc:516-529: nRank when the leading-row scan cut informative rows off (the type-'1'
                                * features' rows, dropped from this update), -1 otherwise */
} rvio_frame_info;

typedef struct rvio_hip rvio_hip;  /* opaque; replaces the System-owned stage objects (System.h:89-92) */

/* --- lifetime ------------------------------------------------------------- */
/* new Tracker/Updater/PreIntegrator (System.cc:96-99).  device = HIP ordinal. */
int rvio_hip_create(const rvio_config* cfg, int device, rvio_hip** out);
void rvio_hip_destroy(rvio_hip* h);
const char* rvio_hip_last_error(const rvio_hip* h);
int rvio_hip_abi_version(void);
/* the hipStream_t (as void*) all work of this handle is enqueued on */
void* rvio_hip_stream(rvio_hip* h);
int rvio_hip_sync(rvio_hip* h);

/* --- filter state --------------------------------------------------------- */
/* System::xkk / Pkk (System.h:85-86).  xdim = 26+7n, d = 24+6n. */
int rvio_hip_set_state(rvio_hip* h, const double* x, int xdim, const double* P, int d);
int rvio_hip_get_state(rvio_hip* h, double* x, int* xdim, double* P, int* d);
/* System::initialize (System.cc:115-170): gravity-aligned x(26), P(24x24). */
int rvio_hip_initialize(rvio_hip* h, const double w[3], const double a[3], int n_imu);

/* --- stage entry points (one per reference call site) ---------------------- */
/* PreIntegrator::propagate (PreIntegrator.h:40, call site System.cc:263).
 * Mutates the handle's (x,P) in place: they become xk1k / Pk1k. */
int rvio_hip_propagate(rvio_hip* h, const rvio_imu* imu, int m);

/* Updater::update (Updater.h:43-44, call site System.cc:268) on host-provided
 * tracks.  (x,P) become xk1k1 / Pk1k1; pass-through if <=2 features survive
 * (Updater.cc:460,621-627). */
int rvio_hip_update(rvio_hip* h, const rvio_tracks* tracks);

/* State augmentation + window slide + robocentric composition
 * (System.cc:279-365).  do_augment mirrors `nImageCountAfterInit>1`. */
int rvio_hip_augment_compose(rvio_hip* h, int do_augment);

/* --- visual front end ------------------------------------------------------ */
/* Tracker::track (Tracker.h:50, call site System.cc:258) on a W x H u8 image.
 * `cand`/`n_cand` are the detector's corner list for this frame, i.e. the
 * output of FeatureDetector::DetectWithSubPix (FeatureDetector.cc:55-75; the
 * detector itself is outside the hot path); the grid-based
 * selection FindNewer (FeatureDetector.cc:97-150) and the refill
 * (Tracker.cc:344-387) run on the device.  Results stay on the device for
 * rvio_hip_update_tracked and can be fetched with rvio_hip_get_tracks. */
int rvio_hip_track(rvio_hip* h, const uint8_t* img, int stride,
                   const rvio_imu* imu, int m, const float* cand_xy, int n_cand);
int rvio_hip_track_dev(rvio_hip* h, const uint8_t* d_img, int stride,
                       const rvio_imu* d_imu, int m, const float* d_cand_xy, int n_cand);
/* Direct-track mode: the caller supplies the result of the
 * calcOpticalFlowPyrLK call (vFeatsTracked, vInlierFlag; Tracker.cc:244) for the
 * n_pts currently tracked features; everything after Tracker.cc:246 (undistort,
 * RANSAC, book-keeping, refill) runs on the device as in rvio_hip_track. */
int rvio_hip_track_points(rvio_hip* h, const float* tracked_xy, const unsigned char* status, int n_pts,
                          const rvio_imu* imu, int m, const float* cand_xy, int n_cand);
/* Corner lists.  Every image entry point (rvio_hip_track, rvio_hip_track_dev, rvio_hip_frame,
 * rvio_hip_frame_dev) accepts cand_xy == NULL: the library then runs
 * FeatureDetector::DetectWithSubPix (FeatureDetector.cc:55-75: goodFeaturesToTrack with
 * s*nMinDist, s = 1 on the first image / 2 on refills, + cornerSubPix) on the device, on the image
 * the tracker sees (after CLAHE), concurrently with pyramid/KLT/RANSAC (two streams).  A non-NULL list
 * replaces the detector (caller-side detection).  rvio_hip_get_corners copies the device
 * detector's last result out: refined corners, the corners before cornerSubPix and the
 * min-eigenvalue map (W*H floats); each pointer may be NULL. */
int rvio_hip_get_corners(rvio_hip* h, int32_t* n, float* xy, float* raw_xy, float* eig);

/* Copy the device-resident mvFeatTypesForUpdate / mvlFeatMeasForUpdate out.
 * Buffers must hold ceil(n_features/2) entries (x max_track_len x 2 floats). */
int rvio_hip_get_tracks(rvio_hip* h, int32_t* n_feat, unsigned char* types, int32_t* len, float* meas);
/* Tracker's persistent members: mvFeatsToTrack (px) and track lengths. */
int rvio_hip_get_tracker_points(rvio_hip* h, int32_t* n, float* xy, int32_t* hist_len);
/* Updater::update on the tracker's device-resident outputs. */
int rvio_hip_update_tracked(rvio_hip* h);

/* --- whole frame ----------------------------------------------------------- */
/* The timed body of System::MonoVIO (System.cc:253-367): track -> propagate ->
 * [update if nCloneStates > nMinTrackingLength-1] -> augment -> compose, all
 * enqueued on the handle's streams with no host synchronisation; the front end of frame k+1
 * overlaps the filter of frame k.  The call returns before the device has read the inputs:
 * d_img / d_imu / d_cand_xy must stay valid (and unchanged) until the frame has completed on the
 * device, i.e. until rvio_hip_sync or any getter (rvio_hip_get_pose, rvio_hip_get_state, ...)
 * issued after this call has returned.  (rvio_hip_frame, below, has no such condition.) */
int rvio_hip_frame_dev(rvio_hip* h, const uint8_t* d_img, int stride,
                       const rvio_imu* d_imu, int m, const float* d_cand_xy, int n_cand);
/* The same body fed from HOST buffers — what System::MonoVIO holds at System.cc:253-258
 * (pImageData->Image, the IMU list of the frame) plus the detector's corners.  The caller's
 * buffers are packed into a pinned ring on the host and consumed before the call returns; the
 * H2D copies then run asynchronously on the tracker stream (device staging double-buffered by
 * frame parity) and overlap the previous frame's filter work. */
int rvio_hip_frame(rvio_hip* h, const uint8_t* img, int stride,
                   const rvio_imu* imu, int m, const float* cand_xy, int n_cand);
/* The pipelined frame split open for callers that run the update themselves (the feature-sharded
 * updater): frame_begin_dev enqueues PreIntegrator::propagate on the filter stream and the front
 * end on its own streams and orders the filter stream after the front end; the caller then issues
 * rvio_hip_frame_plan, the update (rvio_hip_update_local -> collective on rvio_hip_stream ->
 * rvio_hip_update_global, or rvio_hip_update_tracked) and rvio_hip_augment_compose; frame_end
 * closes the frame.  No host synchronisation anywhere. */
int rvio_hip_frame_begin_dev(rvio_hip* h, const uint8_t* d_img, int stride,
                             const rvio_imu* d_imu, int m, const float* d_cand_xy, int n_cand);
int rvio_hip_frame_end(rvio_hip* h);
/* For callers that sequence the frame themselves (per-stage timing, the sharded updater):
 * frame_plan advances nImageCountAfterInit and reports MonoVIO's two data-independent
 * branches (System.cc:266 `nCloneStates > mnMinCloneStates`, System.cc:280
 * `nImageCountAfterInit > 1`); propagate_dev is rvio_hip_propagate on device-resident IMU. */
int rvio_hip_frame_plan(rvio_hip* h, int* do_update, int* do_augment);
int rvio_hip_propagate_dev(rvio_hip* h, const rvio_imu* d_imu, int m);

/* --- batched instances ---------------------------------- */
/* B independent instances (B robots / B replays of the same sensor rate) behind one handle:
 * every buffer of instance i lives in slab i, and every stage is ONE launch with
 * gridDim.z = B, i.e. the single-workgroup stages of one filter become B workgroups.  The
 * arithmetic of an instance is the arithmetic of a plain handle, bit for bit.  The reference
 * runs one System per process (System.cc:38-41 globals); this is the multi-robot form of the
 * same calls.  front_end = 0: the filter only (PreIntegrator::propagate, Updater::update,
 * augmentation/composition; driven by rvio_hip_frame_tracks_dev); front_end = 1: Tracker::track
 * as well (CLAHE, DetectWithSubPix, KLT, RANSAC, book-keeping; driven by
 * rvio_hip_frame_batch_dev).  The single-instance entry points return RVIO_ERR_UNSUPPORTED on
 * a batch handle; the window length is common to all instances (it depends on the frame count
 * only, System.cc:266,280).  rvio_hip_set_state / rvio_hip_initialize give every instance the
 * same state, rvio_hip_get_state reads instance 0. */
int rvio_hip_create_batch(const rvio_config* cfg, int device, int n_instances, int front_end,
                          rvio_hip** out);
int rvio_hip_batch_size(const rvio_hip* h);
int rvio_hip_set_state_at(rvio_hip* h, int instance, const double* x, int xdim, const double* P, int d);
int rvio_hip_get_state_at(rvio_hip* h, int instance, double* x, int* xdim, double* P, int* d);
/* rvio_hip_get_tracker_points of one instance (batch handle with front end): Tracker::mvFeatsToTrack
 * and the length of each feature's tracking history. */
int rvio_hip_get_tracker_points_at(rvio_hip* h, int instance, int32_t* n, float* xy, int32_t* hist_len);
/* The body of System::MonoVIO after Tracker::track (System.cc:263-365: propagate, update if
 * nCloneStates > mnMinCloneStates, augmentation + composition) on DEVICE-resident hand-over
 * tables, for all B instances of the handle (B = 1 for a plain handle):
 *   d_n_feat[B], d_types[B][Fu], d_len[B][Fu], d_meas[B][Fu][max_track_len][2]
 * (Tracker::mvFeatTypesForUpdate / mvlFeatMeasForUpdate, Tracker.h:67-74, Fu = ceil(nFeatures/2));
 * d_imu[B][imu_stride] samples, imu_stride = 0: one IMU batch shared by all instances.
 * Track lengths must satisfy 2 <= len <= window length + 1 (what Tracker::track emits). */
int rvio_hip_frame_tracks_dev(rvio_hip* h, const rvio_imu* d_imu, int imu_stride, int m,
                              const int32_t* d_n_feat, const unsigned char* d_types,
                              const int32_t* d_len, const float* d_meas);
/* One camera frame of every instance (the whole timed body of System::MonoVIO, System.cc:253-367,
 * pipelined like rvio_hip_frame_dev): d_imgs = B mono8 images, `img_stride` bytes apart, rows
 * `stride` bytes apart; d_imu[B][imu_stride] (0: shared).  Corners come from the device detector. */
int rvio_hip_frame_batch_dev(rvio_hip* h, const uint8_t* d_imgs, int stride, size_t img_stride,
                             const rvio_imu* d_imu, int imu_stride, int m);
/* same, direct-track mode, host inputs */
int rvio_hip_frame_points(rvio_hip* h, const float* tracked_xy, const unsigned char* status, int n_pts,
                          const rvio_imu* imu, int m, const float* cand_xy, int n_cand);
int rvio_hip_get_frame_info(rvio_hip* h, rvio_frame_info* info);
/* pose line of stamped_pose_ests.dat (System.cc:371-374): pGk(3), qkG(4) */
int rvio_hip_get_pose(rvio_hip* h, double p[3], double q[4]);

/* --- feature-sharded updater (no reference counterpart) ------ */
/* Stage A: per-feature build + gate on the features f with f % world == rank,
 * then local compression to this shard's share of the information block
 * [A|b] = Hw^T [Hw | r], kept in two parts (the sum over the type-'2' features and the sum over
 * the type-'1' features) plus five counters, so that stage B can apply the reference's rank
 * truncation (Updater.cc:516-529) to the gathered whole.  *d_block is a device pointer owned by
 * the handle, *n_doubles its length: this is the payload of the all-gather, in the wire format of
 * csrc/rvio_dev.h shard_layout (r-vio_amd/abi.py shard_pack / shard_unpack mirror it): 8 counters, then
 * the 16 x 16 tiles (256 doubles each) on and above the diagonal of the type-'2' part that a
 * type-'2' feature can reach, then those of the type-'1' part — 8 + 256 (tiles2 + tiles1) doubles
 * for the CURRENT window (215 KB at 1600 features / 30 clones; rounds 2-5: 2 (6n_max+1)^2 doubles = 524 KB). */
int rvio_hip_update_local(rvio_hip* h, const rvio_tracks* tracks, int rank, int world,
                          double** d_block, int* n_doubles);
/* Stage B: sum `world` gathered blocks (device pointer, rank-major) in rank
 * order and run the EKF update; bit-identical on every rank. */
int rvio_hip_update_global(rvio_hip* h, const double* d_blocks, int world);

/* The whole sharded frame behind ONE call (what a multi-GPU host loop issues per image): rvio_hip_frame_begin_dev, rvio_hip_frame_plan,
 * rvio_hip_update_local, ncclAllGather of the blocks ON THE HANDLE'S FILTER STREAM, rvio_hip_update_global, rvio_hip_augment_compose,
 * rvio_hip_frame_end — no host synchronisation, the collective ordered by plain stream order.  `comm` is the caller's ncclComm_t (RCCL);
 * NULL is allowed with world == 1 only (no collective).  `allgather` = NULL: RCCL's ncclAllGather is resolved from the RCCL the process has
 * loaded (the library has no link-time dependency on it); or the entry point itself,
 *   int (*)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream). */
int rvio_hip_frame_sharded_dev(rvio_hip* h, const uint8_t* d_img, int stride, const rvio_imu* d_imu, int m,
                               const float* d_cand_xy, int n_cand, int rank, int world, void* comm, void* allgather);

/* --- diagnostics for parity tests ------------------------------------------ */
/* per-feature results of the last update: accept flag, Mahalanobis distance,
 * nDOF, inverse-depth estimate (phi,psi,rho). arrays sized n_feat. */
int rvio_hip_get_update_diag(rvio_hip* h, int32_t* n_feat, int32_t* accepted, double* gamma,
                             int32_t* ndof, double* pfinv /* n_feat x 3 */);

/* pyramid level of the most recent image (u8 w*h, int16 (dx,dy) w*h*2) and the raw KLT output
 * (vFeatsTracked + its undistorted-normalised form) of the last track() call */
int rvio_hip_debug_pyramid(rvio_hip* h, int level, int32_t* w, int32_t* hgt, uint8_t* img, int16_t* dxy);
int rvio_hip_debug_tracked(rvio_hip* h, int n, float* xy, float* un_xy);
/* Measurement hook for bench.py's roofline object: average device time (us, HIP events on the handle's stream)
 * of `iters` back-to-back launches of one hot kernel on the operands the last frame left in HBM.
 * which: 0 = solve kernel, 1 = KLT kernel (the current image matched back onto the previous one from the current feature positions),
 * 2 = per-feature Jacobian/nullspace/gate kernel, 3 = reduction of the per-feature information shares (+ rank truncation),
 * 4 = U/G/P1 strips, 5 = Joseph-form kernel (4, 5: the two-launch forms), 6 = cornerSubPix on the last corner list,
 * 7 = U/G/P1 + Joseph form as this handle launches them for an update (one instance, 6n <= 60: one fused kernel). */
int rvio_hip_debug_time_kernel(rvio_hip* h, int which, int iters, float* avg_us);
/* Test hook against results that depend on LEFT-OVER state (scratch in HBM, LDS contents, stale hand-over entries).  Drains every stream of
 * the handle, then: what & 1 fills the filter's scratch and the spare state / covariance buffer with 0xff bytes (NaN); & 2 rewrites the LDS
 * of the whole chip with NaN patterns; & 4 does the same to the Tracker -> Updater hand-over tables (types / len / meas; counts stay) and the
 * tracker's per-frame scratch; & 8 sets the device-side error bit 4 ("a stage counter timed out", RVIO_ERR_STATE at the next
 * rvio_hip_sync) so that the recovery path — rvio_hip_initialize — can be exercised.  1 | 2 | 4 between the frames of a sequence must not
 * change any result. */
int rvio_hip_debug_poison(rvio_hip* h, int what);
/* Test hook against ordering holes between the handle's streams: occupies stream `which` (0 filter, 1 tracker / image chain 0, 2 side stream
 * — KLT, RANSAC, book-keeping —, 3 image chain 1) with a sleeping one-wave kernel for `usec` microseconds, enqueued where the call is made.
 * A frame sequence with stalls sprinkled over the streams must give the results of the synchronised run bit for bit. */
int rvio_hip_debug_stall(rvio_hip* h, int which, int usec);
/* Test hook against timing dependence inside kernels: `wgs` workgroups on a stream of their own load HBM, L2 and LDS for `usec`
 * microseconds beside whatever the handle has in flight (a box under load); results must not move by a bit. */
int rvio_hip_debug_noise(rvio_hip* h, int wgs, int usec);
/* Test hook: selects the throughput forms of the image kernels (the ones a batch handle of >= 8 instances launches: several pixels per
 * thread, four corners / features per wave) on any handle, or the latency forms (0) on a batch handle.  The two families compute the same
 * bits by construction; the tests run the KLT early-out cases and the detector's image set through both. */
int rvio_hip_debug_kernel_forms(rvio_hip* h, int throughput);

#ifdef __cplusplus
}
#endif
#endif /* RVIO_HIP_H */
