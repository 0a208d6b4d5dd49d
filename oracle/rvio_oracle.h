/*
 * rvio_oracle.h — C API of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * A dependency-free restatement (C++17, double precision with the reference's
 * float32 islands reproduced) of the R-VIO per-frame hot path, each function
 * citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (r-vio_amd/) never links or calls it.
 *
 * PARITY PIN: the reference ships no tests / golden vectors and its real dependencies (Eigen, OpenCV, ROS) are not in this
 * image, so the reference cannot be built the way its CMakeLists.txt builds it.  What IS done (oracle/Makefile target `ref`,
 * tests/test_ref_pins.py): the reference's own translation units — Updater.cc, PreIntegrator.cc, Ransac.cc, InputBuffer.cc,
 * FeatureDetector.cc, Tracker.cc, System.cc, util/Numerics.h — are compiled UNMODIFIED, from where they lie, against
 * oracle/refshim/ (a header stand-in for exactly the Eigen / OpenCV-container / ROS surface they use) into oracle/_ref/libref.so,
 * and this restatement is held against it stage by stage (<= 1e-16 observed, identical discrete decisions) and over free-running
 * System::MonoVIO sequences (241 frames: 2.9e-13).  That pins every line the reference's authors wrote.  Still UNPINNED, and said
 * so wherever it matters: (1) the last bits of Eigen's own kernels (refshim/mini_eigen.hpp follows Eigen 3.3's documented
 * semantics, SURVEY.md appendix C, with index-order sums); (2) OpenCV's image algorithms (CLAHE, pyramidal LK, undistortPoints,
 * goodFeaturesToTrack, cornerSubPix), restated in frontend.cpp / detector.cpp from the public OpenCV 3.x implementation
 * (appendix B) — inside libref.so those calls forward to the restatement, so they are pinned only by the analytic / independent
 * checks of tests/test_oracle_pins.py and tests/test_opencv_distance.py; (3) glibc rand() IS pinned bit for bit (Ransac.cc:63-69).
 */
#ifndef RVIO_ORACLE_H
#define RVIO_ORACLE_H
#include "../include/rvio_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- N1: util/Numerics.h:30-167 ---- */
void orc_quat_mul(const double q1[4], const double q2[4], double out[4]);
void orc_quat_to_rot(const double q[4], double R_rowmajor[9]);
void orc_rot_to_quat(const double R_rowmajor[9], double q[4]);
double orc_chi2_95(int dof); /* CHI_THRESHOLD[dof-1], Numerics.h:173-224 */

/* ---- System::initialize, System.cc:115-170 ---- */
void orc_initialize(const rvio_config* cfg, const double w[3], const double a[3], int n_imu,
                    double x[26], double P[24 * 24]);

/* ---- P1: PreIntegrator::propagate, PreIntegrator.cc:51-194 ----
 * x (xdim) in -> x_out; P (d x d col-major) is propagated IN PLACE. */
void orc_propagate(const rvio_config* cfg, const double* x, int xdim, double* P, int d,
                   const rvio_imu* imu, int m, double* x_out);

/* ---- U1..U10: Updater::update, Updater.cc:72-628 ----
 * diag arrays (may be NULL) are sized tracks->n_feat; info[0]=nGoodFeatCount,
 * info[1]=nRowCount, info[2]=nRank (or -1 if fat / no update), info[3]=updated. */
void orc_update(const rvio_config* cfg, const double* x, int xdim, const double* P, int d,
                const rvio_tracks* tracks, double* x_out, double* P_out,
                int32_t* accepted, double* gamma, int32_t* ndof, double* pfinv, int32_t info[4]);

/* U1..U3 of one feature evaluated at a given inverse-depth triple pf = (phi, psi, rho) (NULL: the LM estimate, returned in pf_out):
 * residual, Hx (row-major 2Lu x 6n) and Hf (2Lu x 3) BEFORE the nullspace projection; returns 2Lu.  Finite-difference pin of U3. */
int orc_feature_model(const rvio_config* cfg, const double* x, int xdim, unsigned char type, const float* meas, int L,
                      const double* pf, double* r_out, double* Hx_rowmajor, double* Hf_rowmajor, double* pf_out);

/* The two halves of orc_update (analysis of the rank truncation Updater.cc:516-529, tests/test_truncation.py):
 * orc_update_stack = U1..U6, returns M and the stacked pair (Hw row-major M x 6n, r) of the accepted features;
 * orc_update_from_stack = U7..U10 on a given pair; row_norms (may be NULL) = norms of the first min(M,12n) rows after the sweep */
int orc_update_stack(const rvio_config* cfg, const double* x, int xdim, const double* P, int d,
                     const rvio_tracks* tracks, double* Hw_rowmajor, double* r_out, int32_t* n_good);
void orc_update_from_stack(const rvio_config* cfg, const double* x, int xdim, const double* P, int d,
                           const double* Hw_rowmajor, const double* r_in, int M, int n_good,
                           double* x_out, double* P_out, int32_t info[4], double* row_norms);

/* Same update, but with the information-form compression [A|b]=Hw^T[Hw|r]
 * sharded over `world` ranks (features f%world==rank) and summed in rank
 * order, and the reference's rank truncation (Updater.cc:516-529) applied in its structural form
 * (filter.cpp, comment above orc_update_local) — the CPU mirror of rvio_hip_update_local/_global, used by the gloo
 * tests.  block: 2*6n*(6n+1)+8 doubles (type-'2' sum, type-'1' sum, counters); info[2] = the column the scan stopped at
 * when it discarded the type-'1' rows, else -1. */
void orc_update_local(const rvio_config* cfg, const double* x, int xdim, const double* P, int d,
                      const rvio_tracks* tracks, int rank, int world, double* block);
void orc_update_global(const rvio_config* cfg, const double* x, int xdim, const double* P, int d,
                       const double* blocks, int world, double* x_out, double* P_out, int32_t info[4]);

/* ---- S1+S2: System.cc:279-365.  x/P buffers must hold the grown state. ---- */
void orc_augment_compose(const rvio_config* cfg, double* x, int* xdim, double* P, int* d,
                         int do_augment, double pose_p[3], double pose_q[4]);

/* ---- T4: Tracker::UndistortAndNormalize, Tracker.cc:100-132 (cv::undistortPoints) ---- */
void orc_undistort(const rvio_config* cfg, const float* px_xy, int n, float* out_xy);

/* ---- T5: Ransac, Ransac.cc:50-266.  p1/p2: 3 x n col-major; flags in/out.
 * rng: 35 ints of oracle rand state (see orc_rand_*). returns #inliers, winner idx in *winner. */
void orc_srand(int32_t state[35], unsigned seed);
int orc_rand(int32_t state[35]);
int orc_ransac(const rvio_config* cfg, const double* p1, const double* p2, int n,
               const rvio_imu* imu, int m, unsigned char* flags, int32_t rng[35], int* winner,
               int32_t* pairs /* 32 ints or NULL */);

/* ---- T3: cv::calcOpticalFlowPyrLK restated ---- */
/* pyrDown ([1 4 6 4 1]/16 separable, BORDER_REFLECT_101): (w,h) -> ((w+1)/2,(h+1)/2) */
void orc_pyr_down(const uint8_t* src, int w, int h, int stride, uint8_t* dst);
/* Scharr derivative, int16 interleaved (dx,dy) */
void orc_scharr(const uint8_t* src, int w, int h, int stride, int16_t* dxy);
/* LK on raw images: win 15x15, maxLevel 3, 30 it / eps 0.01, minEig 1e-3 (Tracker.cc:237-244) */
void orc_klt(const uint8_t* prev, const uint8_t* next, int w, int h, int stride,
             const float* pts_xy, int n, float* out_xy, unsigned char* status);
/* measurement only: float accumulators in row-major order, as OpenCV's scalar LKTrackerInvoker holds them */
void orc_klt_float(const uint8_t* prev, const uint8_t* next, int w, int h, int stride,
             const float* pts_xy, int n, float* out_xy, unsigned char* status);

/* ---- T0: cv::createCLAHE(3.0, Size(5,5))->apply, Tracker.cc:198-202 (OpenCV clahe.cpp restated); out is w*h, packed ---- */
void orc_clahe(const uint8_t* img, int w, int h, int stride, uint8_t* out);

/* ---- T7: FeatureDetector::DetectWithSubPix, FeatureDetector.cc:55-75 (OpenCV goodFeaturesToTrack + cornerSubPix restated) ---- */
void orc_min_eig(const uint8_t* img, int w, int h, int stride, float* eig);
int orc_gftt(const uint8_t* img, int w, int h, int stride, int max_corners, double quality, double min_distance, float* out_xy);
void orc_corner_subpix(const uint8_t* img, int w, int h, int stride, float* pts_xy, int n, int win);
/* measurement only (tests/test_opencv_distance.py): OpenCV's own summation orders where the oracle fixes a canonical one */
void orc_min_eig_cvorder(const uint8_t* img, int w, int h, int stride, float* eig);
void orc_corner_subpix_rowmajor(const uint8_t* img, int w, int h, int stride, float* pts_xy, int n, int win);
/* s = 1 on the first image, 2 on refills (Tracker.cc:207,350); out_xy holds n_features points; returns the count */
int orc_detect(const rvio_config* cfg, const uint8_t* img, int stride, int s, float* out_xy);

/* ---- T1/T6: Tracker::track, Tracker.cc:179-396.  cand_xy == NULL (with an image) runs the detector above on the
 * (equalised) image like the reference; a non-NULL list replaces it (detector supplied by the caller) ---- */
typedef struct orc_tracker orc_tracker;
orc_tracker* orc_tracker_create(const rvio_config* cfg);
void orc_tracker_destroy(orc_tracker*);
void orc_tracker_track(orc_tracker*, const uint8_t* img, int stride, const rvio_imu* imu, int m,
                       const float* cand_xy, int n_cand, rvio_frame_info* info);
/* direct-track mode: caller supplies the KLT result (pixel positions + status) */
void orc_tracker_track_points(orc_tracker*, const float* tracked_xy, const unsigned char* status,
                              const rvio_imu* imu, int m, const float* cand_xy, int n_cand, rvio_frame_info* info);
/* outputs: n_feat, types, len, meas[n_feat][max_track_len][2]; buffers sized ceil(F/2) */
void orc_tracker_get_tracks(orc_tracker*, int32_t* n_feat, unsigned char* types, int32_t* len, float* meas);
void orc_tracker_get_points(orc_tracker*, int32_t* n, float* xy, int32_t* hist_len);

/* ---- whole System::MonoVIO timed body (System.cc:253-367) ---- */
typedef struct orc_system orc_system;
orc_system* orc_system_create(const rvio_config* cfg);
void orc_system_destroy(orc_system*);
void orc_system_set_state(orc_system*, const double* x, int xdim, const double* P, int d);
void orc_system_get_state(orc_system*, double* x, int* xdim, double* P, int* d);
/* one frame; t_ms[0]=track, t_ms[1]=propagate, t_ms[2]=update, t_ms[3]=augment+compose */
orc_tracker* orc_system_tracker(orc_system*);
/* analysis: run Updater::update in the information form [A|b] = Hw^T [Hw | r] (orc_update_local/global), which keeps the
 * rows the reference's rank truncation (Updater.cc:516-529) drops; last_rank = nRank of the literal path's last update */
void orc_system_set_information_form(orc_system*, int on);
int orc_system_last_rank(orc_system*);
/* img==NULL selects direct-track mode (tracked_xy/status given) */
void orc_system_frame(orc_system*, const uint8_t* img, int stride, const float* tracked_xy, const unsigned char* status,
                      const rvio_imu* imu, int m,
                      const float* cand_xy, int n_cand, rvio_frame_info* info, double t_ms[4],
                      double pose_p[3], double pose_q[4]);

#ifdef __cplusplus
}
#endif
#endif
