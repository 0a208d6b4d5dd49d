// oracle/ref_capi.cpp — TEST INFRASTRUCTURE: the C API of oracle/_ref/libref.so.
//
// libref.so = the reference's OWN sources (/root/reference/src/rvio/{Updater,PreIntegrator,Ransac,InputBuffer,
// FeatureDetector,Tracker,System}.cc + util/Numerics.h), compiled unmodified from where they lie against the header
// shim in oracle/refshim/ (mini Eigen / OpenCV containers / inert ROS types), plus this file.  It exists to pin
// oracle/filter.cpp + oracle/frontend.cpp — the restatement every parity test of this repository rests on — against
// the code it restates (tests/test_ref_pins.py).  It is built only where /root/reference exists (oracle/Makefile,
// target `ref`), lands in the git-ignored oracle/_ref/, and is never linked, loaded or executed by the product.
//
// What it pins: everything the reference itself wrote (Numerics.h, propagate, RANSAC, the whole Updater, Tracker's
// book-keeping, FeatureDetector's grid selection, System::MonoVIO's sequencing / augmentation / composition).
// What it cannot pin: Eigen's and OpenCV's internal arithmetic.  Eigen is replaced by refshim/mini_eigen.hpp (written
// from SURVEY.md appendix C); the OpenCV image algorithms forward to liborc.so's restatements below.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <string>
#include <vector>

#include "refshim/mini_ros.hpp"
#include "rvio_oracle.h"

#include <deque>
#include <mutex>

// -I$(REF)/src (oracle/Makefile): the reference's own headers, from where they lie
#include "util/Numerics.h"
#define private public  // System::xkk/Pkk/mbIsReady, Tracker's lists, Ransac's model are private; read access for the snapshots below
#define protected public
#include "rvio/PreIntegrator.h"
#include "rvio/Ransac.h"
#include "rvio/Updater.h"
#include "rvio/Tracker.h"
#include "rvio/System.h"
#undef private
#undef protected

// ---------------------------------------------------------------- shim state
namespace cv {
RefshimConfigTable& refshim_config_table() {
    static RefshimConfigTable t;
    return t;
}
}  // namespace cv
namespace refshim {
std::map<std::string, int>& debug_counts() {
    static std::map<std::string, int> m;
    return m;
}
visualization_msgs::Marker& last_marker() {
    static visualization_msgs::Marker m;
    return m;
}
nav_msgs::Odometry& last_odometry() {
    static nav_msgs::Odometry m;
    return m;
}
// direct-track mode: results the image algorithms return instead of running
struct Inject {
    bool klt = false, det = false;
    std::vector<float> klt_xy;
    std::vector<unsigned char> klt_status;
    std::vector<float> det_xy;
} g_inject;
rvio_config g_cfg;
}  // namespace refshim

static void fill_table(const rvio_config* c) {
    refshim::g_cfg = *c;
    std::map<std::string, double>& n = cv::refshim_config_table().num;
    n.clear();
    n["IMU.dps"] = c->imu_rate; n["IMU.sigma_g"] = c->sigma_g; n["IMU.sigma_wg"] = c->sigma_wg; n["IMU.sigma_a"] = c->sigma_a;
    n["IMU.sigma_wa"] = c->sigma_wa; n["IMU.nG"] = c->gravity; n["IMU.nSmallAngle"] = c->small_angle;
    n["Camera.width"] = c->width; n["Camera.height"] = c->height;
    n["Camera.fx"] = c->fx; n["Camera.fy"] = c->fy; n["Camera.cx"] = c->cx; n["Camera.cy"] = c->cy;
    n["Camera.k1"] = c->k1; n["Camera.k2"] = c->k2; n["Camera.p1"] = c->p1; n["Camera.p2"] = c->p2; n["Camera.k3"] = c->k3;
    n["Camera.sigma_px"] = c->sigma_px; n["Camera.sigma_py"] = c->sigma_py;
    n["Camera.Fisheye"] = c->fisheye; n["Camera.RGB"] = 0; n["Camera.fps"] = 20; n["Camera.nTimeOffset"] = 0;
    n["Tracker.nFeatures"] = c->n_features; n["Tracker.nMaxTrackingLength"] = c->max_track_len;
    n["Tracker.nMinTrackingLength"] = c->min_track_len; n["Tracker.nMinDist"] = c->min_dist; n["Tracker.nQualLvl"] = c->qual_lvl;
    n["Tracker.nBlockSizeX"] = c->block_x; n["Tracker.nBlockSizeY"] = c->block_y;
    n["Tracker.EnableEqualizer"] = c->enable_equalizer; n["Tracker.UseSampson"] = c->use_sampson; n["Tracker.nInlierThrd"] = c->inlier_thr;
    n["INI.nThresholdAngle"] = c->ini_thr_angle; n["INI.nThresholdDispl"] = c->ini_thr_displ;
    n["INI.EnableAlignment"] = c->ini_enable_alignment; n["INI.RecordOutputs"] = 0;
    n["Landmark.nScale"] = 0.03; n["Landmark.nPubRate"] = 5;
    for (int i = 0; i < 16; ++i) cv::refshim_config_table().T_BC0[i] = c->T_bc[i];
}

// ---------------------------------------------------------------- OpenCV image algorithms -> liborc.so restatements
#define REFSHIM_REQUIRE(cond)                                                                         \
    do {                                                                                              \
        if (!(cond)) {                                                                                \
            std::fprintf(stderr, "refshim: the reference passed an unexpected parameter: %s\n", #cond); \
            std::abort();                                                                             \
        }                                                                                             \
    } while (0)

namespace cv {
namespace {
class ClaheFwd : public CLAHE {
public:
    void apply(const Mat& src, const Mat& dst) override {
        if (refshim::g_inject.klt || refshim::g_inject.det) return;  // direct-track mode: no image
        REFSHIM_REQUIRE(src.type() == CV_8UC1 && src.ptr() == dst.ptr());
        std::vector<uint8_t> out((size_t)src.rows * src.cols);
        orc_clahe(src.ptr(), src.cols, src.rows, src.cols, out.data());
        std::memcpy(dst.ptr(), out.data(), out.size());
    }
};
}  // namespace
Ptr<CLAHE> createCLAHE(double clipLimit, Size tile) {
    REFSHIM_REQUIRE(clipLimit == 3.0 && tile.width == 5 && tile.height == 5);  // Tracker.cc:200 == oracle/frontend.cpp clahe_apply(…, 3.0, 5, 5)
    return std::make_shared<ClaheFwd>();
}
void cvtColor(const Mat& src, const Mat&, int) { REFSHIM_REQUIRE(src.channels() == 1 && "mono8 input only"); }
void cvtColor(const Mat& src, Mat& dst, int code) {
    REFSHIM_REQUIRE(code == CV_GRAY2BGR);  // DisplayTrack / DisplayNewer: rviz debug image, out of scope
    (void)src; (void)dst;
}
void calcOpticalFlowPyrLK(const Mat& prev, const Mat& next, std::vector<Point2f>& p0, std::vector<Point2f>& p1,
                          std::vector<unsigned char>& status, std::vector<float>& err, Size win, int maxLevel,
                          TermCriteria crit, int flags, double minEig) {
    // Tracker.cc:237-244 == the constants oracle/frontend.cpp's lk_point hard-codes
    REFSHIM_REQUIRE(win.width == 15 && win.height == 15 && maxLevel == 3 && flags == 0 && minEig == 1e-3);
    REFSHIM_REQUIRE(crit.type == (TermCriteria::COUNT + TermCriteria::EPS) && crit.maxCount == 30 && crit.epsilon == 1e-2);
    const int n = (int)p0.size();
    p1.resize(n); status.resize(n); err.assign(n, 0.f);
    if (refshim::g_inject.klt) {
        REFSHIM_REQUIRE((int)refshim::g_inject.klt_status.size() == n);
        for (int i = 0; i < n; ++i) {
            p1[i] = Point2f(refshim::g_inject.klt_xy[2 * i], refshim::g_inject.klt_xy[2 * i + 1]);
            status[i] = refshim::g_inject.klt_status[i];
        }
        return;
    }
    REFSHIM_REQUIRE(prev.type() == CV_8UC1 && next.type() == CV_8UC1 && prev.rows == next.rows && prev.cols == next.cols);
    static_assert(sizeof(Point2f) == 8, "Point2f is two packed floats");
    if (n) orc_klt(prev.ptr(), next.ptr(), next.cols, next.rows, next.cols, &p0[0].x, n, &p1[0].x, status.data());
}
static void undistort_fwd(const Mat& src, Mat& dst, const Mat& K, const Mat& D, int fisheye) {
    REFSHIM_REQUIRE(src.type() == CV_32FC2 && src.cols == 1 && K.type() == CV_32FC1 && D.type() == CV_32FC1);
    rvio_config c;
    std::memset(&c, 0, sizeof c);
    c.fx = K.at<float>(0, 0); c.fy = K.at<float>(1, 1); c.cx = K.at<float>(0, 2); c.cy = K.at<float>(1, 2);
    c.k1 = D.at<float>(0); c.k2 = D.at<float>(1); c.p1 = D.at<float>(2); c.p2 = D.at<float>(3);
    c.k3 = D.rows >= 5 ? D.at<float>(4) : 0.f;
    c.fisheye = fisheye;
    std::vector<float> out((size_t)2 * src.rows);
    orc_undistort(&c, &src.at<float>(0), src.rows, out.data());
    if (dst.ptr() != src.ptr()) dst = Mat(src.rows, 1, CV_32FC2);
    std::memcpy(dst.ptr(), out.data(), out.size() * sizeof(float));
}
void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& D) { undistort_fwd(src, dst, K, D, 0); }
namespace fisheye { void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& D) { undistort_fwd(src, dst, K, D, 1); } }
void goodFeaturesToTrack(const Mat& im, std::vector<Point2f>& corners, int maxCorners, double quality, double minDistance) {
    corners.clear();
    if (refshim::g_inject.det) {  // the caller's corner list stands in for DetectWithSubPix's result
        const int n = std::min((int)refshim::g_inject.det_xy.size() / 2, maxCorners);
        for (int i = 0; i < n; ++i) corners.push_back(Point2f(refshim::g_inject.det_xy[2 * i], refshim::g_inject.det_xy[2 * i + 1]));
        return;
    }
    REFSHIM_REQUIRE(im.type() == CV_8UC1);
    std::vector<float> xy((size_t)2 * maxCorners);
    const int n = orc_gftt(im.ptr(), im.cols, im.rows, im.cols, maxCorners, quality, minDistance, xy.data());
    for (int i = 0; i < n; ++i) corners.push_back(Point2f(xy[2 * i], xy[2 * i + 1]));
}
void cornerSubPix(const Mat& im, std::vector<Point2f>& corners, Size win, Size zero, TermCriteria crit) {
    if (refshim::g_inject.det) return;
    REFSHIM_REQUIRE(win.width == win.height && zero.width == -1 && zero.height == -1);
    REFSHIM_REQUIRE(crit.type == (TermCriteria::COUNT + TermCriteria::EPS) && crit.maxCount == 30 && crit.epsilon == 1e-2);
    if (!corners.empty()) orc_corner_subpix(im.ptr(), im.cols, im.rows, im.cols, &corners[0].x, (int)corners.size(), win.width);
}
}  // namespace cv

// ---------------------------------------------------------------- helpers
static std::list<RVIO::ImuData*> make_imu_list(const rvio_imu* imu, int m, std::vector<RVIO::ImuData>& store) {
    store.resize(m);
    std::list<RVIO::ImuData*> l;
    for (int i = 0; i < m; ++i) {
        store[i].AngularVel = Eigen::Vector3d(imu[i].w[0], imu[i].w[1], imu[i].w[2]);
        store[i].LinearAccel = Eigen::Vector3d(imu[i].a[0], imu[i].a[1], imu[i].a[2]);
        store[i].Timestamp = imu[i].t;
        store[i].TimeInterval = imu[i].dt;
        l.push_back(&store[i]);
    }
    return l;
}
static Eigen::VectorXd vec_in(const double* x, int n) {
    Eigen::VectorXd v(n, 1);
    for (int i = 0; i < n; ++i) v(i) = x[i];
    return v;
}
static Eigen::MatrixXd mat_in(const double* P, int d) {  // col-major
    Eigen::MatrixXd M(d, d);
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) M(i, j) = P[i + (size_t)j * d];
    return M;
}
static void mat_out(const Eigen::MatrixXd& M, double* P) {
    for (int j = 0; j < M.cols(); ++j)
        for (int i = 0; i < M.rows(); ++i) P[i + (size_t)j * M.rows()] = M(i, j);
}

extern "C" {

int ref_debug_count(const char* fmt) {
    std::map<std::string, int>::const_iterator it = refshim::debug_counts().find(fmt);
    return it == refshim::debug_counts().end() ? 0 : it->second;
}
void ref_debug_reset(void) { refshim::debug_counts().clear(); }

// ---- N1: util/Numerics.h:30-224
void ref_quat_mul(const double q1[4], const double q2[4], double out[4]) {
    Eigen::Vector4d q = QuatMul(Eigen::Vector4d(q1[0], q1[1], q1[2], q1[3]), Eigen::Vector4d(q2[0], q2[1], q2[2], q2[3]));
    for (int i = 0; i < 4; ++i) out[i] = q(i);
}
void ref_quat_to_rot(const double q[4], double R[9]) {
    Eigen::Matrix3d M = QuatToRot(Eigen::Vector4d(q[0], q[1], q[2], q[3]));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = M(i, j);
}
void ref_rot_to_quat(const double R[9], double q[4]) {
    Eigen::MatrixXd M(3, 3);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M(i, j) = R[3 * i + j];
    Eigen::Vector4d v = RotToQuat(M);
    for (int i = 0; i < 4; ++i) q[i] = v(i);
}
double ref_chi2_95(int dof) { return CHI_THRESHOLD[dof - 1]; }

// ---- System::initialize, System.cc:115-170
void ref_initialize(const rvio_config* cfg, const double w[3], const double a[3], int n_imu, double x[26], double P[24 * 24]) {
    fill_table(cfg);
    std::streambuf* keep = std::cout.rdbuf(nullptr);  // the constructor's welcome banner
    RVIO::System S("refshim");
    std::cout.rdbuf(keep);
    S.initialize(Eigen::Vector3d(w[0], w[1], w[2]), Eigen::Vector3d(a[0], a[1], a[2]), n_imu, cfg->ini_enable_alignment != 0);
    for (int i = 0; i < 26; ++i) x[i] = S.xkk(i);
    mat_out(S.Pkk, P);
}

// ---- P1: PreIntegrator::propagate, PreIntegrator.cc:51-194 (P in place, like the reference's Pkk)
void ref_propagate(const rvio_config* cfg, const double* x, int xdim, double* P, int d, const rvio_imu* imu, int m, double* x_out) {
    fill_table(cfg);
    cv::FileStorage fs;
    RVIO::PreIntegrator pre(fs);
    Eigen::VectorXd xkk = vec_in(x, xdim);
    Eigen::MatrixXd Pkk = mat_in(P, d);
    std::vector<RVIO::ImuData> store;
    std::list<RVIO::ImuData*> l = make_imu_list(imu, m, store);
    pre.propagate(xkk, Pkk, l);
    for (int i = 0; i < xdim; ++i) x_out[i] = pre.xk1k(i);
    mat_out(Pkk, P);  // the caller's Pkk is mutated (PreIntegrator.cc:142,189-193); Pk1k is a copy of it
}

// ---- U1..U10: Updater::update, Updater.cc:72-628.
// info[0] = points of the published landmark cloud (= accepted features with rho > 0, Updater.cc:430-448),
// info[1] = "Failed in Mahalanobis distance test!" count, info[2] = "Invalid inverse-depth feature estimate (0|1)!" count,
// info[3] = 1 unless "Too few measurements for update!", info[4] = "Hf is rank deficient!" count, info[5] = "Hw is rank deficient!" count.
// cloud (may be NULL): 3 doubles per published point.
void ref_update(const rvio_config* cfg, const double* x, int xdim, const double* P, int d, const rvio_tracks* tracks,
                double* x_out, double* P_out, int32_t info[6], double* cloud) {
    fill_table(cfg);
    cv::FileStorage fs;
    RVIO::Updater upd(fs);
    Eigen::VectorXd xk1k = vec_in(x, xdim);
    Eigen::MatrixXd Pk1k = mat_in(P, d);
    std::vector<unsigned char> types(tracks->types, tracks->types + tracks->n_feat);
    // Tracker hands over a vector sized ceil(F/2) whose first types.size() entries are valid (Tracker.cc:272-274)
    std::vector<std::list<cv::Point2f> > meas(std::max(tracks->n_feat, (int)std::ceil(.5 * cfg->n_features)));
    for (int f = 0; f < tracks->n_feat; ++f)
        for (int k = 0; k < tracks->len[f]; ++k) {
            const float* p = tracks->meas + ((size_t)f * tracks->max_len + k) * 2;
            meas[f].push_back(cv::Point2f(p[0], p[1]));
        }
    refshim::debug_counts().clear();
    refshim::last_marker().points.clear();
    upd.update(xk1k, Pk1k, types, meas);
    for (int i = 0; i < xdim; ++i) x_out[i] = upd.xk1k1(i);
    mat_out(upd.Pk1k1, P_out);
    const std::vector<geometry_msgs::Point>& pts = refshim::last_marker().points;
    info[0] = (int)pts.size();
    info[1] = ref_debug_count("Failed in Mahalanobis distance test!");
    info[2] = ref_debug_count("Invalid inverse-depth feature estimate (0)!") + ref_debug_count("Invalid inverse-depth feature estimate (1)!");
    info[3] = ref_debug_count("Too few measurements for update!") ? 0 : 1;
    info[4] = ref_debug_count("Hf is rank deficient!");
    info[5] = ref_debug_count("Hw is rank deficient!");
    if (cloud)
        for (size_t i = 0; i < pts.size(); ++i) { cloud[3 * i] = pts[i].x; cloud[3 * i + 1] = pts[i].y; cloud[3 * i + 2] = pts[i].z; }
}

// ---- T5: Ransac::FindInliers, Ransac.cc:180-247.  p1/p2: 3 x n col-major; flags in/out; seed -> srand (the reference never seeds: 1)
int ref_ransac(const rvio_config* cfg, const double* p1, const double* p2, int n, const rvio_imu* imu, int m,
               unsigned char* flags, unsigned seed, int32_t* pairs /* 32 or NULL */, int32_t* votes /* 16 or NULL */) {
    fill_table(cfg);
    cv::FileStorage fs;
    RVIO::Ransac R(fs);
    Eigen::MatrixXd P1(3, n), P2(3, n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < 3; ++i) { P1(i, j) = p1[3 * j + i]; P2(i, j) = p2[3 * j + i]; }
    std::vector<unsigned char> f(flags, flags + n);
    std::vector<RVIO::ImuData> store;
    std::list<RVIO::ImuData*> l = make_imu_list(imu, m, store);
    srand(seed);
    const int ninl = R.FindInliers(P1, P2, l, f);
    std::memcpy(flags, f.data(), n);
    if (pairs)
        for (int i = 0; i < 16; ++i) { pairs[2 * i] = R.mRansacModel.twoPoints(i, 0); pairs[2 * i + 1] = R.mRansacModel.twoPoints(i, 1); }
    if (votes)
        for (int i = 0; i < 16; ++i) votes[i] = R.mRansacModel.nInliers(i);
    return ninl;
}

// ---- S1 + S2: System.cc's augmentation / slide / composition block, lifted VERBATIM by oracle/Makefile into
// _ref/system_augment_compose.inc (everything between the "State augmentation" comment and the t3 time stamp).
void ref_augment_compose(const rvio_config* cfg, double* x, int* xdim, double* P, int* d, int do_augment, double pose_p[3], double pose_q[4]) {
    Eigen::VectorXd xkk = vec_in(x, *xdim);
    Eigen::MatrixXd Pkk = mat_in(P, *d);
    int nCloneStates = (*xdim - 26) / 7;
    const int mnSlidingWindowSize = cfg->max_track_len - 1;  // System.cc:71-72
    const int nImageCountAfterInit = do_augment ? 2 : 1;
#include "_ref/system_augment_compose.inc"
    *xdim = xkk.rows();
    *d = Pkk.rows();
    for (int i = 0; i < *xdim; ++i) x[i] = xkk(i);
    mat_out(Pkk, P);
    for (int i = 0; i < 3; ++i) pose_p[i] = pGk(i);
    for (int i = 0; i < 4; ++i) pose_q[i] = qkG(i);
    (void)vk;
}

// ---- T1/T6/T7-grid: Tracker::track, Tracker.cc:179-396
struct ref_tracker {
    RVIO::Tracker* T;
    rvio_config cfg;
};
ref_tracker* ref_tracker_create(const rvio_config* cfg) {
    fill_table(cfg);
    cv::FileStorage fs;
    ref_tracker* t = new ref_tracker();
    t->cfg = *cfg;
    t->T = new RVIO::Tracker(fs);
    srand(1);  // a process that never calls srand() draws the seed-1 sequence (Ransac.cc:63,69)
    return t;
}
void ref_tracker_destroy(ref_tracker* t) { delete t->T; delete t; }

static void set_injection(const float* tracked_xy, const unsigned char* status, int n_tracked, const float* cand_xy, int n_cand) {
    refshim::Inject& I = refshim::g_inject;
    I.klt = tracked_xy != nullptr || status != nullptr;
    if (I.klt) { I.klt_xy.assign(tracked_xy, tracked_xy + 2 * n_tracked); I.klt_status.assign(status, status + n_tracked); }
    I.det = cand_xy != nullptr;
    if (I.det) I.det_xy.assign(cand_xy, cand_xy + 2 * n_cand);
}
static cv::Mat image_in(const rvio_config& c, const uint8_t* img, int stride) {
    cv::Mat im(c.height, c.width, CV_8UC1);
    if (img)
        for (int r = 0; r < c.height; ++r) std::memcpy(im.ptr() + (size_t)r * c.width, img + (size_t)r * stride, c.width);
    return im;
}
// img != NULL: the image path (cand_xy == NULL runs the detector); img == NULL: direct-track mode (tracked_xy / status given)
void ref_tracker_track(ref_tracker* t, const uint8_t* img, int stride, const float* tracked_xy, const unsigned char* status,
                       const rvio_imu* imu, int m, const float* cand_xy, int n_cand) {
    const int n_in = t->T->mbIsTheFirstImage ? 0 : t->T->mnFeatsToTrack;
    set_injection(img ? nullptr : tracked_xy, img ? nullptr : status, n_in, cand_xy, n_cand);
    if (!img && !refshim::g_inject.klt) { refshim::g_inject.klt = true; refshim::g_inject.klt_xy.clear(); refshim::g_inject.klt_status.clear(); }
    cv::Mat im = image_in(t->cfg, img, stride);
    std::vector<RVIO::ImuData> store;
    std::list<RVIO::ImuData*> l = make_imu_list(imu, m, store);
    t->T->track(im, l);
    set_injection(nullptr, nullptr, 0, nullptr, 0);
}
static void tracks_out(RVIO::Tracker* T, int max_len, int32_t* n_feat, unsigned char* types, int32_t* len, float* meas) {
    *n_feat = (int)T->mvFeatTypesForUpdate.size();
    for (int f = 0; f < *n_feat; ++f) {
        types[f] = T->mvFeatTypesForUpdate[f];
        len[f] = (int)T->mvlFeatMeasForUpdate[f].size();
        int k = 0;
        for (const cv::Point2f& p : T->mvlFeatMeasForUpdate[f]) {
            meas[((size_t)f * max_len + k) * 2] = p.x;
            meas[((size_t)f * max_len + k) * 2 + 1] = p.y;
            ++k;
        }
    }
}
void ref_tracker_get_tracks(ref_tracker* t, int32_t* n_feat, unsigned char* types, int32_t* len, float* meas) {
    tracks_out(t->T, t->cfg.max_track_len, n_feat, types, len, meas);
}
static void points_out(RVIO::Tracker* T, int32_t* n, float* xy, int32_t* hist_len) {
    *n = T->mbIsTheFirstImage ? 0 : T->mnFeatsToTrack;
    for (int i = 0; i < *n; ++i) {
        xy[2 * i] = T->mvFeatsToTrack[i].x;
        xy[2 * i + 1] = T->mvFeatsToTrack[i].y;
        hist_len[i] = (int)T->mvlTrackingHistory[T->mvInlierIndices[i]].size();
    }
}
void ref_tracker_get_points(ref_tracker* t, int32_t* n, float* xy, int32_t* hist_len) { points_out(t->T, n, xy, hist_len); }

// ---- the whole System::MonoVIO, System.cc:173-437.  MonoVIO keeps its clone / image counters in function-local statics
// (System.cc:175-176,185-187), so ONE ref_system per loaded copy of libref.so (tests load a private copy per sequence).
struct ref_system {
    RVIO::System* S;
    rvio_config cfg;
    double t_img;
    int frames;
};
ref_system* ref_system_create(const rvio_config* cfg) {
    static int made = 0;
    if (made++) { std::fprintf(stderr, "ref_system_create: one System per loaded libref.so (MonoVIO's static counters)\n"); return nullptr; }
    fill_table(cfg);
    ref_system* s = new ref_system();
    s->cfg = *cfg;
    std::streambuf* keep = std::cout.rdbuf(nullptr);  // the constructor's welcome banner
    s->S = new RVIO::System("refshim");
    std::cout.rdbuf(keep);
    s->t_img = 0;
    s->frames = 0;
    srand(1);
    return s;
}
void ref_system_destroy(ref_system* s) { delete s->S; delete s; }
// start from a given (x, P) instead of the stationary-start detector (System.cc:183-249), like orc_system_set_state; no clones yet
int ref_system_set_state(ref_system* s, const double* x, int xdim, const double* P, int d) {
    if (xdim != 26 || d != 24 || s->frames) return -1;
    s->S->xkk = vec_in(x, xdim);
    s->S->Pkk = mat_in(P, d);
    s->S->mbIsReady = true;
    return 0;
}
void ref_system_get_state(ref_system* s, double* x, int* xdim, double* P, int* d) {
    *xdim = s->S->xkk.rows();
    *d = s->S->Pkk.rows();
    for (int i = 0; i < *xdim; ++i) x[i] = s->S->xkk(i);
    mat_out(s->S->Pkk, P);
}
// one camera frame through PushImuData / PushImageData / MonoVIO; returns 1 if MonoVIO consumed it.  info[0] = landmark-cloud
// points of this frame's update, [1] gate rejects, [2] invalid estimates, [3] updated (0 = too few / no update ran), [4] n tracked out
int ref_system_frame(ref_system* s, const uint8_t* img, int stride, const float* tracked_xy, const unsigned char* status,
                     const rvio_imu* imu, int m, const float* cand_xy, int n_cand, int32_t info[5], double pose_p[3], double pose_q[4]) {
    RVIO::Tracker* T = s->S->mpTracker;
    const int n_in = T->mbIsTheFirstImage ? 0 : T->mnFeatsToTrack;
    set_injection(img ? nullptr : tracked_xy, img ? nullptr : status, n_in, cand_xy, n_cand);
    if (!img && !refshim::g_inject.klt) { refshim::g_inject.klt = true; refshim::g_inject.klt_xy.clear(); refshim::g_inject.klt_status.clear(); }
    std::vector<RVIO::ImuData*> owned;
    for (int i = 0; i < m; ++i) {
        RVIO::ImuData* d = new RVIO::ImuData();
        d->AngularVel = Eigen::Vector3d(imu[i].w[0], imu[i].w[1], imu[i].w[2]);
        d->LinearAccel = Eigen::Vector3d(imu[i].a[0], imu[i].a[1], imu[i].a[2]);
        d->Timestamp = imu[i].t;
        d->TimeInterval = imu[i].dt;
        s->S->PushImuData(d);
        owned.push_back(d);
    }
    RVIO::ImageData* I = new RVIO::ImageData();
    I->Image = image_in(s->cfg, img, stride);
    I->Timestamp = imu[m - 1].t;  // the frame's last IMU sample carries the image stamp (InputBuffer.cc:53-81)
    s->S->PushImageData(I);
    refshim::debug_counts().clear();
    refshim::last_marker().points.clear();
    const int before = (int)s->S->mpInputBuffer->mlImageFIFO.size();
    s->S->MonoVIO();
    const int consumed = before - (int)s->S->mpInputBuffer->mlImageFIFO.size();
    s->frames++;
    set_injection(nullptr, nullptr, 0, nullptr, 0);
    if (info) {
        info[0] = (int)refshim::last_marker().points.size();
        info[1] = ref_debug_count("Failed in Mahalanobis distance test!");
        info[2] = ref_debug_count("Invalid inverse-depth feature estimate (0)!") + ref_debug_count("Invalid inverse-depth feature estimate (1)!");
        info[3] = ref_debug_count("Too few measurements for update!") ? 0 : 1;
        info[4] = T->mbIsTheFirstImage ? 0 : T->mnFeatsToTrack;
    }
    const nav_msgs::Odometry& o = refshim::last_odometry();
    if (pose_p) { pose_p[0] = o.pose.pose.position.x; pose_p[1] = o.pose.pose.position.y; pose_p[2] = o.pose.pose.position.z; }
    if (pose_q) { pose_q[0] = o.pose.pose.orientation.x; pose_q[1] = o.pose.pose.orientation.y; pose_q[2] = o.pose.pose.orientation.z; pose_q[3] = o.pose.pose.orientation.w; }
    if (consumed == 1 && s->S->mpInputBuffer->mlImuFIFO.empty()) {  // the reference never frees its packets
        for (RVIO::ImuData* d : owned) delete d;
        delete I;
    }
    return consumed;
}
void ref_system_get_tracks(ref_system* s, int32_t* n_feat, unsigned char* types, int32_t* len, float* meas) {
    tracks_out(s->S->mpTracker, s->cfg.max_track_len, n_feat, types, len, meas);
}
void ref_system_get_points(ref_system* s, int32_t* n, float* xy, int32_t* hist_len) { points_out(s->S->mpTracker, n, xy, hist_len); }

}  // extern "C"
