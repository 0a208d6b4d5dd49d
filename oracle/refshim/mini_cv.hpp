// oracle/refshim/mini_cv.hpp — TEST INFRASTRUCTURE (builds oracle/_ref), not product code.
//
// The OpenCV surface the reference's sources touch, so that they compile
// unmodified into oracle/_ref/libref.so.  OpenCV is not installed here and is not
// copied: containers (Mat, Point_, Size, FileStorage ...) are minimal
// re-implementations; the IMAGE ALGORITHMS the reference calls
// (createCLAHE/apply, calcOpticalFlowPyrLK, undistortPoints, goodFeaturesToTrack,
// cornerSubPix) forward to the oracle's restatements in liborc.so
//  after asserting that the reference passed exactly the
// parameters the oracle hard-codes — so what _ref pins is the reference's OWN code
// (book-keeping, grid selection, RANSAC, filter); the OpenCV-internal arithmetic
// stays "restated, unpinned" and is said so in DESIGN.md.
#ifndef RVIO_REFSHIM_MINI_CV_HPP
#define RVIO_REFSHIM_MINI_CV_HPP
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)

enum { CV_BGR2GRAY = 6, CV_RGB2GRAY = 7, CV_GRAY2BGR = 8, CV_BGRA2GRAY = 10, CV_RGBA2GRAY = 11 };

struct CvScalar { double val[4]; };
inline CvScalar cvScalar(double a, double b, double c, double d) { CvScalar s = {{a, b, c, d}}; return s; }
#define CV_RGB(r, g, b) cvScalar((b), (g), (r), 0)

namespace cv {

template <class T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
template <class T> Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x - b.x, a.y - b.y); }
template <class T> Point_<T> operator+(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x + b.x, a.y + b.y); }
typedef Point_<float> Point2f;
typedef Point_<int> Point;
// cv::norm(Point_<T>) (core/types.hpp): sqrt in double of the double-converted squares
template <class T> double norm(const Point_<T>& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};
struct Scalar {
    double val[4];
    Scalar(const CvScalar& s) { std::memcpy(val, s.val, sizeof val); }
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
};
struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = 1, EPS = 2 };
    int type, maxCount;
    double epsilon;
    TermCriteria(int t, int n, double e) : type(t), maxCount(n), epsilon(e) {}
};

template <class T> using Ptr = std::shared_ptr<T>;

// continuous, reference-counted, row-major; just what the reference uses
class Mat {
public:
    int rows, cols;
    Mat() : rows(0), cols(0), type_(0) {}
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type), buf_(std::make_shared<std::vector<unsigned char> >((size_t)r * c * elemSize(), 0)) {}
    static Mat eye(int r, int c, int type) {
        Mat m(r, c, type);
        assert(depth_of(type) == CV_32F && channels_of(type) == 1);
        for (int i = 0; i < (r < c ? r : c); ++i) m.at<float>(i, i) = 1.f;
        return m;
    }
    int type() const { return type_; }
    int depth() const { return depth_of(type_); }
    int channels() const { return channels_of(type_); }
    bool empty() const { return !buf_ || rows * cols == 0; }
    size_t elemSize() const { return (size_t)channels() * (depth() == CV_8U ? 1 : depth() == CV_32F ? 4 : 8); }
    unsigned char* ptr() const { return buf_ ? buf_->data() : nullptr; }
    template <class T> T& at(int i, int j) const { return reinterpret_cast<T*>(ptr())[(size_t)i * cols * (elemSize() / sizeof(T)) + j]; }
    template <class T> T& at(int i) const { return reinterpret_cast<T*>(ptr())[i]; }
    void copyTo(Mat& dst) const {
        dst.rows = rows; dst.cols = cols; dst.type_ = type_;
        dst.buf_ = buf_ ? std::make_shared<std::vector<unsigned char> >(*buf_) : nullptr;
    }
    Mat clone() const { Mat m; copyTo(m); return m; }
    // reshape(cn): same data, new channel count, same rows
    Mat reshape(int cn) const {
        Mat m = *this;
        int total_ch = cols * channels();
        assert(total_ch % cn == 0);
        m.cols = total_ch / cn;
        m.type_ = CV_MAKETYPE(depth(), cn);
        return m;
    }
    // Mat::resize(sz): change the number of rows, keeping the leading data
    void resize(size_t nrows) {
        auto nb = std::make_shared<std::vector<unsigned char> >(nrows * cols * elemSize(), 0);
        if (buf_) std::memcpy(nb->data(), buf_->data(), std::min(nb->size(), buf_->size()));
        buf_ = nb;
        rows = (int)nrows;
    }
private:
    static int depth_of(int t) { return t & 7; }
    static int channels_of(int t) { return (t >> 3) + 1; }
    int type_;
    std::shared_ptr<std::vector<unsigned char> > buf_;
};

// ---- FileStorage: backed by the key/value table refshim_set_config() fills from an rvio_config (ref_capi.cpp)
struct RefshimConfigTable {
    std::map<std::string, double> num;
    double T_BC0[16];
};
RefshimConfigTable& refshim_config_table();

class FileNode {
    const RefshimConfigTable* t_;
    std::string key_;
    double get() const {
        std::map<std::string, double>::const_iterator it = t_->num.find(key_);
        assert(it != t_->num.end() && "refshim: config key not provided");
        return it == t_->num.end() ? 0.0 : it->second;
    }
public:
    FileNode(const RefshimConfigTable* t, const std::string& k) : t_(t), key_(k) {}
    operator int() const { return (int)std::lrint(get()); }  // cv::FileNode: cvRound for reals
    operator float() const { return (float)get(); }
    operator double() const { return get(); }
    void operator>>(Mat& m) const {
        assert(key_ == "Camera.T_BC0");
        // cv::read(FileNode, Mat&) REPLACES the destination with the stored matrix: config/rvio_euroc.yaml:56-62 stores `dt: d`, so
        // the CV_32F header the caller pre-allocated (Updater.cc:46, Ransac.cc:41) becomes CV_64F and cv2eigen copies doubles
        m = Mat(4, 4, CV_64F);
        for (int i = 0; i < 16; ++i) m.at<double>(i) = t_->T_BC0[i];
    }
};
class FileStorage {
    bool open_;
public:
    enum { READ = 0 };
    FileStorage() : open_(true) {}
    FileStorage(const std::string&, int) : open_(true) {}
    bool isOpened() const { return open_; }
    FileNode operator[](const char* key) const { return FileNode(&refshim_config_table(), key); }
    FileNode operator[](const std::string& key) const { return FileNode(&refshim_config_table(), key); }
};

// ---- image algorithms: forwarded to liborc.so's restatements (refshim_cv.cpp)
class CLAHE {
public:
    virtual ~CLAHE() {}
    virtual void apply(const Mat& src, const Mat& dst) = 0;
};
Ptr<CLAHE> createCLAHE(double clipLimit, Size tileGridSize);
void cvtColor(const Mat& src, const Mat& dst, int code);
void cvtColor(const Mat& src, Mat& dst, int code);
void calcOpticalFlowPyrLK(const Mat& prev, const Mat& next, std::vector<Point2f>& prevPts, std::vector<Point2f>& nextPts,
                          std::vector<unsigned char>& status, std::vector<float>& err, Size winSize, int maxLevel,
                          TermCriteria criteria, int flags, double minEigThreshold);
void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& D);
namespace fisheye { void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& D); }
void goodFeaturesToTrack(const Mat& im, std::vector<Point2f>& corners, int maxCorners, double qualityLevel, double minDistance);
void cornerSubPix(const Mat& im, std::vector<Point2f>& corners, Size winSize, Size zeroZone, TermCriteria criteria);
inline void circle(Mat&, Point2f, int, const Scalar&, int = 1) {}
inline void line(Mat&, Point2f, Point2f, const Scalar&, int = 1) {}

}  // namespace cv

using cv::cvtColor;  // Tracker.cc calls it unqualified with cv:: arguments (ADL would find it; keep it explicit)

// cv::cv2eigen (opencv2/core/eigen.hpp)
#include "mini_eigen.hpp"
namespace cv {
template <class S, int R, int C> void cv2eigen(const Mat& src, Eigen::Matrix<S, R, C>& dst) {
    dst.resize(src.rows, src.cols);
    for (int i = 0; i < src.rows; ++i)
        for (int j = 0; j < src.cols; ++j)
            dst(i, j) = src.depth() == CV_32F ? (S)src.at<float>(i, j) : (S)src.at<double>(i, j);
}
}  // namespace cv
#endif
