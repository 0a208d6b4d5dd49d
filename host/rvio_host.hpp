// host/rvio_host.hpp — the C++ host side above the C-ABI: the reference's System / InputBuffer / settings reader without
// ROS, OpenCV or Eigen (replay and I/O formats).  The per-frame hot path is ONE call into
// librvio_hip.so (rvio_hip_frame); everything here is plumbing with the reference's semantics:
//   Settings + read_settings   System::System / Tracker / Updater / PreIntegrator / FeatureDetector constructors reading the
//                              OpenCV-YAML settings file (System.cc:44-103, Tracker.cc:37-90, Updater.cc:38-69,
//                              PreIntegrator.cc:30-48, FeatureDetector.cc:28-52; keys: config/rvio_euroc.yaml)
//   InputBuffer                InputBuffer.cc:29-81 (time-sorted FIFOs, GetMeasurements)
//   System::MonoVIO            System.cc:172-436: static-start detection and initialisation gate (:185-250), then the timed
//                              body (:253-367) = rvio_hip_frame, then the pose line of stamped_pose_ests.dat (:369-374)
//   read_image / AslDataset    EuRoC ASL folder: mav0/cam0/data.csv + data/*.png (8-bit gray; also binary PGM),
//                              mav0/imu0/data.csv  [ns, wx, wy, wz, ax, ay, az]
// Unlike the reference, sensor packets are owned by value (upstream never frees them).
#pragma once
#include <cstdint>
#include <deque>
#include <list>
#include <string>
#include <utility>
#include <vector>

#include "../include/rvio_hip.h"

namespace rvio {

struct ImuData {          // InputBuffer.h:35-51
    double w[3], a[3];
    double t, dt;
};
struct ImageData {        // InputBuffer.h:53-63 (cv::Mat -> packed bytes, `channels` interleaved bytes per pixel: 1, 3 or 4)
    std::vector<uint8_t> px;
    int width = 0, height = 0;
    int channels = 1;
    double t = 0;
};
// Tracker::track's "Convert to gray scale" (Tracker.cc:182-196): cvtColor CV_RGB2GRAY / CV_BGR2GRAY (3 channels) or the RGBA / BGRA forms
// (4 channels, alpha ignored) on 8-bit data = OpenCV's fixed-point  (R 4899 + G 9617 + B 1868 + 8192) >> 14.  In place; 1 channel: no-op.
void to_gray(ImageData* im, bool is_rgb);

struct Settings {
    rvio_config cfg;              // everything the hot path needs
    double cam_time_offset = 0;   // Camera.nTimeOffset
    int record_outputs = 0;       // INI.RecordOutputs
    int is_rgb = 0;               // Camera.RGB (Tracker.cc:64-65): channel order of a 3 / 4-channel image, 1 = RGB(A), 0 = BGR(A)
    std::vector<std::string> missing;   // keys the reference reads (cv::FileStorage would yield 0 for them) that the file does not hold:
                                        // they keep the EuRoC defaults here, and the caller is told (rvio_replay prints them)
};
// Parses the OpenCV-YAML (1.0) subset the reference's settings files use: "Key: scalar" lines, '#' comments and
// "Key: !!opencv-matrix" blocks (rows / cols / dt / data: [ ... ]).  Missing keys keep the EuRoC defaults.
bool read_settings(const std::string& path, Settings* out, std::string* err);
bool parse_settings(const std::string& text, Settings* out, std::string* err);

class InputBuffer {       // InputBuffer.cc:29-81
public:
    void PushImuData(const ImuData& d);
    void PushImageData(ImageData&& d);
    // false if there is no image, not enough IMU data yet, or fewer than 2 samples precede the image
    bool GetMeasurements(double time_offset, ImageData* image, std::vector<ImuData>* imus);
    size_t images() const { return img_.size(); }
    size_t imus() const { return imu_.size(); }
private:
    std::list<ImuData> imu_;
    std::list<ImageData> img_;
};

struct PoseLine { double t, p[3], q[4]; };   // System.cc:369-374: timestamp pGk(3) qkG(4)

class System {
public:
    System(const Settings& s, int device);
    ~System();
    System(const System&) = delete;
    System& operator=(const System&) = delete;
    bool ok() const { return h_ != nullptr; }
    const std::string& error() const { return err_; }

    void PushImuData(const ImuData& d) { buf_.PushImuData(d); }          // System::PushImuData, System.h:60
    void PushImageData(ImageData&& d) { buf_.PushImageData(std::move(d)); }
    // System::MonoVIO.  Returns 1 if a frame went through the filter (pose valid), 0 if nothing was processed or the filter is
    // still waiting for motion, <0 on a library error.
    int MonoVIO(PoseLine* pose);
    // INI.RecordOutputs (System.cc:81-88): open <dir>/stamped_pose_ests.dat and <dir>/time_cost.dat (upstream: the package directory).
    // While recording, a frame runs stage by stage like upstream — Tracker::track, then propagate / update / augment / compose, the host
    // waiting behind each — so that the two spans of time_cost.dat (System.cc:254-260,367: t2-t1 and t3-t2, milliseconds) mean what they
    // mean there; without it the frame is one pipelined rvio_hip_frame call.  No-op unless the settings ask for it (or `force`).
    bool record_to(const std::string& dir, bool force = false);
    // the handle's sticky device-side flags (rvio_frame_info.reserved[0]: 1 singular pivot, 2 a track dropped, 4 a stage counter timed out,
    // 8 a non-positive gate pivot; 0 = none) — waits for everything in flight, so a replay reads it once at its end.  -1: the query failed.
    int device_flags();
    bool recording() const { return rec_; }
    bool is_ready() const { return ready_; }
    int frames_after_init() const { return n_img_; }
    rvio_hip* handle() { return h_; }

private:
    Settings s_;
    rvio_hip* h_ = nullptr;
    std::string err_;
    InputBuffer buf_;
    bool moving_ = false, ready_ = false;
    double wm_[3] = {0, 0, 0}, am_[3] = {0, 0, 0};
    int n_imu_ = 0, n_img_ = 0;
    bool rec_ = false;
    void* f_pose_ = nullptr;      // std::ofstream* (kept out of the header)
    void* f_time_ = nullptr;
};

// 8-bit PNG (non-interlaced; gray, RGB or RGBA: channels in file order = RGB) or binary PGM / PPM (P5 / P6, maxval 255)
bool read_image(const std::string& path, ImageData* out, std::string* err);
bool decode_png_gray8(const uint8_t* data, size_t n, ImageData* out, std::string* err);

struct AslDataset {
    std::vector<std::pair<double, std::string>> images;   // (t [s], absolute path)
    std::vector<ImuData> imu;                             // dt = t - previous t (0 for the first), rvio_mono.cc:97-106
};
bool read_asl(const std::string& root, AslDataset* out, std::string* err);

std::string format_pose(const PoseLine& p);               // one line of stamped_pose_ests.dat, setprecision(19)
std::string format_time_cost(int n_img, double track_ms, double filter_ms);   // one line of time_cost.dat (System.cc:376-378)

}  // namespace rvio
