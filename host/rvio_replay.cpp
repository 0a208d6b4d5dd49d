// host/rvio_replay.cpp — offline replay of a EuRoC ASL folder through the MI355X hot path: the role of the reference's
// rvio_mono ROS node + rosbag play (rvio_mono.cc:54-137), without ROS.  Sensor packets are pushed in time order exactly as
// the two ROS callbacks would (every image triggers System::MonoVIO, rvio_mono.cc:78-80); poses are written in the format
// of stamped_pose_ests.dat (System.cc:369-374).
//
//   rvio_replay <settings.yaml> <asl_root> [<poses_out.dat>] [--device N] [--max-frames K]
//   rvio_replay --check-settings <settings.yaml>        print the parsed configuration (no GPU needed)
//   rvio_replay --check-dataset <asl_root>              print what the dataset reader found (no GPU needed)
//   rvio_replay --check-image <file.png|.pgm>           decode one image and print its size and checksum (no GPU needed)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "rvio_host.hpp"

using namespace rvio;

static int check_settings(const char* path) {
    Settings s; std::string err;
    if (!read_settings(path, &s, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    for (const std::string& k : s.missing) std::fprintf(stderr, "settings: %s is missing (upstream would read 0); the EuRoC default is used\n", k.c_str());
    const rvio_config& c = s.cfg;
    std::printf("{\"imu_rate\": %.17g, \"sigma_g\": %.17g, \"sigma_wg\": %.17g, \"sigma_a\": %.17g, \"sigma_wa\": %.17g, \"gravity\": %.17g, "
                "\"small_angle\": %.17g, \"width\": %d, \"height\": %d, \"fx\": %.9g, \"fy\": %.9g, \"cx\": %.9g, \"cy\": %.9g, "
                "\"k1\": %.9g, \"k2\": %.9g, \"p1\": %.9g, \"p2\": %.9g, \"k3\": %.9g, \"sigma_px\": %.9g, \"sigma_py\": %.9g, \"fisheye\": %d, "
                "\"n_features\": %d, \"max_track_len\": %d, \"min_track_len\": %d, \"min_dist\": %.9g, \"qual_lvl\": %.9g, \"block_x\": %.9g, "
                "\"block_y\": %.9g, \"enable_equalizer\": %d, \"use_sampson\": %d, \"inlier_thr\": %.17g, \"ini_thr_angle\": %.17g, "
                "\"ini_thr_displ\": %.17g, \"ini_enable_alignment\": %d, \"cam_time_offset\": %.17g, \"record_outputs\": %d, \"is_rgb\": %d, \"T_bc\": [",
                c.imu_rate, c.sigma_g, c.sigma_wg, c.sigma_a, c.sigma_wa, c.gravity, c.small_angle, c.width, c.height, c.fx, c.fy, c.cx, c.cy,
                c.k1, c.k2, c.p1, c.p2, c.k3, c.sigma_px, c.sigma_py, c.fisheye, c.n_features, c.max_track_len, c.min_track_len, c.min_dist,
                c.qual_lvl, c.block_x, c.block_y, c.enable_equalizer, c.use_sampson, c.inlier_thr, c.ini_thr_angle, c.ini_thr_displ,
                c.ini_enable_alignment, s.cam_time_offset, s.record_outputs, s.is_rgb);
    for (int i = 0; i < 16; ++i) std::printf("%s%.17g", i ? ", " : "", c.T_bc[i]);
    std::printf("]}\n");
    return 0;
}

static int check_dataset(const char* root) {
    AslDataset d; std::string err;
    if (!read_asl(root, &d, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    std::printf("{\"images\": %zu, \"imu\": %zu", d.images.size(), d.imu.size());
    if (!d.images.empty()) std::printf(", \"t_first_image\": %.9f, \"first_image\": \"%s\"", d.images.front().first, d.images.front().second.c_str());
    if (d.imu.size() > 1) std::printf(", \"t_first_imu\": %.9f, \"dt1\": %.9f, \"w1\": [%.17g, %.17g, %.17g]", d.imu[0].t, d.imu[1].dt, d.imu[1].w[0], d.imu[1].w[1], d.imu[1].w[2]);
    std::printf("}\n");
    return 0;
}

static int check_image(const char* path, bool is_rgb) {
    ImageData im; std::string err;
    if (!read_image(path, &im, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    const int channels = im.channels;
    to_gray(&im, is_rgb);
    unsigned long long sum = 0, wsum = 0;
    for (size_t i = 0; i < im.px.size(); ++i) { sum += im.px[i]; wsum += (unsigned long long)im.px[i] * (i % 251 + 1); }
    std::printf("{\"width\": %d, \"height\": %d, \"channels\": %d, \"sum\": %llu, \"wsum\": %llu}\n", im.width, im.height, channels, sum, wsum);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !std::strcmp(argv[1], "--check-settings")) return check_settings(argv[2]);
    if (argc >= 3 && !std::strcmp(argv[1], "--check-dataset")) return check_dataset(argv[2]);
    if (argc >= 3 && !std::strcmp(argv[1], "--check-image")) return check_image(argv[2], !(argc >= 4 && !std::strcmp(argv[3], "--bgr")));
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <settings.yaml> <asl_root> [<poses_out.dat>] [--device N] [--max-frames K] [--record-dir DIR] [--record]\n", argv[0]);
        return 2;
    }
    const char* out_path = nullptr;
    int device = 0; long max_frames = -1;
    const char* record_dir = "."; bool force_record = false;
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--max-frames") && i + 1 < argc) max_frames = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--record-dir") && i + 1 < argc) record_dir = argv[++i];   // where INI.RecordOutputs: 1 writes its two files
        else if (!std::strcmp(argv[i], "--record")) force_record = true;                          // as if the settings said INI.RecordOutputs: 1
        else out_path = argv[i];
    }
    Settings s; std::string err;
    if (!read_settings(argv[1], &s, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }   // System.cc:54-58: exit(-1)
    for (const std::string& k : s.missing) std::fprintf(stderr, "settings: %s is missing (upstream would read 0); the EuRoC default is used\n", k.c_str());
    AslDataset d;
    if (!read_asl(argv[2], &d, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    System sys(s, device);
    if (!sys.ok()) { std::fprintf(stderr, "%s\n", sys.error().c_str()); return 1; }
    if (!sys.record_to(record_dir, force_record)) { std::fprintf(stderr, "%s\n", sys.error().c_str()); return 1; }
    std::ofstream out;
    if (out_path) { out.open(out_path); if (!out) { std::fprintf(stderr, "cannot write %s\n", out_path); return 1; } }

    size_t ii = 0;
    long n_frames = 0, n_images = 0;
    double t_filter = 0;
    for (const auto& im : d.images) {
        if (max_frames >= 0 && n_images >= max_frames) break;
        // IMU callbacks that precede this image (plus one sample beyond it, so that GetMeasurements sees enough data)
        while (ii < d.imu.size() && (d.imu[ii].t <= im.first + s.cam_time_offset || (ii > 0 && d.imu[ii - 1].t <= im.first + s.cam_time_offset))) sys.PushImuData(d.imu[ii++]);
        ImageData img;
        if (!read_image(im.second, &img, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        img.t = im.first;
        sys.PushImageData(std::move(img));
        ++n_images;
        PoseLine p;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = sys.MonoVIO(&p);
        t_filter += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc < 0) { std::fprintf(stderr, "%s\n", sys.error().c_str()); return 1; }
        if (rc == 1) { ++n_frames; if (out.is_open()) out << format_pose(p); }
    }
    std::fprintf(stderr, "rvio_replay: %ld images, %ld filtered frames, %.3f ms per MonoVIO call (host wall clock, pose read-back included)\n",
                 n_images, n_frames, n_images ? 1e3 * t_filter / n_images : 0.0);
    // anything only the device saw (0 in every test and bench run): said out loud, the poses above are suspect if it is not 0
    const int flags = sys.device_flags();
    if (flags) std::fprintf(stderr, "rvio_replay: DEVICE-SIDE FLAGS %d (1 singular pivot in the solve, 2 a track the window cannot hold was dropped, "
                                    "4 a stage counter timed out, 8 non-positive gate pivot)%s%s\n", flags, flags < 0 ? ": " : "", flags < 0 ? sys.error().c_str() : "");
    return 0;
}
