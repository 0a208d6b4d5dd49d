// host/rvio_replay.cpp — offline replay of a EuRoC ASL folder through the MI355X hot path: the role of the reference's
// rvio_mono ROS node + rosbag play (rvio_mono.cc:54-137), without ROS.  Sensor packets are pushed in time order exactly as
// the two ROS callbacks would (every image triggers System::MonoVIO, rvio_mono.cc:78-80); poses are written in the format
// of stamped_pose_ests.dat (System.cc:369-374).
//
//   rvio_replay <settings.yaml> <asl_root> [<poses_out.dat>] [--device N] [--max-frames K]
//   rvio_replay --check-settings <settings.yaml>        print the parsed configuration (no GPU needed)
//   rvio_replay --check-dataset <asl_root>              print what the dataset reader found (no GPU needed)
//   rvio_replay --check-image <file.png|.pgm>           decode one image and print its size and checksum (no GPU needed)
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

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

// ---------------------------------------------------------------- replay passes
struct FrameDump {            // what the device holds behind a frame (read with every stream drained)
    rvio_frame_info info{};
    int n_feat = 0;
    std::vector<unsigned char> types;
    std::vector<int32_t> len;
    std::vector<float> meas;
};
struct PassOptions {
    bool sync_every_frame = false;   // rvio_hip_sync behind every MonoVIO call (the reference pacing: nothing overlaps)
    long stall_seed = -1;            // >= 0: sleeping kernels on random streams in front of random frames (rvio_hip_debug_stall)
    int noise_wgs = 0;               // > 0: that many workgroups of HBM / L2 / LDS traffic beside every fourth frame (rvio_hip_debug_noise)
    long dump_at = -1;               // filtered-frame index behind which the streams are drained and the hand-over is read back
    long max_frames = -1;
};
static bool dump_frame(System& sys, const rvio_config& c, FrameDump* d) {
    rvio_hip* h = sys.handle();
    const int Fu = (c.n_features + 1) / 2, ML = c.max_track_len;
    if (rvio_hip_sync(h) != RVIO_OK) return false;
    if (rvio_hip_get_frame_info(h, &d->info) != RVIO_OK) return false;
    d->types.assign(Fu, 0); d->len.assign(Fu, 0); d->meas.assign((size_t)Fu * ML * 2, 0.f);
    int32_t nf = 0;
    if (rvio_hip_get_tracks(h, &nf, d->types.data(), d->len.data(), d->meas.data()) != RVIO_OK) return false;
    d->n_feat = nf;
    return true;
}
// one pass over the dataset with a fresh System; poses of the filtered frames, optionally the dump of every frame (sync_every_frame) or of one
static int run_pass(const Settings& s, const AslDataset& d, int device, const PassOptions& o, std::vector<PoseLine>* poses,
                    std::vector<FrameDump>* dumps, double* ms_per_call, int* flags, std::string* err) {
    System sys(s, device);
    if (!sys.ok()) { *err = sys.error(); return 1; }
    size_t ii = 0;
    long n_images = 0, n_frames = 0;
    double t_filter = 0;
    unsigned long long lcg = 0x9e3779b97f4a7c15ull ^ (unsigned long long)(o.stall_seed + 1);
    auto rnd = [&](unsigned mod) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)((lcg >> 33) % mod); };
    for (const auto& im : d.images) {
        if (o.max_frames >= 0 && n_images >= o.max_frames) break;
        while (ii < d.imu.size() && (d.imu[ii].t <= im.first + s.cam_time_offset || (ii > 0 && d.imu[ii - 1].t <= im.first + s.cam_time_offset))) sys.PushImuData(d.imu[ii++]);
        ImageData img;
        if (!read_image(im.second, &img, err)) return 1;
        img.t = im.first;
        sys.PushImageData(std::move(img));
        ++n_images;
        if (o.stall_seed >= 0 && sys.is_ready())
            for (unsigned k = rnd(3); k > 0; --k) rvio_hip_debug_stall(sys.handle(), (int)rnd(4), 30 + (int)rnd(870));
        if (o.noise_wgs > 0 && sys.is_ready() && n_images % 4 == 0) rvio_hip_debug_noise(sys.handle(), o.noise_wgs, 1500);
        PoseLine p;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = sys.MonoVIO(&p);
        t_filter += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc < 0) { *err = sys.error(); return 1; }
        if (rc == 1) {
            poses->push_back(p);
            if (o.sync_every_frame || o.dump_at == n_frames) {
                FrameDump fd;
                if (!dump_frame(sys, s.cfg, &fd)) { *err = rvio_hip_last_error(sys.handle()); return 1; }
                if (dumps) dumps->push_back(std::move(fd));
            }
            ++n_frames;
            if (o.dump_at >= 0 && n_frames > o.dump_at) break;
        }
    }
    if (ms_per_call) *ms_per_call = n_images ? 1e3 * t_filter / n_images : 0.0;
    if (flags) *flags = sys.device_flags();
    return 0;
}
static double pose_diff(const PoseLine& a, const PoseLine& b) {
    double m = 0;
    for (int i = 0; i < 3; ++i) m = std::max(m, std::fabs(a.p[i] - b.p[i]));
    const double sa = a.q[3] < 0 ? -1 : 1, sb = b.q[3] < 0 ? -1 : 1;
    for (int i = 0; i < 4; ++i) m = std::max(m, std::fabs(sa * a.q[i] - sb * b.q[i]));
    return m;
}
// which HIP runtime this process ended up with (a Python process loads torch's bundled copy first; this binary takes /opt/rocm's through
// the library's DT_NEEDED) — resolved from the process image, no link-time dependency of the host on the runtime
static void print_runtime() {
    typedef int (*ver_fn)(int*);
    int rt = 0, drv = 0;
    const char* path = "?";
    if (ver_fn f = (ver_fn)dlsym(RTLD_DEFAULT, "hipRuntimeGetVersion")) {
        (void)f(&rt);
        Dl_info di{};
        if (dladdr((void*)f, &di) && di.dli_fname) path = di.dli_fname;
    }
    if (ver_fn f = (ver_fn)dlsym(RTLD_DEFAULT, "hipDriverGetVersion")) (void)f(&drv);
    std::fprintf(stderr, "rvio_replay: HIP runtime %d, driver %d, libamdhip64 = %s, RVIO_PARANOID=%s\n", rt, drv, path,
                 getenv("RVIO_PARANOID") ? getenv("RVIO_PARANOID") : "");
}
// --selfcheck: the pipelined replay against the SAME binary's synchronised pass (rvio_hip_sync behind every frame), pose by pose and bit for
// bit; on a mismatch the pipelined pass is repeated up to the first differing frame, drained there, and the first table that differs —
// front-end counters, hand-over count / types / lengths / measurements — is printed beside the synchronised pass's.  Exit status 3.
static int selfcheck(const Settings& s, const AslDataset& d, int device, long max_frames, long stall_seed, int noise_wgs) {
    std::string err;
    std::vector<PoseLine> ref, got;
    std::vector<FrameDump> dref;
    PassOptions o; o.max_frames = max_frames;
    PassOptions os = o; os.sync_every_frame = true;
    if (run_pass(s, d, device, os, &ref, &dref, nullptr, nullptr, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    PassOptions op = o; op.stall_seed = stall_seed; op.noise_wgs = noise_wgs;
    int flags = 0;
    if (run_pass(s, d, device, op, &got, nullptr, nullptr, &flags, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    long first = -1; double worst = 0;
    for (size_t i = 0; i < std::min(ref.size(), got.size()); ++i) {
        const double df = pose_diff(ref[i], got[i]);
        if (df > 0 && first < 0) first = (long)i;
        worst = std::max(worst, df);
    }
    if (ref.size() != got.size() && first < 0) first = (long)std::min(ref.size(), got.size());
    std::fprintf(stderr, "rvio_replay --selfcheck: %zu / %zu filtered frames (synchronised / pipelined%s), max |pose diff| %.3e, first differing frame %ld, device flags %d\n",
                 ref.size(), got.size(), stall_seed >= 0 ? " with stalled queues" : "", worst, first, flags);
    if (first < 0) return 0;
    std::vector<PoseLine> again; std::vector<FrameDump> dgot;
    PassOptions od = op; od.dump_at = first;
    if (run_pass(s, d, device, od, &again, &dgot, nullptr, nullptr, &err) || dgot.empty() || (size_t)first >= dref.size()) {
        std::fprintf(stderr, "  (could not re-run up to frame %ld: %s)\n", first, err.c_str());
        return 3;
    }
    const FrameDump &a = dref[first], &b = dgot[0];
    std::fprintf(stderr, "  frame %ld           synchronised   pipelined\n", first);
#define ROW(name, f) std::fprintf(stderr, "  %-18s %12d %12d%s\n", name, (int)a.f, (int)b.f, a.f != b.f ? "   <--" : "")
    ROW("n_tracked_in", info.n_tracked_in); ROW("n_klt_ok", info.n_klt_ok); ROW("n_ransac_inliers", info.n_ransac_inliers); ROW("ransac_winner", info.ransac_winner);
    ROW("n_feat_update", info.n_feat_update); ROW("n_tracked_out", info.n_tracked_out); ROW("n_feat_accepted", info.n_feat_accepted); ROW("n_rows", info.n_rows);
    ROW("updated", info.updated); ROW("hand-over n_feat", n_feat);
#undef ROW
    const int ML = s.cfg.max_track_len;
    for (int f = 0; f < std::min(a.n_feat, b.n_feat); ++f) {
        bool same = a.types[f] == b.types[f] && a.len[f] == b.len[f];
        for (int k = 0; same && k < 2 * a.len[f]; ++k) same = a.meas[(size_t)f * ML * 2 + k] == b.meas[(size_t)f * ML * 2 + k];
        if (!same) { std::fprintf(stderr, "  hand-over feature %d differs: type %c / %c, len %d / %d\n", f, a.types[f], b.types[f], a.len[f], b.len[f]); break; }
    }
    return 3;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !std::strcmp(argv[1], "--check-settings")) return check_settings(argv[2]);
    if (argc >= 3 && !std::strcmp(argv[1], "--check-dataset")) return check_dataset(argv[2]);
    if (argc >= 3 && !std::strcmp(argv[1], "--check-image")) return check_image(argv[2], !(argc >= 4 && !std::strcmp(argv[3], "--bgr")));
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <settings.yaml> <asl_root> [<poses_out.dat>] [--device N] [--max-frames K] [--record-dir DIR] [--record]\n"
                             "          [--sync-every-frame] [--stall-seed S] [--noise WGS] [--selfcheck]\n", argv[0]);
        return 2;
    }
    const char* out_path = nullptr;
    int device = 0; long max_frames = -1, stall_seed = -1;
    const char* record_dir = "."; bool force_record = false, sync_every = false, self = false; int noise_wgs = 0;
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--max-frames") && i + 1 < argc) max_frames = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--record-dir") && i + 1 < argc) record_dir = argv[++i];   // where INI.RecordOutputs: 1 writes its two files
        else if (!std::strcmp(argv[i], "--record")) force_record = true;                          // as if the settings said INI.RecordOutputs: 1
        else if (!std::strcmp(argv[i], "--sync-every-frame")) sync_every = true;                   // rvio_hip_sync behind every frame (A/B of the pipelining)
        else if (!std::strcmp(argv[i], "--stall-seed") && i + 1 < argc) stall_seed = std::atol(argv[++i]);   // sleeping kernels on random streams (A/B of the ordering)
        else if (!std::strcmp(argv[i], "--noise") && i + 1 < argc) noise_wgs = std::atoi(argv[++i]);         // a loaded chip beside the replay (A/B of timing inside kernels)
        else if (!std::strcmp(argv[i], "--selfcheck")) self = true;
        else out_path = argv[i];
    }
    Settings s; std::string err;
    if (!read_settings(argv[1], &s, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }   // System.cc:54-58: exit(-1)
    for (const std::string& k : s.missing) std::fprintf(stderr, "settings: %s is missing (upstream would read 0); the EuRoC default is used\n", k.c_str());
    AslDataset d;
    if (!read_asl(argv[2], &d, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    print_runtime();
    if (self) return selfcheck(s, d, device, max_frames, stall_seed, noise_wgs);
    if (sync_every || stall_seed >= 0 || noise_wgs > 0) {   // the plain replay with one of the two A/B pacings
        std::vector<PoseLine> poses; double ms = 0; int flags = 0;
        PassOptions o; o.max_frames = max_frames; o.sync_every_frame = sync_every; o.stall_seed = stall_seed; o.noise_wgs = noise_wgs;
        if (run_pass(s, d, device, o, &poses, nullptr, &ms, &flags, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (out_path) { std::ofstream out(out_path); if (!out) { std::fprintf(stderr, "cannot write %s\n", out_path); return 1; } for (const PoseLine& p : poses) out << format_pose(p); }
        std::fprintf(stderr, "rvio_replay: %zu filtered frames, %.3f ms per MonoVIO call, device flags %d\n", poses.size(), ms, flags);
        return 0;
    }
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
