// host/rvio_host.cpp — see rvio_host.hpp.  No ROS / OpenCV / Eigen; zlib for PNG.
#include "rvio_host.hpp"

#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace rvio {

// ------------------------------------------------------------------ settings (OpenCV-YAML 1.0 subset)
namespace {
std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
std::string strip_comment(const std::string& s) {
    const size_t p = s.find('#');
    return p == std::string::npos ? s : s.substr(0, p);
}
struct Yaml {
    std::map<std::string, double> num;
    std::map<std::string, std::vector<double>> mat;
};
bool parse_yaml(const std::string& text, Yaml* y, std::string* err) {
    std::istringstream in(text);
    std::string line, cur_mat;
    bool in_data = false;
    std::string data;
    auto finish_data = [&]() {
        std::vector<double> v;
        std::string tok;
        for (char& c : data) if (c == ',' || c == '[' || c == ']') c = ' ';
        std::istringstream ds(data);
        while (ds >> tok) v.push_back(std::atof(tok.c_str()));
        y->mat[cur_mat] = v;
        in_data = false; data.clear(); cur_mat.clear();
    };
    while (std::getline(in, line)) {
        if (line.rfind("%YAML", 0) == 0 || line.rfind("---", 0) == 0) continue;
        std::string s = trim(strip_comment(line));
        if (s.empty()) continue;
        if (in_data) {
            data += " " + s;
            if (s.find(']') != std::string::npos) finish_data();
            continue;
        }
        const size_t c = s.find(':');
        if (c == std::string::npos) { if (err) *err = "settings: cannot parse line '" + s + "'"; return false; }
        const std::string key = trim(s.substr(0, c)), val = trim(s.substr(c + 1));
        if (val.rfind("!!opencv-matrix", 0) == 0) { cur_mat = key; continue; }
        if (!cur_mat.empty()) {
            if (key == "data") {
                data = val; in_data = true;
                if (val.find(']') != std::string::npos) finish_data();
            }
            continue;   // rows / cols / dt
        }
        char* end = nullptr;
        const double d = std::strtod(val.c_str(), &end);
        if (end != val.c_str()) y->num[key] = d;   // non-numeric scalars (strings) are not used by the reference's hot path
    }
    return true;
}
}  // namespace

bool parse_settings(const std::string& text, Settings* out, std::string* err) {
    Yaml y;
    if (!parse_yaml(text, &y, err)) return false;
    rvio_config& c = out->cfg;
    rvio_config_euroc(&c);
    out->missing.clear();
    // (Camera.k3, Camera.Fisheye, Camera.nTimeOffset and INI.RecordOutputs are optional upstream too: absent in some of its settings files)
    auto num = [&](const char* k, double dflt) {
        auto it = y.num.find(k);
        if (it != y.num.end()) return it->second;
        const std::string ks(k);
        if (ks != "Camera.k3" && ks != "Camera.Fisheye" && ks != "Camera.nTimeOffset" && ks != "INI.RecordOutputs" && ks != "Camera.RGB") out->missing.push_back(ks);
        return dflt;
    };
    c.imu_rate = num("IMU.dps", c.imu_rate);
    c.sigma_g = num("IMU.sigma_g", c.sigma_g); c.sigma_wg = num("IMU.sigma_wg", c.sigma_wg);
    c.sigma_a = num("IMU.sigma_a", c.sigma_a); c.sigma_wa = num("IMU.sigma_wa", c.sigma_wa);
    c.gravity = num("IMU.nG", c.gravity); c.small_angle = num("IMU.nSmallAngle", c.small_angle);
    c.width = (int)num("Camera.width", c.width); c.height = (int)num("Camera.height", c.height);
    // float32, as Tracker.cc:39-62 / Updater.cc:42-44 store them
    c.fx = (float)num("Camera.fx", c.fx); c.fy = (float)num("Camera.fy", c.fy); c.cx = (float)num("Camera.cx", c.cx); c.cy = (float)num("Camera.cy", c.cy);
    c.k1 = (float)num("Camera.k1", c.k1); c.k2 = (float)num("Camera.k2", c.k2); c.p1 = (float)num("Camera.p1", c.p1); c.p2 = (float)num("Camera.p2", c.p2);
    c.k3 = (float)num("Camera.k3", c.k3);
    c.sigma_px = (float)num("Camera.sigma_px", c.sigma_px); c.sigma_py = (float)num("Camera.sigma_py", c.sigma_py);
    c.fisheye = (int)num("Camera.Fisheye", c.fisheye);
    auto m = y.mat.find("Camera.T_BC0");
    if (m == y.mat.end()) out->missing.push_back("Camera.T_BC0");
    if (m != y.mat.end()) {
        if (m->second.size() != 16) { if (err) *err = "settings: Camera.T_BC0 must hold 16 values"; return false; }
        for (int i = 0; i < 16; ++i) c.T_bc[i] = m->second[i];
    }
    c.n_features = (int)num("Tracker.nFeatures", c.n_features);
    c.max_track_len = (int)num("Tracker.nMaxTrackingLength", c.max_track_len);
    c.min_track_len = (int)num("Tracker.nMinTrackingLength", c.min_track_len);
    c.min_dist = (float)num("Tracker.nMinDist", c.min_dist); c.qual_lvl = (float)num("Tracker.nQualLvl", c.qual_lvl);
    c.block_x = (float)num("Tracker.nBlockSizeX", c.block_x); c.block_y = (float)num("Tracker.nBlockSizeY", c.block_y);
    c.enable_equalizer = (int)num("Tracker.EnableEqualizer", c.enable_equalizer);
    c.use_sampson = (int)num("Tracker.UseSampson", c.use_sampson);
    c.inlier_thr = num("Tracker.nInlierThrd", c.inlier_thr);
    c.ini_thr_angle = num("INI.nThresholdAngle", c.ini_thr_angle); c.ini_thr_displ = num("INI.nThresholdDispl", c.ini_thr_displ);
    c.ini_enable_alignment = (int)num("INI.EnableAlignment", c.ini_enable_alignment);
    out->cam_time_offset = num("Camera.nTimeOffset", 0.0);
    out->record_outputs = (int)num("INI.RecordOutputs", 0.0);
    out->is_rgb = (int)num("Camera.RGB", 0.0);
    return true;
}

bool read_settings(const std::string& path, Settings* out, std::string* err) {
    std::ifstream f(path);
    if (!f) { if (err) *err = "Failed to open settings file at: " + path; return false; }   // System.cc:54-58
    std::stringstream ss; ss << f.rdbuf();
    return parse_settings(ss.str(), out, err);
}

// ------------------------------------------------------------------ InputBuffer (InputBuffer.cc:29-81)
void InputBuffer::PushImuData(const ImuData& d) {
    imu_.push_back(d);
    imu_.sort([](const ImuData& a, const ImuData& b) { return a.t < b.t; });
}
void InputBuffer::PushImageData(ImageData&& d) {
    img_.push_back(std::move(d));
    img_.sort([](const ImageData& a, const ImageData& b) { return a.t < b.t; });
}
bool InputBuffer::GetMeasurements(double off, ImageData* image, std::vector<ImuData>* imus) {
    if (imu_.empty() || img_.empty()) return false;
    if (imu_.back().t < img_.front().t + off) return false;      // not enough IMU data for this image yet
    *image = std::move(img_.front());
    img_.pop_front();
    imus->clear();
    while (!imu_.empty() && imu_.front().t <= image->t + off) { imus->push_back(imu_.front()); imu_.pop_front(); }
    return imus->size() >= 2;
}

// ------------------------------------------------------------------ System
System::System(const Settings& s, int device) : s_(s) {
    const int rc = rvio_hip_create(&s_.cfg, device, &h_);
    if (rc != RVIO_OK) {
        err_ = std::string("rvio_hip_create: ") + (h_ ? rvio_hip_last_error(h_) : "invalid configuration") + " (status " + std::to_string(rc) + ")";
        if (h_) { rvio_hip_destroy(h_); h_ = nullptr; }
    }
}
System::~System() {
    delete static_cast<std::ofstream*>(f_pose_);
    delete static_cast<std::ofstream*>(f_time_);
    if (h_) rvio_hip_destroy(h_);
}

bool System::record_to(const std::string& dir, bool force) {
    if (!s_.record_outputs && !force) return true;
    auto* fp = new std::ofstream(dir + "/stamped_pose_ests.dat", std::ofstream::out);   // System.cc:86-87
    auto* ft = new std::ofstream(dir + "/time_cost.dat", std::ofstream::out);
    if (!*fp || !*ft) { delete fp; delete ft; err_ = "cannot write the record files in " + dir; return false; }
    delete static_cast<std::ofstream*>(f_pose_); delete static_cast<std::ofstream*>(f_time_);
    f_pose_ = fp; f_time_ = ft; rec_ = true;
    return true;
}

int System::device_flags() {
    if (!h_) return 0;
    rvio_frame_info fi{};
    const int rc = rvio_hip_get_frame_info(h_, &fi);
    if (rc != RVIO_OK && rc != RVIO_ERR_STATE) { err_ = rvio_hip_last_error(h_); return -1; }
    return fi.reserved[0];
}

int System::MonoVIO(PoseLine* pose) {
    ImageData image;
    std::vector<ImuData> imus;
    if (!buf_.GetMeasurements(s_.cam_time_offset, &image, &imus)) return 0;
    size_t first = 0;                                  // samples before `first` were consumed by the start-up average
    if (!ready_) {                                     // System.cc:185-250
        if (!moving_) {
            double ang[3] = {0, 0, 0}, vel[3] = {0, 0, 0}, displ[3] = {0, 0, 0};
            for (const ImuData& d : imus) {
                const double an = std::sqrt(d.a[0] * d.a[0] + d.a[1] * d.a[1] + d.a[2] * d.a[2]);
                for (int i = 0; i < 3; ++i) {
                    const double a = d.a[i] - s_.cfg.gravity * d.a[i] / an;
                    ang[i] += d.dt * d.w[i];
                    vel[i] += d.dt * a;
                    displ[i] += d.dt * vel[i] + .5 * d.dt * d.dt * a;
                }
            }
            const double na = std::sqrt(ang[0] * ang[0] + ang[1] * ang[1] + ang[2] * ang[2]);
            const double nd = std::sqrt(displ[0] * displ[0] + displ[1] * displ[1] + displ[2] * displ[2]);
            if (na > s_.cfg.ini_thr_angle || nd > s_.cfg.ini_thr_displ) moving_ = true;
        }
        while (first < imus.size()) {
            if (!moving_) {
                for (int i = 0; i < 3; ++i) { wm_[i] += imus[first].w[i]; am_[i] += imus[first].a[i]; }
                ++first; ++n_imu_;
            } else {
                if (n_imu_ == 0) {
                    for (int i = 0; i < 3; ++i) { wm_[i] = imus[first].w[i]; am_[i] = imus[first].a[i]; }
                    n_imu_ = 1;
                } else
                    for (int i = 0; i < 3; ++i) { wm_[i] /= n_imu_; am_[i] /= n_imu_; }
                if (rvio_hip_initialize(h_, wm_, am_, n_imu_) != RVIO_OK) { err_ = rvio_hip_last_error(h_); return -1; }
                ready_ = true;
                break;
            }
        }
        if (!ready_) return 0;
    }
    ++n_img_;
    static_assert(sizeof(ImuData) == sizeof(rvio_imu), "ImuData mirrors rvio_imu");
    const rvio_imu* pi = reinterpret_cast<const rvio_imu*>(imus.data() + first);
    const int m = (int)(imus.size() - first);
    // (any number of samples: a gap of a few dropped images is integrated in one go as upstream does, PreIntegrator.cc:96-97; beyond
    // RVIO_HIP_MAX_IMU = 192 the library grows its staging once)
    if (image.width != s_.cfg.width || image.height != s_.cfg.height) { err_ = "image size does not match Camera.width/height"; return -1; }
    to_gray(&image, s_.is_rgb != 0);                   // Tracker.cc:182-196
    // the timed body of MonoVIO (System.cc:253-367): track -> propagate -> update -> augment -> compose
    if (!rec_) {
        if (rvio_hip_frame(h_, image.px.data(), image.width, pi, m, nullptr, 0) != RVIO_OK) { err_ = rvio_hip_last_error(h_); return -1; }
        if (pose) {
            pose->t = image.t;
            if (rvio_hip_get_pose(h_, pose->p, pose->q) != RVIO_OK) { err_ = rvio_hip_last_error(h_); return -1; }
        }
        return 1;
    }
    // INI.RecordOutputs: the same body stage by stage with the host waiting behind each, t1 / t2 / t3 taken where upstream takes them
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto fail = [&] { err_ = rvio_hip_last_error(h_); return -1; };
    const double t1 = now();
    if (rvio_hip_track(h_, image.px.data(), image.width, pi, m, nullptr, 0) != RVIO_OK || rvio_hip_sync(h_) != RVIO_OK) return fail();   // System.cc:258
    const double t2 = now();
    int do_update = 0, do_augment = 0;
    if (rvio_hip_frame_plan(h_, &do_update, &do_augment) != RVIO_OK) return fail();
    if (rvio_hip_propagate(h_, pi, m) != RVIO_OK) return fail();                          // System.cc:263
    if (do_update && rvio_hip_update_tracked(h_) != RVIO_OK) return fail();               // System.cc:266-268
    if (rvio_hip_augment_compose(h_, do_augment) != RVIO_OK) return fail();               // System.cc:279-365
    PoseLine pl;
    pl.t = image.t;
    if (rvio_hip_get_pose(h_, pl.p, pl.q) != RVIO_OK) return fail();                      // (waits for the filter stream)
    const double t3 = now();
    auto& fp = *static_cast<std::ofstream*>(f_pose_);
    auto& ft = *static_cast<std::ofstream*>(f_time_);
    fp << format_pose(pl); fp.flush();
    ft << format_time_cost(n_img_, t2 - t1, t3 - t2); ft.flush();
    if (pose) *pose = pl;
    return 1;
}

std::string format_time_cost(int n_img, double track_ms, double filter_ms) {   // nImageCountAfterInit, 1e3 (t2-t1), 1e3 (t3-t2); setprecision(19)
    char buf[160];
    std::snprintf(buf, sizeof buf, "%d %.19g %.19g\n", n_img, track_ms, filter_ms);
    return buf;
}

std::string format_pose(const PoseLine& p) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%.19g %.19g %.19g %.19g %.19g %.19g %.19g %.19g\n", p.t, p.p[0], p.p[1], p.p[2], p.q[0], p.q[1], p.q[2], p.q[3]);
    return buf;
}

// ------------------------------------------------------------------ images
void to_gray(ImageData* im, bool is_rgb) {
    const int c = im->channels;
    if (c != 3 && c != 4) return;
    const size_t n = (size_t)im->width * im->height;
    const int ir = is_rgb ? 0 : 2, ib = is_rgb ? 2 : 0;          // byte position of R and B inside a pixel
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* p = &im->px[i * c];
        im->px[i] = (uint8_t)((p[ir] * 4899 + p[1] * 9617 + p[ib] * 1868 + (1 << 13)) >> 14);   // cv::cvtColor 8u: R2Y, G2Y, B2Y, yuv_shift = 14
    }
    im->px.resize(n);
    im->channels = 1;
}

bool decode_png_gray8(const uint8_t* d, size_t n, ImageData* out, std::string* err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 || std::memcmp(d, sig, 8) != 0) { if (err) *err = "not a PNG"; return false; }
    auto be32 = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
    size_t o = 8;
    uint32_t w = 0, h = 0, bpp = 1;
    std::vector<uint8_t> z;
    bool have_hdr = false;
    while (o + 12 <= n) {
        const uint32_t len = be32(o);
        const char* type = (const char*)d + o + 4;
        if (o + 12 + len > n) break;
        const uint8_t* p = d + o + 8;
        if (!std::memcmp(type, "IHDR", 4)) {
            w = be32(o + 8); h = be32(o + 12);
            if (p[8] != 8 || (p[9] != 0 && p[9] != 2 && p[9] != 6) || p[12] != 0) { if (err) *err = "PNG: only 8-bit gray / RGB / RGBA, non-interlaced images are supported"; return false; }
            bpp = p[9] == 0 ? 1 : (p[9] == 2 ? 3 : 4);
            have_hdr = true;
        } else if (!std::memcmp(type, "IDAT", 4)) z.insert(z.end(), p, p + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        o += 12 + len;
    }
    if (!have_hdr || w == 0 || h == 0) { if (err) *err = "PNG: no header"; return false; }
    const size_t rowb = (size_t)w * bpp;
    std::vector<uint8_t> raw((rowb + 1) * h);
    uLongf rl = (uLongf)raw.size();
    if (uncompress(raw.data(), &rl, z.data(), (uLong)z.size()) != Z_OK || rl != raw.size()) { if (err) *err = "PNG: inflate failed"; return false; }
    out->width = (int)w; out->height = (int)h; out->channels = (int)bpp; out->px.assign(rowb * h, 0);
    for (uint32_t y = 0; y < h; ++y) {                 // un-filter
        const uint8_t ft = raw[(size_t)y * (rowb + 1)];
        const uint8_t* s = &raw[(size_t)y * (rowb + 1) + 1];
        uint8_t* r = &out->px[(size_t)y * rowb];
        const uint8_t* up = y ? r - rowb : nullptr;
        for (size_t x = 0; x < rowb; ++x) {
            const int a = x >= bpp ? r[x - bpp] : 0, b = up ? up[x] : 0, c = (x >= bpp && up) ? up[x - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: if (err) *err = "PNG: bad filter type"; return false;
            }
            r[x] = (uint8_t)(s[x] + pred);
        }
    }
    return true;
}

bool read_image(const std::string& path, ImageData* out, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (buf.size() >= 2 && buf[0] == 'P' && (buf[1] == '5' || buf[1] == '6')) {   // binary PGM / PPM: P5|P6 <w> <h> <maxval> <single whitespace> data
        const size_t ch = buf[1] == '6' ? 3 : 1;
        size_t o = 2;
        long v[3];
        for (int k = 0; k < 3; ++k) {
            while (o < buf.size() && (std::isspace(buf[o]) || buf[o] == '#')) { if (buf[o] == '#') while (o < buf.size() && buf[o] != '\n') ++o; else ++o; }
            long x = 0; bool any = false;
            while (o < buf.size() && std::isdigit(buf[o])) { x = 10 * x + (buf[o] - '0'); ++o; any = true; }
            if (!any) { if (err) *err = "PGM: bad header in " + path; return false; }
            v[k] = x;
        }
        ++o;
        if (v[2] != 255 || o + (size_t)v[0] * v[1] * ch > buf.size()) { if (err) *err = "PGM/PPM: only maxval 255 is supported (" + path + ")"; return false; }
        out->width = (int)v[0]; out->height = (int)v[1]; out->channels = (int)ch;
        out->px.assign(buf.begin() + o, buf.begin() + o + (size_t)v[0] * v[1] * ch);
        return true;
    }
    std::string e;
    if (!decode_png_gray8(buf.data(), buf.size(), out, &e)) { if (err) *err = e + " (" + path + ")"; return false; }
    return true;
}

// ------------------------------------------------------------------ EuRoC ASL folder
bool read_asl(const std::string& root, AslDataset* out, std::string* err) {
    auto open = [&](const std::string& rel, std::ifstream& f) {
        f.open(root + "/" + rel);
        if (!f) { if (err) *err = "cannot open " + root + "/" + rel; return false; }
        return true;
    };
    std::ifstream fi, fc;
    if (!open("mav0/imu0/data.csv", fi) || !open("mav0/cam0/data.csv", fc)) return false;
    std::string line;
    double last = -1;
    while (std::getline(fi, line)) {
        if (line.empty() || line[0] == '#') continue;
        for (char& c : line) if (c == ',') c = ' ';
        std::istringstream ls(line);
        long long ns; ImuData d;
        if (!(ls >> ns >> d.w[0] >> d.w[1] >> d.w[2] >> d.a[0] >> d.a[1] >> d.a[2])) continue;
        d.t = (double)(ns / 1000000000LL) + 1e-9 * (double)(ns % 1000000000LL);   // ros::Time::toSec() of the message stamp
        d.dt = last < 0 ? 0.0 : d.t - last;                     // rvio_mono.cc:97-106
        last = d.t;
        out->imu.push_back(d);
    }
    while (std::getline(fc, line)) {
        if (line.empty() || line[0] == '#') continue;
        const size_t c = line.find(',');
        if (c == std::string::npos) continue;
        const long long ns = std::atoll(line.substr(0, c).c_str());
        std::string name = trim(line.substr(c + 1));
        out->images.emplace_back((double)(ns / 1000000000LL) + 1e-9 * (double)(ns % 1000000000LL), root + "/mav0/cam0/data/" + name);
    }
    std::sort(out->images.begin(), out->images.end());
    return true;
}

}  // namespace rvio
