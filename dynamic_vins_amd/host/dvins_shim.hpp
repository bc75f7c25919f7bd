// dvins_shim.hpp — header-only C++17 host shim: the reference's class surface for the hot path, on top of the C ABI
// (include/dvins.h, libdvins_hip.so).  A dynamic_vins maintainer includes this instead of
// front_end/background_tracker.h + estimator/estimator.h for the two objects constructed in Run()
// (system/main.cpp:363-369); INTEGRATION.md shows the exact binding.  Depends on the STL only — the OpenCV / Eigen
// types of the reference appear as small views and std::array, and conversion helpers are enabled when the
// reference's own headers are in the translation unit (DVINS_SHIM_WITH_OPENCV / DVINS_SHIM_WITH_EIGEN).
//
//   dynamic_vins::FeatureTracker   front_end/background_tracker.h:41-88
//        FeatureTracker(const std::string& config_path)
//        FeatureBackground TrackImage(SemanticImage&)          background_tracker.cpp:52-158
//        FeatureBackground TrackImageNaive(SemanticImage&)     background_tracker.cpp:400-516
//        FeatureBackground TrackSemanticImage(SemanticImage&)  background_tracker.cpp:757-837
//   dynamic_vins::Estimator        estimator/estimator.h:55-164
//        Estimator(const std::string& config_path); SetParameter(); ClearState();
//        InputIMU(double t, const Vec3d& acc, const Vec3d& gyr)                     estimator.cpp:1765-1779
//        ProcessMeasurements()  — blocking loop over the global feature_queue until cfg::ok is false   estimator.cpp:1786-1863
//        ProcessMeasurements(const FrontendFeature&) — one iteration of that loop
//        public solver_flag, margin_flag, key_poses, frame (estimator.h:150-164)
//   dynamic_vins::FeatureQueue feature_queue (basic/feature_queue.h:19-73), dynamic_vins::cfg::ok (utils/parameters.h)
//   BASELINE.json's north_star spells the two entry points trackImage / processImage (the VINS-Fusion names): both exist as aliases
//   Errors: std::runtime_error, as the reference throws (bad config, empty input); the C ABI itself never throws.
//   Threading: one FeatureTracker per tracking thread, one Estimator per back-end thread (each owns a dv_ctx);
//   InputIMU may be called from another thread (guarded like the reference's buf_mutex).
#pragma once
#include <array>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cmath>
#include <memory>
#include <thread>
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <list>
#include <map>
#include <mutex>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "dvins.h"

namespace dynamic_vins {

using Vec3d = std::array<double, 3>;   // Eigen::Vector3d            (utils/def.h)
using Vec7d = std::array<double, 7>;   // Eigen::Matrix<double,7,1>: x_n y_n 1 u v vx vy
using Mat3d = std::array<double, 9>;   // row-major

// ---------------------------------------------------------------------------------------------------------------
// Minimal reader of the cv::FileStorage YAML 1.0 files the reference ships (config/**/*.yaml): top-level scalars,
// one level of nested maps ("projection_parameters: { fx, fy, cx, cy }" written as an indented block) and
// "!!opencv-matrix" blocks with a (possibly multi-line) "data: [ ... ]".  Keys keep the reference's names.
class YamlFile {
public:
    explicit YamlFile(const std::string& path) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("dvins: cannot open config " + path);
        std::string line, section;
        while (std::getline(f, line)) {
            const size_t hash = find_comment(line);
            if (hash != std::string::npos) line.erase(hash);
            if (trim(line).empty() || line[0] == '%' || trim(line) == "---") continue;
            const size_t indent = line.find_first_not_of(" \t");
            const size_t colon = line.find(':');
            if (colon == std::string::npos) { append_data(section, line); continue; }
            std::string key = trim(line.substr(indent, colon - indent)), val = trim(line.substr(colon + 1));
            if (indent == 0) {
                section = key;
                if (!val.empty() && val.rfind("!!", 0) != 0) { scalars_[key] = unquote(val); section.clear(); }
            } else if (!section.empty()) {
                if (key == "data") { data_open_ = true; append_data(section, val); }
                else scalars_[section + "." + key] = unquote(val);
            }
        }
    }
    bool has(const std::string& k) const { return scalars_.count(k) != 0; }
    std::string str(const std::string& k) const {
        auto it = scalars_.find(k);
        if (it == scalars_.end()) throw std::runtime_error("dvins: config key missing: " + k);
        return it->second;
    }
    std::string str(const std::string& k, const std::string& dflt) const { return has(k) ? str(k) : dflt; }
    double num(const std::string& k) const { return std::strtod(str(k).c_str(), nullptr); }
    double num(const std::string& k, double dflt) const { return has(k) ? num(k) : dflt; }
    int integer(const std::string& k, int dflt) const { return has(k) ? (int)std::lround(num(k)) : dflt; }
    std::vector<int> int_list(const std::string& k) const {          // a flow sequence on one line: "dynamic_label_id: [241,242,243]"
        std::vector<int> out;
        if (!has(k)) return out;
        std::string t = str(k);
        for (char& c : t) if (c == '[' || c == ']' || c == ',') c = ' ';
        std::istringstream is(t); double v;
        while (is >> v) out.push_back((int)v);
        return out;
    }
    const std::vector<double>& matrix(const std::string& k) const {
        auto it = mats_.find(k);
        if (it == mats_.end()) throw std::runtime_error("dvins: config matrix missing: " + k);
        return it->second;
    }
    bool has_matrix(const std::string& k) const { return mats_.count(k) != 0; }

private:
    static size_t find_comment(const std::string& s) {
        bool q = false;
        for (size_t i = 0; i < s.size(); ++i) { if (s[i] == '"') q = !q; if (s[i] == '#' && !q) return i; }
        return std::string::npos;
    }
    static std::string trim(const std::string& s) {
        const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    }
    static std::string unquote(const std::string& s) { return s.size() >= 2 && s.front() == '"' && s.back() == '"' ? s.substr(1, s.size() - 2) : s; }
    void append_data(const std::string& section, const std::string& text) {
        if (!data_open_ || section.empty()) return;
        std::string t = text;
        for (char& c : t) if (c == '[' || c == ',') c = ' ';
        const size_t close = t.find(']');
        if (close != std::string::npos) { t.erase(close); }
        std::istringstream is(t);
        double v;
        while (is >> v) mats_[section].push_back(v);
        if (close != std::string::npos) data_open_ = false;
    }
    std::map<std::string, std::string> scalars_;
    std::map<std::string, std::vector<double>> mats_;
    bool data_open_ = false;
};

inline std::string dir_of(const std::string& path) {
    const size_t s = path.find_last_of('/');
    return s == std::string::npos ? std::string(".") : path.substr(0, s);
}

// camodocal PinholeCamera yaml (camera_models/src/camera_models/PinholeCamera.cc:210-260)
inline dv_cam ReadPinholeCamera(const std::string& path) {
    YamlFile y(path);
    if (y.has("model_type") && y.str("model_type") != "PINHOLE") throw std::runtime_error("dvins: only PINHOLE cameras are on the path (" + path + ")");
    dv_cam c{};
    c.fx = y.num("projection_parameters.fx"); c.fy = y.num("projection_parameters.fy");
    c.cx = y.num("projection_parameters.cx"); c.cy = y.num("projection_parameters.cy");
    c.k1 = y.num("distortion_parameters.k1", 0); c.k2 = y.num("distortion_parameters.k2", 0);
    c.p1 = y.num("distortion_parameters.p1", 0); c.p2 = y.num("distortion_parameters.p2", 0);
    return c;
}

// KITTI calibration file (utils/dataset/kitti_utils.cpp:23-100, "P2: 12 numbers" ...): the two colour cameras' projection matrices.  The reference takes the
// intrinsics from P2 / P3 (utils/camera_model.cpp:41-75) and puts the body frame ON camera 2: R_IC = I, T_IC0 = 0, T_IC1 = (|b3 - b2|, 0, 0) with
// b = P(0,3) / P(0,0) (utils/camera_model.cpp:219-266).
struct KittiCalib { dv_cam cam0{}, cam1{}; double baseline = 0; };
inline KittiCalib ReadKittiCalib(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("dvins: cannot open KITTI calib file " + path);
    double P2[12] = {0}, P3[12] = {0}; bool h2 = false, h3 = false;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream is(line); std::string tag; is >> tag;
        double* dst = tag == "P2:" ? P2 : tag == "P3:" ? P3 : nullptr;
        if (!dst) continue;
        for (int i = 0; i < 12; ++i) if (!(is >> dst[i])) throw std::runtime_error("dvins: malformed " + tag + " in " + path);
        (tag == "P2:" ? h2 : h3) = true;
    }
    if (!h2 || !h3) throw std::runtime_error("dvins: P2 / P3 missing in " + path);
    KittiCalib k;
    k.cam0 = dv_cam{P2[0], P2[5], P2[2], P2[6], 0, 0, 0, 0}; k.cam1 = dv_cam{P3[0], P3[5], P3[2], P3[6], 0, 0, 0, 0};
    k.baseline = std::fabs(P3[3] / P3[0] - P2[3] / P2[0]);
    return k;
}

// Everything the hot path reads from a reference config file, with the reference's own defaults and failure rules:
// Config::Config (utils/parameters.cpp:18-150), FrontendParemater::SetParameters (front_end/front_end_parameters.cpp:17-40), VioParameters::SetParameters
// (estimator/vio_parameters.cpp:17-85), ReadExtrinsicParameters / GetCameraPath (utils/camera_model.cpp:205-340), the topic names of system_call_back.cpp:18-35.
struct Config {
    dv_config front{}; dv_est_config est{};
    std::string slam_type, dataset_type;
    bool dynamic = false, naive = false, input_seg = false;
    bool every_frame = false;            // system/main.cpp:300-307: KITTI forwards every frame to the back end, every other dataset every second one
    int max_dynamic_cnt = 0, min_dynamic_dist = 0, use_det3d = 0, undistort_input = 0, only_frontend = 0;
    bool static_inst_as_background = true;      // vio_parameters.h:86 (default TRUE), vio_parameters.cpp:69-71 -> estimator.cpp:1583, system/main.cpp:217-245
    // VIODE (utils/dataset/viode_utils.cpp:254-276 SetParameters): the keys VIODE::IsDynamic accepts — PixelToKey(r, g, b) = r * 1000000 + g * 1000 * b (sic) of every
    // rgb_ids.txt line whose label id is one of dynamic_label_id; the first line of a key wins (unordered_map::insert).  Ascending.  Empty unless input_seg.
    std::vector<uint32_t> viode_dynamic_keys; std::string rgb_to_label_file;
    double baseline = 0, max_solver_time = 0;
    std::map<std::string, std::string> topics;  // image0_topic, image1_topic, image0_segmentation_topic, image1_segmentation_topic, imu_topic
};
inline Config ReadConfig(const std::string& config_path, int device = 0, const std::string& seq_name = "", const std::string& kitti_calib_dir = "") {
    YamlFile y(config_path);
    Config c;
    // a non-string slam_type node (the stale "slam_type: 0 / 2" files) reads as "" through cv::FileStorage >> std::string: neither raw nor naive -> dynamic
    c.slam_type = y.str("slam_type", "");
    { char* end = nullptr; (void)std::strtod(c.slam_type.c_str(), &end); if (!c.slam_type.empty() && end && *end == 0) c.slam_type.clear(); }
    c.naive = c.slam_type == "naive"; c.dynamic = c.slam_type != "raw" && !c.naive;
    c.dataset_type = y.str("dataset_type", "");
    for (char& ch : c.dataset_type) ch = (char)std::tolower((unsigned char)ch);
    if (c.dataset_type != "kitti" && c.dataset_type != "euroc" && c.dataset_type != "custom") c.dataset_type = "viode";      // parameters.cpp:44-53: anything else is VIODE
    c.input_seg = c.dataset_type == "viode" && (c.dynamic || c.naive);
    c.every_frame = c.dataset_type == "kitti";
    const int ncam = y.integer("num_of_cam", 0);
    if (ncam != 1 && ncam != 2) throw std::runtime_error("num_of_cam should be 1 or 2");
    dv_config& f = c.front;
    f.width = y.integer("image_width", 0); f.height = y.integer("image_height", 0);
    f.max_cnt = y.integer("max_cnt", 0); f.min_dist = y.integer("min_dist", 0);
    f.flow_back = y.integer("flow_back", 0); f.stereo = ncam == 2; f.device = device;
    f.mask_morphology_size = y.integer("use_mask_morphology", 0) ? y.integer("mask_morphology_size", 0) : 0;      // an absent node reads as 0 / false through FileStorage
    c.max_dynamic_cnt = y.integer("max_dynamic_cnt", 0); c.min_dynamic_dist = y.integer("min_dynamic_dist", 0);
    c.undistort_input = y.integer("undistort_input", 0); c.only_frontend = y.integer("dst_mode", 0) ? 1 : y.integer("only_frontend", 0);
    dv_est_config& e = c.est;
    std::memset(&e, 0, sizeof(e));
    e.use_imu = y.integer("imu", 0); e.stereo = f.stereo;
    e.plane_constraint = y.integer("plane_constraint", 0);
    e.max_iters = y.integer("max_num_iterations", 0); c.max_solver_time = y.num("max_solver_time", 0.0);
    e.keyframe_parallax = y.num("keyframe_parallax", 0.0);
    e.init_depth = y.num("INIT_DEPTH", 5.0);
    e.g_norm = 9.81007; e.acc_n = 0.1; e.gyr_n = 0.01; e.acc_w = 0.001; e.gyr_w = 1e-4;                           // vio_parameters.h defaults, overwritten only with an IMU
    if (e.use_imu) { e.acc_n = y.num("acc_n", 0); e.gyr_n = y.num("gyr_n", 0); e.acc_w = y.num("acc_w", 0); e.gyr_w = y.num("gyr_w", 0); e.g_norm = y.num("g_norm", 0); }
    e.td = y.num("td", 0.0);
    e.use_line = y.integer("use_line", 0); e.line_min_obs = y.integer("line_min_obs", 5);
    e.dynamic = c.dynamic ? 1 : 0;
    e.instance_init_min_num = 4; e.static_inst_threshold = 10.0;
    if (c.dynamic) {
        if (!y.has("use_det3d")) throw std::runtime_error("Config::Config(() fs[\"use_det3d\"].isNone()");                                     // parameters.cpp:134-137
        c.use_det3d = e.use_det3d = y.integer("use_det3d", 0);
        if (!y.has("instance_init_min_num")) throw std::runtime_error("VioParameters::SetParameters() fs[\"instance_init_min_num\"].isNone()");   // vio_parameters.cpp:54-61
        e.instance_init_min_num = y.integer("instance_init_min_num", 4);
    }
    if (y.has("static_inst_threshold")) e.static_inst_threshold = y.num("static_inst_threshold");
    if (y.has("static_inst_as_background")) c.static_inst_as_background = y.integer("static_inst_as_background", 1) != 0;
    e.estimate = 0;                         // parameters.cpp:82-99: both switches are forced to 0 without an IMU
    if (e.use_imu) {
        const int ex = y.integer("estimate_extrinsic", 0);
        if (ex == 2) throw std::runtime_error("dvins: estimate_extrinsic 2 (calibration without an initial guess, estimator.cpp:1426-1445) is not on the accelerated path");
        e.estimate = (ex == 1 ? 1 : 0) | (y.integer("estimate_td", 0) != 0 ? 2 : 0);
    }
    for (int k = 0; k < 2; ++k) for (int i = 0; i < 9; ++i) e.ric[k][i] = (i % 4 == 0) ? 1.0 : 0.0;
    const std::string dir = dir_of(config_path);
    if (c.dataset_type == "kitti" && y.has("kitti_calib_path")) {
        const std::string base = kitti_calib_dir.empty() ? y.str("kitti_calib_path") : kitti_calib_dir;
        const KittiCalib k = ReadKittiCalib(base + (seq_name.empty() ? std::string("0000") : seq_name) + ".txt");
        f.cam0 = k.cam0; f.cam1 = k.cam1; c.baseline = k.baseline; e.tic[1][0] = k.baseline;
    } else {
        f.cam0 = ReadPinholeCamera(dir + "/" + y.str("cam0_calib"));
        f.cam1 = (f.stereo && y.has("cam1_calib")) ? ReadPinholeCamera(dir + "/" + y.str("cam1_calib")) : f.cam0;
        for (int k = 0; k < (f.stereo ? 2 : 1); ++k) {
            const std::string key = k == 0 ? "body_T_cam0" : "body_T_cam1";
            if (!y.has_matrix(key)) throw std::runtime_error("dvins: " + key + " missing");
            const std::vector<double>& T = y.matrix(key);
            if (T.size() != 16) throw std::runtime_error("dvins: " + key + " must be 4x4");
            for (int r = 0; r < 3; ++r) { for (int q = 0; q < 3; ++q) e.ric[k][r * 3 + q] = T[r * 4 + q]; e.tic[k][r] = T[r * 4 + 3]; }
        }
        if (f.stereo) {      // cam.baseline = |(body_T_cam0^-1 body_T_cam1)(0,3)| (camera_model.cpp:286-288)
            double d[3] = { e.tic[1][0] - e.tic[0][0], e.tic[1][1] - e.tic[0][1], e.tic[1][2] - e.tic[0][2] };
            c.baseline = std::fabs(e.ric[0][0] * d[0] + e.ric[0][3] * d[1] + e.ric[0][6] * d[2]);
        }
    }
    for (const char* k : { "image0_topic", "image1_topic", "image0_segmentation_topic", "image1_segmentation_topic", "imu_topic" }) if (y.has(k)) c.topics[k] = y.str(k);
    if (c.input_seg && y.has("rgb_to_label_file")) {          // VIODE::SetParameters: basic_dir + rgb_to_label_file; a file next to the config is accepted too (the reference's basic_dir is the author's home)
        const std::string rel = y.str("rgb_to_label_file");
        std::string path = y.str("basic_dir", "") + rel;
        { std::string up = dir; for (int lvl = 0; lvl < 5 && !std::ifstream(path); ++lvl) { path = up + "/" + rel; up = dir_of(up); } }      // the config's directory, then its ancestors (the repository root the shipped files are relative to)
        std::ifstream f(path);
        if (!f) throw std::runtime_error("Can not open:" + path);                                   // ReadViodeRgbIds (viode_utils.cpp:225-227)
        c.rgb_to_label_file = path;
        const std::vector<int> dyn = y.int_list("dynamic_label_id");
        std::map<uint32_t, int> key_to_id;
        std::string line; std::getline(f, line);                                                    // the header line
        while (std::getline(f, line)) {
            int v[4] = { 0, 0, 0, 0 }; std::istringstream is(line); std::string tok;
            for (int j = 0; j < 4 && std::getline(is, tok, ','); ++j) v[j] = std::atoi(tok.c_str());
            key_to_id.insert({ (uint32_t)v[1] * 1000000u + (uint32_t)v[2] * 1000u * (uint32_t)v[3], v[0] });
        }
        for (const auto& kv : key_to_id) if (std::find(dyn.begin(), dyn.end(), kv.second) != dyn.end()) c.viode_dynamic_keys.push_back(kv.first);
    }
    return c;
}

// ---------------------------------------------------------------------------------------------------------------
// data types of the boundary
struct ImageView {                      // a CV_8UC1 cv::Mat: gray0 / gray1 / inv_merge_mask of SemanticImage
    const uint8_t* data = nullptr; int width = 0, height = 0, stride = 0;
    bool device = false;                // true: data is an HBM pointer (DV_MEM_DEVICE)
    bool bgr = false;                   // true: 8-bit BGR (SemanticImage::color0 / color1): converted — and undistorted, if maps are installed — on the device
    bool empty() const { return data == nullptr; }
};
struct SemanticImage {                  // basic/semantic_image.h:30-65 (the fields the path reads)
    ImageView gray0, gray1, inv_merge_mask;
    double time0 = 0; unsigned int seq = 0;
};
struct Line {                           // line_detector/line.h: id + undistorted normalised end points (StartPt / EndPt of FrameLines::un_lines)
    unsigned int id = 0; double x1 = 0, y1 = 0, x2 = 0, y2 = 0;
};
struct LineSegment { unsigned int id = 0; float x1 = 0, y1 = 0, x2 = 0, y2 = 0; };      // a matched segment of the CPU detector (LSD + LBD, line_detector.cpp), pixels
struct FeatureBackground {              // basic/frontend_feature.h:34-44
    std::map<unsigned int, std::vector<std::pair<int, Vec7d>>> points;
    std::map<unsigned int, std::vector<std::pair<int, Line>>> lines;      // {line_id, [(camera_id, line)]}
};
struct FrontendFeature {                // basic/frontend_feature.h:52-75
    FeatureBackground features; double time = 0; unsigned int seq_id = 0;
};
// cfg::ok (utils/parameters.h: inline static std::atomic_bool ok{true}): the flag the reference's three thread loops poll; /vins_terminal clears it
namespace cfg { inline std::atomic<bool> ok{true}; }
constexpr int kImageQueueSize = 100;      // utils/parameters.h:48
// FeatureQueue (basic/feature_queue.h:19-73), the queue between thread T2 (FeatureTrack pushes, system/main.cpp:300-307) and T3 (ProcessMeasurements pops):
// push_back DROPS the frame silently when kImageQueueSize frames are waiting, request() waits at most 30 ms, front_time() peeks the oldest stamp.
class FeatureQueue {
public:
    using Ptr = std::shared_ptr<FeatureQueue>;
    void push_back(FrontendFeature& frame) {
        std::unique_lock<std::mutex> lock(queue_mutex);
        if (frame_list.size() < (size_t)kImageQueueSize) frame_list.push_back(frame);
        queue_cond.notify_one();
    }
    std::optional<FrontendFeature> request() {
        std::unique_lock<std::mutex> lock(queue_mutex);
        if (!queue_cond.wait_for(lock, std::chrono::milliseconds(30), [&] { return !frame_list.empty(); })) return std::nullopt;
        FrontendFeature frame = std::move(frame_list.front());
        frame_list.pop_front();
        return frame;
    }
    int size() { std::unique_lock<std::mutex> lock(queue_mutex); return (int)frame_list.size(); }
    bool empty() { std::unique_lock<std::mutex> lock(queue_mutex); return frame_list.empty(); }
    void clear() { std::unique_lock<std::mutex> lock(queue_mutex); frame_list.clear(); }
    std::optional<double> front_time() {
        std::unique_lock<std::mutex> lock(queue_mutex);
        if (frame_list.empty()) return std::nullopt;
        return frame_list.front().time;
    }
private:
    std::mutex queue_mutex;
    std::condition_variable queue_cond;
    std::list<FrontendFeature> frame_list;
};
inline FeatureQueue feature_queue;      // the reference's global (basic/feature_queue.h:73, defined estimator.cpp:38)

#ifdef DVINS_SHIM_WITH_OPENCV
inline ImageView View(const cv::Mat& m) { return ImageView{m.data, m.cols, m.rows, (int)m.step, false, m.channels() == 3}; }
#endif
#ifdef DVINS_SHIM_WITH_EIGEN
inline Eigen::Matrix<double, 7, 1> ToEigen(const Vec7d& v) { return Eigen::Map<const Eigen::Matrix<double, 7, 1>>(v.data()); }
inline Vec3d FromEigen(const Eigen::Vector3d& v) { return Vec3d{v.x(), v.y(), v.z()}; }
#endif

namespace detail {
inline void check(dv_ctx* ctx, int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string("dvins: ") + what + ": " + dv_last_error(ctx));
}
inline FeatureBackground to_points(const dv_feat* rows, int n) {      // FeatureTracker::SetOutputFeats (background_tracker.cpp:340-392)
    FeatureBackground fb;
    for (int i = 0; i < n; ++i) {
        Vec7d l; std::memcpy(l.data(), rows[i].left, sizeof(double) * 7);
        auto& v = fb.points[rows[i].id];
        v.emplace_back(0, l);
        if (rows[i].has_right) { Vec7d r; std::memcpy(r.data(), rows[i].right, sizeof(double) * 7); v.emplace_back(1, r); }
    }
    return fb;
}
inline std::vector<dv_feat> to_rows(const FeatureBackground& fb) {
    std::vector<dv_feat> rows;
    rows.reserve(fb.points.size());
    for (const auto& kv : fb.points) {
        dv_feat f{};
        f.id = kv.first; f.track_cnt = 1;
        for (const auto& ob : kv.second) {
            if (ob.first == 0) std::memcpy(f.left, ob.second.data(), sizeof(double) * 7);
            else { std::memcpy(f.right, ob.second.data(), sizeof(double) * 7); f.has_right = 1; }
        }
        rows.push_back(f);
    }
    return rows;
}
}  // namespace detail

// ---------------------------------------------------------------------------------------------------------------
class FeatureTracker {
public:
    using Ptr = std::shared_ptr<FeatureTracker>;
    explicit FeatureTracker(const std::string& config_path, int device = 0, const std::string& seq_name = "") { init(ReadConfig(config_path, device, seq_name).front); }
    explicit FeatureTracker(const dv_config& c) { init(c); }
    ~FeatureTracker() { if (ctx_) dv_destroy(ctx_); }
    FeatureTracker(const FeatureTracker&) = delete;
    FeatureTracker& operator=(const FeatureTracker&) = delete;

    FeatureBackground TrackImage(SemanticImage& img) { return track(img, DV_MODE_RAW); }
    FeatureBackground trackImage(SemanticImage& img) { return TrackImage(img); }      // north_star's spelling (VINS-Fusion's FeatureTracker::trackImage)
    FeatureBackground TrackImageNaive(SemanticImage& img) { return track(img, DV_MODE_NAIVE); }
    FeatureBackground TrackSemanticImage(SemanticImage& img) { return track(img, DV_MODE_SEMANTIC); }      // background_tracker.cpp:757-837 (background half)
    // FeatureTracker::TrackImageLine (background_tracker.cpp:198-333): the point half is TrackImage's; the line half is the reference's CPU detector thread
    // (LineDetector::Detect / TrackLeftLine / TrackRightLine — OpenCV line_descriptor, upstream of this path) whose matched segments the caller hands in;
    // FrameLines::UndistortedLineEndPoints (cam0 / cam1) and SetOutputFeats' `lines` map (:373-392) are done here.  The detector may run on another thread
    // while this call tracks the points, as in the reference: pass the segments of THIS frame.
    FeatureBackground TrackImageLine(SemanticImage& img, const std::vector<LineSegment>& left, const std::vector<LineSegment>& right = {}) {
        FeatureBackground fb = track(img, DV_MODE_RAW);
        auto put = [&](const std::vector<LineSegment>& segs, int cam_id) {
            if (segs.empty()) return;
            std::vector<float> px(4 * segs.size()); std::vector<double> un(4 * segs.size());
            for (size_t i = 0; i < segs.size(); ++i) { px[4 * i] = segs[i].x1; px[4 * i + 1] = segs[i].y1; px[4 * i + 2] = segs[i].x2; px[4 * i + 3] = segs[i].y2; }
            detail::check(ctx_, dv_undistort_lines(ctx_, cam_id == 0 ? &cfg_.cam0 : &cfg_.cam1, px.data(), (int)segs.size(), un.data()), "TrackImageLine");
            for (size_t i = 0; i < segs.size(); ++i) {
                const Line l{ segs[i].id, un[4 * i], un[4 * i + 1], un[4 * i + 2], un[4 * i + 3] };
                if (cam_id == 0) fb.lines.insert({ l.id, { { 0, l } } });
                else fb.lines[l.id].push_back({ 1, l });
            }
        };
        put(left, 0);
        if (cfg_.stereo) put(right, 1);
        return fb;
    }
    // two-phase form: lets the caller overlap the front end of frame k+1 with the back end of frame k
    void TrackImageEnqueue(SemanticImage& img, int mode = DV_MODE_RAW) {
        check_image(img);
        remember(img);
        cur_time = img.time0;
        detail::check(ctx_, dv_track_stereo_enqueue(ctx_, img.gray0.data, img.gray1.data, img.gray0.width, img.gray0.height, img.gray0.stride, img.time0,
                                                    img.inv_merge_mask.data, mode, mem_of(img.gray0)), "TrackImage");
    }
    FeatureBackground TrackImageCollect() {
        int n = 0;
        detail::check(ctx_, dv_track_stereo_collect(ctx_, rows_.data(), &n), "TrackImage");
        n_rows_ = n;
        return detail::to_points(rows_.data(), n);
    }
    // FeatureTracker::img_track() / prev_img / cur_img (background_tracker.h:41-88): the publishers' `image_track` picture.  The shim keeps host copies of
    // the last two left frames when keep_images is set (host gray input only) and DrawTrack()s the tracked points onto a BGR copy: colour by track age
    // (blue = new ... red = 20+ frames) like FeatureTracker::DrawTrack (background_tracker.cpp:598-650).  Visualisation only: not on the measured path.
    bool keep_images = false;
    struct HostImage { int width = 0, height = 0, channels = 1; std::vector<uint8_t> data; bool empty() const { return data.empty(); } };
    HostImage prev_img, cur_img;
    const HostImage& img_track() {
        img_track_.width = cur_img.width; img_track_.height = cur_img.height; img_track_.channels = 3;
        img_track_.data.resize((size_t)cur_img.width * cur_img.height * 3);
        for (size_t i = 0; i < (size_t)cur_img.width * cur_img.height; ++i) { const uint8_t g = cur_img.data.empty() ? 0 : cur_img.data[i]; img_track_.data[3 * i] = img_track_.data[3 * i + 1] = img_track_.data[3 * i + 2] = g; }
        for (int k = 0; k < n_rows_; ++k) {
            const double len = std::min(1.0, rows_[k].track_cnt / 20.0);
            const uint8_t b = (uint8_t)(255 * (1 - len)), r = (uint8_t)(255 * len);
            const int cx = (int)std::lround(rows_[k].left[3]), cy = (int)std::lround(rows_[k].left[4]);
            for (int dy = -2; dy <= 2; ++dy) for (int dx = -2; dx <= 2; ++dx) {
                const int x = cx + dx, y = cy + dy;
                if (dx * dx + dy * dy > 5 || x < 0 || y < 0 || x >= img_track_.width || y >= img_track_.height) continue;
                uint8_t* px = &img_track_.data[3 * ((size_t)y * img_track_.width + x)]; px[0] = b; px[1] = 0; px[2] = r;
            }
        }
        return img_track_;
    }
    const dv_feat* rows() const { return rows_.data(); }      // the same output as flat rows (what Estimator::ProcessMeasurements consumes)
    int n_rows() const { return n_rows_; }
    dv_ctx* ctx() { return ctx_; }
    double cur_time = 0;

    // cfg::is_undistort_input (utils/camera_model.cpp:481-499): hand over cam_s.{left,right}_undist_map1 / _map2 (CV_16SC2 / CV_16UC1, image
    // size); from then on TrackImage* takes the DISTORTED frames — gray, or BGR with ImageView::bgr — and undistorts them on the way into
    // pyramid level 0 (ImageProcessor::Run's cv::remap + SetGrayImageGpu fused).  The config's camera files must then describe the new,
    // distortion-free intrinsics, as the reference resets them.  map1 == nullptr removes the maps.
    void SetUndistortMaps(int cam, const int16_t* map1_xy, const uint16_t* map2) {
        detail::check(ctx_, dv_set_undistort_maps(ctx_, cam, map1_xy, map2, cfg_.width, cfg_.height), "SetUndistortMaps");
    }
#ifdef DVINS_SHIM_WITH_OPENCV
    void SetUndistortMaps(int cam, const cv::Mat& map1, const cv::Mat& map2) {
        if (map1.type() != CV_16SC2 || map2.type() != CV_16UC1 || !map1.isContinuous() || !map2.isContinuous() || map1.cols != cfg_.width || map1.rows != cfg_.height)
            throw std::runtime_error("dvins: SetUndistortMaps: maps must be continuous CV_16SC2 / CV_16UC1 of the image size (initUndistortRectifyMap(..., CV_16SC2, ...))");
        SetUndistortMaps(cam, map1.ptr<int16_t>(), map2.ptr<uint16_t>());
    }
#endif
    // cv::remap(src, dst, map1, map2, INTER_LINEAR) of one 8-bit image with 1 or 3 channels (e.g. the merged instance mask, semantic_image.cpp:86-89)
    void Remap(const uint8_t* src, int stride, int channels, const int16_t* map1_xy, const uint16_t* map2, uint8_t* dst) {
        detail::check(ctx_, dv_remap(ctx_, src, cfg_.width, cfg_.height, stride, channels, map1_xy, map2, dst, DV_MEM_HOST), "Remap");
    }

private:
    void init(const dv_config& c) {
        cfg_ = c;
        ctx_ = dv_create(&cfg_);
        if (!ctx_) throw std::runtime_error(std::string("dvins: FeatureTracker: ") + dv_last_error(nullptr));
        rows_.resize(DV_MAX_FEATS);
    }
    static int mem_of(const ImageView& v) { return (v.device ? DV_MEM_DEVICE : DV_MEM_HOST) | (v.bgr ? DV_FMT_BGR : 0); }
    void check_image(const SemanticImage& img) const {
        if (img.gray0.empty() || (cfg_.stereo && img.gray1.empty())) throw std::runtime_error("dvins: TrackImage: empty image");
        if (img.gray0.width != cfg_.width || img.gray0.height != cfg_.height) throw std::runtime_error("dvins: TrackImage: image size differs from image_width/image_height");   // main.cpp:95-99
    }
    void remember(const SemanticImage& img) {
        if (!keep_images || img.gray0.device || img.gray0.bgr) return;
        prev_img = std::move(cur_img);
        cur_img.width = img.gray0.width; cur_img.height = img.gray0.height; cur_img.channels = 1; cur_img.data.resize((size_t)cur_img.width * cur_img.height);
        for (int y = 0; y < cur_img.height; ++y) std::memcpy(&cur_img.data[(size_t)y * cur_img.width], img.gray0.data + (size_t)y * img.gray0.stride, cur_img.width);
    }
    HostImage img_track_;
    FeatureBackground track(SemanticImage& img, int mode) {
        check_image(img);
        remember(img);
        cur_time = img.time0;
        int n = 0;
        detail::check(ctx_, dv_track_stereo(ctx_, img.gray0.data, img.gray1.data, img.gray0.width, img.gray0.height, img.gray0.stride, img.time0,
                                            img.inv_merge_mask.data, mode, mem_of(img.gray0), rows_.data(), &n), "TrackImage");
        n_rows_ = n;
        return detail::to_points(rows_.data(), n);
    }
    dv_config cfg_{};
    dv_ctx* ctx_ = nullptr;
    std::vector<dv_feat> rows_;
    int n_rows_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------
// InstsFeatManager (front_end/dynamic_tracker.h:42-101): the per-object tracker of dynamic mode on the FeatureTracker's device context.
// The reference fills `instances` from SemanticImage::boxes2d after MOT association (AddInstancesByTracking / AddViodeInstances); here the detections
// arrive as dv_inst_det (track id, class, rectangle, ROI mask, optional extra 3-D points).  InstsTrack must follow the TrackSemanticImage enqueue of the
// same frame (the reference runs the two on two threads over the same image, system/main.cpp:247-250).
struct FeatureInstance {                // basic/frontend_feature.h:58-66, flat
    unsigned int id = 0; bool has_box3d = false; dv_box3d box3d{}; float rect[4] = {0, 0, 0, 0};
    std::vector<dv_feat> features;      // left = (x_n, y_n, 1, u, v, vx, vy) in ROI pixel coordinates for u, v; right likewise when has_right
    std::vector<std::array<double, 3>> points;
};
class InstsFeatManager {
public:
    using Ptr = std::shared_ptr<InstsFeatManager>;
    // max_dynamic_cnt / min_dynamic_dist / use_det3d of the config file (front_end_parameters.cpp:30-36, utils/parameters.cpp)
    InstsFeatManager(FeatureTracker& tracker, const std::string& config_path) : ctx_(tracker.ctx()) {
        const Config c = ReadConfig(config_path);
        detail::check(ctx_, dv_inst_config(ctx_, c.max_dynamic_cnt, c.min_dynamic_dist, c.use_det3d), "InstsFeatManager");
        alloc();
    }
    InstsFeatManager(FeatureTracker& tracker, int max_dynamic_cnt, int min_dynamic_dist, int use_det3d) : ctx_(tracker.ctx()) {
        detail::check(ctx_, dv_inst_config(ctx_, max_dynamic_cnt, min_dynamic_dist, use_det3d), "InstsFeatManager");
        alloc();
    }
    // SemanticImage::disp (CV_32F) of the frame the next InstsTrack call processes: the library then runs InstFeat::DetectExtraPoints + ProcessExtraPoints
    // (instance_feature.cpp:413-461, dynamic_tracker.cpp:159-340) for every visible object on the device; without it dv_inst_det::points are handed through
    void SetDisparity(const float* disp, int stride_bytes, double baseline, bool device = false) {
        detail::check(ctx_, dv_inst_set_disparity(ctx_, disp, stride_bytes, device ? DV_MEM_DEVICE : DV_MEM_HOST, baseline), "SetDisparity");
    }
    // cfg::dataset == kViode: the keys of SemanticImage::seg1 (VIODE::PixelToKey per pixel, dv_viode_mask's key image) of the frame the next InstsTrack processes —
    // TrackRightByPad keeps a right-image point only where seg1 carries the object's key (front_end/instance_feature.cpp:263-268)
    void SetRightKeys(const uint32_t* key_image, int stride_bytes = 0, bool device = false) {
        detail::check(ctx_, dv_inst_set_right_keys(ctx_, key_image, stride_bytes, device ? DV_MEM_DEVICE : DV_MEM_HOST), "SetRightKeys");
    }
    // SetEstimatedInstancesInfo(estimator->im.GetOutputInstInfo()) + the unmasking of FeatureTrack (system/main.cpp:194,217-245; para::is_static_inst_as_background):
    // call BEFORE the frame's TrackSemanticImage with the frame's detections and Estimator::StaticInstances() — the pixels of the static ones leave the merged mask
    void UnmaskStaticInstances(const std::vector<dv_inst_det>& dets, const std::vector<uint32_t>& static_ids) {
        detail::check(ctx_, dv_track_unmask_static(ctx_, dets.empty() ? nullptr : dets.data(), (int)dets.size(), static_ids.empty() ? nullptr : static_ids.data(), (int)static_ids.size()), "UnmaskStaticInstances");
    }
    void InstsTrack(double time, const std::vector<dv_inst_det>& dets, const std::vector<dv_box3d>& boxes3d = {}) {
        detail::check(ctx_, dv_inst_track_enqueue(ctx_, time, dets.empty() ? nullptr : dets.data(), (int)dets.size(), boxes3d.empty() ? nullptr : boxes3d.data(), (int)boxes3d.size()), "InstsTrack");
        pending_ = true;
    }
    // flat form (what Estimator::ProcessMeasurements(dynamic) consumes) — valid until the next Output()
    void Collect() {
        if (!pending_) return;
        detail::check(ctx_, dv_inst_track_collect(ctx_, insts_.data(), (int)insts_.size(), &n_insts_, feats_.data(), (int)feats_.size(), &n_feats_, points_.data(), (int)points_.size() / 3, &n_points_), "Output");
        pending_ = false;
    }
    std::map<unsigned int, FeatureInstance> Output() {      // front_end/dynamic_tracker.cpp:521-577
        Collect();
        std::map<unsigned int, FeatureInstance> out;
        for (int i = 0; i < n_insts_; ++i) {
            FeatureInstance f; f.id = insts_[i].id; f.has_box3d = insts_[i].has_box3d != 0; f.box3d = insts_[i].box3d; std::memcpy(f.rect, insts_[i].rect, sizeof(f.rect));
            f.features.assign(feats_.begin() + insts_[i].first_feat, feats_.begin() + insts_[i].first_feat + insts_[i].n_feats);
            for (int k = 0; k < insts_[i].n_points; ++k) { const double* p = &points_[3 * (size_t)(insts_[i].first_point + k)]; f.points.push_back({p[0], p[1], p[2]}); }
            out.emplace(f.id, std::move(f));
        }
        return out;
    }
    const dv_inst_obs* insts() const { return insts_.data(); } int n_insts() const { return n_insts_; }
    const dv_feat* feats() const { return feats_.data(); } int n_feats() const { return n_feats_; }
    const double* points() const { return points_.data(); } int n_points() const { return n_points_; }
private:
    void alloc() { insts_.resize(64); feats_.resize(64 * 256); points_.resize(3 * 65536); }
    dv_ctx* ctx_; bool pending_ = false;
    std::vector<dv_inst_obs> insts_; std::vector<dv_feat> feats_; std::vector<double> points_; int n_insts_ = 0, n_feats_ = 0, n_points_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------
class Estimator {
public:
    using Ptr = std::shared_ptr<Estimator>;
    enum SolverFlag { kInitial = 0, kNonLinear = 1 };
    explicit Estimator(const std::string& config_path, int device = 0, const std::string& seq_name = "") {
        cfg_ = ReadConfig(config_path, device, seq_name).est;
        dv_config fc{};
        fc.width = 64; fc.height = 48; fc.max_cnt = 8; fc.min_dist = 8; fc.flow_back = 1; fc.stereo = 1; fc.device = device;
        fc.cam0 = dv_cam{1, 1, 0, 0, 0, 0, 0, 0}; fc.cam1 = fc.cam0;
        ctx_ = dv_create(&fc);
        if (!ctx_) throw std::runtime_error(std::string("dvins: Estimator: ") + dv_last_error(nullptr));
        SetParameter();
    }
    ~Estimator() { if (ctx_) dv_destroy(ctx_); }
    Estimator(const Estimator&) = delete;
    Estimator& operator=(const Estimator&) = delete;

    void SetParameter() {                                    // estimator.cpp:1701-1716
        std::lock_guard<std::mutex> lk(process_mutex_);
        detail::check(ctx_, created_ ? dv_est_reset(ctx_) : dv_est_create(ctx_, &cfg_), "SetParameter");
        created_ = true;
    }
    void ClearState() {                                      // estimator.cpp:1719-1757
        std::lock_guard<std::mutex> lk(process_mutex_);
        detail::check(ctx_, dv_est_reset(ctx_), "ClearState");
        state_ = dv_est_state{};
    }
    void InputIMU(double t, const Vec3d& acc, const Vec3d& gyr) {
        std::lock_guard<std::mutex> lk(buf_mutex_);
        detail::check(ctx_, dv_est_input_imu(ctx_, t, acc.data(), gyr.data()), "InputIMU");
    }
    // one iteration of the reference's ProcessMeasurements loop: returns false when the IMU stream does not yet cover
    // the frame (the reference sleeps 5 ms and retries, estimator.cpp:1800-1812)
    bool ProcessMeasurements(const FrontendFeature& f) {
        const std::vector<dv_feat> rows = detail::to_rows(f.features);
        if (cfg_.use_line) {                                 // AddFeatureCheckParallax(frame, image.features, td) (estimator.cpp:1524-1527, feature_manager.cpp:124-160)
            std::vector<dv_line_row> lr;
            for (auto& [id, v] : f.features.lines) {
                if (v.empty()) continue;
                dv_line_row r{}; r.id = id; r.left[0] = v[0].second.x1; r.left[1] = v[0].second.y1; r.left[2] = v[0].second.x2; r.left[3] = v[0].second.y2;
                if (v.size() > 1) { r.has_right = 1; r.right[0] = v[1].second.x1; r.right[1] = v[1].second.y1; r.right[2] = v[1].second.x2; r.right[3] = v[1].second.y2; }
                lr.push_back(r);
            }
            std::lock_guard<std::mutex> lk(process_mutex_);
            detail::check(ctx_, dv_est_set_lines(ctx_, lr.data(), (int)lr.size()), "ProcessMeasurements");
        }
        return ProcessMeasurements(rows.data(), (int)rows.size(), f.time);
    }
    bool ProcessMeasurements(const dv_feat* rows, int n, double time) {
        std::lock_guard<std::mutex> lk(process_mutex_);
        int rc;
        { std::lock_guard<std::mutex> lk2(buf_mutex_); rc = dv_est_process(ctx_, rows, n, time, &state_); }
        if (rc == 1) return false;
        detail::check(ctx_, rc, "ProcessMeasurements");
        after_frame();
        return true;
    }
    // dynamic mode: FrontendFeature::instances travel with the background features (estimator.cpp:1562-1622)
    bool ProcessMeasurements(const dv_feat* rows, int n, double time, const InstsFeatManager& insts) {
        std::lock_guard<std::mutex> lk(process_mutex_);
        int rc;
        { std::lock_guard<std::mutex> lk2(buf_mutex_);
          rc = dv_est_process_dynamic(ctx_, rows, n, time, insts.n_insts() ? insts.insts() : nullptr, insts.n_insts(), insts.n_feats() ? insts.feats() : nullptr,
                                      insts.n_points() ? insts.points() : nullptr, &state_); }
        if (rc == 1) return false;
        detail::check(ctx_, rc, "ProcessMeasurements");
        after_frame();
        return true;
    }
    // FeatureManager::line_landmarks after the last frame (Publisher::PubLines reads ptw1 / ptw2 of the triangulated ones)
    std::vector<dv_line_landmark> Lines() {
        std::vector<dv_line_landmark> v(4096); int n = 0;
        detail::check(ctx_, dv_est_get_lines(ctx_, v.data(), (int)v.size(), &n), "Lines");
        v.resize(n); return v;
    }
    // Estimator::im.instances after the last frame (InstanceManager::SetOutputInstInfo / the instance publishers)
    std::vector<dv_inst_state> Instances() {
        std::vector<dv_inst_state> v(64); int n = 0;
        detail::check(ctx_, dv_est_get_instances(ctx_, v.data(), (int)v.size(), &n, nullptr), "Instances");
        v.resize(n); return v;
    }
    // estimator->im.GetOutputInstInfo() as FeatureTrack uses it (system/main.cpp:194,217-245): the ids of the instances reported static by the last dynamic frame
    std::vector<uint32_t> StaticInstances() {
        std::lock_guard<std::mutex> lk(process_mutex_);
        std::vector<uint32_t> v(256); int n = 0;
        detail::check(ctx_, dv_est_get_static_instances(ctx_, v.data(), (int)v.size(), &n), "StaticInstances");
        v.resize(n); return v;
    }
    // Estimator::ChangeSensorType (estimator.cpp:697-726)
    void ChangeSensorType(int use_imu, int use_stereo) {
        std::lock_guard<std::mutex> lk(process_mutex_);
        detail::check(ctx_, dv_est_change_sensor_type(ctx_, use_imu, use_stereo), "ChangeSensorType");
        cfg_.use_imu = use_imu;
    }
    // latest_P / latest_Q / latest_V: FastPredictIMU's output, published on `imu_propagate` (estimator.cpp:729-742); false until the estimator is initialised
    bool LatestState(double& t, Vec3d& P, std::array<double, 4>& Q_xyzw, Vec3d& V) {
        std::lock_guard<std::mutex> lk(buf_mutex_);
        const int rc = dv_est_get_latest(ctx_, &t, P.data(), Q_xyzw.data(), V.data());
        if (rc == 1) return false;
        detail::check(ctx_, rc, "LatestState");
        return true;
    }
    // Set/GetOutputEgoInfo (estimator.h:71-82): pose of the newest frame + extrinsic, written after every frame (estimator.cpp:1856) and read by PubTF
    struct EgoInfo { std::array<double, 9> R; Vec3d P; std::array<double, 9> R_bc; Vec3d P_bc; };
    void SetOutputEgoInfo(const EgoInfo& e) { std::lock_guard<std::mutex> lk(out_pose_mutex_); ego_ = e; }
    EgoInfo GetOutputEgoInfo() { std::lock_guard<std::mutex> lk(out_pose_mutex_); return ego_; }
    std::vector<Vec3d> key_poses;                            // body.Ps[0..kWinSize] after every frame (estimator.cpp:1681-1683)
    // feat_manager.point_landmarks for the `point_cloud` / `margin_cloud` publishers (utils/io/visualization.cpp:214-249)
    std::vector<dv_landmark> Landmarks() {
        std::vector<dv_landmark> v(2048); int n = 0;
        detail::check(ctx_, dv_est_get_landmarks(ctx_, v.data(), (int)v.size(), &n), "Landmarks");
        v.resize(n); return v;
    }
    // Estimator::ProcessMeasurements() (estimator.cpp:1786-1863) as the reference runs it on thread T3: blocks until cfg::ok turns false, over the global
    // feature_queue.  Same order of tests as the reference's loop body: empty queue -> sleep 2 ms; front_time(); IMU stream not yet past cur_time = stamp + td ->
    // sleep 5 ms WITHOUT popping (the frame keeps its queue slot, so the producer's drop rule sees the same occupancy); then request() pops and the frame is processed.
    void ProcessMeasurements() { ProcessMeasurements(feature_queue, cfg::ok); }
    void ProcessMeasurements(FeatureQueue& queue, const std::atomic<bool>& ok) {
        while (ok.load()) {
            if (queue.empty()) { std::this_thread::sleep_for(std::chrono::milliseconds(2)); continue; }
            const std::optional<double> front_time = queue.front_time();
            if (!front_time) continue;
            if (cfg_.use_imu) {
                int avail;
                { std::lock_guard<std::mutex> lk(buf_mutex_); avail = dv_est_imu_available(ctx_, *front_time); }
                if (avail < 0) detail::check(ctx_, avail, "ProcessMeasurements");
                if (!avail) { std::this_thread::sleep_for(std::chrono::milliseconds(5)); continue; }
            }
            std::optional<FrontendFeature> f = queue.request();
            if (!f) continue;
            feature_frame = std::move(*f);
            if (!ProcessMeasurements(feature_frame)) throw std::runtime_error("dvins: ProcessMeasurements: IMU interval vanished between IMUAvailable and GetIMUInterval");
            ++processed_frames;
        }
    }
    // north_star's spelling (VINS-Fusion's Estimator::processImage): one ProcessImage call on an already-popped frame; header = image.time as in the reference
    bool processImage(const FrontendFeature& image, double header) { FrontendFeature f = image; f.time = header; return ProcessMeasurements(f); }
    bool ProcessImage(const FrontendFeature& image, double header) { return processImage(image, header); }      // estimator.cpp:1516
    FrontendFeature feature_frame;                           // the frame ProcessMeasurements() popped last (estimator.h public member)
    std::atomic<long long> processed_frames{0};
    // private queue form kept for callers that do not want the global: pops until `ok` turns false
    void PushFeature(FrontendFeature f) { { std::lock_guard<std::mutex> lk(q_mutex_); queue_.push_back(std::move(f)); } q_cv_.notify_one(); }
    void ProcessMeasurements(const std::atomic<bool>& ok) {
        while (ok.load()) {
            FrontendFeature f;
            {
                std::unique_lock<std::mutex> lk(q_mutex_);
                if (!q_cv_.wait_for(lk, std::chrono::milliseconds(2), [&] { return !queue_.empty(); })) continue;
                f = std::move(queue_.front()); queue_.pop_front();
            }
            while (ok.load() && !ProcessMeasurements(f)) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
    }
    // body.Ps / body.Rs / Vs / Bas / Bgs of window slot i (body.h:94): [px py pz qx qy qz qw vx vy vz bax bay baz bgx bgy bgz]
    std::array<double, 16> WindowState(int i) const { std::array<double, 16> a; std::memcpy(a.data(), state_.window[i], sizeof(double) * 16); return a; }
    const dv_est_state& state() const { return state_; }
    dv_ctx* ctx() { return ctx_; }

    enum MarginFlag { kMarginOld = 0, kMarginSecondNew = 1 };      // vio_parameters.h:57-60
    SolverFlag solver_flag = kInitial;
    MarginFlag margin_flag = kMarginOld;                       // estimator.h:154
    bool margin_old = false;
    int frame = 0;

private:
    static std::array<double, 9> quat_to_R(const double* q /* x y z w */) {
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        return { 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y) };
    }
    void after_frame() {
        solver_flag = state_.nonlinear ? kNonLinear : kInitial;
        margin_old = state_.margin_old != 0;
        margin_flag = margin_old ? kMarginOld : kMarginSecondNew;
        frame = state_.frame;
        key_poses.clear();
        for (int i = 0; i <= 10; ++i) key_poses.push_back(Vec3d{ state_.window[i][0], state_.window[i][1], state_.window[i][2] });
        if (state_.nonlinear) {          // SetOutputEgoInfo(body.Rs[kWinSize], body.Ps[kWinSize], body.ric[0], body.tic[0]) (estimator.cpp:1856)
            EgoInfo e; e.R = quat_to_R(&state_.window[10][3]); e.P = Vec3d{ state_.window[10][0], state_.window[10][1], state_.window[10][2] };
            for (int k = 0; k < 9; ++k) e.R_bc[k] = cfg_.ric[0][k];
            e.P_bc = Vec3d{ cfg_.tic[0][0], cfg_.tic[0][1], cfg_.tic[0][2] };
            SetOutputEgoInfo(e);
        }
    }
    std::mutex out_pose_mutex_; EgoInfo ego_{};
    dv_est_config cfg_;
    dv_est_state state_{};
    dv_ctx* ctx_ = nullptr;
    bool created_ = false;
    std::mutex buf_mutex_, process_mutex_, q_mutex_;
    std::condition_variable q_cv_;
    std::deque<FrontendFeature> queue_;
};

// ---- the per-frame object solve of dynamic mode: what InstanceManager::Optimization hands to ceres (estimator_insts.cpp:772-807) ----
// state: n_obj x 11 x 7 = Instance::para_state after SetOptimizeParameters, dims: n_obj x 3 = para_box; both are updated in place and read
// back by Instance::GetOptimizationParameters (estimator/instance.cpp:421-503).  boxes: one record per (object, frame) with a 3-D detection
// (BoxDimsFactor + BoxOrientationFactor), points: one per triangulated observation (BoxEncloseStereoPointFactor).
struct InstanceSolveSummary { int iterations = 0, successful = 0, termination = 0; double initial_cost = 0, final_cost = 0; };
inline InstanceSolveSummary OptimizeInstances(dv_ctx* ctx, int n_obj, double* state, double* dims, const double* body_para_pose /* 11 x 7 */,
                                              const double* ric0 /* 3 x 3 row-major */, const std::vector<dv_obj_box>& boxes,
                                              const std::vector<dv_obj_point>& points, int max_num_iterations, int plane_kind = 0) {
    dv_obj_problem p{};
    p.n_obj = n_obj; p.n_boxes = (int)boxes.size(); p.n_points = (int)points.size(); p.max_iters = max_num_iterations; p.plane_kind = plane_kind;
    p.state = state; p.dims = dims; p.body_pose = body_para_pose;
    std::memcpy(p.R_bc, ric0, sizeof(p.R_bc));
    p.boxes = boxes.empty() ? nullptr : boxes.data(); p.points = points.empty() ? nullptr : points.data();
    dv_ba_summary s{};
    detail::check(ctx, dv_obj_solve(ctx, &p, &s), "OptimizeInstances");
    InstanceSolveSummary r; r.iterations = s.iterations; r.successful = s.successful; r.termination = s.termination; r.initial_cost = s.initial_cost; r.final_cost = s.final_cost;
    return r;
}

// ---- the two frame-flow policies that sit between the ROS callbacks and the path, ROS-free ----
// StereoSync: the left / right part of SystemCallBack::SyncProcess (utils/io/system_call_back.cpp:97-135): the oldest left image is the
// reference; a right image within kDelay = 5 ms (utils/parameters.h:47) pairs with it, older right images are discarded, and a left image
// that is more than kDelay OLDER than the oldest right image is dropped.  TryPop() is one trip of the reference's loop body.
template <class Img>
class StereoSync {
public:
    static constexpr double kDelay = 0.005;
    void PushLeft(double t, Img img) { std::lock_guard<std::mutex> lk(m_); left_.emplace_back(t, std::move(img)); }
    void PushRight(double t, Img img) { std::lock_guard<std::mutex> lk(m_); right_.emplace_back(t, std::move(img)); }
    // true: (t0, left, t1, right) is a synchronised pair.  false: wait for more input (the reference sleeps 2 ms) — possibly after having
    // dropped a too-early left image.
    bool TryPop(double& t0, Img& left, double& t1, Img& right) {
        std::lock_guard<std::mutex> lk(m_);
        if (left_.empty() || right_.empty()) return false;
        t0 = left_.front().first;
        t1 = right_.front().first;
        if (t0 + kDelay < t1) { left_.pop_front(); ++dropped_left; return false; }          // img0 too early: it is gone (:118-122)
        if (t1 + kDelay < t0) {                                                              // right images too early: discard them (:123-128)
            while (!right_.empty() && t0 - right_.front().first > kDelay) { right_.pop_front(); ++dropped_right; }
            if (right_.empty()) return false;      // the reference reads front() of the emptied deque here (undefined); the shim keeps the left image and waits
            t1 = right_.front().first;
        }
        left = std::move(left_.front().second); left_.pop_front();
        right = std::move(right_.front().second); right_.pop_front();
        return true;
    }
    // (under the lock, and atomic counters: a monitoring thread reads these beside the callbacks — unguarded reads of the deques' sizes were a data race ThreadSanitizer
    //  reported, tests/host/shim_tsan.cpp)
    size_t pending_left() const { std::lock_guard<std::mutex> lk(m_); return left_.size(); }
    size_t pending_right() const { std::lock_guard<std::mutex> lk(m_); return right_.size(); }
    std::atomic<int> dropped_left{0}, dropped_right{0};
private:
    mutable std::mutex m_;
    std::deque<std::pair<double, Img>> left_, right_;
};
// FrameGate: which tracked frames reach the estimator (system/main.cpp:297-307): every frame on KITTI, every second one otherwise.
struct FrameGate {
    bool every_frame = false;              // cfg::dataset == DatasetType::kKitti
    int cnt = 0;
    bool Pass() { const bool p = every_frame || (cnt % 2 == 0); ++cnt; return p; }
};

// ---- on-disk formats on either side of the path ----
// trajectory line of SaveBodyTrajectory (utils/io/output.cpp:189-227): "<sec>.<nsec 9 digits> px py pz qx qy qz qw", fixed, 6 decimals
inline std::string TumLine(double stamp, const std::array<double, 16>& s) {
    long long sec = (long long)std::floor(stamp), nsec = std::llround((stamp - (double)sec) * 1e9);
    if (nsec >= 1000000000LL) { sec += 1; nsec -= 1000000000LL; }
    char buf[320];
    std::snprintf(buf, sizeof(buf), "%lld.%09lld %.6f %.6f %.6f %.6f %.6f %.6f %.6f", sec, nsec, s[0], s[1], s[2], s[3], s[4], s[5], s[6]);
    return buf;
}
// point features of one frame, one line per id: "<0|1> id x y z u v vx vy [x y z u v vx vy]" (utils/io/feature_serialization.cpp:26-75)
inline void SerializePointFeature(const std::string& path, const std::map<unsigned int, std::vector<std::pair<int, Vec7d>>>& points) {
    std::ofstream f(path);
    if (!f) throw std::runtime_error("dvins: cannot write " + path);
    auto num = [](double v) { char b[40]; std::snprintf(b, sizeof(b), "%.17g", v); double back = std::strtod(b, nullptr);
                              for (int p = 1; p < 17; ++p) { char c[40]; std::snprintf(c, sizeof(c), "%.*g", p, v); if (std::strtod(c, nullptr) == v) return std::string(c); }
                              (void)back; return std::string(b); };      // shortest digits that round-trip
    for (const auto& kv : points) {
        f << (kv.second.size() == 1 ? "0 " : "1 ") << kv.first;
        for (size_t o = 0; o < kv.second.size() && o < 2; ++o) for (double v : kv.second[o].second) f << ' ' << num(v);
        f << '\n';
    }
}
inline std::map<unsigned int, std::vector<std::pair<int, Vec7d>>> DeserializePointFeature(const std::string& path) {
    std::map<unsigned int, std::vector<std::pair<int, Vec7d>>> points;
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream is(line);
        int flag; unsigned int id; Vec7d v;
        if (!(is >> flag >> id)) continue;
        for (double& x : v) is >> x;
        points[id].emplace_back(0, v);
        if (flag == 1) { for (double& x : v) is >> x; points[id].emplace_back(1, v); }
    }
    return points;
}

}  // namespace dynamic_vins
