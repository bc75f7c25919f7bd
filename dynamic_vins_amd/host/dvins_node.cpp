// dvins_node — the ROS-free node: what the reference's `dynamic_vins` executable does between its inputs and its trajectory file when `use_dataloader` is set
// (system/main.cpp:334-421 Run, :59-171 ImageProcess with Dataloader::LoadStereo utils/io/dataloader.cpp:62-88, :178-330 FeatureTrack, :394-404 the estimator
// thread, Estimator::Output -> SaveBodyTrajectory utils/io/output.cpp:189-227), on the C++ runner of the library (dv_runner, csrc/runner.hip).
//
//   dvins_node <config.yaml> <sequence dir> [output dir] [--seq NAME] [--kitti-calib DIR] [--max-frames N] [--device D]
//
//   <sequence dir>/left/*.{pgm,png}  <sequence dir>/right/*.{pgm,png}   stereo pairs, sorted by name like Dataloader's std::sort (8-bit gray, or RGB which is
//                                                                        reduced with cvtColor's BGR2GRAY fixed-point weights)
//   <sequence dir>/times.txt          optional, one stamp [s] per pair; otherwise 0.05 s per pair from 0 as Dataloader::LoadStereo does (time += 0.05)
//   <sequence dir>/imu.csv            with `imu: 1`: EuRoC layout  t[ns], wx, wy, wz, ax, ay, az  (lines starting with # are comments)
//   output: <output dir>/<seq>_<VIO|VO>_<raw|naive|dynamic>_<LinePoint|PointOnly>_Odometry.txt (utils/io/io_parameters.cpp:18-80), one line per frame handed to the
//           back end: "<sec>.<nsec> px py pz qx qy qz qw" (output.cpp:199-227)
//   <sequence dir>/segmentation0/*.png  <sequence dir>/segmentation1/*.png   dataset_type viode with slam_type naive / dynamic: the RGB label images of the two cameras
//                                                                        (the bag's image0/1_segmentation_topic, utils/io/system_call_back.cpp:18-35), one per pair
// Frame flow: every pair is tracked; every 2nd tracked pair goes to the back end unless dataset_type is kitti (system/main.cpp:300-307).
// VIODE (config 3) needs no network in the reference and none here: segmentation image -> VIODE::SetViodeMaskSimple (naive) / SetViodeMaskAndRoi (dynamic)
// (image_process/image_process.cpp:161-178, utils/dataset/viode_utils.cpp:21-218; dv_viode_mask does the per-pixel work) -> the inverse merged mask for TrackImageNaive
// resp. TrackSemanticImage, and in dynamic mode one instance per key present (AddViodeInstances, front_end/dynamic_tracker.cpp:585-605: track_id = key, rect =
// cv::Rect(min_pt, max_pt), ROI mask = the key's pixels) for InstsTrack, whose right-image points must carry the object's key in segmentation1
// (front_end/instance_feature.cpp:263-268).  No disparity network: the objects get no extra points (SemanticImage::disp stays empty).
// Other data sets' detector / segmentation / stereo networks are upstream of the path: such a config runs on the background tracker alone and says so.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <vector>
#include <zlib.h>

#include "dvins_shim.hpp"

using namespace dynamic_vins;

namespace {
struct Gray { int w = 0, h = 0; std::vector<uint8_t> d; std::vector<uint8_t> bgr; };      // bgr: filled for colour files when asked (segmentation images), B G R per pixel like cv::imread

std::vector<std::string> list_images(const std::string& dir) {
    std::vector<std::string> out;
    DIR* d = opendir(dir.c_str());
    if (!d) throw std::runtime_error("dvins_node: cannot open " + dir);
    while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        const size_t dot = n.rfind('.');
        if (dot == std::string::npos) continue;
        std::string ext = n.substr(dot + 1);
        for (char& c : ext) c = (char)std::tolower((unsigned char)c);
        if (ext == "pgm" || ext == "png") out.push_back(dir + "/" + n);
    }
    closedir(d);
    std::sort(out.begin(), out.end());
    return out;
}
std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("dvins_node: cannot read " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
inline uint8_t bgr2gray(int r, int g, int b) { return (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14); }      // cvtColor(BGR2GRAY), 14-bit fixed point
Gray read_pgm(const std::vector<uint8_t>& buf, const std::string& path) {
    size_t p = 0;
    auto token = [&]() {
        std::string t;
        for (;;) {
            while (p < buf.size() && std::isspace(buf[p])) ++p;
            if (p < buf.size() && buf[p] == '#') { while (p < buf.size() && buf[p] != '\n') ++p; continue; }
            break;
        }
        while (p < buf.size() && !std::isspace(buf[p])) t += (char)buf[p++];
        return t;
    };
    const std::string magic = token();
    if (magic != "P5" && magic != "P6") throw std::runtime_error("dvins_node: " + path + " is not a binary PGM / PPM");
    auto number = [&]() { const std::string t = token(); try { return std::stoi(t); } catch (const std::exception&) { throw std::runtime_error("dvins_node: " + path + ": bad PGM / PPM header"); } };
    Gray g; g.w = number(); g.h = number();
    if (number() != 255) throw std::runtime_error("dvins_node: " + path + ": only 8-bit images");
    ++p;
    const size_t ch = magic == "P5" ? 1 : 3;
    if (g.w <= 0 || g.h <= 0 || g.w > 65535 || g.h > 65535) throw std::runtime_error("dvins_node: " + path + ": bad image size");
    if (p > buf.size() || buf.size() - p < (size_t)g.w * g.h * ch) throw std::runtime_error("dvins_node: " + path + " is truncated");
    g.d.resize((size_t)g.w * g.h);
    if (ch == 1) std::memcpy(g.d.data(), buf.data() + p, g.d.size());
    else for (size_t i = 0; i < g.d.size(); ++i) g.d[i] = bgr2gray(buf[p + 3 * i], buf[p + 3 * i + 1], buf[p + 3 * i + 2]);
    return g;
}
// non-interlaced 8-bit PNG, colour type 0 (gray), 2 (RGB), 4 (gray + alpha), 6 (RGBA): what EuRoC / KITTI / VIODE ship
Gray read_png(const std::vector<uint8_t>& buf, const std::string& path, bool keep_colour = false) {
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (buf.size() < 33 || std::memcmp(buf.data(), sig, 8)) throw std::runtime_error("dvins_node: " + path + " is not a PNG");
    auto be32 = [&](size_t o) { return ((uint32_t)buf[o] << 24) | ((uint32_t)buf[o + 1] << 16) | ((uint32_t)buf[o + 2] << 8) | buf[o + 3]; };
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> z;
    for (size_t o = 8; o + 12 <= buf.size();) {
        const uint32_t len = be32(o); const char* tag = (const char*)&buf[o + 4];
        if (o + 12 + len > buf.size()) break;
        if (!std::memcmp(tag, "IHDR", 4)) { if (len < 13) throw std::runtime_error("dvins_node: " + path + ": short IHDR chunk"); w = (int)be32(o + 8); h = (int)be32(o + 12); depth = buf[o + 16]; ctype = buf[o + 17]; interlace = buf[o + 20]; }
        else if (!std::memcmp(tag, "IDAT", 4)) z.insert(z.end(), buf.begin() + o + 8, buf.begin() + o + 8 + len);
        else if (!std::memcmp(tag, "IEND", 4)) break;
        o += 12 + len;
    }
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (w <= 0 || h <= 0 || depth != 8 || !ch || interlace) throw std::runtime_error("dvins_node: " + path + ": only non-interlaced 8-bit gray / RGB(A) PNGs");
    const size_t row = (size_t)w * ch;
    // the header is input like any other: deflate expands at most ~1032 : 1, so a size the compressed stream cannot fill is refused BEFORE anything of that size is allocated
    if (w > 65535 || h > 65535 || (row + 1) * (size_t)h > z.size() * 1032 + 1024) throw std::runtime_error("dvins_node: " + path + ": image size and compressed data do not fit together");
    std::vector<uint8_t> raw((row + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, z.data(), (uLong)z.size()) != Z_OK || out_len != raw.size()) throw std::runtime_error("dvins_node: " + path + ": inflate failed");
    std::vector<uint8_t> prev(row, 0), cur(row);
    Gray g; g.w = w; g.h = h; g.d.resize((size_t)w * h);
    if (keep_colour) g.bgr.resize((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) {
        const uint8_t* in = raw.data() + (row + 1) * y; const int f = in[0];
        for (size_t x = 0; x < row; ++x) {
            const int a = x >= (size_t)ch ? cur[x - ch] : 0, b = prev[x], c = x >= (size_t)ch ? prev[x - ch] : 0;
            int v = in[1 + x];
            if (f == 1) v += a; else if (f == 2) v += b; else if (f == 3) v += (a + b) >> 1;
            else if (f == 4) { const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            cur[x] = (uint8_t)v;
        }
        for (int x = 0; x < w; ++x) g.d[(size_t)y * w + x] = ch <= 2 ? cur[(size_t)x * ch] : bgr2gray(cur[(size_t)x * ch], cur[(size_t)x * ch + 1], cur[(size_t)x * ch + 2]);
        if (keep_colour) for (int x = 0; x < w; ++x) {
            uint8_t* o = &g.bgr[((size_t)y * w + x) * 3];
            if (ch <= 2) o[0] = o[1] = o[2] = cur[(size_t)x * ch]; else { o[0] = cur[(size_t)x * ch + 2]; o[1] = cur[(size_t)x * ch + 1]; o[2] = cur[(size_t)x * ch]; }
        }
        prev.swap(cur);
    }
    return g;
}
Gray read_image(const std::string& path, bool keep_colour = false) {
    const std::vector<uint8_t> buf = read_file(path);
    if (buf.size() > 1 && buf[0] == 'P') { if (keep_colour) throw std::runtime_error("dvins_node: " + path + ": segmentation images must be PNG"); return read_pgm(buf, path); }
    return read_png(buf, path, keep_colour);
}
struct Pinned {          // dv_pinned_alloc / dv_pinned_free (hipHostMalloc behind the C ABI: the node itself does not link the HIP runtime)
    uint8_t* p = nullptr;
    explicit Pinned(size_t bytes) { p = bytes ? static_cast<uint8_t*>(dv_pinned_alloc(bytes)) : nullptr; }
    ~Pinned() { if (p) dv_pinned_free(p); }
    Pinned(const Pinned&) = delete; Pinned& operator=(const Pinned&) = delete;
};
constexpr int kMinInstSize = 8;          // a VIODE instance's rectangle must be at least this many pixels on both sides to be handed to the object tracker
std::string stem(const std::string& path) {
    std::string s = path;
    while (s.size() > 1 && s.back() == '/') s.pop_back();
    const size_t a = s.find_last_of('/');
    return a == std::string::npos ? s : s.substr(a + 1);
}
}

int main(int argc, char** argv) {
    try {
        std::vector<std::string> pos; std::string seq_name, kitti_calib; int max_frames = 1 << 30, device = 0;
        if (argc >= 3 && std::string(argv[1]) == "--decode") {      // dvins_node --decode <image>...: size and a checksum of the decoded gray image (CPU; tests/test_node.py)
            for (int i = 2; i < argc; ++i) {
                const Gray g = read_image(argv[i]);
                unsigned long long sum = 0, wsum = 0;
                for (size_t k = 0; k < g.d.size(); ++k) { sum += g.d[k]; wsum += (unsigned long long)g.d[k] * (k % 65521 + 1); }
                std::printf("%d %d %llu %llu\n", g.w, g.h, sum, wsum);
            }
            return 0;
        }
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "--seq" && i + 1 < argc) seq_name = argv[++i];
            else if (a == "--kitti-calib" && i + 1 < argc) kitti_calib = argv[++i];
            else if (a == "--max-frames" && i + 1 < argc) max_frames = std::atoi(argv[++i]);
            else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
            else pos.push_back(a);
        }
        if (pos.size() < 2) { std::fprintf(stderr, "usage: dvins_node <config.yaml> <sequence dir> [output dir] [--seq NAME] [--kitti-calib DIR] [--max-frames N] [--device D]\n"); return 2; }
        const std::string cfg_path = pos[0], seq_dir = pos[1], out_dir = pos.size() > 2 ? pos[2] : ".";
        if (seq_name.empty()) seq_name = stem(seq_dir);
        Config cfg = ReadConfig(cfg_path, device, seq_name, kitti_calib);
        // dataset_type viode with slam_type naive / dynamic: everything the mode needs comes out of the segmentation images (cfg::is_input_seg, utils/parameters.cpp:59-64)
        const bool viode = cfg.input_seg && !cfg.viode_dynamic_keys.empty();
        if (cfg.input_seg && !viode) std::fprintf(stderr, "dvins_node: dataset_type viode with slam_type %s, but rgb_to_label_file / dynamic_label_id give no dynamic key: running without masks\n", cfg.slam_type.c_str());
        if ((cfg.dynamic || cfg.naive) && !cfg.input_seg) std::fprintf(stderr, "dvins_node: slam_type %s on this data set needs the detector / segmentation outputs (upstream of this path): running the background tracker without masks\n", cfg.slam_type.c_str());
        if (cfg.est.use_line) std::fprintf(stderr, "dvins_node: use_line needs the LSD / LBD detector's segments (upstream of this path): running without lines\n");
        const bool run_dynamic = cfg.dynamic && viode, run_naive = cfg.naive && viode;
        if (viode && cfg.viode_dynamic_keys.size() > 64) throw std::runtime_error("dvins_node: more than 64 dynamic keys (dv_viode_mask takes at most 64)");
        cfg.est.dynamic = run_dynamic ? 1 : 0; cfg.est.use_line = 0;
        if (run_dynamic) cfg.est.use_det3d = 0;          // (no 3-D detector output in a VIODE directory; viode.yaml ships use_det3d: 0)

        std::vector<std::string> lf = list_images(seq_dir + "/left"), rf = list_images(seq_dir + "/right");
        if (lf.empty() || lf.size() != rf.size()) throw std::runtime_error("dvins_node: left/ and right/ must hold the same, non-zero number of images");
        const int n = (int)std::min<size_t>(lf.size(), (size_t)max_frames);
        std::vector<Gray> L(n), R(n);
        for (int k = 0; k < n; ++k) {
            L[k] = read_image(lf[k]); R[k] = read_image(rf[k]);
            const bool left_ok = L[k].w == cfg.front.width && L[k].h == cfg.front.height;      // (system/main.cpp:95-99 enforces the configured size)
            if (!left_ok || R[k].w != L[k].w || R[k].h != L[k].h)
                throw std::runtime_error("dvins_node: " + (left_ok ? rf[k] : lf[k]) + " is not image_width x image_height of the config");
        }
        std::vector<double> times(n);
        { std::ifstream tf(seq_dir + "/times.txt"); double t = 0.0; for (int k = 0; k < n; ++k) { if (tf && (tf >> times[k])) continue; times[k] = t; t += 0.05; } }      // Dataloader::LoadStereo: time += 0.05
        std::vector<double> imu_t, imu_a, imu_g;
        if (cfg.est.use_imu) {
            std::ifstream f(seq_dir + "/imu.csv");
            if (!f) throw std::runtime_error("dvins_node: imu: 1 but " + seq_dir + "/imu.csv is missing");
            std::string line;
            while (std::getline(f, line)) {
                if (line.empty() || line[0] == '#') continue;
                for (char& c : line) if (c == ',') c = ' ';
                std::istringstream is(line); double t, w[3], a[3];
                if (!(is >> t >> w[0] >> w[1] >> w[2] >> a[0] >> a[1] >> a[2])) continue;
                imu_t.push_back(t > 1e12 ? t * 1e-9 : t); for (int i = 0; i < 3; ++i) { imu_g.push_back(w[i]); imu_a.push_back(a[i]); }
            }
        }

        dv_ctx* ctx = dv_create(&cfg.front);
        if (!ctx) throw std::runtime_error(std::string("dvins_node: ") + dv_last_error(nullptr));
        if (dv_est_create(ctx, &cfg.est)) throw std::runtime_error(std::string("dvins_node: ") + dv_last_error(ctx));
        // Everything the device reads per frame — frames, inverse masks, key images — lives in ONE pinned, device-mapped arena (dv_pinned_alloc) and is handed over as
        // DV_MEM_PINNED: the kernels read it in place over PCIe, no staging copy and no copy engine in the per-frame path.  If the arena cannot be had (a very long sequence),
        // the buffers stay pageable and travel as DV_MEM_HOST (hipMemcpy2DAsync per frame).
        const int W = cfg.front.width, H = cfg.front.height;
        const size_t px = (size_t)W * H;
        const bool want_masks = viode, want_keys = run_dynamic;
        const size_t arena_bytes = (size_t)n * px * (2 + (want_masks ? 1 : 0) + (want_keys ? 4 : 0));
        Pinned arena(arena_bytes);
        const int mem = arena.p ? DV_MEM_PINNED : DV_MEM_HOST;
        if (!arena.p) std::fprintf(stderr, "dvins_node: %zu MB of pinned memory are not available: pageable buffers (DV_MEM_HOST)\n", arena_bytes >> 20);
        std::vector<const uint8_t*> lp(n), rp(n);
        for (int k = 0; k < n; ++k) {
            if (arena.p) {
                uint8_t* a = arena.p + (size_t)k * 2 * px;
                std::memcpy(a, L[k].d.data(), px); std::memcpy(a + px, R[k].d.data(), px);
                lp[k] = a; rp[k] = a + px; L[k].d.clear(); L[k].d.shrink_to_fit(); R[k].d.clear(); R[k].d.shrink_to_fit();
            } else { lp[k] = L[k].d.data(); rp[k] = R[k].d.data(); }
        }
        uint8_t* mask_arena = arena.p ? arena.p + (size_t)n * 2 * px : nullptr;
        uint32_t* key_arena = (arena.p && want_keys) ? reinterpret_cast<uint32_t*>(arena.p + (size_t)n * (2 + (want_masks ? 1 : 0)) * px) : nullptr;
        dv_seq_input in{};
        in.left = lp.data(); in.right = rp.data(); in.times = times.data(); in.n_frames = n; in.mem = mem; in.stride = 0; in.ba_stride = cfg.every_frame ? 1 : 2;
        in.imu_t = imu_t.data(); in.imu_acc = imu_a.data(); in.imu_gyr = imu_g.data(); in.n_imu = (int)imu_t.size();

        // ---- VIODE: thread T1's part of the frame (ImageProcessor::Run, image_process.cpp:161-178) for every pair, before the run ----
        std::vector<std::vector<uint8_t>> inv_mask;                       // img.inv_merge_mask per frame
        std::vector<std::vector<uint32_t>> right_keys;                    // PixelToKey of every pixel of seg1
        std::vector<std::vector<dv_inst_det>> dets; std::vector<std::vector<std::vector<uint8_t>>> det_masks;
        std::vector<const uint8_t*> mask_ptr; std::vector<const uint32_t*> keys_ptr; std::vector<const dv_inst_det*> det_ptr; std::vector<int32_t> n_dets;
        if (viode) {
            const std::vector<std::string> s0 = list_images(seq_dir + "/segmentation0");
            const std::vector<std::string> s1 = run_dynamic ? list_images(seq_dir + "/segmentation1") : std::vector<std::string>();
            if ((int)s0.size() < n || (run_dynamic && (int)s1.size() < n)) throw std::runtime_error("dvins_node: segmentation0/ (and, in dynamic mode, segmentation1/) must hold one label image per pair");
            const std::vector<uint32_t>& keys = cfg.viode_dynamic_keys; const int nk = (int)keys.size();
            inv_mask.resize(n); right_keys.resize(n); dets.resize(n); det_masks.resize(n); mask_ptr.resize(n); keys_ptr.resize(n); det_ptr.resize(n); n_dets.assign(n, 0);
            std::vector<uint8_t> merge((size_t)W * H); std::vector<uint32_t> kimg((size_t)W * H); std::vector<int32_t> boxes(4 * (size_t)nk);
            for (int k = 0; k < n; ++k) {
                const Gray seg = read_image(s0[k], true);
                if (seg.w != W || seg.h != H) throw std::runtime_error("dvins_node: " + s0[k] + " is not image_width x image_height of the config");
                uint8_t* inv_k = mask_arena ? mask_arena + (size_t)k * px : (inv_mask[k].resize(px), inv_mask[k].data());
                if (dv_viode_mask(ctx, seg.bgr.data(), W, H, 3 * W, keys.data(), nk, merge.data(), inv_k, run_dynamic ? kimg.data() : nullptr, boxes.data()))
                    throw std::runtime_error(std::string("dvins_node: dv_viode_mask: ") + dv_last_error(ctx));
                mask_ptr[k] = inv_k;
                if (!run_dynamic) continue;
                // VIODE::SetViodeMaskAndRoi (viode_utils.cpp:177-218): one Box2D per key present; ascending key (the reference walks an unordered_map)
                for (int q = 0; q < nk; ++q) {
                    const int r0 = boxes[4 * q], r1 = boxes[4 * q + 1], c0 = boxes[4 * q + 2], c1 = boxes[4 * q + 3];
                    if (r1 < r0 || c1 < c0) continue;
                    const int bw = c1 - c0, bh = r1 - r0;                        // cv::Rect(min_pt, max_pt): the max row / column is excluded
                    if (bw < kMinInstSize || bh < kMinInstSize) continue;        // (declared deviation: see dynamic_vins_amd/viode.py detections())
                    std::vector<uint8_t> m((size_t)bw * bh);
                    for (int y = 0; y < bh; ++y) for (int x = 0; x < bw; ++x) m[(size_t)y * bw + x] = kimg[(size_t)(r0 + y) * W + c0 + x] == keys[q] ? 255 : 0;
                    det_masks[k].push_back(std::move(m));
                    dv_inst_det d{}; d.track_id = keys[q]; d.class_id = 0; d.x = c0; d.y = r0; d.w = bw; d.h = bh; d.points = nullptr; d.n_points = 0;
                    dets[k].push_back(d);
                }
                for (size_t q = 0; q < dets[k].size(); ++q) dets[k][q].mask = det_masks[k][q].data();
                det_ptr[k] = dets[k].empty() ? nullptr : dets[k].data(); n_dets[k] = (int)dets[k].size();
                const Gray seg1 = read_image(s1[k], true);
                if (seg1.w != W || seg1.h != H) throw std::runtime_error("dvins_node: " + s1[k] + " is not image_width x image_height of the config");
                uint32_t* keys_k = key_arena ? key_arena + (size_t)k * px : (right_keys[k].resize(px), right_keys[k].data());
                std::vector<uint8_t> tmp_inv(px);
                if (dv_viode_mask(ctx, seg1.bgr.data(), W, H, 3 * W, keys.data(), nk, merge.data(), tmp_inv.data(), keys_k, boxes.data()))
                    throw std::runtime_error(std::string("dvins_node: dv_viode_mask: ") + dv_last_error(ctx));
                keys_ptr[k] = keys_k;
            }
            if (run_dynamic && dv_inst_config(ctx, cfg.max_dynamic_cnt, cfg.min_dynamic_dist, 0)) throw std::runtime_error(std::string("dvins_node: ") + dv_last_error(ctx));
        }

        dv_runner* runner = dv_runner_create(&ctx, &in, 1, 0, 1);
        if (!runner) throw std::runtime_error(std::string("dvins_node: ") + dv_last_error(nullptr));
        dv_seq_dynamic dyn{};
        if (run_dynamic) {
            dyn.inv_mask = mask_ptr.data(); dyn.mask_mem = mem; dyn.mode = DV_MODE_SEMANTIC;
            dyn.dets = det_ptr.data(); dyn.n_dets = n_dets.data(); dyn.boxes3d = nullptr; dyn.n_boxes3d = nullptr; dyn.disp = nullptr; dyn.baseline = cfg.baseline;
            dyn.right_keys = keys_ptr.data(); dyn.right_keys_mem = mem;
            dyn.static_as_background = cfg.static_inst_as_background ? 1 : 0;      // (vio_parameters.h:86: on unless the YAML says static_inst_as_background: 0)
            if (dv_runner_set_dynamic(runner, 0, &dyn)) throw std::runtime_error(std::string("dvins_node: ") + dv_runner_error(runner));
        } else if (run_naive) {
            if (dv_runner_set_mask(runner, 0, mask_ptr.data(), mem, DV_MODE_NAIVE)) throw std::runtime_error(std::string("dvins_node: ") + dv_runner_error(runner));
        }
        double wall = 0;
        if (dv_runner_run(runner, n, &wall)) throw std::runtime_error(std::string("dvins_node: ") + dv_runner_error(runner));
        int rows = 0;
        dv_runner_get_frames(runner, 0, nullptr, 0, &rows);
        std::vector<double> fr(9 * (size_t)std::max(rows, 1));
        dv_runner_get_frames(runner, 0, fr.data(), rows, &rows);
        const std::string mode = std::string(cfg.est.use_imu ? "VIO" : "VO") + "_" + (cfg.naive ? "naive" : (cfg.dynamic ? "dynamic" : "raw")) + "_" + (ReadConfig(cfg_path, device, seq_name, kitti_calib).est.use_line ? "LinePoint" : "PointOnly");
        const std::string out_path = out_dir + "/" + seq_name + "_" + mode + "_Odometry.txt";
        std::ofstream out(out_path);
        if (!out) throw std::runtime_error("dvins_node: cannot write " + out_path);
        for (int i = 0; i < rows; ++i) { std::array<double, 16> s{}; for (int k = 0; k < 7; ++k) s[k] = fr[9 * (size_t)i + 1 + k]; out << TumLine(fr[9 * (size_t)i], s) << "\n"; }
        std::printf("dvins_node: %d pairs tracked, %d frames through the back end in %.3f s (%.1f pairs/s) -> %s\n", n, rows, wall, n / std::max(wall, 1e-9), out_path.c_str());
        if (run_dynamic) {
            long long nd = 0, nf = 0, fo = 0; int mn = 0;
            dv_runner_dynamic_stats(runner, 0, &nd, &nf, &fo, &mn);
            std::printf("dvins_node: dynamic mode from the segmentation images: %lld detections, %lld object feature rows, %lld frames with objects\n", nd, nf, fo);
        }
        dv_runner_destroy(runner);
        dv_destroy(ctx);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
