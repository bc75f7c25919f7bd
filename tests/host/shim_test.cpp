// test driver of dynamic_vins_amd/host/dvins_shim.hpp (built by tests/test_host_shim.py with g++):
//   shim_test parse <config.yaml>                      -> prints the parsed configuration (CPU)
//   shim_test nogpu <config.yaml>                      -> constructing the tracker without a device must throw (CPU)
//   shim_test sync <config.yaml>                       -> StereoSync / FrameGate on scripted time stamps (CPU)
//   shim_test track <config.yaml> <frames.raw> <n> <w> <h>  -> runs TrackImage over n stereo pairs read from a raw file,
//                                                         feeds a constant-gravity IMU stream and runs the Estimator (GPU)
//   shim_test members <config.yaml> <frames.raw> <n> <w> <h> -> the callback / publisher members + dynamic mode through the shim (GPU)
//   shim_test queue <config.yaml>                      -> FeatureQueue push/pop/drop semantics (CPU)
//   shim_test blocking <config.yaml> <frames.raw> <n> <w> <h> -> thread T2 pushes trackImage() output into the global feature_queue, thread T3 runs the blocking
//                                                         Estimator::ProcessMeasurements() until cfg::ok is cleared; compared with the per-iteration path (GPU)
//   shim_test extras <config.yaml> <frames.raw> <n> <w> <h> -> SetUndistortMaps + BGR views, OptimizeInstances (GPU)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>
#include "dvins_shim.hpp"

using namespace dynamic_vins;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string mode = argv[1], cfg = argv[2];
    try {
        if (mode == "parse") {
            YamlFile y(cfg);
            const dv_cam c = ReadPinholeCamera(dir_of(cfg) + "/" + y.str("cam0_calib"));
            std::printf("width %d height %d max_cnt %d min_dist %d flow_back %d stereo %d imu %d iters %d\n", y.integer("image_width", 0), y.integer("image_height", 0),
                        y.integer("max_cnt", 0), y.integer("min_dist", 0), y.integer("flow_back", 0), y.integer("num_of_cam", 0) == 2, y.integer("imu", 0), y.integer("max_num_iterations", 0));
            std::printf("slam_type %s acc_n %.17g g_norm %.17g parallax %.17g\n", y.str("slam_type").c_str(), y.num("acc_n"), y.num("g_norm"), y.num("keyframe_parallax"));
            std::printf("cam %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", c.fx, c.fy, c.cx, c.cy, c.k1, c.k2, c.p1, c.p2);
            for (const char* k : {"body_T_cam0", "body_T_cam1"}) { std::printf("%s", k); for (double v : y.matrix(k)) std::printf(" %g", v); std::printf("\n"); }
            return 0;
        }
        if (mode == "config") {       // shim_test config <config.yaml> [seq] [kitti_calib_dir]: ReadConfig as key=value lines, or "THROWN <what>" (CPU)
            Config c;
            try { c = ReadConfig(cfg, 0, argc > 3 ? argv[3] : "", argc > 4 ? argv[4] : ""); } catch (const std::runtime_error& e) { std::printf("THROWN %s\n", e.what()); return 0; }
            const dv_config& f = c.front; const dv_est_config& e = c.est;
            std::printf("slam_type=%s\ndataset_type=%s\ndynamic=%d\nnaive=%d\ninput_seg=%d\nevery_frame=%d\n", c.slam_type.c_str(), c.dataset_type.c_str(), (int)c.dynamic, (int)c.naive, (int)c.input_seg, (int)c.every_frame);
            std::printf("width=%d\nheight=%d\nmax_cnt=%d\nmin_dist=%d\nflow_back=%d\nstereo=%d\nmask_morphology_size=%d\n", f.width, f.height, f.max_cnt, f.min_dist, f.flow_back, f.stereo, f.mask_morphology_size);
            std::printf("max_dynamic_cnt=%d\nmin_dynamic_dist=%d\nuse_det3d=%d\nundistort_input=%d\nstatic_inst_as_background=%d\nbaseline=%.17g\n", c.max_dynamic_cnt, c.min_dynamic_dist, c.use_det3d, c.undistort_input, (int)c.static_inst_as_background, c.baseline);
            std::printf("use_imu=%d\nplane_constraint=%d\nmax_iters=%d\nkeyframe_parallax=%.17g\ninit_depth=%.17g\ng_norm=%.17g\ntd=%.17g\n", e.use_imu, e.plane_constraint, e.max_iters, e.keyframe_parallax, e.init_depth, e.g_norm, e.td);
            std::printf("acc_n=%.17g\ngyr_n=%.17g\nacc_w=%.17g\ngyr_w=%.17g\nestimate=%d\n", e.acc_n, e.gyr_n, e.acc_w, e.gyr_w, e.estimate);
            std::printf("est_dynamic=%d\nest_use_det3d=%d\ninstance_init_min_num=%d\nstatic_inst_threshold=%.17g\nuse_line=%d\nline_min_obs=%d\n", e.dynamic, e.use_det3d, e.instance_init_min_num, e.static_inst_threshold, e.use_line, e.line_min_obs);
            for (int k = 0; k < 2; ++k) { const dv_cam& m = k ? f.cam1 : f.cam0; std::printf("cam%d=%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", k, m.fx, m.fy, m.cx, m.cy, m.k1, m.k2, m.p1, m.p2); }
            for (int k = 0; k < 2; ++k) { std::printf("ric%d=", k); for (int i = 0; i < 9; ++i) std::printf("%.17g ", e.ric[k][i]); std::printf("\ntic%d=%.17g %.17g %.17g\n", k, e.tic[k][0], e.tic[k][1], e.tic[k][2]); }
            for (const auto& kv : c.topics) std::printf("topic.%s=%s\n", kv.first.c_str(), kv.second.c_str());
            return 0;
        }
        if (mode == "formats") {      // shim_test formats <dummy> <in.txt> <out.txt>: read a feature file, write it back, print a trajectory line
            auto pts = DeserializePointFeature(argv[3]);
            SerializePointFeature(argv[4], pts);
            std::array<double, 16> st{}; st[0] = 1.0; st[1] = -2.5; st[2] = 0.125; st[5] = 0.70710678; st[6] = 0.70710678;
            std::printf("%zu\n%s\n", pts.size(), TumLine(1403636579.763555992, st).c_str());
            return 0;
        }
        if (mode == "sync") {         // scripted time stamps through StereoSync / FrameGate (CPU)
            StereoSync<int> sync;
            double t0, t1; int l, r;
            std::printf("%d", (int)sync.TryPop(t0, l, t1, r));                      // nothing queued
            sync.PushLeft(1.000, 10); sync.PushLeft(1.050, 11); sync.PushLeft(1.100, 12); sync.PushLeft(1.150, 13);
            sync.PushRight(0.900, 19); sync.PushRight(0.950, 20); sync.PushRight(1.052, 21); sync.PushRight(1.100, 22); sync.PushRight(1.149, 23);
            // left 1.000: right 0.900, 0.950 are too old -> discarded; right 1.052 is > 5 ms newer than 1.000?  no: |1.000 - 1.052| — the rule only
            // discards OLDER right images, so 1.000 pairs with 1.052 (the reference's behaviour)
            for (int k = 0; k < 6; ++k) {
                const bool ok = sync.TryPop(t0, l, t1, r);
                if (ok) std::printf(" | %d %d %.3f %.3f", l, r, t0, t1); else std::printf(" | -");
            }
            std::printf(" | dropped %d %d pending %zu %zu\n", sync.dropped_left.load(), sync.dropped_right.load(), sync.pending_left(), sync.pending_right());
            StereoSync<int> s2;                                                      // a left image far older than every right image is dropped
            s2.PushLeft(2.000, 1); s2.PushLeft(2.100, 2); s2.PushRight(2.099, 7);
            const bool a = s2.TryPop(t0, l, t1, r); const bool b = s2.TryPop(t0, l, t1, r);
            std::printf("%d %d %d %d %.3f dropped %d\n", (int)a, (int)b, l, r, t1, s2.dropped_left.load());
            FrameGate g, gk; gk.every_frame = true;
            for (int k = 0; k < 5; ++k) std::printf("%d%d ", (int)g.Pass(), (int)gk.Pass());
            std::printf("\n");
            return 0;
        }
        if (mode == "queue") {        // basic/feature_queue.h:19-73 on scripted input (CPU)
            FeatureQueue q;
            std::printf("empty %d size %d front %d", (int)q.empty(), q.size(), (int)q.front_time().has_value());
            const auto t0 = std::chrono::steady_clock::now();
            const bool none = !q.request().has_value();
            const double waited = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::printf(" request_empty %d waited_30ms %d\n", (int)none, waited >= 29.0 && waited < 500.0);
            for (int k = 0; k < kImageQueueSize + 5; ++k) { FrontendFeature f; f.time = 10.0 + k; f.seq_id = k; q.push_back(f); }
            std::printf("size %d front %.1f", q.size(), *q.front_time());
            auto a = q.request(), b = q.request();
            std::printf(" pop %u %u size %d front %.1f", a->seq_id, b->seq_id, q.size(), *q.front_time());
            { FrontendFeature f; f.time = 999.0; f.seq_id = 999; q.push_back(f); }      // room again: accepted at the back
            int last = -1; while (!q.empty()) last = (int)q.request()->seq_id;
            std::printf(" last %d", last);
            { FrontendFeature f; f.time = 1.0; q.push_back(f); } q.clear();
            std::printf(" cleared %d\n", (int)q.empty());
            std::printf("global %d ok %d\n", (int)feature_queue.empty(), (int)cfg::ok.load());
            return 0;
        }
        if (mode == "blocking" && argc >= 7) {
            const int n = std::atoi(argv[4]), w = std::atoi(argv[5]), h = std::atoi(argv[6]);
            std::ifstream f(argv[3], std::ios::binary);
            std::vector<uint8_t> buf((size_t)n * 2 * w * h);
            f.read((char*)buf.data(), buf.size());
            dv_config c{};
            {
                YamlFile y(cfg);
                c.width = w; c.height = h; c.max_cnt = 30; c.min_dist = 10; c.flow_back = y.integer("flow_back", 1); c.stereo = 1;
                c.cam0 = ReadPinholeCamera(dir_of(cfg) + "/" + y.str("cam0_calib"));
                c.cam0.fx *= w / 1280.0; c.cam0.cx *= w / 1280.0; c.cam0.fy *= h / 720.0; c.cam0.cy *= h / 720.0;
                c.cam1 = c.cam0;
            }
            // reference run: one ProcessMeasurements(frame) per tracked frame, IMU fed in front of every frame
            std::vector<FrontendFeature> frames;
            std::string ref_line; int ref_frame = 0, ref_margin = -1;
            {
                FeatureTracker tracker(c);
                Estimator est(cfg);
                double t_imu = 0.95;
                for (int k = 0; k < n; ++k) {
                    SemanticImage img;
                    img.gray0 = ImageView{buf.data() + (size_t)(2 * k) * w * h, w, h, w, false};
                    img.gray1 = ImageView{buf.data() + (size_t)(2 * k + 1) * w * h, w, h, w, false};
                    img.time0 = 1.0 + 0.05 * k; img.seq = k;
                    FrontendFeature ff; ff.features = tracker.trackImage(img); ff.time = img.time0; ff.seq_id = k;      // north_star's spelling
                    frames.push_back(ff);
                    for (; t_imu <= img.time0 + 0.006; t_imu += 0.005) est.InputIMU(t_imu, Vec3d{0, 0, 9.81007}, Vec3d{0, 0, 0});
                    if (!est.processImage(ff, ff.time)) { std::printf("reference run: IMU missing\n"); return 1; }
                }
                ref_line = TumLine(1.0, est.WindowState(est.frame)); ref_frame = est.frame; ref_margin = (int)est.margin_flag;
            }
            // the reference's threading: T3 blocks in ProcessMeasurements(); T2 (here: this thread) pushes the frames AHEAD of their IMU data, so the loop's
            // "wait for imu" branch runs (the frame must stay queued meanwhile); the IMU callback thread delivers the samples late
            Estimator est(cfg);
            cfg::ok = true;
            std::thread t3([&] { est.ProcessMeasurements(); });
            int max_queued = 0;
            double t_imu = 0.95;
            for (int k = 0; k < n; ++k) {
                feature_queue.push_back(frames[k]);
                std::this_thread::sleep_for(std::chrono::milliseconds(12));      // T3 has seen the frame and is waiting for its IMU interval
                max_queued = std::max(max_queued, feature_queue.size());
                for (; t_imu <= frames[k].time + 0.006; t_imu += 0.005) est.InputIMU(t_imu, Vec3d{0, 0, 9.81007}, Vec3d{0, 0, 0});
            }
            for (int spin = 0; spin < 2000 && est.processed_frames.load() < n; ++spin) std::this_thread::sleep_for(std::chrono::milliseconds(5));
            cfg::ok = false;
            t3.join();
            std::printf("processed %lld of %d queued_while_waiting %d left %d seq %u\n", est.processed_frames.load(), n, max_queued >= 1, feature_queue.size(), est.feature_frame.seq_id);
            std::printf("same_state %d frame %d %d margin_flag %d %d solver %d\n", (int)(TumLine(1.0, est.WindowState(est.frame)) == ref_line), est.frame, ref_frame, (int)est.margin_flag, ref_margin, (int)est.solver_flag);
            return 0;
        }
        if (mode == "nogpu") {
            try { FeatureTracker t(cfg); } catch (const std::runtime_error& e) { std::printf("THROWN %s\n", e.what()); return 0; }
            std::printf("NO EXCEPTION\n");
            return 1;
        }
        if (mode == "track" && argc >= 7) {
            const int n = std::atoi(argv[4]), w = std::atoi(argv[5]), h = std::atoi(argv[6]);
            std::ifstream f(argv[3], std::ios::binary);
            std::vector<uint8_t> buf((size_t)n * 2 * w * h);
            f.read((char*)buf.data(), buf.size());
            dv_config c{};
            {   // the YAML gives 1280x720; the test frames are smaller, so build the config explicitly but through the same readers
                YamlFile y(cfg);
                c.width = w; c.height = h; c.max_cnt = 30; c.min_dist = 10; c.flow_back = y.integer("flow_back", 1); c.stereo = 1;
                c.cam0 = ReadPinholeCamera(dir_of(cfg) + "/" + y.str("cam0_calib"));
                c.cam0.fx *= w / 1280.0; c.cam0.cx *= w / 1280.0; c.cam0.fy *= h / 720.0; c.cam0.cy *= h / 720.0;
                c.cam1 = c.cam0;
            }
            FeatureTracker tracker(c);
            Estimator est(cfg);
            double t_imu = 0.95;
            for (int k = 0; k < n; ++k) {
                SemanticImage img;
                img.gray0 = ImageView{buf.data() + (size_t)(2 * k) * w * h, w, h, w, false};
                img.gray1 = ImageView{buf.data() + (size_t)(2 * k + 1) * w * h, w, h, w, false};
                img.time0 = 1.0 + 0.05 * k; img.seq = k;
                FeatureBackground fb = tracker.TrackImage(img);
                std::printf("frame %d n %zu ids", k, fb.points.size());
                unsigned long long idsum = 0; int stereo = 0;
                for (auto& kv : fb.points) { idsum += kv.first; stereo += kv.second.size() == 2; }
                std::printf(" %llu stereo %d first %.9f %.9f\n", idsum, stereo, fb.points.begin()->second[0].second[3], fb.points.begin()->second[0].second[4]);
                for (; t_imu <= img.time0 + 0.006; t_imu += 0.005) est.InputIMU(t_imu, Vec3d{0, 0, 9.81007}, Vec3d{0, 0, 0});
                FrontendFeature ff; ff.features = fb; ff.time = img.time0; ff.seq_id = k;
                const bool ok = est.ProcessMeasurements(ff);
                std::printf("est ok %d frame %d nonlinear %d\n", (int)ok, est.frame, (int)est.solver_flag);
            }
            std::printf("%s\n", TumLine(1.0, est.WindowState(est.frame)).c_str());
            return 0;
        }
        if (mode == "members" && argc >= 7) {
            // the members the reference's callbacks / publishers use: keep_images + img_track, LatestState (FastPredictIMU), key_poses, Set/GetOutputEgoInfo,
            // Landmarks, ChangeSensorType; and dynamic mode through the shim: InstsFeatManager::InstsTrack / Output + Estimator::ProcessMeasurements(dynamic)
            const int n = std::atoi(argv[4]), w = std::atoi(argv[5]), h = std::atoi(argv[6]);
            std::ifstream f(argv[3], std::ios::binary);
            std::vector<uint8_t> buf((size_t)n * 2 * w * h);
            f.read((char*)buf.data(), buf.size());
            dv_config c{};
            YamlFile y(cfg);
            c.width = w; c.height = h; c.max_cnt = 30; c.min_dist = 10; c.flow_back = 1; c.stereo = 1;
            c.cam0 = ReadPinholeCamera(dir_of(cfg) + "/" + y.str("cam0_calib"));
            c.cam0.fx *= w / 1280.0; c.cam0.cx *= w / 1280.0; c.cam0.fy *= h / 720.0; c.cam0.cy *= h / 720.0; c.cam1 = c.cam0;
            FeatureTracker tracker(c);
            tracker.keep_images = true;
            InstsFeatManager insts(tracker, 20, 4, 0);
            Estimator est(cfg);
            std::vector<uint8_t> mask((size_t)60 * 40, 255), inv((size_t)w * h, 255);
            const int bx = w / 2 - 30, by = h / 2 - 20;
            for (int yy = 0; yy < 40; ++yy) for (int xx = 0; xx < 60; ++xx) inv[(size_t)(by + yy) * w + bx + xx] = 0;
            double t_imu = 0.95; int obj_feats = 0, frames_with_obj = 0;
            for (int k = 0; k < n; ++k) {
                SemanticImage img;
                img.gray0 = ImageView{buf.data() + (size_t)(2 * k) * w * h, w, h, w, false};
                img.gray1 = ImageView{buf.data() + (size_t)(2 * k + 1) * w * h, w, h, w, false};
                img.inv_merge_mask = ImageView{inv.data(), w, h, w, false};
                img.time0 = 1.0 + 0.05 * k; img.seq = k;
                tracker.TrackImageEnqueue(img, DV_MODE_SEMANTIC);
                dv_inst_det d{}; d.track_id = 7; d.class_id = 2; d.x = bx; d.y = by; d.w = 60; d.h = 40; d.mask = mask.data();
                insts.InstsTrack(img.time0, {d});
                FeatureBackground fb = tracker.TrackImageCollect();
                auto objs = insts.Output();
                if (!objs.empty()) { frames_with_obj++; obj_feats += (int)objs.begin()->second.features.size(); }
                for (; t_imu <= img.time0 + 0.006; t_imu += 0.005) est.InputIMU(t_imu, Vec3d{0, 0, 9.81007}, Vec3d{0, 0, 0});
                est.ProcessMeasurements(tracker.rows(), tracker.n_rows(), img.time0);
            }
            const auto& it = tracker.img_track();
            size_t coloured = 0; for (size_t i = 0; i < it.data.size(); i += 3) coloured += it.data[i] != it.data[i + 1];
            std::printf("img_track %dx%dx%d coloured %d prev %d cur %d\n", it.width, it.height, it.channels, coloured > 0, !tracker.prev_img.empty(), !tracker.cur_img.empty());
            std::printf("objects frames %d feats_positive %d\n", frames_with_obj, obj_feats > 0);
            double tl; Vec3d P, V; std::array<double, 4> Q;
            const bool have = est.LatestState(tl, P, Q, V);
            std::printf("latest %d key_poses %zu landmarks_positive %d\n", (int)have, est.key_poses.size(), (int)!est.Landmarks().empty());
            const Estimator::EgoInfo e = est.GetOutputEgoInfo();
            std::printf("ego R00 %.3f P_bc %.3f %.3f %.3f\n", e.R[0], e.P_bc[0], e.P_bc[1], e.P_bc[2]);
            {   // TrackImageLine: one more frame with two detector segments (the second only seen on the left); the principal point maps to (0, 0)
                SemanticImage img;
                img.gray0 = ImageView{buf.data() + (size_t)(2 * (n - 1)) * w * h, w, h, w, false};
                img.gray1 = ImageView{buf.data() + (size_t)(2 * (n - 1) + 1) * w * h, w, h, w, false};
                img.time0 = 1.0 + 0.05 * n; img.seq = n;
                const float cx = (float)c.cam0.cx, cy = (float)c.cam0.cy;
                FeatureBackground fb = tracker.TrackImageLine(img, { LineSegment{ 5, cx, cy, cx + 50, cy }, LineSegment{ 9, cx, cy - 20, cx, cy + 20 } }, { LineSegment{ 5, cx - 4, cy, cx + 46, cy } });
                const Line& l5 = fb.lines.at(5)[0].second;
                std::printf("lines %zu stereo %zu %zu start %.6f %.6f end_x_positive %d points_positive %d\n", fb.lines.size(), fb.lines.at(5).size(), fb.lines.at(9).size(),
                            std::fabs(l5.x1), std::fabs(l5.y1), l5.x2 > 0.01, !fb.points.empty());
            }
            est.ChangeSensorType(0, 1);
            std::printf("changed\n");
            return 0;
        }
        if (mode == "extras" && argc >= 7) {
            // (1) identity undistortion maps + BGR views: TrackImage must yield what the plain gray path yields
            // (2) OptimizeInstances on a fixed two-detection problem (tests/test_host_shim.py runs the same through ctypes)
            const int n = std::atoi(argv[4]), w = std::atoi(argv[5]), h = std::atoi(argv[6]);
            std::ifstream f(argv[3], std::ios::binary);
            std::vector<uint8_t> buf((size_t)n * 2 * w * h);
            f.read((char*)buf.data(), buf.size());
            dv_config c{};
            c.width = w; c.height = h; c.max_cnt = 30; c.min_dist = 10; c.flow_back = 1; c.stereo = 1;
            c.cam0 = dv_cam{0.55 * w, 0.55 * w, 0.5 * w, 0.5 * h, 0, 0, 0, 0}; c.cam1 = c.cam0;
            FeatureTracker plain(c), undist(c);
            std::vector<int16_t> m1((size_t)2 * w * h); std::vector<uint16_t> m2((size_t)w * h, 0);
            for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) { m1[((size_t)y * w + x) * 2] = (int16_t)x; m1[((size_t)y * w + x) * 2 + 1] = (int16_t)y; }
            undist.SetUndistortMaps(0, m1.data(), m2.data());
            undist.SetUndistortMaps(1, m1.data(), m2.data());
            std::vector<uint8_t> bgr0((size_t)3 * w * h), bgr1((size_t)3 * w * h);
            int same = 0;
            for (int k = 0; k < n; ++k) {
                const uint8_t* g0 = buf.data() + (size_t)(2 * k) * w * h; const uint8_t* g1 = buf.data() + (size_t)(2 * k + 1) * w * h;
                for (size_t i = 0; i < (size_t)w * h; ++i) for (int ch = 0; ch < 3; ++ch) { bgr0[3 * i + ch] = g0[i]; bgr1[3 * i + ch] = g1[i]; }
                SemanticImage a, b;
                a.gray0 = ImageView{g0, w, h, w, false}; a.gray1 = ImageView{g1, w, h, w, false}; a.time0 = 0.05 * k;
                b.gray0 = ImageView{bgr0.data(), w, h, 3 * w, false, true}; b.gray1 = ImageView{bgr1.data(), w, h, 3 * w, false, true}; b.time0 = 0.05 * k;
                const FeatureBackground fa = plain.TrackImage(a), fb = undist.TrackImage(b);
                same += fa.points == fb.points && !fa.points.empty();
            }
            std::printf("undistort+bgr identical frames %d of %d\n", same, n);
            std::vector<double> state(77), dims = {4.0, 2.0, 1.5}, body(77, 0.0);
            for (int i = 0; i < 11; ++i) { state[7 * i] = 5; state[7 * i + 1] = 1; state[7 * i + 6] = 1; body[7 * i + 6] = 1; }
            const double ric[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            std::vector<dv_obj_box> boxes(2);
            const double ang[2] = {0.1, -0.05}; const int fr[2] = {3, 7};
            for (int k = 0; k < 2; ++k) {
                boxes[k].obj = 0; boxes[k].frame = fr[k]; boxes[k].dims[0] = 4.2; boxes[k].dims[1] = 1.9; boxes[k].dims[2] = 1.6;
                const double cs = std::cos(ang[k]), sn = std::sin(ang[k]);
                const double R[9] = {cs, -sn, 0, sn, cs, 0, 0, 0, 1};
                std::memcpy(boxes[k].R_cioi, R, sizeof(R));
            }
            const InstanceSolveSummary r = OptimizeInstances(plain.ctx(), 1, state.data(), dims.data(), body.data(), ric, boxes, {}, 10);
            std::printf("instances %d %d %d %.17g %.17g %.17g %.17g %.17g\n", r.iterations, r.successful, r.termination, r.initial_cost, r.final_cost, dims[0], state[7 * 3 + 5], state[7 * 7 + 6]);
            return 0;
        }
    } catch (const std::exception& e) { std::printf("EXCEPTION %s\n", e.what()); return 3; }
    return 2;
}
