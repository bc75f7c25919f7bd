// inst_host.h — host side of dynamic mode in the back end (product code): the object bookkeeping of dynamic_vins'
// InstanceManager / Instance around the device's object solve (be_objsolve.hip).  Mirrors, on flat vectors instead of
// std::list<LandmarkPoint> / shared_ptr<FeaturePoint> / unordered_map:
//   InstanceManager::{PushBack,PropagatePose,Triangulate,BoxFitPoints,InitialInstance,InitialInstanceVelocity,SetDynamicOrStatic,
//                     Optimization (problem assembly + read-back),ManageTriangulatePoint,SlideWindow,SetOutputInstInfo}   estimator/estimator_insts.cpp:54-1249
//   Instance::{SlideWindowOld,SlideWindowNew,OutlierRejection,OutlierRejectionByBox3d,DeleteBadLandmarks,SetOptimizeParameters,
//              GetOptimizationParameters,DeleteOutdatedLandmarks,ClearState,IsInBoxPw,IsInBoxPc}                          estimator/instance.cpp:19-537, instance.h
//   FitBox3DWithRANSAC / FitBox3DFromCameraFrame                                                                          estimator/vio_util.cpp:209-332
// Everything here is O(objects x points) scalar work per frame (a few thousand flops); the numeric solve runs on the GPU.
// Canonical choices: objects are visited in ascending id (the reference iterates an unordered_map, i.e. libstdc++ bucket
// order); FitBox3DWithRANSAC's std::random_device is replaced by a xorshift generator seeded from (object id, frame sequence number,
// call site) with a Fisher-Yates shuffle — the reference's result is not reproducible run to run.
// Quirks kept bug-for-bug are marked (sic).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>
#include "be_math.h"
#include "../../include/dvins.h"

namespace dvi {
using namespace be;

constexpr int kW = 10;                       // kWinSize
constexpr double kDynDepthMin = 0.1, kDynDepthMax = 100.0;      // vio_parameters.h:21-22

struct BodyView {        // the slice of BodyState (estimator/body.h) the object code reads
    const m33* Rs; const d3* Ps; const m33* ric; const d3* tic; const double* headers; double td; int frame;
    d3 cam_to_world(d3 p, int f, int c = 0) const { return mul(Rs[f], mul(ric[c], p) + tic[c]) + Ps[f]; }
    d3 world_to_cam(d3 p, int f, int c = 0) const { return mul(tr(ric[c]), mul(tr(Rs[f]), p - Ps[f]) - tic[c]); }
    void cam34(int k, int cam, double P[3][4]) const {      // GetCamPose34d
        const d3 t0 = Ps[k] + mul(Rs[k], tic[cam]); const m33 Rt = tr(mul(Rs[k], ric[cam])); const d3 t = -mul(Rt, t0);
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P[i][j] = Rt.m[i * 3 + j]; P[i][3] = get(t, i); }
    }
};

struct IObs {            // FeaturePoint (basic/point_feature.h)
    int frame = 0; d3 pt, pt_r; double vel[2] = { 0, 0 }, vel_r[2] = { 0, 0 }, td = 0; bool stereo = false, tri = false; d3 pw;
};
struct ILm {             // LandmarkPoint (basic/point_landmark.h)
    unsigned id = 0; std::vector<IObs> obs; bool bad = false; double depth = -1.0;
    int frame() const { return obs.front().frame; }
    void erase_at(size_t i) { obs.erase(obs.begin() + (long)i); if (obs.empty()) bad = true; }
    void erase_front(size_t n = 1) { obs.erase(obs.begin(), obs.begin() + (long)n); if (obs.empty()) bad = true; }
};
struct IBox { bool valid = false; double dims[3] = { 0, 0, 0 }, center[3] = { 0, 0, 0 }, yaw = 0;
    m33 R_cioi() const { m33 r = zero3(); const double c = cos(yaw), s = sin(yaw); r.m[0] = c; r.m[2] = s; r.m[4] = 1; r.m[6] = -s; r.m[8] = c; return r; }      // Box3D::R_cioi (box3d.h:79-83): [c 0 -s; 0 1 0; s 0 c]^T
};
struct Inst {            // Instance (estimator/instance.h)
    unsigned id = 0; std::vector<ILm> lms; std::vector<d3> pts_extra[kW + 1];
    bool is_initial = false, is_tracking = true, is_curr_visible = false, is_static = false, is_init_velocity = false;
    m33 R[kW + 1]; d3 P[kW + 1]; double time[kW + 1];
    d3 vel_v, vel_a, last_v, last_a, point_v, point_a;
    double dims[3] = { 0, 0, 0 };                  // box3d->dims
    IBox boxes[kW + 1];
    int triangle_num = 0, static_frame = 1, age = 0, lost_number = 0;
    double para_state[kW + 1][7], para_box[3];
    Inst() { for (int i = 0; i <= kW; ++i) { R[i] = eye3(); P[i] = mk3(0, 0, 0); time[i] = 0; } vel_v = vel_a = last_v = last_a = point_v = point_a = mk3(0, 0, 0); }
    int valid_size() const { int c = 0; for (auto& l : lms) if (!l.bad) ++c; return c; }
    int set_triangle_num() { triangle_num = 0; for (auto& l : lms) if (!l.bad && l.depth > 0) ++triangle_num; return triangle_num; }
    int extra_frames() const { int n = 0; for (int i = 0; i <= kW; ++i) if (!pts_extra[i].empty()) ++n; return n; }
    void clear_state() { is_init_velocity = is_initial = is_tracking = is_curr_visible = is_static = false; age = 0; vel_v = vel_a = mk3(0, 0, 0); }
    d3 world_to_object(d3 p, int f) const { return mul(tr(R[f]), p - P[f]); }
    d3 object_to_world(d3 p, int f) const { return mul(R[f], p) + P[f]; }
    bool in_box(d3 po, double factor) const { return fabs(po.x) < factor * dims[0] && fabs(po.y) < factor * dims[1] && fabs(po.z) < factor * dims[2]; }
};

struct Rng {             // stand-in for std::random_device in FitBox3DWithRANSAC (documented canonical choice)
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed ? seed : 0x9E3779B97F4A7C15ull) {}
    uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
};
inline uint64_t ransac_seed(unsigned inst_id, uint64_t seq, int site) { return 0x9E3779B97F4A7C15ull ^ ((uint64_t)inst_id * 0xD1B54A32D192ED03ull) ^ (seq * 0x94D049BB133111EBull) ^ ((uint64_t)site << 56); }

// FitBox3DWithRANSAC (vio_util.cpp:209-264): 20 draws of <= 10 points; the centre whose axis-aligned half-box holds most points wins (> 10 inliers needed).
// The reference shuffles the WHOLE index vector with std::random_device per draw and uses its first 10 entries; the canonical stand-in (same in the oracle) is the
// seeded xorshift driving a PARTIAL Fisher-Yates shuffle from the front: `batch` swaps give the same distribution for those first entries (a uniform sample
// without replacement; the permutation carries over between draws as in the reference) for 10 random numbers per draw instead of one per point (1 250 per draw on a
// 1280x720 object: the full shuffle was 60 us per object and frame on the host).
inline d3 fit_box_ransac(const std::vector<d3>& pts, const double dims[3], uint64_t seed) {
    const int size = (int)pts.size();
    d3 best = mk3(0, 0, 0); int best_in = 10;
    for (auto& p : pts) best = best + p;
    best = best / (double)size;
    const double bx = dims[0] / 2, by = dims[1] / 2, bz = dims[2] / 2;
    std::vector<int> idx(size); for (int i = 0; i < size; ++i) idx[i] = i;
    const int batch = std::min(10, size);
    Rng rng(seed);
    for (int iter = 0; iter < 20; ++iter) {
        for (int i = 0; i < batch && i < size - 1; ++i) { const int j = i + (int)(rng.next() % (uint64_t)(size - i)); std::swap(idx[i], idx[j]); }
        d3 c = mk3(0, 0, 0);
        for (int i = 0; i < batch; ++i) c = c + pts[idx[i]];
        c = c / (double)batch;
        int inl = 0;
        for (int i = 0; i < size; ++i) { const d3 a = pts[i] - c; if (fabs(a.x) <= bx && fabs(a.y) <= by && fabs(a.z) <= bz) ++inl; }
        if (inl > best_in) { best_in = inl; best = c; }
    }
    return best;
}

// FitBox3DFromCameraFrame (vio_util.cpp:274-332); false = no box found
inline bool fit_box_camera(const std::vector<d3>& points, const double dims[3], d3& out) {
    if (points.empty()) return false;
    std::vector<d3> rest(points);
    d3 c = mk3(0, 0, 0); for (auto& p : points) c = c + p; c = c / (double)points.size();
    const double dn = norm(mk3(dims[0] / 2, dims[1] / 2, dims[2] / 2));
    bool found = false;
    for (int iter = 0; iter < 10; ++iter) {
        std::vector<std::pair<double, d3>> wd; wd.reserve(rest.size());
        for (auto& p : rest) wd.push_back({ norm(p - c), p });
        std::stable_sort(wd.begin(), wd.end(), [](const std::pair<double, d3>& a, const std::pair<double, d3>& b) { return a.first < b.first; });
        c = mk3(0, 0, 0);
        const double len = (double)wd.size(); const int used = (int)(len * 0.8);
        for (int i = 0; i < used; ++i) c = c + wd[i].second;
        if (used < 2) break;
        c = c / (double)used;
        if (wd[used].first <= dn) { found = true; break; }
        std::vector<std::pair<double, d3>> cd; cd.reserve(rest.size());
        for (auto& p : rest) cd.push_back({ norm(p), p });
        std::stable_sort(cd.begin(), cd.end(), [](const std::pair<double, d3>& a, const std::pair<double, d3>& b) { return a.first < b.first; });
        rest.clear();
        for (int i = 0; i < len * 0.5; ++i) rest.push_back(cd[i].second);
    }
    out = c; return found;
}

struct InstCfg { int use_det3d = 0, init_min_num = 4, plane_kind = 0, max_iters = 10; double static_threshold = 10.0; };

struct InstMgr {
    std::map<unsigned, Inst> insts;      // ascending id = the canonical visiting order
    InstCfg cfg; int tracking_num = 0, frame = 0; uint64_t seq = 0;

    void clear() { insts.clear(); tracking_num = 0; frame = 0; seq = 0; static_out.clear(); }
    // ---- SetOutputInstInfo (estimator_insts.cpp:967-990), as far as the front end reads it back: the ids of the instances InstExec visits (initialised and tracking) that
    // are_static at this point — right behind PushBack (estimator.cpp:1579-1586), i.e. with the flags the PREVIOUS frame's SetDynamicOrStatic left.  system/main.cpp:217-245
    // takes the pixels of those instances out of the merged mask (para::is_static_inst_as_background, default true: vio_parameters.h:86).
    std::vector<uint32_t> static_out;
    void set_output_inst_info() {
        static_out.clear();
        if (tracking_num < 1) return;
        for (auto& kv : insts) { if (!kv.second.is_initial || !kv.second.is_tracking) continue; if (kv.second.is_static) static_out.push_back(kv.first); }
    }

    // ---- PushBack (estimator_insts.cpp:54-170) ----
    void push_back(int frame_id, const BodyView& B, const dv_inst_obs* in, int n_in, const dv_feat* feats, const double* points) {
        frame = frame_id; tracking_num = 0; ++seq;
        for (auto& kv : insts) { kv.second.lost_number++; kv.second.is_curr_visible = false; }
        if (n_in <= 0) return;
        std::vector<const dv_inst_obs*> order(n_in);
        for (int i = 0; i < n_in; ++i) order[i] = &in[i];
        std::sort(order.begin(), order.end(), [](const dv_inst_obs* a, const dv_inst_obs* b) { return a->id < b->id; });      // std::map<unsigned, FeatureInstance>
        for (const dv_inst_obs* io : order) {
            auto it = insts.find(io->id);
            const bool created = it == insts.end();
            if (created) { it = insts.emplace(io->id, Inst()).first; it->second.id = io->id; }
            Inst& I = it->second;
            if (io->has_box3d) {
                IBox& b = I.boxes[frame]; b.valid = true; b.yaw = io->box3d.yaw;
                for (int k = 0; k < 3; ++k) { b.dims[k] = io->box3d.dims[k]; b.center[k] = io->box3d.center[k]; }
            }
            if (!created) { I.lost_number = 0; if (!I.is_tracking) I.is_tracking = true; }
            I.is_curr_visible = true;
            // FeatureInstance::features is a std::map keyed by feature id
            std::vector<const dv_feat*> fo(io->n_feats);
            for (int k = 0; k < io->n_feats; ++k) fo[k] = &feats[io->first_feat + k];
            std::sort(fo.begin(), fo.end(), [](const dv_feat* a, const dv_feat* b) { return a->id < b->id; });
            for (const dv_feat* f : fo) {
                IObs o; o.frame = frame; o.td = B.td; o.pt = mk3(f->left[0], f->left[1], 1.0); o.vel[0] = f->left[5]; o.vel[1] = f->left[6];
                o.stereo = f->has_right != 0; o.pt_r = mk3(0, 0, 0);
                if (o.stereo) { o.pt_r = mk3(f->right[0], f->right[1], 1.0); o.vel_r[0] = f->right[5]; o.vel_r[1] = f->right[6]; }
                ILm* L = nullptr;
                if (!created) for (auto& l : I.lms) if (l.id == f->id) { L = &l; break; }      // find_if over ALL landmarks, bad ones included
                if (!L) { I.lms.emplace_back(); L = &I.lms.back(); L->id = f->id; }
                L->obs.push_back(o);
            }
            std::vector<d3>& pe = I.pts_extra[B.frame];      // ProcessExtraPoint: camera -> world at body.frame
            pe.resize(io->n_points);
            for (int k = 0; k < io->n_points; ++k) { const double* p = points + 3 * (size_t)(io->first_point + k); pe[k] = B.cam_to_world(mk3(p[0], p[1], p[2]), B.frame); }
        }
        for (auto& kv : insts) if (kv.second.is_curr_visible || kv.second.is_tracking) ++tracking_num;
    }

    // ---- BoxFitPoints (estimator_insts.cpp:463-489); R_arg is what the caller passes as "R_cioi" (a world rotation at both call sites, sic) ----
    d3 box_fit_points(const BodyView& B, const std::vector<d3>& pts, const m33& R_arg, const double dims[3], uint64_t seed) const {
        if (pts.empty()) return mk3(0, 0, 0);
        const m33 R_woi = mul(mul(B.Rs[frame], B.ric[0]), R_arg);
        std::vector<d3> pr(pts.size());
        for (size_t i = 0; i < pts.size(); ++i) pr[i] = mul(R_woi, pts[i]);
        return mul(inv3(R_woi), fit_box_ransac(pr, dims, seed));
    }

    // ---- PropagatePose (estimator_insts.cpp:210-310) ----
    void propagate_pose(const BodyView& B) {
        if (tracking_num < 1) return;
        const int last = frame - 1; const double tij = B.headers[frame] - B.headers[last];
        for (auto& kv : insts) {
            Inst& I = kv.second;
            if (!I.is_tracking) continue;
            I.time[frame] = B.headers[frame];
            if (!I.is_curr_visible || I.is_static) { I.R[frame] = I.R[last]; I.P[frame] = I.P[last]; continue; }
            if (!I.pts_extra[frame].empty()) {
                if (cfg.use_det3d && I.boxes[frame].valid) for (int k = 0; k < 3; ++k) I.dims[k] = I.boxes[frame].dims[k];
                I.P[frame] = box_fit_points(B, I.pts_extra[frame], I.R[frame], I.dims, ransac_seed(I.id, seq, 0));
            } else if (!I.is_init_velocity && I.age > 5) {
                const d3 dp = (I.P[frame - 1] - I.P[frame - 4]) / 3.0;
                I.R[frame] = mul(eye3(), I.R[last]); I.P[frame] = mul(eye3(), I.P[last]) + dp;
            } else if (I.is_init_velocity) {
                const m33 Ro = so3_exp(I.vel_a * tij); const d3 Po = I.vel_v * tij;      // Velocity::RelativePose
                I.R[frame] = mul(Ro, I.R[last]); I.P[frame] = mul(Ro, I.P[last]) + Po;
            } else { I.R[frame] = I.R[last]; I.P[frame] = I.P[last]; }
            I.vel_v = I.point_v; I.vel_a = I.point_a;          // inst.vel = inst.point_vel (never assigned elsewhere: zero, sic)
        }
    }

    // TriangulatePoint (vio_util.cpp:30-45): null vector of the 4x4 DLT matrix (right singular vector of the smallest singular value,
    // here the eigenvector of D^T D by cyclic Jacobi), de-homogenised
    static d3 triangulate_point(const double L[3][4], const double R[3][4], double x0, double y0, double x1, double y1) {
        double D[4][4], A[4][4], V[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
        for (int c = 0; c < 4; ++c) { D[0][c] = x0 * L[2][c] - L[0][c]; D[1][c] = y0 * L[2][c] - L[1][c]; D[2][c] = x1 * R[2][c] - R[0][c]; D[3][c] = y1 * R[2][c] - R[1][c]; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += D[k][i] * D[k][j]; A[i][j] = s; }
        const double tr2 = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2] + A[3][3] * A[3][3];
        for (int sweep = 0; sweep < 60; ++sweep) {
            double off = 0; for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
            if (off <= 1e-34 * tr2 || off < 1e-300) break;      // off-diagonal mass below double precision of the diagonal: further sweeps rotate by angles < 1e-17
            for (int p = 0; p < 3; ++p) for (int q = p + 1; q < 4; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double th = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1)), c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s * b; A[k][q] = s * a + c * b; }
                for (int k = 0; k < 4; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s * b; A[q][k] = s * a + c * b; }
                for (int k = 0; k < 4; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
            }
        }
        int m = 0; for (int i = 1; i < 4; ++i) if (A[i][i] < A[m][m]) m = i;
        return mk3(V[0][m] / V[3][m], V[1][m] / V[3][m], V[2][m] / V[3][m]);
    }

    // ---- Triangulate (estimator_insts.cpp:316-453), stereo branch (the multi-view branch is commented out in the reference) ----
    void triangulate(const BodyView& B) {
        if (tracking_num < 1) return;
        for (auto& kv : insts) {
            Inst& I = kv.second;
            if (!I.is_tracking) continue;
            int done = 0;
            for (auto& lm : I.lms) {
                if (lm.bad) continue;
                for (size_t k = 0; k < lm.obs.size(); ++k) {
                    IObs& o = lm.obs[k];
                    if (o.tri || !o.stereo) continue;
                    double L[3][4], R[3][4];
                    B.cam34(o.frame, 0, L); B.cam34(o.frame, 1, R);
                    const d3 pw = triangulate_point(L, R, o.pt.x, o.pt.y, o.pt_r.x, o.pt_r.y);
                    const double depth = L[2][0] * pw.x + L[2][1] * pw.y + L[2][2] * pw.z + L[2][3];
                    ++done;
                    if (depth > kDynDepthMin && depth < kDynDepthMax && I.in_box(I.world_to_object(pw, o.frame), 4.0)) {
                        o.tri = true; o.pw = pw;
                        if (lm.depth <= 0) { lm.erase_front(k); k = 0; lm.depth = depth; }       // observations before the first triangulated one are dropped
                    } else o.stereo = false;
                }
            }
            if (done == 0) continue;
            I.set_triangle_num();
        }
    }

    // ---- InitialInstance (estimator_insts.cpp:495-576) ----
    void initial_instance(const BodyView& B) {
        for (auto& kv : insts) {
            Inst& I = kv.second;
            if (I.is_initial) I.age++;
            if (I.is_initial || !I.is_tracking || I.pts_extra[frame].empty()) continue;
            if ((int)I.pts_extra[frame].size() <= cfg.init_min_num) continue;
            m33 R0; d3 P0;
            if (cfg.use_det3d) {
                if (!I.boxes[frame].valid) continue;
                for (int k = 0; k < 3; ++k) I.dims[k] = I.boxes[frame].dims[k];
                R0 = mul(mul(B.Rs[frame], B.ric[0]), I.boxes[frame].R_cioi());
                P0 = box_fit_points(B, I.pts_extra[frame], R0, I.dims, ransac_seed(I.id, seq, 1));
            } else {
                I.dims[0] = 2; I.dims[1] = 4; I.dims[2] = 1.5;
                if (!fit_box_camera(I.pts_extra[frame], I.dims, P0)) continue;       // fed the WORLD points (sic: the camera-frame call is commented out)
                R0 = eye3();
            }
            I.vel_v = I.vel_a = mk3(0, 0, 0);
            for (int i = 0; i <= kW; ++i) { I.R[i] = R0; I.P[i] = P0; I.time[i] = B.headers[i]; }
            I.is_initial = true;
            delete_outdated(I, frame);
        }
    }
    static void delete_outdated(Inst& I, int critical) {      // Instance::DeleteOutdatedLandmarks (instance.cpp:512-537)
        for (auto& lm : I.lms) {
            if (lm.bad || lm.frame() == critical) continue;
            if (lm.obs.size() == 1) { lm.bad = true; continue; }
            lm.obs.erase(std::remove_if(lm.obs.begin(), lm.obs.end(), [&](const IObs& o) { return o.frame < critical; }), lm.obs.end());
            if (lm.obs.empty()) lm.bad = true;
            if (lm.depth > 0) lm.depth = -1.0;
        }
    }

    // ---- InitialInstanceVelocity (estimator_insts.cpp:582-604) ----
    void initial_velocity(const BodyView& B) {
        for (auto& kv : insts) {
            Inst& I = kv.second;
            if (!I.is_initial || I.is_init_velocity || I.age < 3) continue;
            const int i = B.frame - 1, j = B.frame;
            const m33 Rit = tr(I.R[i]);                                   // Isometry3d::inverse(): (R^T, -R^T t)
            const m33 Rij = mul(Rit, I.R[j]); const d3 tij = mul(Rit, I.P[j]) + (-mul(Rit, I.P[i]));
            const double dt = B.headers[B.frame] - B.headers[B.frame - 1];
            I.vel_v = tij / dt; I.vel_a = so3_log(Rij) / dt;              // Velocity::SetVel
            I.is_init_velocity = true;
        }
    }

    // ---- SetDynamicOrStatic (estimator_insts.cpp:610-677) ----
    void set_dynamic_or_static(const BodyView& B) {
        if (tracking_num < 1) return;
        for (auto& kv : insts) {
            Inst& I = kv.second;
            if (!I.is_initial || !I.is_tracking || !I.is_curr_visible) continue;
            int n = 0; d3 scene = mk3(0, 0, 0);
            for (auto& lm : I.lms) {
                if (lm.bad || lm.obs.size() <= 1 || lm.obs.back().frame != B.frame) continue;
                const IObs* nxt = nullptr;
                for (size_t k = lm.obs.size(); k-- > 0;) {
                    const IObs& o = lm.obs[k];
                    if (!o.tri) continue;
                    if (!nxt) nxt = &o;
                    else { scene = scene + (nxt->pw - o.pw) / (B.headers[nxt->frame] - B.headers[o.frame]); ++n; break; }
                }
            }
            const d3 vel = (I.P[B.frame] - I.P[B.frame - 1]) / (I.time[B.frame] - I.time[B.frame - 1]);
            if (n < 5) continue;
            scene = scene / (double)n;
            if (norm(vel) > 15 || norm(scene) > cfg.static_threshold) I.static_frame--; else I.static_frame++;
            if (I.static_frame >= 2) { I.is_static = true; I.static_frame = 2; }
            else if (I.static_frame <= 0) { I.is_static = false; I.static_frame = 0; }
        }
    }

    // ---- Optimization: Instance::SetOptimizeParameters + the residual blocks of AddResidualBlockForInstOpt (estimator_insts.cpp:772-807,997-1249) ----
    // fills the flat problem of dv_obj_solve; `slot` maps problem object index -> instance
    std::vector<Inst*> slot; std::vector<double> P_state, P_dims; std::vector<dv_obj_box> P_boxes; std::vector<dv_obj_point> P_points;
    static void set_optimize_parameters(Inst& I) {
        for (int k = 0; k < 3; ++k) I.para_box[k] = I.dims[k];
        for (int i = 0; i <= kW; ++i) { const quat q = qfromR(I.R[i]); double* p = I.para_state[i]; p[0] = I.P[i].x; p[1] = I.P[i].y; p[2] = I.P[i].z; p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w; }
    }
    static void get_optimization_parameters(Inst& I) {      // Instance::GetOptimizationParameters (instance.cpp:458-506)
        I.last_v = I.vel_v; I.last_a = I.vel_a;              // para_speed is written from vel and no live factor touches it
        for (int k = 0; k < 3; ++k) I.dims[k] = I.para_box[k];
        for (int i = 0; i <= kW; ++i) {
            const double* p = I.para_state[i];
            d3 step = mk3(p[0], p[1], p[2]) - I.P[i];
            if (norm(step) > 10) step = step / norm(step) * 10.0;
            I.P[i] = I.P[i] + step;
            I.R[i] = qR(qnormalized(mkq(p[6], p[3], p[4], p[5])));
        }
    }
    bool build_problem(dv_obj_problem& P, const double* body_pose77, const m33& ric0) {      // false: nothing to solve
        slot.clear(); P_boxes.clear(); P_points.clear();
        if (tracking_num < 1) return false;
        for (auto& kv : insts) { Inst& I = kv.second; if (I.is_initial && I.is_tracking) { set_optimize_parameters(I); slot.push_back(&I); } }
        if (slot.empty()) return false;
        P_state.resize(slot.size() * 77); P_dims.resize(slot.size() * 3);
        for (size_t o = 0; o < slot.size(); ++o) {
            Inst& I = *slot[o];
            std::memcpy(&P_state[o * 77], I.para_state, sizeof(I.para_state)); std::memcpy(&P_dims[o * 3], I.para_box, 24);
            if (I.valid_size() < 1) continue;
            for (int i = 0; i <= kW; ++i) if (I.boxes[i].valid) {
                dv_obj_box b{}; b.obj = (int)o; b.frame = i; std::memcpy(b.dims, I.boxes[i].dims, 24); const m33 Rc = I.boxes[i].R_cioi(); std::memcpy(b.R_cioi, Rc.m, 72);
                P_boxes.push_back(b);
            }
            for (auto& lm : I.lms) {
                if (lm.bad || lm.depth < 0.2) continue;
                for (auto& ob : lm.obs) if (ob.tri) { dv_obj_point p{}; p.obj = (int)o; p.frame = ob.frame; p.p_w[0] = ob.pw.x; p.p_w[1] = ob.pw.y; p.p_w[2] = ob.pw.z; P_points.push_back(p); }
            }
        }
        std::memset(&P, 0, sizeof(P));
        P.n_obj = (int)slot.size(); P.n_boxes = (int)P_boxes.size(); P.n_points = (int)P_points.size(); P.max_iters = cfg.max_iters; P.plane_kind = cfg.plane_kind;
        P.state = P_state.data(); P.dims = P_dims.data(); P.body_pose = body_pose77; std::memcpy(P.R_bc, ric0.m, 72);
        P.boxes = P_boxes.data(); P.points = P_points.data();
        return true;
    }
    void read_back(bool solved) {        // InstanceManager::GetOptimizationParameters
        if (tracking_num < 1) return;
        for (size_t o = 0; o < slot.size(); ++o) {
            Inst& I = *slot[o];
            if (solved) { std::memcpy(I.para_state, &P_state[o * 77], sizeof(I.para_state)); std::memcpy(I.para_box, &P_dims[o * 3], 24); }
            get_optimization_parameters(I);
        }
    }
    // the main window solve re-registers the object blocks (AddInstanceParameterBlock) and reads them back unchanged (estimator.cpp:272-276,321-323)
    void touch_in_main_optimization() {
        if (tracking_num < 1) return;
        for (auto& kv : insts) { Inst& I = kv.second; if (I.is_initial && I.is_tracking) { set_optimize_parameters(I); get_optimization_parameters(I); } }
    }

    // ---- Instance::OutlierRejection (instance.cpp:236-314) over InstExec ----
    void outliers_rejection(const BodyView& B) {
        if (tracking_num < 1) return;
        for (auto& kv : insts) {
            Inst& I = kv.second;
            if (!I.is_initial || !I.is_tracking) continue;
            auto cam_to_object = [&](d3 p, int f, int c) { return I.world_to_object(B.cam_to_world(p, f, c), f); };
            auto object_to_cam = [&](d3 p, int f, int c) { return B.world_to_cam(I.object_to_world(p, f), f, c); };
            for (auto& lm : I.lms) {
                if (lm.bad) continue;
                bool drop = false;
                if (std::isfinite(lm.depth)) {
                    if (!I.in_box(cam_to_object(lm.obs.front().pt * lm.depth, lm.frame(), 0), 4.0)) drop = true;
                    else {
                        double err = 0; int cnt = 0;
                        const int fi = lm.obs.front().frame; const d3 start = lm.obs.front().pt;
                        for (size_t k = 1; k < lm.obs.size(); ++k) {
                            const d3 pc = object_to_cam(cam_to_object(start * lm.depth, fi, 0), lm.obs[k].frame, 0);
                            const double rx = pc.x / pc.z - lm.obs[k].pt.x, ry = pc.y / pc.z - lm.obs[k].pt.y;
                            err += sqrt(rx * rx + ry * ry); ++cnt;
                        }
                        for (size_t k = 1; k < lm.obs.size(); ++k) if (lm.obs[k].stereo) {
                            const d3 pc = object_to_cam(cam_to_object(start * lm.depth, fi, 0), lm.obs[k].frame, 1);
                            const double rx = pc.x / pc.z - lm.obs[k].pt.x, ry = pc.y / pc.z - lm.obs[k].pt.y;      // compared with the LEFT observation (sic)
                            err += sqrt(rx * rx + ry * ry); ++cnt;
                        }
                        const double ave = err / cnt * 460.0;
                        if (ave > 30) drop = true;
                    }
                } else drop = true;
                if (drop) { lm.erase_front(); lm.depth = -1; }
            }
        }
    }

    // ---- Instance::OutlierRejectionByBox3d (instance.cpp:321-395) ----
    static int reject_by_box(Inst& I, const BodyView& B) {
        int del = 0;
        const double bn = norm(mk3(I.dims[0], I.dims[1], I.dims[2]));
        auto outside = [&](d3 po) { return (fabs(po.x) >= 3 * I.dims[0] || fabs(po.y) > 3 * I.dims[1] || fabs(po.z) > 3 * I.dims[2]) || norm(po) > 3 * bn; };
        for (auto& lm : I.lms) {
            if (lm.bad) continue;
            for (auto& o : lm.obs) if (o.tri && o.frame != B.frame) {
                bool out = outside(I.world_to_object(o.pw, o.frame));
                if (out && I.boxes[o.frame].valid) {
                    const IBox& bx = I.boxes[o.frame];
                    const d3 pc = B.world_to_cam(o.pw, B.frame);
                    if (norm(pc - mk3(bx.center[0], bx.center[1], bx.center[2])) > 3 * norm(mk3(bx.dims[0], bx.dims[1], bx.dims[2]))) out = false;
                }
                if (out) { o.tri = false; o.stereo = false; ++del; }
            }
            if (lm.depth > 0) {
                const d3 po = I.world_to_object(B.cam_to_world(lm.obs.front().pt * lm.depth, B.frame), B.frame);      // body.frame, not the landmark's frame (sic)
                if (outside(po)) { lm.erase_front(); lm.depth = -1; ++del; }
            }
        }
        return del;
    }

    // ---- ManageTriangulatePoint (estimator_insts.cpp:813-903) ----
    void manage_triangulate_point(const BodyView& B) {
        for (auto& kv : insts) {
            Inst& I = kv.second;
            if (I.lms.empty() || !I.is_initial || I.valid_size() < 10) continue;
            int stat[kW + 1] = { 0 };
            for (auto& lm : I.lms) if (!lm.bad) for (auto& o : lm.obs) stat[o.frame]++;
            if (stat[kW - 1] <= 2) continue;
            reject_by_box(I, B);
        }
        for (auto& kv : insts) {
            Inst& I = kv.second;
            I.set_triangle_num();
            if (I.lms.empty()) continue;
            if (I.triangle_num > 100) for (auto& lm : I.lms) if (!lm.bad && lm.depth <= 0 && lm.frame() != frame) lm.bad = true;
            I.set_triangle_num();          // (the second pruning pass only drops "extra" landmarks, which the front end never produces)
        }
    }

    // ---- SlideWindow (estimator_insts.cpp:910-960) + Instance::SlideWindowOld / SlideWindowNew (instance.cpp:35-188) ----
    static void margin_transform(const Inst& I, const BodyView& B, int f, m33& Rm, d3& tm) {      // camera frame 0 -> camera frame f THROUGH the object (appendix eq. 12 of the reference)
        const m33 R_bc = B.ric[0], R_cb = tr(R_bc); const d3 P_bc = B.tic[0];
        const d3 t5 = -mul(R_cb, P_bc);
        const m33 CB = mul(R_cb, tr(B.Rs[f]));
        const d3 t4 = mul(CB, I.P[f] - B.Ps[f]);
        const m33 RR = mul(mul(CB, I.R[f]), tr(I.R[0]));
        const d3 t3 = mul(RR, B.Ps[0] - I.P[0]);
        const d3 t2 = mul(mul(RR, B.Rs[0]), P_bc);
        Rm = mul(mul(RR, B.Rs[0]), R_bc);
        tm = t2 + t3 + t4 + t5;
    }
    static void slide_old(Inst& I, const BodyView& B) {
        m33 Rm = eye3(); d3 tm = mk3(0, 0, 0);
        for (auto& lm : I.lms) { if (lm.bad) continue; if (lm.frame() == 0 && lm.obs.size() > 1 && lm.obs[1].frame == 1) { margin_transform(I, B, 1, Rm, tm); break; } }
        for (auto& lm : I.lms) {
            if (lm.bad) continue;
            if (lm.frame() != 0) { for (auto& o : lm.obs) o.frame--; continue; }
            if (lm.obs.size() <= 1) { lm.bad = true; continue; }
            const d3 old = lm.obs.front().pt;
            lm.erase_front();
            if (lm.depth > 0) {
                const d3 pj = old * lm.depth;
                if (lm.frame() != 1) margin_transform(I, B, lm.frame(), Rm, tm);      // overwrites the shared transform: later frame-1 landmarks use it too (sic)
                const d3 pi = mul(Rm, pj) + tm;
                lm.depth = pi.z > 0 ? pi.z : -1;
            }
            for (auto& o : lm.obs) o.frame--;
        }
        for (int i = 0; i < kW; ++i) { std::swap(I.R[i], I.R[i + 1]); std::swap(I.P[i], I.P[i + 1]); std::swap(I.time[i], I.time[i + 1]); std::swap(I.boxes[i], I.boxes[i + 1]); I.pts_extra[i] = I.pts_extra[i + 1]; }
        I.R[kW] = I.R[kW - 1]; I.P[kW] = I.P[kW - 1]; I.time[kW] = I.time[kW - 1];
        I.boxes[kW] = IBox(); I.pts_extra[kW].clear();
    }
    static void slide_new(Inst& I, const BodyView& B) {
        for (auto& lm : I.lms) {
            if (lm.bad) continue;
            if (lm.obs.empty()) { lm.bad = true; continue; }
            if (lm.obs.size() == 1 && lm.frame() == B.frame - 1) { lm.bad = true; continue; }
            for (size_t k = 0; k < lm.obs.size(); ++k) if (lm.obs[k].frame == B.frame - 1) { lm.erase_at(k); break; }
            for (auto& o : lm.obs) if (o.frame == B.frame) { o.frame--; break; }
        }
        I.boxes[kW - 1] = I.boxes[kW]; I.boxes[kW] = IBox();
        I.pts_extra[kW - 1] = I.pts_extra[kW]; I.pts_extra[kW].clear();
        I.R[kW - 1] = I.R[kW]; I.P[kW - 1] = I.P[kW]; I.time[kW - 1] = I.time[kW];
    }
    void slide_window(const BodyView& B, bool margin_old) {
        if (frame != kW) return;
        for (auto& kv : insts) {
            Inst& I = kv.second;
            if (!I.is_tracking && I.lms.empty()) continue;
            if (margin_old) slide_old(I, B); else slide_new(I, B);
            const int pc = I.extra_frames();
            I.set_triangle_num();
            if (I.lms.empty()) I.clear_state();
            else if (I.is_tracking && I.triangle_num == 0 && pc == 0) I.is_initial = false;
        }
    }

    // ---- the tail of ProcessImage's dynamic branch (estimator.cpp:1663-1676): OutliersRejection, DeleteBadLandmarks, ClearState of empty objects ----
    void finish_frame(const BodyView& B) {
        outliers_rejection(B);
        for (auto& kv : insts) { auto& v = kv.second.lms; if (!v.empty()) v.erase(std::remove_if(v.begin(), v.end(), [](const ILm& l) { return l.bad; }), v.end()); }
        for (auto& kv : insts) if (kv.second.lms.empty() && kv.second.extra_frames() == 0) kv.second.clear_state();
    }
};

}  // namespace dvi
