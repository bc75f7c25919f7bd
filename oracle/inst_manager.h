// inst_manager.h — CPU ORACLE (test infrastructure, NOT the product): the object (instance) half of dynamic_vins' back end,
// restated statement by statement from
//   InstanceManager::{PushBack,PropagatePose,Triangulate,BoxFitPoints,InitialInstance,InitialInstanceVelocity,SetDynamicOrStatic,
//                     Optimization,ManageTriangulatePoint,SlideWindow,AddInstanceParameterBlock,AddResidualBlockForInstOpt}   estimator/estimator_insts.cpp:54-1249
//   Instance::*                                                                                                          estimator/instance.cpp:19-537, instance.h:36-203
//   LandmarkPoint / FeaturePoint / State / Velocity                                                                      basic/point_landmark.h, point_feature.h, state.h, velocity.h
//   FitBox3DWithRANSAC / FitBox3DFromCameraFrame                                                                          estimator/vio_util.cpp:209-332
// on the oracle's own la.h types (std::list + shared_ptr like the reference).  PARITY UNPINNED (dvo.h): the reference cannot be built here and
// ships no vectors for this path.  Two things the reference leaves to chance are fixed (and documented in DESIGN.md): the unordered_map of
// instances is visited in ascending id, and FitBox3DWithRANSAC's std::random_device + std::shuffle becomes a seeded xorshift + (partial, front) Fisher-Yates
// (seed = f(instance id, frame sequence number, call site)).  Sophus SO3 exp/log are restated from their published algorithm.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <tuple>
#include <vector>
#include "dvo.h"
#include "la.h"

namespace oim {
using namespace ola;

constexpr int kWinSize = 10;
constexpr double kDynamicDepthMin = 0.1, kDynamicDepthMax = 100, kFocalLength = 460.0;

struct Body {      // the global `body` (estimator/body.h:25-92) as a view of the oracle estimator's members
    M3* Rs; V3* Ps; M3* ric; V3* tic; double* headers; double td; int frame; double (*para_pose)[7];
    V3 CamToWorld(const V3& pt, int f, int c = 0) const { return Rs[f] * (ric[c] * pt + tic[c]) + Ps[f]; }
    V3 WorldToCam(const V3& pt, int f, int c = 0) const { return ric[c].t() * (Rs[f].t() * (pt - Ps[f]) - tic[c]); }
    void GetCamPose34d(int index, int cam, double P[3][4]) const {
        const V3 t0 = Ps[index] + Rs[index] * tic[cam]; const M3 R0 = Rs[index] * ric[cam]; const M3 Rt = R0.t(); const V3 t = -(Rt * t0);
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P[i][j] = Rt(i, j); P[i][3] = t[i]; }
    }
};

inline M3 inverse3(const M3& a) {       // Eigen::Matrix3d::inverse()
    const double c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1), c01 = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2), c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
    const double det = a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02, id = 1.0 / det;
    M3 r;
    r(0, 0) = c00 * id; r(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * id; r(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * id;
    r(1, 0) = c01 * id; r(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * id; r(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * id;
    r(2, 0) = c02 * id; r(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * id; r(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * id;
    return r;
}
inline M3 SO3exp(const V3& w) {        // Sophus::SO3d::exp(w).matrix()
    const double th2 = w.dot(w);
    double imag, real;
    if (th2 < 1e-10 * 1e-10) { const double th4 = th2 * th2; imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4; real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4; }
    else { const double th = std::sqrt(th2), half = 0.5 * th; imag = std::sin(half) / th; real = std::cos(half); }
    return Q(real, imag * w.x, imag * w.y, imag * w.z).R();
}
inline V3 SO3log(const M3& R) {        // Sophus::SO3d(R).log()
    Q q = Q::fromR(R).normalized();
    const double sq = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w;
    double two_atan;
    if (sq < 1e-10 * 1e-10) two_atan = 2.0 / w - 2.0 / 3.0 * sq / (w * w * w);
    else { const double n = std::sqrt(sq); two_atan = std::fabs(w) < 1e-10 ? (w > 0 ? M_PI : -M_PI) / n : 2.0 * std::atan(n / w) / n; }
    return V3(q.x, q.y, q.z) * two_atan;
}

struct FeaturePoint {      // basic/point_feature.h:21-100
    using Ptr = std::shared_ptr<FeaturePoint>;
    V3 point, point_right; bool is_stereo = false, is_extra = false; int frame = 0; double vel[2] = { 0, 0 }, vel_right[2] = { 0, 0 }, td = 0;
    V3 p_w; bool is_triangulated = false;
};
struct LandmarkPoint {     // basic/point_landmark.h:22-88
    explicit LandmarkPoint(unsigned id_) : id(id_) {}
    FeaturePoint::Ptr front() { return feats.front(); }
    int frame() const { return feats.front()->frame; }
    int size() const { return (int)feats.size(); }
    bool is_extra() { return feats.front()->is_extra; }
    void EraseBegin() { erase(feats.begin()); }
    void erase(std::list<FeaturePoint::Ptr>::iterator it) { feats.erase(it); if (feats.empty()) bad = true; }
    void erase(std::list<FeaturePoint::Ptr>::iterator l, std::list<FeaturePoint::Ptr>::iterator r) { feats.erase(l, r); if (feats.empty()) bad = true; }
    FeaturePoint::Ptr& operator[](int index) { auto it = feats.begin(); std::advance(it, index); return *it; }
    bool bad = false; unsigned id; std::list<FeaturePoint::Ptr> feats; double depth = -1.0;
};
struct State { M3 R = M3::identity(); V3 P; double time = 0; void swap(State& o) { std::swap(*this, o); } };      // basic/state.h
struct Velocity {          // basic/velocity.h
    V3 v, a;
    void SetZero() { v = V3(); a = V3(); }
    std::tuple<M3, V3> RelativePose(double t) const { return { SO3exp(a * t), v * t }; }
};
struct Box3D { double dims[3] = { 0, 0, 0 }, center_pt[3] = { 0, 0, 0 }, yaw = 0;      // basic/box3d.h (fields read by the path)
    M3 R_cioi() const { M3 R; R(0, 0) = std::cos(yaw); R(0, 2) = -std::sin(yaw); R(1, 1) = 1; R(2, 0) = std::sin(yaw); R(2, 2) = std::cos(yaw); return R.t(); }
    V3 dimsv() const { return V3(dims[0], dims[1], dims[2]); } V3 center() const { return V3(center_pt[0], center_pt[1], center_pt[2]); }
};
using Box3DPtr = std::shared_ptr<Box3D>;

struct Rng { uint64_t s; explicit Rng(uint64_t seed) : s(seed ? seed : 0x9E3779B97F4A7C15ull) {} uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; } };
inline uint64_t ransac_seed(unsigned inst_id, uint64_t seq, int site) { return 0x9E3779B97F4A7C15ull ^ ((uint64_t)inst_id * 0xD1B54A32D192ED03ull) ^ (seq * 0x94D049BB133111EBull) ^ ((uint64_t)site << 56); }

// vio_util.cpp:209-264
inline bool FitBox3DWithRANSAC(const std::vector<V3>& points, const V3& dims, uint64_t seed, V3& out) {
    if (points.empty()) return false;
    const int size = (int)points.size();
    V3 best_center; int best_inlines = 10;
    for (int i = 0; i < size; ++i) best_center += points[i];
    best_center = best_center / (double)size;
    const V3 box = dims / 2;
    Rng rd(seed);
    std::vector<int> random_indices(size);
    for (int i = 0; i < size; ++i) random_indices[i] = i;
    const int batch_size = std::min(10, size);
    for (int iter = 0; iter < 20; ++iter) {
        // std::shuffle(rd) stand-in: only the first batch_size entries are read, so a partial Fisher-Yates from the front (a uniform sample without replacement,
        // the permutation carrying over between draws like the reference's) replaces the full shuffle; the product draws the same numbers (inst_host.h)
        for (int i = 0; i < batch_size && i < size - 1; ++i) { const int j = i + (int)(rd.next() % (uint64_t)(size - i)); std::swap(random_indices[i], random_indices[j]); }
        V3 center;
        for (int i = 0; i < batch_size; ++i) center += points[random_indices[i]];
        center = center / (double)batch_size;
        int inliers = 0;
        for (int i = 0; i < size; ++i) {
            const V3 d = points[i] - center;
            if (std::fabs(d.x) <= box.x && std::fabs(d.y) <= box.y && std::fabs(d.z) <= box.z) inliers++;
        }
        if (inliers > best_inlines) { best_inlines = inliers; best_center = center; }
    }
    out = best_center; return true;
}
// vio_util.cpp:274-332
inline bool FitBox3DFromCameraFrame(std::vector<V3>& points, const V3& dims, V3& out) {
    if (points.empty()) return false;
    std::list<V3> points_rest; V3 center_pt;
    for (auto& p : points) { center_pt += p; points_rest.push_back(p); }
    center_pt = center_pt / (double)points.size();
    bool is_find = false;
    const double dims_norm = (dims / 2.).norm();
    auto by_dist = [](const std::tuple<double, V3>& a, const std::tuple<double, V3>& b) { return std::get<0>(a) < std::get<0>(b); };
    for (int iter = 0; iter < 10; ++iter) {
        std::vector<std::tuple<double, V3>> points_with_dist;
        for (auto& p : points_rest) points_with_dist.emplace_back((p - center_pt).norm(), p);
        std::stable_sort(points_with_dist.begin(), points_with_dist.end(), by_dist);
        center_pt = V3();
        const double len = (double)points_with_dist.size(); const int len_used = (int)(len * 0.8);
        for (int i = 0; i < len_used; ++i) center_pt += std::get<1>(points_with_dist[i]);
        if (len_used < 2) break;
        center_pt = center_pt / (double)len_used;
        if (std::get<0>(points_with_dist[len_used]) <= dims_norm) { is_find = true; break; }
        std::vector<std::tuple<double, V3>> points_cam_dist;
        for (auto& p : points_rest) points_cam_dist.emplace_back(p.norm(), p);
        std::stable_sort(points_cam_dist.begin(), points_cam_dist.end(), by_dist);
        points_rest.clear();
        for (int i = 0; i < len * 0.5; ++i) points_rest.push_back(std::get<1>(points_cam_dist[i]));
    }
    out = center_pt; return is_find;
}

struct Instance {          // estimator/instance.h:36-203
    Instance() = default;
    explicit Instance(unsigned id_) : id(id_) {}
    V3 WorldToObject(const V3& pt, int f) const { return state[f].R.t() * (pt - state[f].P); }
    V3 ObjectToWorld(const V3& pt, int f) const { return state[f].R * pt + state[f].P; }
    V3 CamToObject(const Body& body, const V3& pt, int f, int c = 0) const { return WorldToObject(body.CamToWorld(pt, f, c), f); }
    V3 ObjectToCam(const Body& body, const V3& pt, int f, int c = 0) const { return body.WorldToCam(ObjectToWorld(pt, f), f, c); }
    int valid_size() { int cnt = 0; for (auto& lm : landmarks) if (!lm.bad) cnt++; return cnt; }
    void ClearState() { is_init_velocity = false; is_initial = false; is_tracking = false; is_curr_visible = false; is_static = false; age = 0; vel.SetZero(); }
    int GetPointsExtraFrames() { int num = 0; for (int i = 0; i <= kWinSize; ++i) if (!points_extra[i].empty()) num++; return num; }
    bool InBox(const V3& pts_oi, double factor) const {
        return (std::fabs(pts_oi.x) < factor * box3d->dims[0]) && (std::fabs(pts_oi.y) < factor * box3d->dims[1]) && (std::fabs(pts_oi.z) < factor * box3d->dims[2]);
    }
    bool IsInBoxPw(const V3& pw, int f, double factor = 4.) { return InBox(WorldToObject(pw, f), factor); }
    bool IsInBoxPc(const Body& body, const V3& pc, int f, double factor = 4.) { return InBox(CamToObject(body, pc, f), factor); }
    int set_triangle_num() { triangle_num = 0; for (auto& lm : landmarks) { if (lm.bad) continue; else if (lm.depth > 0) triangle_num++; } return triangle_num; }

    // instance.cpp:35-138
    int SlideWindowOld(const Body& body) {
        M3 R_margin; V3 t_margin;
        auto margin = [&](int f) {
            const M3& R_bc = body.ric[0]; const M3 R_cb = R_bc.t(); const V3& P_bc = body.tic[0];
            const V3 temp_5 = -(R_cb * P_bc);
            const M3 temp_RcbRbiw = R_cb * body.Rs[f].t();
            const V3 temp_4 = temp_RcbRbiw * (state[f].P - body.Ps[f]);
            const M3 temp_RcbRbiwRwoiRojw = temp_RcbRbiw * state[f].R * state[0].R.t();
            const V3 temp_3 = temp_RcbRbiwRwoiRojw * (body.Ps[0] - state[0].P);
            const V3 temp_2 = temp_RcbRbiwRwoiRojw * body.Rs[0] * P_bc;
            const M3 temp_1 = temp_RcbRbiwRwoiRojw * body.Rs[0] * R_bc;
            R_margin = temp_1; t_margin = temp_2 + temp_3 + temp_4 + temp_5;
        };
        for (auto& lm : landmarks) { if (lm.bad) continue; if (lm.frame() == 0 && lm.size() > 1 && lm[1]->frame == 1) { margin(1); break; } }
        int debug_num = 0;
        for (auto& lm : landmarks) {
            if (lm.bad) continue;
            else if (lm.frame() != 0) { for (auto& feat : lm.feats) feat->frame--; continue; }
            else if (lm.size() <= 1) { lm.bad = true; debug_num++; continue; }
            else {
                const V3 point_old = lm.front()->point;
                lm.EraseBegin();
                if (lm.depth > 0) {
                    const V3 pts_cam_j = point_old * lm.depth; V3 pts_cam_i;
                    if (lm.frame() == 1) pts_cam_i = R_margin * pts_cam_j + t_margin;
                    else { margin(lm.frame()); pts_cam_i = R_margin * pts_cam_j + t_margin; }        // R_margin / t_margin are overwritten for everyone after (sic)
                    lm.depth = pts_cam_i.z > 0 ? pts_cam_i.z : -1;
                }
                for (auto& feat : lm.feats) feat->frame--;
            }
        }
        for (int i = 0; i < kWinSize; i++) { state[i].swap(state[i + 1]); boxes3d[i].swap(boxes3d[i + 1]); points_extra[i] = points_extra[i + 1]; }
        state[kWinSize] = state[kWinSize - 1]; boxes3d[kWinSize].reset(); points_extra[kWinSize].clear();
        return debug_num;
    }
    // instance.cpp:144-188
    int SlideWindowNew(const Body& body) {
        int debug_num = 0;
        for (auto& lm : landmarks) {
            if (lm.bad) continue;
            if (lm.feats.empty()) { lm.bad = true; debug_num++; continue; }
            if (lm.size() == 1 && lm.frame() == body.frame - 1) { lm.bad = true; debug_num++; continue; }
            for (auto it = lm.feats.begin(); it != lm.feats.end(); ++it) if ((*it)->frame == body.frame - 1) { lm.erase(it); break; }
            for (auto& feat : lm.feats) if (feat->frame == body.frame) { feat->frame--; break; }
        }
        boxes3d[kWinSize - 1] = boxes3d[kWinSize]; boxes3d[kWinSize].reset();
        points_extra[kWinSize - 1] = points_extra[kWinSize]; points_extra[kWinSize].clear();
        state[kWinSize - 1] = state[kWinSize];
        return debug_num;
    }
    // instance.cpp:236-314
    void OutlierRejection(const Body& body) {
        if (!is_initial || !is_tracking) return;
        for (auto& lm : landmarks) {
            if (lm.bad) continue;
            if (std::isfinite(lm.depth)) {
                if (!IsInBoxPc(body, lm.front()->point * lm.depth, lm.frame())) { lm.EraseBegin(); lm.depth = -1; if (lm.feats.empty()) lm.bad = true; continue; }
                double err = 0; int err_cnt = 0;
                auto feat_it = lm.feats.begin();
                const int imu_i = (*feat_it)->frame; const V3 start_observe = (*feat_it)->point;
                for (++feat_it; feat_it != lm.feats.end(); ++feat_it) {
                    const int imu_j = (*feat_it)->frame;
                    const V3 pts_cj = ObjectToCam(body, CamToObject(body, start_observe * lm.depth, imu_i), imu_j);
                    const double rx = pts_cj.x / pts_cj.z - (*feat_it)->point.x, ry = pts_cj.y / pts_cj.z - (*feat_it)->point.y;
                    err += std::sqrt(rx * rx + ry * ry); err_cnt++;
                }
                feat_it = lm.feats.begin();
                for (++feat_it; feat_it != lm.feats.end(); ++feat_it) if ((*feat_it)->is_stereo) {
                    const int imu_j = (*feat_it)->frame;
                    const V3 pts_cj = ObjectToCam(body, CamToObject(body, start_observe * lm.depth, imu_i, 0), imu_j, 1);
                    const double rx = pts_cj.x / pts_cj.z - (*feat_it)->point.x, ry = pts_cj.y / pts_cj.z - (*feat_it)->point.y;      // ->point, not ->point_right (sic)
                    err += std::sqrt(rx * rx + ry * ry); err_cnt++;
                }
                const double ave_err = err / err_cnt * kFocalLength;
                if (ave_err > 30) { lm.EraseBegin(); lm.depth = -1; if (lm.feats.empty()) lm.bad = true; }
            } else { lm.EraseBegin(); lm.depth = -1; if (lm.feats.empty()) lm.bad = true; }
        }
    }
    // instance.cpp:321-395
    int OutlierRejectionByBox3d(const Body& body) {
        int del_num = 0;
        const V3 dims = box3d->dimsv(); const double box_norm = dims.norm();
        auto outbox = [&](const V3& po) { return (std::fabs(po.x) >= 3 * dims.x || std::fabs(po.y) > 3 * dims.y || std::fabs(po.z) > 3 * dims.z) || (po.norm() > 3 * box_norm); };
        for (auto& lm : landmarks) {
            if (lm.bad) continue;
            if (lm.is_extra()) continue;          // never true: nothing sets FeaturePoint::is_extra
            for (auto& feat : lm.feats) {
                if (feat->is_triangulated && feat->frame != body.frame) {
                    bool is_outbox = outbox(WorldToObject(feat->p_w, feat->frame));
                    if (is_outbox && boxes3d[feat->frame]) {
                        const V3 pts_cam = body.WorldToCam(feat->p_w, body.frame);
                        if ((pts_cam - boxes3d[feat->frame]->center()).norm() > 3 * boxes3d[feat->frame]->dimsv().norm()) is_outbox = false;
                    }
                    if (is_outbox) { feat->is_triangulated = false; feat->is_stereo = false; del_num++; }
                }
            }
            if (lm.depth > 0) {
                auto feat = lm.front();
                if (outbox(CamToObject(body, feat->point * lm.depth, body.frame))) { lm.EraseBegin(); lm.depth = -1; del_num++; }
            }
        }
        return del_num;
    }
    int DeleteBadLandmarks() { int cnt = 0; for (auto it = landmarks.begin(); it != landmarks.end();) { if (it->bad) { it = landmarks.erase(it); cnt++; } else ++it; } return cnt; }
    // instance.cpp:421-453 / 458-506
    void SetOptimizeParameters() {
        para_speed[0] = vel.v.x; para_speed[1] = vel.v.y; para_speed[2] = vel.v.z; para_speed[3] = vel.a.x; para_speed[4] = vel.a.y; para_speed[5] = vel.a.z;
        for (int k = 0; k < 3; ++k) para_box[k] = box3d->dims[k];
        for (int i = 0; i <= kWinSize; ++i) {
            para_state[i][0] = state[i].P.x; para_state[i][1] = state[i].P.y; para_state[i][2] = state[i].P.z;
            const Q q = Q::fromR(state[i].R);
            para_state[i][3] = q.x; para_state[i][4] = q.y; para_state[i][5] = q.z; para_state[i][6] = q.w;
        }
    }
    void GetOptimizationParameters() {
        last_vel = vel;
        vel.v = V3(para_speed[0], para_speed[1], para_speed[2]); vel.a = V3(para_speed[3], para_speed[4], para_speed[5]);
        for (int k = 0; k < 3; ++k) box3d->dims[k] = para_box[k];
        for (int i = 0; i <= kWinSize; ++i) {
            V3 step = V3(para_state[i][0], para_state[i][1], para_state[i][2]) - state[i].P;
            if (step.norm() > 10) step = step.normalized() * 10.;
            state[i].P = state[i].P + step;
            state[i].R = Q(para_state[i][6], para_state[i][3], para_state[i][4], para_state[i][5]).normalized().R();
        }
    }
    // instance.cpp:512-537
    void DeleteOutdatedLandmarks(int critical_frame) {
        for (auto& lm : landmarks) {
            if (lm.bad || lm.frame() == critical_frame) continue;
            if (lm.size() == 1) { lm.bad = true; continue; }
            for (auto it = lm.feats.begin(), it_next = it; it != lm.feats.end(); it = it_next) { it_next++; if ((*it)->frame < critical_frame) lm.erase(it); }
            if (lm.feats.empty()) lm.bad = true;
            if (lm.depth > 0) lm.depth = -1.0;
        }
    }

    std::list<LandmarkPoint> landmarks; std::vector<V3> points_extra[kWinSize + 1];
    unsigned id = 0; Box3DPtr box3d;
    bool is_initial = false, is_tracking = true, is_curr_visible = false, is_static = false, is_init_velocity = false;
    State state[kWinSize + 1]; Velocity vel, last_vel, point_vel;
    double para_state[kWinSize + 1][7] = { { 0 } }, para_speed[6] = { 0 }, para_box[3] = { 0 };
    int triangle_num = 0, static_frame = 1, age = 0, lost_number = 0;
    Box3DPtr boxes3d[kWinSize + 1];
};

struct FeatureInstance { std::map<unsigned, FeaturePoint::Ptr> features; Box3DPtr box3d; std::vector<V3> points; };      // basic/frontend_feature.h:58-66

struct Params { int use_det3d = 0, kInstanceInitMinNum = 4, plane_kind = 0, KNumIter = 10; double kStaticInstThreshold = 10.0; };

struct InstanceManager {   // estimator/estimator_insts.h
    std::map<unsigned, Instance> instances;      // canonical: ascending id (reference: unordered_map)
    Params para; int tracking_num = 0, frame = 0; uint64_t seq = 0; dvo_ba_summary last_summary{};

    // SetOutputInstInfo (:967-990) as far as the front end reads it back (system/main.cpp:194,217-245): the ids of the instances InstExec visits that are is_static
    std::vector<uint32_t> insts_output_static;
    void SetOutputInstInfo() {
        insts_output_static.clear();
        if (tracking_num < 1) return;
        InstExec([this](unsigned, Instance& inst) { if (inst.is_static) insts_output_static.push_back(inst.id); });
    }
    template <class F> void InstExec(F function, bool exec_all = false) {
        if (tracking_num < 1) return;
        for (auto& kv : instances) { if (!exec_all && (!kv.second.is_initial || !kv.second.is_tracking)) continue; function(kv.first, kv.second); }
    }
    // :54-170
    void PushBack(const Body& body, unsigned frame_id, std::map<unsigned, FeatureInstance>& input_insts) {
        frame = (int)frame_id; tracking_num = 0; ++seq;
        for (auto& p : instances) { p.second.lost_number++; p.second.is_curr_visible = false; }
        if (input_insts.empty()) return;
        auto extra = [&](const std::vector<V3>& pts) { std::vector<V3> w(pts.size()); for (size_t i = 0; i < pts.size(); ++i) w[i] = body.CamToWorld(pts[i], body.frame); return w; };      // ProcessExtraPoint (:33-45)
        for (auto& [instance_id, inst_feat] : input_insts) {
            auto inst_iter = instances.find(instance_id);
            if (inst_iter == instances.end()) {
                auto it = instances.insert({ instance_id, Instance(instance_id) }).first;
                it->second.is_curr_visible = true;
                it->second.box3d = std::make_shared<Box3D>();
                if (inst_feat.box3d) it->second.boxes3d[frame] = inst_feat.box3d;
                for (auto& [feat_id, feat_ptr] : inst_feat.features) {
                    LandmarkPoint lm(feat_id);
                    feat_ptr->frame = frame; feat_ptr->td = body.td;
                    lm.feats.push_back(feat_ptr);
                    it->second.landmarks.push_back(lm);
                }
                it->second.points_extra[body.frame] = extra(inst_feat.points);
            } else {
                auto& landmarks = inst_iter->second.landmarks;
                if (inst_feat.box3d) inst_iter->second.boxes3d[frame] = inst_feat.box3d;
                inst_iter->second.lost_number = 0; inst_iter->second.is_curr_visible = true;
                if (!inst_iter->second.is_tracking) inst_iter->second.is_tracking = true;
                for (auto& [feat_id, feat_ptr] : inst_feat.features) {
                    feat_ptr->frame = frame; feat_ptr->td = body.td;
                    auto it = std::find_if(landmarks.begin(), landmarks.end(), [id = feat_id](const LandmarkPoint& l) { return l.id == id; });
                    if (it == landmarks.end()) { landmarks.emplace_back(feat_id); it = std::prev(landmarks.end()); }
                    it->feats.push_back(feat_ptr);
                }
                inst_iter->second.points_extra[body.frame] = extra(inst_feat.points);
            }
        }
        for (auto& p : instances) if (p.second.is_curr_visible || p.second.is_tracking) tracking_num++;
    }
    // :463-489
    V3 BoxFitPoints(const Body& body, const std::vector<V3>& points3d, const M3& R_cioi, const V3& dims, uint64_t seed) const {
        if (points3d.empty()) return V3();
        const M3 R_woi = body.Rs[frame] * body.ric[0] * R_cioi;
        std::vector<V3> points_r(points3d.size());
        for (size_t i = 0; i < points3d.size(); ++i) points_r[i] = R_woi * points3d[i];
        V3 P, init_cam_pt;
        if (FitBox3DWithRANSAC(points_r, dims, seed, init_cam_pt)) P = inverse3(R_woi) * init_cam_pt;
        else { for (auto& p : points3d) P += p; P = P / (double)points3d.size(); }
        return P;
    }
    // :210-310
    void PropagatePose(const Body& body) {
        if (tracking_num < 1) return;
        const int last_frame = frame - 1;
        const double time_ij = body.headers[frame] - body.headers[last_frame];
        InstExec([&](unsigned, Instance& inst) {
            if (!inst.is_tracking) return;
            inst.state[frame].time = body.headers[frame];
            if (!inst.is_curr_visible) { inst.state[frame].R = inst.state[last_frame].R; inst.state[frame].P = inst.state[last_frame].P; return; }
            if (inst.is_static) { inst.state[frame].R = inst.state[last_frame].R; inst.state[frame].P = inst.state[last_frame].P; return; }
            else if (!inst.points_extra[frame].empty()) {
                State init_state = inst.state[frame];
                if (para.use_det3d && inst.boxes3d[frame]) for (int k = 0; k < 3; ++k) inst.box3d->dims[k] = inst.boxes3d[frame]->dims[k];
                init_state.R = inst.state[frame].R;
                init_state.P = BoxFitPoints(body, inst.points_extra[frame], init_state.R, inst.box3d->dimsv(), ransac_seed(inst.id, seq, 0));
                inst.state[frame] = init_state;
            } else if (!inst.is_init_velocity && inst.age > 5) {
                const M3 Roioj = M3::identity();
                V3 Poioj = inst.state[frame - 1].P - inst.state[frame - 4].P;
                Poioj = Poioj / 3.;
                inst.state[frame].R = Roioj * inst.state[last_frame].R; inst.state[frame].P = Roioj * inst.state[last_frame].P + Poioj;
            } else if (inst.is_init_velocity) {
                auto [Roioj, Poioj] = inst.vel.RelativePose(time_ij);
                inst.state[frame].R = Roioj * inst.state[last_frame].R; inst.state[frame].P = Roioj * inst.state[last_frame].P + Poioj;
            } else { inst.state[frame].R = inst.state[last_frame].R; inst.state[frame].P = inst.state[last_frame].P; }
            inst.vel = inst.point_vel;
        }, true);
    }
    static V3 TriangulatePoint(const double P0[3][4], const double P1[3][4], double x0, double y0, double x1, double y1) {      // vio_util.cpp:30-45
        Mat D(4, 4);
        for (int c = 0; c < 4; ++c) { D(0, c) = x0 * P0[2][c] - P0[0][c]; D(1, c) = y0 * P0[2][c] - P0[1][c]; D(2, c) = x1 * P1[2][c] - P1[0][c]; D(3, c) = y1 * P1[2][c] - P1[1][c]; }
        double v[4]; smallest_right_singular4(D, v);
        return { v[0] / v[3], v[1] / v[3], v[2] / v[3] };
    }
    // :316-453
    void Triangulate(const Body& body) {
        if (tracking_num < 1) return;
        for (auto& [key, inst] : instances) {
            if (!inst.is_tracking) continue;
            int stereo_triangle_succeed = 0, stereo_triangle_failed = 0;
            for (auto& lm : inst.landmarks) {
                if (lm.bad) continue;
                for (auto it = lm.feats.begin(), it_next = it; it != lm.feats.end(); it = it_next) {
                    it_next++;
                    auto& feat = *it;
                    if (feat->is_triangulated) continue;
                    if (feat->is_stereo) {
                        double leftPose[3][4], rightPose[3][4];
                        body.GetCamPose34d(feat->frame, 0, leftPose); body.GetCamPose34d(feat->frame, 1, rightPose);
                        const V3 point3d_w = TriangulatePoint(leftPose, rightPose, feat->point.x, feat->point.y, feat->point_right.x, feat->point_right.y);
                        const double depth = leftPose[2][0] * point3d_w.x + leftPose[2][1] * point3d_w.y + leftPose[2][2] * point3d_w.z + leftPose[2][3];
                        if (depth > kDynamicDepthMin && depth < kDynamicDepthMax && inst.IsInBoxPw(point3d_w, feat->frame)) {
                            feat->is_triangulated = true; feat->p_w = point3d_w; stereo_triangle_succeed++;
                            if (lm.depth <= 0) { lm.erase(lm.feats.begin(), it); lm.depth = depth; }
                        } else { stereo_triangle_failed++; feat->is_stereo = false; }
                    }
                }
            }
            if (stereo_triangle_succeed + stereo_triangle_failed == 0) continue;
            inst.set_triangle_num();
        }
    }
    // :495-576
    void InitialInstance(const Body& body) {
        for (auto& [inst_id, inst] : instances) {
            if (inst.is_initial) inst.age++;
            if (inst.is_initial || !inst.is_tracking || inst.points_extra[frame].empty()) continue;
            if ((int)inst.points_extra[frame].size() <= para.kInstanceInitMinNum) continue;
            std::vector<V3>& points3d = inst.points_extra[frame];
            State init_state;
            if (para.use_det3d) {
                if (!inst.boxes3d[frame]) continue;
                for (int k = 0; k < 3; ++k) inst.box3d->dims[k] = inst.boxes3d[frame]->dims[k];
                init_state.R = body.Rs[frame] * body.ric[0] * inst.boxes3d[frame]->R_cioi();
                init_state.P = BoxFitPoints(body, inst.points_extra[frame], init_state.R, inst.box3d->dimsv(), ransac_seed(inst.id, seq, 1));
            } else {
                inst.box3d->dims[0] = 2; inst.box3d->dims[1] = 4; inst.box3d->dims[2] = 1.5;
                V3 init_cam_pt;
                if (!FitBox3DFromCameraFrame(points3d, inst.box3d->dimsv(), init_cam_pt)) continue;
                init_state.P = init_cam_pt; init_state.R = M3::identity();
            }
            inst.state[frame].time = body.headers[0];
            inst.vel.SetZero();
            for (int i = 0; i <= kWinSize; i++) { inst.state[i] = init_state; inst.state[i].time = body.headers[i]; }
            inst.is_initial = true;
            inst.DeleteOutdatedLandmarks(frame);
        }
    }
    // :582-604
    void InitialInstanceVelocity(const Body& body) {
        for (auto& [inst_id, inst] : instances) {
            if (!inst.is_initial || inst.is_init_velocity) continue;
            if (inst.age < 3) continue;
            const State& si = inst.state[body.frame - 1]; const State& sj = inst.state[body.frame];
            const M3 Rinv = si.R.t(); const V3 tinv = -(Rinv * si.P);            // Isometry3d::inverse()
            const M3 R_oioj = Rinv * sj.R; const V3 t_oioj = Rinv * sj.P + tinv;
            const double time_ij = body.headers[body.frame] - body.headers[body.frame - 1];
            inst.vel.v = t_oioj / time_ij; inst.vel.a = SO3log(R_oioj) / time_ij;      // Velocity::SetVel
            inst.is_init_velocity = true;
        }
    }
    // :610-677
    void SetDynamicOrStatic(const Body& body) {
        InstExec([&](unsigned, Instance& inst) {
            if (!inst.is_curr_visible) return;
            int vec_size = 0; V3 scene_vec;
            FeaturePoint::Ptr next_ptr;
            for (auto& lm : inst.landmarks) {
                bool is_found_next = false;
                if (!lm.bad && lm.feats.size() > 1 && lm.feats.back()->frame == body.frame) {
                    for (auto it = lm.feats.rbegin(); it != lm.feats.rend(); it++) {
                        if ((*it)->is_triangulated) {
                            if (!is_found_next) { next_ptr = *it; is_found_next = true; }
                            else { scene_vec += (next_ptr->p_w - (*it)->p_w) / (body.headers[next_ptr->frame] - body.headers[(*it)->frame]); vec_size++; break; }
                        }
                    }
                }
            }
            const V3 vel = (inst.state[body.frame].P - inst.state[body.frame - 1].P) / (inst.state[body.frame].time - inst.state[body.frame - 1].time);
            scene_vec = scene_vec / (double)vec_size;
            if (vec_size < 5) return;
            if (vel.norm() > 15 || scene_vec.norm() > para.kStaticInstThreshold) inst.static_frame--; else inst.static_frame++;
            if (inst.static_frame >= 2) { inst.is_static = true; inst.static_frame = 2; }
            else if (inst.static_frame <= 0) { inst.is_static = false; inst.static_frame = 0; }
        });
    }
    // :772-807 with AddInstanceParameterBlock (:997-1015) and AddResidualBlockForInstOpt (:1018-1249); the ceres::Problem is expressed as the flat
    // problem of dvo_obj_solve (obj_solve.cpp), which adds the same residual blocks to the generic ceres restatement
    void Optimization(const Body& body) {
        std::vector<Instance*> objs;
        InstExec([&](unsigned, Instance& inst) { inst.SetOptimizeParameters(); objs.push_back(&inst); });
        std::vector<dvo_obj_box> boxes; std::vector<dvo_obj_point> points;
        if (tracking_num >= 1) for (size_t o = 0; o < objs.size(); ++o) {
            Instance& inst = *objs[o];
            if (!inst.is_initial || !inst.is_tracking) continue;
            if (inst.valid_size() < 1) continue;
            for (int i = 0; i <= kWinSize; ++i) if (inst.boxes3d[i]) {
                dvo_obj_box b{}; b.obj = (int)o; b.frame = i; std::memcpy(b.dims, inst.boxes3d[i]->dims, 24);
                const M3 R = inst.boxes3d[i]->R_cioi(); for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) b.R_cioi[r * 3 + c] = R(r, c);
                boxes.push_back(b);
            }
            for (auto& lm : inst.landmarks) {
                if (lm.bad || lm.depth < 0.2 || lm.is_extra()) continue;
                for (auto& feat : lm.feats) if (feat->is_triangulated) { dvo_obj_point p{}; p.obj = (int)o; p.frame = feat->frame; p.p_w[0] = feat->p_w.x; p.p_w[1] = feat->p_w.y; p.p_w[2] = feat->p_w.z; points.push_back(p); }
            }
        }
        last_summary = dvo_ba_summary{};
        if (!objs.empty() && (!boxes.empty() || !points.empty())) {
            std::vector<double> st(objs.size() * 77), dm(objs.size() * 3);
            for (size_t o = 0; o < objs.size(); ++o) { std::memcpy(&st[o * 77], objs[o]->para_state, 77 * 8); std::memcpy(&dm[o * 3], objs[o]->para_box, 24); }
            dvo_obj_problem P{};
            P.n_obj = (int)objs.size(); P.n_boxes = (int)boxes.size(); P.n_points = (int)points.size(); P.max_iters = para.KNumIter; P.plane_kind = para.plane_kind;
            P.state = st.data(); P.dims = dm.data(); P.body_pose = &body.para_pose[0][0];
            // variant "obj_perturb" n (sensitivity only, dvo.h): the world points the enclose factors read are moved by n x 1e-7 m (a fixed pattern) — the size of the ego-state
            // difference between two correct implementations of the window solve (marginalization forms), which reaches the objects through p_w; tests/tools/obj_sensitivity.py
            if (const int np = dvo_get_variant("obj_perturb"); np > 0 && np < 1000)
                for (size_t i = 0; i < points.size(); ++i) for (int c = 0; c < 3; ++c) points[i].p_w[c] += 1e-7 * np * std::sin(1.7 * (double)i + 2.3 * c + 0.37 * seq);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) P.R_bc[r * 3 + c] = body.ric[0](r, c);
            P.boxes = boxes.data(); P.points = points.data();
            // hook "obj_dump" (tests/tools/obj_problem_dump.py; not a variant of the arithmetic): while set, the problem as it enters the solve is written under $DVO_OBJ_DUMP_DIR —
            // fixtures for the operator-level comparison of dv_obj_solve with dvo_obj_solve on problems a real sequence produced (tests/test_obj_sequence_problems.py)
            if (dvo_get_variant("obj_dump") > 0) {
                const char* dir = std::getenv("DVO_OBJ_DUMP_DIR");
                const std::string path = std::string(dir ? dir : "/tmp") + "/obj_" + std::to_string((long long)seq) + ".bin";
                if (FILE* f = std::fopen(path.c_str(), "wb")) {
                    const int32_t hdr[6] = { P.n_obj, P.n_boxes, P.n_points, P.max_iters, P.plane_kind, (int32_t)seq };
                    std::fwrite(hdr, 4, 6, f); std::fwrite(st.data(), 8, st.size(), f); std::fwrite(dm.data(), 8, dm.size(), f); std::fwrite(P.body_pose, 8, 77, f); std::fwrite(P.R_bc, 8, 9, f);
                    std::fwrite(boxes.data(), sizeof(dvo_obj_box), boxes.size(), f); std::fwrite(points.data(), sizeof(dvo_obj_point), points.size(), f);
                    std::fclose(f);
                }
            }
            dvo_obj_solve(&P, &last_summary);
            for (size_t o = 0; o < objs.size(); ++o) { std::memcpy(objs[o]->para_state, &st[o * 77], 77 * 8); std::memcpy(objs[o]->para_box, &dm[o * 3], 24); }
        }
        InstExec([](unsigned, Instance& inst) { inst.GetOptimizationParameters(); });
    }
    void OutliersRejection(const Body& body) { InstExec([&](unsigned, Instance& inst) { inst.OutlierRejection(body); }); }
    // :813-903
    void ManageTriangulatePoint(const Body& body) {
        for (auto& [key, inst] : instances) {
            if (inst.landmarks.empty() || !inst.is_initial || inst.valid_size() < 10) continue;
            int statistics[11] = { 0 };
            for (auto& lm : inst.landmarks) { if (lm.bad) continue; for (auto& feat : lm.feats) statistics[feat->frame]++; }
            if (statistics[kWinSize - 1] <= 2) continue;
            inst.OutlierRejectionByBox3d(body);
        }
        for (auto& [key, inst] : instances) {
            inst.set_triangle_num();
            if (inst.landmarks.empty()) continue;
            constexpr int KEEP_SIZE = 100;
            if (inst.triangle_num > KEEP_SIZE) {
                for (auto& lm : inst.landmarks) {
                    if (lm.bad) continue;
                    else if (lm.depth <= 0) { if ((lm.frame() != frame) || (lm.frame() == frame && lm.is_extra())) lm.bad = true; }
                }
            }
            if (int cnt = inst.valid_size(); cnt > KEEP_SIZE) {
                cnt -= KEEP_SIZE;
                for (auto& lm : inst.landmarks) {
                    if (lm.bad) continue;
                    if (lm.is_extra() && lm.frame() != frame) { lm.bad = true; cnt--; }
                    if (cnt <= 0) break;
                }
            }
            inst.set_triangle_num();
        }
    }
    // :910-960
    void SlideWindow(const Body& body, bool margin_old) {
        if (frame != kWinSize) return;
        for (auto& [key, inst] : instances) {
            if (!inst.is_tracking && inst.landmarks.empty()) continue;
            if (margin_old) inst.SlideWindowOld(body); else inst.SlideWindowNew(body);
            int pc_frame_num = 0;
            for (int i = 0; i <= kWinSize; ++i) if (!inst.points_extra[i].empty()) pc_frame_num++;
            inst.set_triangle_num();
            if (inst.landmarks.empty()) inst.ClearState();
            else if (inst.is_tracking && (inst.triangle_num == 0 && pc_frame_num == 0)) inst.is_initial = false;
        }
    }
    void DeleteBadLandmarks() { for (auto& [key, inst] : instances) { if (inst.landmarks.empty()) continue; inst.DeleteBadLandmarks(); } }
};

}  // namespace oim
