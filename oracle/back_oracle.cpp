// back_oracle.cpp — CPU ORACLE (test infrastructure, not the product) for the back end: restates
//   Estimator::{ProcessMeasurements,ProcessImage,InitEstimator,Optimization,SetMarginalizationInfo,SlideWindow,
//              InitFramePoseByPnP,InputIMU,GetIMUInterval,InitFirstIMUPose,AddIMU,ProcessIMU,Vector2double,Double2vector}
//                                                   estimator/estimator.cpp:68-339,403-619,729-847,1087-1154,1201-1366,1424-1696,1765-1863
//   BodyState::{Set,Get}OptimizeParameters           estimator/body.cpp:17-132
//   FeatureManager::*                                estimator/feature_manager.cpp:42-333,568-778
//   TriangulatePoint / OutliersRejection / ReprojectionError / SolvePoseByPnP / CompensatedParallax2
//                                                   estimator/vio_util.cpp:30-45,381-443,637-712
//   MarginalizationInfo / MarginalizationFactor / ResidualBlockInfo   estimator/factor/marginalization_factor.cpp:18-396
//   SolveGyroscopeBias                               estimator/initial/initial_aligment.cpp:29-61
//   Utility::{R2ypr,ypr2R,g2R}                       estimator/utility.h:86-131, utility.cpp:24-35
// PARITY UNPINNED (dvo.h).  Canonical choices in addition to back_solver.h S1/S2:
//   M1  marginalization orders blocks by first insertion (dropped first, then kept) instead of the
//       pointer-keyed unordered_map iteration order (Q9).
//   P1  cv::solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess) is restated as Levenberg-Marquardt on the
//       Rodrigues/translation parameters (<= 20 iterations, CvLevMarq update rule), validated by
//       reprojection error, not by bit parity (App. A.5).
//   Scope: stereo (+IMU or vision-only) initialisation, i.e. every BASELINE config; monocular SFM
//   initialisation (initial_sfm / solve_5pts) is not restated.
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <list>
#include <map>
#include <set>
#include <thread>
#include "back_factors.h"
#include "back_solver.h"
#include "dvo.h"
#include "inst_manager.h"
extern "C" int dvo_get_threads();

namespace obe {

constexpr int kWin = 10;     // kWinSize (utils/parameters.h:42)

static V3 R2ypr(const M3& R) {
    V3 n = R.col(0), o = R.col(1), a = R.col(2);
    double y = std::atan2(n.y, n.x);
    double p = std::atan2(-n.z, n.x * std::cos(y) + n.y * std::sin(y));
    double r = std::atan2(a.x * std::sin(y) - a.y * std::cos(y), -o.x * std::sin(y) + o.y * std::cos(y));
    return V3(y, p, r) / M_PI * 180.0;
}
static M3 ypr2R(const V3& ypr) {
    double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
    M3 Rz, Ry, Rx;
    Rz(0, 0) = std::cos(y); Rz(0, 1) = -std::sin(y); Rz(1, 0) = std::sin(y); Rz(1, 1) = std::cos(y); Rz(2, 2) = 1;
    Ry(0, 0) = std::cos(p); Ry(0, 2) = std::sin(p); Ry(1, 1) = 1; Ry(2, 0) = -std::sin(p); Ry(2, 2) = std::cos(p);
    Rx(0, 0) = 1; Rx(1, 1) = std::cos(r); Rx(1, 2) = -std::sin(r); Rx(2, 1) = std::sin(r); Rx(2, 2) = std::cos(r);
    return Rz * Ry * Rx;
}
static M3 g2R(const V3& g) {
    M3 R0 = Q::fromTwoVectors(g.normalized(), V3(0, 0, 1)).R();
    double yaw = R2ypr(R0).x;
    return ypr2R(V3(-yaw, 0, 0)) * R0;
}

struct Feat {                 // StaticPointFeature (basic/static_point_feature.h)
    V3 point, point_right; V3 vel, vel_right; double cur_td = 0; bool is_stereo = false;
};
struct Landmark {             // StaticPointLandmark
    int feature_id; int start_frame; std::vector<Feat> feats; double depth = -1.0; int solve_flag = 0;
    int endFrame() const { return start_frame + (int)feats.size() - 1; }
};

struct LineFeat { double line_obs[4], line_obs_right[4]; bool is_stereo = false; };      // LineFeature (basic/line_feature.h)
struct LineLandmark {         // basic/line_landmark.h
    int feature_id, start_frame; std::vector<LineFeat> feats; unsigned used_num = 0; bool is_triangulation = false; double line_plucker[6] = { 0 }; V3 ptw1, ptw2;
    int endFrame() const { return start_frame + (int)feats.size() - 1; }
};
struct LineCost : CostFunction {      // lineProjectionFactor on (pose, ex_pose, line) through dvo_line_eval (obj_factors.cpp)
    double obs[4], si[4];
    LineCost(const double* o, const double* w) { nres = 2; sizes = { 7, 7, 4 }; std::memcpy(obs, o, 32); std::memcpy(si, w, 32); }
    void Evaluate(const double* const* par, double* res, double** J) const override { dvo_line_eval(obs, si, par, res, J); }
};
struct ProjCost : CostFunction {
    int kind; ProjObs o;
    ProjCost(int k, const ProjObs& ob) : kind(k), o(ob) {
        nres = 2;
        if (k == 0) sizes = { 7, 7, 7, 1, 1 }; else if (k == 1) sizes = { 7, 7, 7, 7, 1, 1 }; else sizes = { 7, 7, 1, 1 };
    }
    void Evaluate(const double* const* par, double* res, double** J) const override { proj_eval(kind, o, par, res, J); }
};
struct ImuCost : CostFunction {
    const Integration* pre; V3 G;
    ImuCost(const Integration* p, const V3& g) : pre(p), G(g) { nres = 15; sizes = { 7, 9, 7, 9 }; }
    void Evaluate(const double* const* par, double* res, double** J) const override { imu_eval(*pre, G, par, res, J); }
};

struct MargInfo {             // MarginalizationInfo after marginalize() + getParameterBlocks()
    bool valid = true; int m = 0, n = 0;
    std::vector<int> keep_size, keep_idx; std::vector<std::vector<double>> keep_data;
    std::vector<double*> keep_addr;       // last_marg_para_blocks
    Mat J0; std::vector<double> r0;       // linearized_jacobians / linearized_residuals
};
struct MargCost : CostFunction {          // MarginalizationFactor (marginalization_factor.cpp:339-396)
    const MargInfo* mi;
    explicit MargCost(const MargInfo* m) : mi(m) { nres = m->n; sizes = m->keep_size; }
    void Evaluate(const double* const* par, double* res, double** J) const override {
        const int n = mi->n, m = mi->m;
        std::vector<double> dx(n, 0.0);
        for (size_t i = 0; i < mi->keep_size.size(); ++i) {
            const int size = mi->keep_size[i], idx = mi->keep_idx[i] - m;
            const double* x = par[i]; const double* x0 = mi->keep_data[i].data();
            if (size != 7) for (int k = 0; k < size; ++k) dx[idx + k] = x[k] - x0[k];
            else {
                for (int k = 0; k < 3; ++k) dx[idx + k] = x[k] - x0[k];
                Q dq = Q(x0[6], x0[3], x0[4], x0[5]).inverse() * Q(x[6], x[3], x[4], x[5]);
                V3 v = dq.vec() * 2.0;
                if (!(dq.w >= 0)) v = -v;
                for (int k = 0; k < 3; ++k) dx[idx + 3 + k] = v[k];
            }
        }
        for (int i = 0; i < n; ++i) { double s = mi->r0[i]; for (int k = 0; k < n; ++k) s += mi->J0(i, k) * dx[k]; res[i] = s; }
        if (!J) return;
        for (size_t b = 0; b < mi->keep_size.size(); ++b) {
            if (!J[b]) continue;
            const int size = mi->keep_size[b], ls = size == 7 ? 6 : size, idx = mi->keep_idx[b] - m;
            for (int i = 0; i < n; ++i) { for (int k = 0; k < size; ++k) J[b][i * size + k] = 0.0; for (int k = 0; k < ls; ++k) J[b][i * size + k] = mi->J0(i, idx + k); }
        }
    }
};

struct RBInfo {               // ResidualBlockInfo
    std::shared_ptr<CostFunction> f; int loss; std::vector<double*> blocks; std::vector<int> drop;
    std::vector<double> res; std::vector<std::vector<double>> J;
    void Evaluate() {         // marginalization_factor.cpp:18-79
        const int nb = (int)blocks.size();
        res.assign(f->nres, 0.0); J.assign(nb, {});
        std::vector<const double*> par(nb); std::vector<double*> Jp(nb);
        for (int k = 0; k < nb; ++k) { par[k] = blocks[k]; J[k].assign((size_t)f->nres * f->sizes[k], 0.0); Jp[k] = J[k].data(); }
        f->Evaluate(par.data(), res.data(), Jp.data());
        if (loss != kNoLoss) correct(loss, f->nres, res.data(), J, f->sizes, nullptr);
    }
};

struct Marginalizer {         // MarginalizationInfo::{addResidualBlockInfo,preMarginalize,marginalize,getParameterBlocks}
    std::vector<RBInfo> factors;
    std::vector<double*> order; std::map<double*, int> size, idx; std::set<double*> dropped; std::map<double*, std::vector<double>> data;
    void add(RBInfo rb) {
        for (size_t i = 0; i < rb.blocks.size(); ++i) if (!size.count(rb.blocks[i])) { size[rb.blocks[i]] = rb.f->sizes[i]; order.push_back(rb.blocks[i]); }
        for (int d : rb.drop) dropped.insert(rb.blocks[d]);
        factors.push_back(std::move(rb));
    }
    static int ls(int s) { return s == 7 ? 6 : s; }
    std::unique_ptr<MargInfo> run(const std::map<double*, double*>& addr_shift) {
        auto out = std::make_unique<MargInfo>();
        for (auto& f : factors) {     // preMarginalize
            f.Evaluate();
            for (size_t i = 0; i < f.blocks.size(); ++i) if (!data.count(f.blocks[i])) data[f.blocks[i]] = std::vector<double>(f.blocks[i], f.blocks[i] + f.f->sizes[i]);
        }
        int pos = 0;
        for (double* p : order) if (dropped.count(p)) { idx[p] = pos; pos += ls(size[p]); }       // M1
        const int m = pos;
        for (double* p : order) if (!dropped.count(p)) { idx[p] = pos; pos += ls(size[p]); }
        const int n = pos - m;
        out->m = m; out->n = n;
        if (m == 0) { out->valid = false; return out; }
        Mat A(pos, pos); std::vector<double> b(pos, 0.0);
        // marginalization_factor.cpp:254-281 builds A, b on 4 pthreads (residual block i goes to thread i % 4, partial sums added in thread order);
        // here that form is used for the multi-threaded CPU-baseline TIMING (dvo_set_threads > 1), the default is the sequential sum over the blocks
        const int NT = dvo_get_threads() > 1 ? 4 : 1;
        std::vector<Mat> Ap_(NT > 1 ? NT : 0, Mat(pos, pos)); std::vector<std::vector<double>> bp_(NT > 1 ? NT : 0, std::vector<double>(pos, 0.0));
        auto build = [&](Mat& A, std::vector<double>& b, int first, int stride) {
          for (size_t fi = first; fi < factors.size(); fi += stride) {
            auto& f = factors[fi];
            const int nr = f.f->nres;
            for (size_t i = 0; i < f.blocks.size(); ++i) {
                const int ii = idx[f.blocks[i]], si = ls(size[f.blocks[i]]), ci = f.f->sizes[i];
                for (size_t j = i; j < f.blocks.size(); ++j) {
                    const int jj = idx[f.blocks[j]], sj = ls(size[f.blocks[j]]), cj = f.f->sizes[j];
                    for (int a = 0; a < si; ++a) for (int c = 0; c < sj; ++c) {
                        double s = 0; for (int r = 0; r < nr; ++r) s += f.J[i][r * ci + a] * f.J[j][r * cj + c];
                        A(ii + a, jj + c) += s;
                        if (i != j) A(jj + c, ii + a) = A(ii + a, jj + c);
                    }
                }
                for (int a = 0; a < si; ++a) { double s = 0; for (int r = 0; r < nr; ++r) s += f.J[i][r * ci + a] * f.res[r]; b[ii + a] += s; }
            }
          }
        };
        if (NT == 1) build(A, b, 0, 1);
        else {
            std::vector<std::thread> th;
            for (int k = 0; k < NT; ++k) th.emplace_back([&, k] { build(Ap_[k], bp_[k], k, NT); });
            for (auto& t : th) t.join();
            for (int k = NT - 1; k >= 0; --k) { for (size_t e = 0; e < A.d.size(); ++e) A.d[e] += Ap_[k].d[e]; for (int e = 0; e < pos; ++e) b[e] += bp_[k][e]; }
        }
        Mat Amm(m, m);
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Amm(i, j) = 0.5 * (A(i, j) + A(j, i));
        std::vector<double> ev; Mat V;
        sym_eig(Amm, ev, V);
        const double eps = 1e-8;
        Mat Amm_inv(m, m);
        for (int k = 0; k < m; ++k) { if (!(ev[k] > eps)) continue; const double inv = 1.0 / ev[k]; for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Amm_inv(i, j) += V(i, k) * inv * V(j, k); }
        // A' = Arr - Arm Amm^-1 Amr ; b' = brr - Arm Amm^-1 bmm
        Mat Arm(n, m), Amr(m, n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) { Arm(i, j) = A(m + i, j); Amr(j, i) = A(j, m + i); }
        Mat T = matmul(Arm, Amm_inv);
        Mat TA = matmul(T, Amr);
        Mat Ap(n, n); std::vector<double> bp(n);
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) Ap(i, j) = A(m + i, m + j) - TA(i, j);
            double s = b[m + i]; for (int k = 0; k < m; ++k) s -= T(i, k) * b[k]; bp[i] = s;
        }
        for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { double s = 0.5 * (Ap(i, j) + Ap(j, i)); Ap(i, j) = Ap(j, i) = s; }   // SelfAdjointEigenSolver reads one triangle
        std::vector<double> ev2; Mat V2;
        sym_eig(Ap, ev2, V2);
        out->J0 = Mat(n, n); out->r0.assign(n, 0.0);
        for (int k = 0; k < n; ++k) {
            const double S = ev2[k] > eps ? ev2[k] : 0.0, Sinv = ev2[k] > eps ? 1.0 / ev2[k] : 0.0;
            const double ss = std::sqrt(S), sis = std::sqrt(Sinv);
            double vb = 0; for (int i = 0; i < n; ++i) { out->J0(k, i) = ss * V2(i, k); vb += V2(i, k) * bp[i]; }
            out->r0[k] = sis * vb;
        }
        for (double* p : order) if (!dropped.count(p)) {       // getParameterBlocks
            out->keep_size.push_back(size[p]); out->keep_idx.push_back(idx[p]); out->keep_data.push_back(data[p]);
            auto it = addr_shift.find(p);
            out->keep_addr.push_back(it == addr_shift.end() ? nullptr : it->second);
        }
        return out;
    }
};

// TriangulatePoint (vio_util.cpp:30-45); Pose = 3x4 [R^T | -R^T t]
static V3 triangulate(const double P0[3][4], const double P1[3][4], double x0, double y0, double x1, double y1) {
    Mat D(4, 4);
    for (int c = 0; c < 4; ++c) {
        D(0, c) = x0 * P0[2][c] - P0[0][c]; D(1, c) = y0 * P0[2][c] - P0[1][c];
        D(2, c) = x1 * P1[2][c] - P1[0][c]; D(3, c) = y1 * P1[2][c] - P1[1][c];
    }
    double v[4]; smallest_right_singular4(D, v);
    return { v[0] / v[3], v[1] / v[3], v[2] / v[3] };
}

struct Config {
    int use_imu = 1, stereo = 1, plane_constraint = 0, max_iters = 8;
    double min_parallax = 10.0 / kFocalLength, init_depth = 5.0, g_norm = 9.81, td = 0.0;
    ImuNoise noise{ 0.1, 0.01, 0.001, 1e-4 };
    M3 ric[2]; V3 tic[2];
    int use_line = 0, line_min_obs = 5; double line_sqrt_info[4] = { 0, 0, 0, 0 };      // cfg::use_line, para::kLineMinObs, lineProjectionFactor::sqrt_info (never assigned in the reference: zero)
    int estimate = 0;      // bit 0 cfg::is_estimate_ex (1: optimise the extrinsics around the initial guess; 2 = calibrate from scratch is not restated), bit 1 cfg::is_estimate_td
    int dynamic = 0, use_det3d = 0, instance_init_min_num = 4; double static_inst_threshold = 10.0;      // cfg::slam == kDynamic, use_det3d, para::kInstanceInitMinNum, kStaticInstThreshold
};

struct Estimator {
    Config cfg;
    // BodyState (estimator/body.h)
    M3 ric[2]; V3 tic[2]; V3 Ps[kWin + 1], Vs[kWin + 1], Bas[kWin + 1], Bgs[kWin + 1]; M3 Rs[kWin + 1];
    V3 g; double td = 0; double headers[kWin + 1] = { 0 };
    double para_ex_pose[2][7], para_pose[kWin + 1][7], para_speed_bias[kWin + 1][9], para_feature[1000][1], para_td[1][1];
    int frame = 0; bool open_ex_estimation = false;      // Estimator::openExEstimation (estimator.h:173; reset by ClearState, estimator.cpp:632)
    std::list<Landmark> lms;                   // FeatureManager::point_landmarks
    int last_track_num = 0, new_feature_num = 0, long_track_num = 0;
    std::deque<std::pair<double, V3>> acc_buf, gyr_buf;
    double prev_time = -1, cur_time = 0; bool first_imu = false, init_first_pose = false; V3 acc_0, gyr_0;
    std::unique_ptr<Integration> pre[kWin + 1], tmp_pre;
    std::vector<double> dt_buf[kWin + 1]; std::vector<V3> la_buf[kWin + 1], av_buf[kWin + 1];
    std::vector<std::pair<double, std::shared_ptr<Integration>>> all_frames;       // all_image_frame (header, pre_integration)
    bool nonlinear = false; bool margin_old = true;
    std::unique_ptr<MargInfo> last_marg;
    M3 back_R0; V3 back_P0;
    SolveSummary last_summary; int n_solves = 0;
    std::list<LineLandmark> line_landmarks;    // FeatureManager::line_landmarks
    std::vector<dvo_line_row> pending_lines; double para_line_features[1000][4];
    oim::InstanceManager im;                   // Estimator::im (estimator.h)
    // variant "obj_perturb" 1000 + n (sensitivity only, dvo.h): the WHOLE object branch (triangulation, initialisation, solve, outlier tests) sees body positions moved by
    // n x 1e-7 m, a pattern keyed to the frame's time stamp so that a frame keeps its offset while the window slides — what the ego-state difference between two correct
    // window solves (4e-7 m between the HIP path and this oracle) is worth to the objects through the closed loop pose -> landmark depths -> world points -> pose
    V3 Ps_pert[kWin + 1]; double para_pert[kWin + 1][7];
    oim::Body body() {
        const int np = dvo_get_variant("obj_perturb");
        if (np >= 1000) {
            for (int f = 0; f <= kWin; ++f) {
                std::memcpy(para_pert[f], para_pose[f], sizeof(para_pert[f]));
                const double d[3] = { std::sin(37.0 * headers[f] + 0.3), std::sin(37.0 * headers[f] + 2.6), std::sin(37.0 * headers[f] + 4.9) }, a = 1e-7 * (np - 1000);
                Ps_pert[f] = Ps[f]; Ps_pert[f].x += a * d[0]; Ps_pert[f].y += a * d[1]; Ps_pert[f].z += a * d[2];
                for (int c = 0; c < 3; ++c) para_pert[f][c] += a * d[c];
            }
            return oim::Body{ Rs, Ps_pert, ric, tic, headers, td, frame, para_pert };
        }
        return oim::Body{ Rs, Ps, ric, tic, headers, td, frame, para_pose };
    }

    explicit Estimator(const Config& c) : cfg(c) { clear(); set_parameter(); }
    void clear() {
        for (int i = 0; i <= kWin; ++i) { Rs[i] = M3::identity(); Ps[i] = Vs[i] = Bas[i] = Bgs[i] = V3(); pre[i].reset(); dt_buf[i].clear(); la_buf[i].clear(); av_buf[i].clear(); headers[i] = 0; }
        lms.clear(); acc_buf.clear(); gyr_buf.clear(); all_frames.clear(); tmp_pre.reset(); last_marg.reset(); open_ex_estimation = false;
        prev_time = -1; cur_time = 0; first_imu = false; init_first_pose = false; frame = 0; nonlinear = false;
        line_landmarks.clear(); pending_lines.clear();
        im = oim::InstanceManager();
        im.para.use_det3d = cfg.use_det3d; im.para.kInstanceInitMinNum = cfg.instance_init_min_num; im.para.kStaticInstThreshold = cfg.static_inst_threshold;
        im.para.plane_kind = cfg.plane_constraint ? (cfg.use_imu ? 1 : 2) : 0; im.para.KNumIter = cfg.max_iters;
        std::memset(para_pose, 0, sizeof(para_pose));
    }
    void set_parameter() { for (int i = 0; i < 2; ++i) { ric[i] = cfg.ric[i]; tic[i] = cfg.tic[i]; } td = cfg.td; g = V3(0, 0, cfg.g_norm); }

    // ------------------------------ IMU ------------------------------
    void input_imu(double t, const V3& a, const V3& w) { acc_buf.push_back({ t, a }); gyr_buf.push_back({ t, w }); }
    bool imu_available(double t) const { return !acc_buf.empty() && t <= acc_buf.back().first; }
    bool get_imu_interval(double t0, double t1, std::vector<std::pair<double, V3>>& av, std::vector<std::pair<double, V3>>& gv) {
        if (acc_buf.empty()) return false;
        if (t1 <= acc_buf.back().first) {
            while (acc_buf.front().first <= t0) { acc_buf.pop_front(); gyr_buf.pop_front(); }
            while (acc_buf.front().first < t1) { av.push_back(acc_buf.front()); acc_buf.pop_front(); gv.push_back(gyr_buf.front()); gyr_buf.pop_front(); }
            av.push_back(acc_buf.front()); gv.push_back(gyr_buf.front());
            return true;
        }
        return false;
    }
    void process_imu(double dt, const V3& la, const V3& av) {       // ProcessIMU (estimator.cpp:811-847)
        if (!first_imu) { first_imu = true; acc_0 = la; gyr_0 = av; }
        if (!pre[frame]) pre[frame] = std::make_unique<Integration>(acc_0, gyr_0, Bas[frame], Bgs[frame], cfg.noise);
        if (frame != 0) {
            pre[frame]->push_back(dt, la, av);
            tmp_pre->push_back(dt, la, av);
            dt_buf[frame].push_back(dt); la_buf[frame].push_back(la); av_buf[frame].push_back(av);
            const int j = frame;
            V3 un_acc_0 = Rs[j] * (acc_0 - Bas[j]) - g;
            V3 un_gyr = (gyr_0 + av) * 0.5 - Bgs[j];
            Rs[j] = Rs[j] * deltaQ(un_gyr * dt).R();      // un-normalised quaternion -> toRotationMatrix, as in the reference
            V3 un_acc_1 = Rs[j] * (la - Bas[j]) - g;
            V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
            Ps[j] += Vs[j] * dt + un_acc * (0.5 * dt * dt);
            Vs[j] += un_acc * dt;
        }
        acc_0 = la; gyr_0 = av;
    }
    void add_imu(std::vector<std::pair<double, V3>>& av, std::vector<std::pair<double, V3>>& gv) {
        if (!init_first_pose) {               // InitFirstIMUPose
            init_first_pose = true;
            V3 aver; for (auto& a : av) aver += a.second; aver = aver / (double)av.size();
            M3 R0 = g2R(aver);
            double yaw = R2ypr(R0).x;
            Rs[0] = ypr2R(V3(-yaw, 0, 0)) * R0;
        }
        for (size_t i = 0; i < av.size(); ++i) {
            double dt;
            if (i == 0) dt = av[i].first - prev_time;
            else if (i == av.size() - 1) dt = cur_time - av[i - 1].first;
            else dt = av[i].first - av[i - 1].first;
            process_imu(dt, av[i].second, gv[i].second);
        }
    }

    // ------------------------------ FeatureManager ------------------------------
    int feature_count() { int c = 0; for (auto& l : lms) if (l.feats.size() >= 4) ++c; return c; }
    bool add_feature_check_parallax(int frame_count, const dvo_feat* feats, int n, double td_) {
        double parallax_sum = 0; int parallax_num = 0;
        last_track_num = 0; new_feature_num = 0; long_track_num = 0;
        std::map<unsigned, const dvo_feat*> image;        // the reference iterates a std::map keyed by id
        for (int i = 0; i < n; ++i) image[feats[i].id] = &feats[i];
        for (auto& kv : image) {
            const dvo_feat& f = *kv.second;
            Feat ft; ft.point = V3(f.left[0], f.left[1], f.left[2]); ft.vel = V3(f.left[5], f.left[6], 0); ft.cur_td = td_;
            if (f.has_right) { ft.point_right = V3(f.right[0], f.right[1], f.right[2]); ft.vel_right = V3(f.right[5], f.right[6], 0); ft.is_stereo = true; }
            auto it = std::find_if(lms.begin(), lms.end(), [&](const Landmark& l) { return l.feature_id == (int)kv.first; });
            if (it == lms.end()) { lms.push_back(Landmark{ (int)kv.first, frame_count, {}, -1.0, 0 }); lms.back().feats.push_back(ft); new_feature_num++; }
            else { it->feats.push_back(ft); last_track_num++; if (it->feats.size() >= 4) long_track_num++; }
        }
        if (frame_count < 2 || last_track_num < 20 || long_track_num < 40 || new_feature_num > 0.5 * last_track_num) return true;
        for (auto& lm : lms)
            if (lm.start_frame <= frame_count - 2 && lm.start_frame + (int)lm.feats.size() - 1 >= frame_count - 1) {
                const Feat& fi = lm.feats[frame_count - 2 - lm.start_frame]; const Feat& fj = lm.feats[frame_count - 1 - lm.start_frame];
                double u_j = fj.point.x, v_j = fj.point.y, dep_i = fi.point.z, u_i = fi.point.x / dep_i, v_i = fi.point.y / dep_i;
                double du = u_i - u_j, dv = v_i - v_j;
                parallax_sum += std::max(0.0, std::sqrt(std::min(du * du + dv * dv, du * du + dv * dv)));   // CompensatedParallax2
                parallax_num++;
            }
        if (parallax_num == 0) return true;
        return parallax_sum / parallax_num >= cfg.min_parallax;
    }
    void cam_pose34(int index, int cam, double P[3][4]) const {     // BodyState::GetCamPose34d
        V3 t0 = Ps[index] + Rs[index] * tic[cam]; M3 R0 = Rs[index] * ric[cam]; M3 Rt = R0.t(); V3 t = -(Rt * t0);
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P[i][j] = Rt(i, j); P[i][3] = t[i]; }
    }
    void triangulate_points() {       // FeatureManager::TriangulatePoints (feature_manager.cpp:244-333)
        for (auto& lm : lms) {
            if (lm.depth > 0) continue;
            if (cfg.stereo && lm.feats[0].is_stereo) {
                double L[3][4], Rr[3][4]; cam_pose34(lm.start_frame, 0, L); cam_pose34(lm.start_frame, 1, Rr);
                V3 p = triangulate(L, Rr, lm.feats[0].point.x, lm.feats[0].point.y, lm.feats[0].point_right.x, lm.feats[0].point_right.y);
                double depth = L[2][0] * p.x + L[2][1] * p.y + L[2][2] * p.z + L[2][3];
                lm.depth = depth > 0 ? depth : cfg.init_depth;
                continue;
            } else if (lm.feats.size() > 1) {
                double L[3][4], Rr[3][4]; cam_pose34(lm.start_frame, 0, L); cam_pose34(lm.start_frame + 1, 0, Rr);
                V3 p = triangulate(L, Rr, lm.feats[0].point.x, lm.feats[0].point.y, lm.feats[1].point.x, lm.feats[1].point.y);
                double depth = L[2][0] * p.x + L[2][1] * p.y + L[2][2] * p.z + L[2][3];
                lm.depth = depth > 0 ? depth : cfg.init_depth;
                continue;
            }
            // (single mono observation: the multi-view branch below needs >= 4 observations, unreachable with 1)
        }
    }
    void remove_back_shift_depth(const M3& marg_R, const V3& marg_P, const M3& new_R, const V3& new_P) {
        for (auto it = lms.begin(); it != lms.end();) {
            auto cur = it++;
            if (cur->start_frame != 0) { cur->start_frame--; continue; }
            V3 uv_i = cur->feats[0].point;
            cur->feats.erase(cur->feats.begin());
            if (cur->feats.size() < 2) { lms.erase(cur); continue; }
            V3 w = marg_R * (uv_i * cur->depth) + marg_P;
            V3 pj = new_R.t() * (w - new_P);
            cur->depth = pj.z > 0 ? pj.z : cfg.init_depth;
        }
    }
    void remove_back() {
        for (auto it = lms.begin(); it != lms.end();) {
            auto cur = it++;
            if (cur->start_frame != 0) cur->start_frame--;
            else { cur->feats.erase(cur->feats.begin()); if (cur->feats.empty()) lms.erase(cur); }
        }
    }
    void remove_front(int frame_count) {
        for (auto it = lms.begin(); it != lms.end();) {
            auto cur = it++;
            if (cur->start_frame == frame_count) cur->start_frame--;
            else {
                int j = kWin - 1 - cur->start_frame;
                if (cur->endFrame() < frame_count - 1) continue;
                cur->feats.erase(cur->feats.begin() + j);
                if (cur->feats.empty()) lms.erase(cur);
            }
        }
    }

    // ------------------------------ PnP (P1) ------------------------------
    static void rodrigues(const V3& r, M3& R) {
        double th = r.norm();
        if (th < 1e-12) { R = M3::identity() + skew(r); return; }
        V3 k = r / th; M3 K = skew(k);
        R = M3::identity() + K * std::sin(th) + K * K * (1 - std::cos(th));
    }
    static V3 inv_rodrigues(const M3& R) {
        Q q = Q::fromR(R).normalized(); if (q.w < 0) q = Q(-q.w, -q.x, -q.y, -q.z);
        double s = q.vec().norm(); if (s < 1e-12) return q.vec() * 2.0;
        double th = 2 * std::atan2(s, q.w); return q.vec() / s * th;
    }
    bool solve_pnp(M3& R, V3& P, const std::vector<V3>& p3, const std::vector<std::pair<float, float>>& p2) {
        if ((int)p2.size() < 4) return false;
        M3 Ri = R.t(); V3 ti = -(Ri * P);             // w_T_cam -> cam_T_w
        V3 rv = inv_rodrigues(Ri);
        double x[6] = { rv.x, rv.y, rv.z, ti.x, ti.y, ti.z };
        auto cost = [&](const double* p, std::vector<double>* res) {
            M3 Rm; rodrigues(V3(p[0], p[1], p[2]), Rm); V3 t(p[3], p[4], p[5]);
            double c = 0; if (res) res->resize(p2.size() * 2);
            for (size_t i = 0; i < p2.size(); ++i) { V3 q = Rm * p3[i] + t; double ex = q.x / q.z - p2[i].first, ey = q.y / q.z - p2[i].second; c += ex * ex + ey * ey; if (res) { (*res)[2 * i] = ex; (*res)[2 * i + 1] = ey; } }
            return c;
        };
        double lambda = 1e-3; std::vector<double> r0, r1;
        double c0 = cost(x, &r0);
        for (int it = 0; it < 20; ++it) {
            Mat JtJ(6, 6); double Jtr[6] = { 0 };
            for (size_t i = 0; i < p2.size(); ++i) for (int rr = 0; rr < 2; ++rr) {
                double Jrow[6];
                for (int k = 0; k < 6; ++k) { double xp[6]; std::memcpy(xp, x, sizeof(xp)); const double h = 1e-7; xp[k] += h; M3 Rm; rodrigues(V3(xp[0], xp[1], xp[2]), Rm); V3 q = Rm * p3[i] + V3(xp[3], xp[4], xp[5]); double e = (rr == 0 ? q.x / q.z - p2[i].first : q.y / q.z - p2[i].second); Jrow[k] = (e - r0[2 * i + rr]) / h; }
                for (int a = 0; a < 6; ++a) { Jtr[a] += Jrow[a] * r0[2 * i + rr]; for (int b = 0; b < 6; ++b) JtJ(a, b) += Jrow[a] * Jrow[b]; }
            }
            bool improved = false;
            for (int tries = 0; tries < 10 && !improved; ++tries) {
                Mat A = JtJ; for (int a = 0; a < 6; ++a) A(a, a) *= (1 + lambda);
                Mat L; std::vector<double> d(Jtr, Jtr + 6);
                if (!cholesky(A, L)) { lambda *= 10; continue; }
                chol_solve(L, d);
                double xn[6]; for (int a = 0; a < 6; ++a) xn[a] = x[a] - d[a];
                double c1 = cost(xn, &r1);
                if (c1 < c0) { std::memcpy(x, xn, sizeof(xn)); r0 = r1; double dc = c0 - c1; c0 = c1; lambda = std::max(lambda / 10, 1e-16); improved = true; if (dc < 1e-20) it = 100; }
                else lambda *= 10;
            }
            if (!improved) break;
        }
        M3 Rp; rodrigues(V3(x[0], x[1], x[2]), Rp);
        R = Rp.t(); P = R * (-V3(x[3], x[4], x[5]));
        return true;
    }
    void init_frame_pose_by_pnp(int fc) {        // InitFramePoseByPnP (estimator.cpp:1323-1366)
        if (fc <= 0) return;
        std::vector<V3> p3; std::vector<std::pair<float, float>> p2;
        for (auto& lm : lms) if (lm.depth > 0) {
            int index = fc - lm.start_frame;
            if ((int)lm.feats.size() >= index + 1) {
                V3 w = Rs[lm.start_frame] * (ric[0] * (lm.feats[0].point * lm.depth) + tic[0]) + Ps[lm.start_frame];
                p3.push_back(V3((float)w.x, (float)w.y, (float)w.z));        // cv::Point3f
                p2.push_back({ (float)lm.feats[index].point.x, (float)lm.feats[index].point.y });
            }
        }
        M3 RCam = Rs[fc - 1] * ric[0]; V3 PCam = Rs[fc - 1] * tic[0] + Ps[fc - 1];
        if (solve_pnp(RCam, PCam, p3, p2)) {
            Rs[fc] = RCam * ric[0].t();
            Ps[fc] = -(RCam * ric[0].t() * tic[0]) + PCam;
            if (cfg.plane_constraint) { if (cfg.use_imu) Ps[fc].z = 0; else Ps[fc].y = 0; }
        }
    }

    // ------------------------------ line landmarks (feature_manager.cpp:124-160,339-356,392-560,611-778) ------------------------------
    void add_line_features(int frame_count) {
        std::map<unsigned, const dvo_line_row*> image;      // FeatureBackground::lines
        for (auto& r : pending_lines) image[r.id] = &r;
        for (auto& [line_id, r] : image) {
            LineFeat feat; std::memcpy(feat.line_obs, r->left, 32); feat.is_stereo = r->has_right != 0; if (feat.is_stereo) std::memcpy(feat.line_obs_right, r->right, 32);
            auto it = std::find_if(line_landmarks.begin(), line_landmarks.end(), [id = (int)line_id](const LineLandmark& l) { return l.feature_id == id; });
            if (it == line_landmarks.end()) { line_landmarks.push_back(LineLandmark{ (int)line_id, frame_count }); line_landmarks.back().feats.push_back(feat); }
            else it->feats.push_back(feat);
        }
        pending_lines.clear();
    }
    void body_arrays(double* R99, double* P33) const { for (int i = 0; i <= kWin; ++i) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R99[9 * i + 3 * r + c] = Rs[i](r, c); for (int k = 0; k < 3; ++k) P33[3 * i + k] = Ps[i][k]; } }
    void triangulate_line_mono() {       // TriangulateLineMono + TriangulateOneLine (vio_util.cpp:447-561) through dvo_triangulate_line
        double R99[99], P33[33], ric9[9], tic3[3]; body_arrays(R99, P33);
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) ric9[3 * r + c] = ric[0](r, c); tic3[r] = tic[0][r]; }
        for (auto& landmark : line_landmarks) {
            landmark.used_num = (unsigned)landmark.feats.size();
            if ((int)landmark.used_num < cfg.line_min_obs) continue;
            if (landmark.is_triangulation) continue;
            std::vector<double> obs; for (auto& f : landmark.feats) obs.insert(obs.end(), f.line_obs, f.line_obs + 4);
            double plk[6], w1[3], w2[3];
            if (dvo_triangulate_line(obs.data(), (int)landmark.feats.size(), landmark.start_frame, R99, P33, ric9, tic3, plk, w1, w2)) {
                std::memcpy(landmark.line_plucker, plk, 48); landmark.ptw1 = V3(w1[0], w1[1], w1[2]); landmark.ptw2 = V3(w2[0], w2[1], w2[2]); landmark.is_triangulation = true;
            }
        }
    }
    bool line_in_problem(LineLandmark& l) { l.used_num = (unsigned)l.feats.size(); return (int)l.used_num >= cfg.line_min_obs && l.start_frame < kWin - 2 && l.is_triangulation; }
    int line_feature_count() { int c = 0; for (auto& l : line_landmarks) if (line_in_problem(l)) ++c; return c; }
    static void plk_to_pose_o(const double* plk, const M3& Rcw, const V3& tcw, double* out) {      // line_geometry.cpp:252-262
        const V3 nw(plk[0], plk[1], plk[2]), vw(plk[3], plk[4], plk[5]);
        const V3 nc = Rcw * nw + skew(tcw) * (Rcw * vw), vc = Rcw * vw;
        for (int k = 0; k < 3; ++k) { out[k] = nc[k]; out[3 + k] = vc[k]; }
    }
    static void plk_from_pose_o(const double* plk, const M3& Rcw, const V3& tcw, double* out) { const M3 Rwc = Rcw.t(); plk_to_pose_o(plk, Rwc, -(Rwc * tcw), out); }
    void get_line_orth() {             // GetLineOrthVector -> para_line_features (Vector2double)
        int k = -1;
        for (auto& l : line_landmarks) {
            if (!line_in_problem(l)) continue;
            const V3 twc = Ps[l.start_frame] + Rs[l.start_frame] * tic[0]; const M3 Rwc = Rs[l.start_frame] * ric[0];
            double lw[6]; plk_to_pose_o(l.line_plucker, Rwc, twc, lw);
            dvo_plk_to_orth(lw, para_line_features[++k]);
        }
    }
    void set_line_orth() {             // SetLineOrth (Double2vector)
        int k = -1;
        for (auto& l : line_landmarks) {
            if (!line_in_problem(l)) continue;
            double lw[6]; dvo_orth_to_plk(para_line_features[++k], lw);
            const V3 twc = Ps[l.start_frame] + Rs[l.start_frame] * tic[0]; const M3 Rwc = Rs[l.start_frame] * ric[0];
            plk_from_pose_o(lw, Rwc, twc, l.line_plucker);
        }
    }
    void remove_line_outlier() {       // feature_manager.cpp:518-560
        for (auto it = line_landmarks.begin(); it != line_landmarks.end();) {
            auto landmark = it++;
            if (!line_in_problem(*landmark)) continue;
            int imu_i = landmark->start_frame, imu_j = imu_i - 1;
            const V3 twc = Ps[imu_i] + Rs[imu_i] * tic[0]; const M3 Rwc = Rs[imu_i] * ric[0];
            double p1[3], p2[3];
            const int valid = dvo_line_trimming(landmark->line_plucker, landmark->feats[0].line_obs, p1, p2);
            if (!valid || (V3(p1[0], p1[1], p1[2]) - V3(p2[0], p2[1], p2[2])).norm() > 10) { line_landmarks.erase(landmark); continue; }
            double line_w[6]; plk_to_pose_o(landmark->line_plucker, Rwc, twc, line_w);
            double allerr = 0;
            for (auto& feat : landmark->feats) {
                imu_j++;
                const V3 t1 = Ps[imu_j] + Rs[imu_j] * tic[0]; const M3 R1 = Rs[imu_j] * ric[0];
                double lc[6]; plk_from_pose_o(line_w, R1, t1, lc);        // LineReprojectionError (line_geometry.cpp:272-287)
                const double sql = std::sqrt(lc[0] * lc[0] + lc[1] * lc[1]);
                const V3 nc(lc[0] / sql, lc[1] / sql, lc[2] / sql);
                const double err = (std::fabs(nc.dot(V3(feat.line_obs[0], feat.line_obs[1], 1))) + std::fabs(nc.dot(V3(feat.line_obs[2], feat.line_obs[3], 1)))) / 2.0;
                if (allerr < err) allerr = err;
            }
            if (allerr > 3.0 / 500.0) line_landmarks.erase(landmark);
        }
    }
    void lines_remove_back_shift(const M3& marg_R, const V3& marg_P, const M3& new_R, const V3& new_P) {
        for (auto it = line_landmarks.begin(); it != line_landmarks.end();) {
            auto cur = it++;
            if (cur->start_frame != 0) { cur->start_frame--; continue; }
            cur->feats.erase(cur->feats.begin());
            if (cur->feats.size() < 2) { line_landmarks.erase(cur); continue; }
            const M3 Rji = new_R.t() * marg_R; const V3 tji = new_R.t() * (marg_P - new_P);
            double plk_j[6]; plk_to_pose_o(cur->line_plucker, Rji, tji, plk_j); std::memcpy(cur->line_plucker, plk_j, 48);
        }
    }
    void lines_remove_back() {
        for (auto it = line_landmarks.begin(); it != line_landmarks.end();) {
            auto cur = it++;
            if (cur->start_frame != 0) cur->start_frame--;
            else { cur->feats.erase(cur->feats.begin()); if (cur->feats.empty()) line_landmarks.erase(cur); }
        }
    }
    void lines_remove_front(int frame_count) {
        for (auto it = line_landmarks.begin(); it != line_landmarks.end();) {
            auto cur = it++;
            if (cur->start_frame == frame_count) { cur->start_frame--; continue; }
            const int j = kWin - 1 - cur->start_frame;
            if (cur->endFrame() < frame_count - 1) continue;
            cur->feats.erase(cur->feats.begin() + j);
            if (cur->feats.empty()) line_landmarks.erase(cur);
        }
    }
    // AddLineResidualBlock (estimator.cpp:224-253)
    void add_line_residual_blocks(Problem& prob, int loss) {
        if (!cfg.use_line) return;
        int feature_index = -1;
        for (auto& landmark : line_landmarks) {
            if (!line_in_problem(landmark)) continue;
            ++feature_index;
            prob.AddParameterBlock(para_line_features[feature_index], 4, kLineOrth);
            int imu_j = landmark.start_frame - 1;
            for (auto& feat : landmark.feats) {
                imu_j++;
                prob.AddResidualBlock(std::make_shared<LineCost>(feat.line_obs, cfg.line_sqrt_info), loss, { para_pose[imu_j], para_ex_pose[0], para_line_features[feature_index] });
            }
        }
    }
    // OptimizationWithOnlyLine (estimator.cpp:345-395): poses 0..kWinSize-1 and the extrinsics are added constant; para_pose[kWinSize] is not, so it enters as a free
    // 7-wide block without parameterisation whenever a line is seen in the newest frame (sic)
    void optimization_with_only_line() {
        vector2double();
        Problem prob;
        for (int i = 0; i < kWin; ++i) { prob.AddParameterBlock(para_pose[i], 7, kPose); prob.SetConstant(para_pose[i]); }
        for (int i = 0; i < 2; ++i) { prob.AddParameterBlock(para_ex_pose[i], 7, kPose); prob.SetConstant(para_ex_pose[i]); }
        prob.AddParameterBlock(para_pose[kWin], 7, kPlain);
        add_line_residual_blocks(prob, kCauchy1);
        Solver solver(prob);
        SolveOptions so; so.max_num_iterations = cfg.max_iters;
        solver.solve(so);
        double2vector();
        remove_line_outlier();
    }

    // ------------------------------ optimisation ------------------------------
    void vector2double() {       // BodyState::SetOptimizeParameters + Estimator::Vector2double
        for (int i = 0; i <= kWin; ++i) {
            para_pose[i][0] = Ps[i].x; para_pose[i][1] = Ps[i].y; para_pose[i][2] = Ps[i].z;
            Q q = Q::fromR(Rs[i]); para_pose[i][3] = q.x; para_pose[i][4] = q.y; para_pose[i][5] = q.z; para_pose[i][6] = q.w;
            if (cfg.use_imu) for (int k = 0; k < 3; ++k) { para_speed_bias[i][k] = Vs[i][k]; para_speed_bias[i][3 + k] = Bas[i][k]; para_speed_bias[i][6 + k] = Bgs[i][k]; }
        }
        for (int i = 0; i < 2; ++i) {
            para_ex_pose[i][0] = tic[i].x; para_ex_pose[i][1] = tic[i].y; para_ex_pose[i][2] = tic[i].z;
            Q q = Q::fromR(ric[i]); para_ex_pose[i][3] = q.x; para_ex_pose[i][4] = q.y; para_ex_pose[i][5] = q.z; para_ex_pose[i][6] = q.w;
        }
        para_td[0][0] = td;
        int k = -1;
        for (auto& lm : lms) if (lm.feats.size() >= 4) para_feature[++k][0] = 1.0 / lm.depth;
        if (cfg.use_line) get_line_orth();
    }
    void double2vector() {       // Estimator::Double2vector + BodyState::GetOptimizationParameters (body.cpp:61-132)
        V3 origin_R0 = R2ypr(Rs[0]), origin_P0 = Ps[0];
        auto qp = [&](int i) { return Q(para_pose[i][6], para_pose[i][3], para_pose[i][4], para_pose[i][5]); };
        if (cfg.use_imu) {
            V3 origin_R00 = R2ypr(qp(0).R());
            double y_diff = origin_R0.x - origin_R00.x;
            M3 rot_diff = ypr2R(V3(y_diff, 0, 0));
            if (std::fabs(std::fabs(origin_R0.y) - 90) < 1.0 || std::fabs(std::fabs(origin_R00.y) - 90) < 1.0) rot_diff = Rs[0] * qp(0).R().t();
            for (int i = 0; i <= kWin; ++i) {
                Rs[i] = rot_diff * qp(i).normalized().R();
                Ps[i] = rot_diff * V3(para_pose[i][0] - para_pose[0][0], para_pose[i][1] - para_pose[0][1], para_pose[i][2] - para_pose[0][2]) + origin_P0;
                Vs[i] = rot_diff * V3(para_speed_bias[i][0], para_speed_bias[i][1], para_speed_bias[i][2]);
                Bas[i] = V3(para_speed_bias[i][3], para_speed_bias[i][4], para_speed_bias[i][5]);
                Bgs[i] = V3(para_speed_bias[i][6], para_speed_bias[i][7], para_speed_bias[i][8]);
            }
            for (int i = 0; i < 2; ++i) {
                tic[i] = V3(para_ex_pose[i][0], para_ex_pose[i][1], para_ex_pose[i][2]);
                ric[i] = Q(para_ex_pose[i][6], para_ex_pose[i][3], para_ex_pose[i][4], para_ex_pose[i][5]).normalized().R();
            }
            td = para_td[0][0];
        } else {
            for (int i = 0; i <= kWin; ++i) { Rs[i] = qp(i).normalized().R(); Ps[i] = V3(para_pose[i][0], para_pose[i][1], para_pose[i][2]); }
        }
        int k = -1;
        for (auto& lm : lms) if (lm.feats.size() >= 4) { lm.depth = 1.0 / para_feature[++k][0]; lm.solve_flag = lm.depth < 0 ? 2 : 1; }    // SetDepth
        if (cfg.use_line) set_line_orth();       // body.cpp / estimator.cpp:1141
    }
    static ProjObs obs(const Feat& f0, const Feat& f, bool right) {
        ProjObs o; o.pts_i = f0.point; o.pts_j = right ? f.point_right : f.point; o.vel_i = f0.vel; o.vel_j = right ? f.vel_right : f.vel; o.td_i = f0.cur_td; o.td_j = f.cur_td; return o;
    }
    void optimization() {        // Estimator::Optimization (estimator.cpp:261-339)
        vector2double();
        Problem prob;
        const int pose_kind = cfg.plane_constraint ? (cfg.use_imu ? kPosePlaneImu : kPosePlaneVo) : kPose;
        for (int i = 0; i < frame + 1; ++i) {
            prob.AddParameterBlock(para_pose[i], 7, pose_kind);
            if (cfg.use_imu) prob.AddParameterBlock(para_speed_bias[i], 9);
        }
        if (!cfg.use_imu) prob.SetConstant(para_pose[0]);
        for (int i = 0; i < 2; ++i) {          // AddBodyParameterBlock, estimator.cpp:87-95 (estimate_extrinsic is 0 in every shipped config)
            prob.AddParameterBlock(para_ex_pose[i], 7, kPose);
            if (((cfg.estimate & 1) && frame == kWin && Vs[0].norm() > 0.2) || open_ex_estimation) open_ex_estimation = true;
            else prob.SetConstant(para_ex_pose[i]);
        }
        prob.AddParameterBlock(para_td[0], 1);          // :98-100 (estimate_td: 0 in every shipped config)
        if (!(cfg.estimate & 2) || Vs[0].norm() < 0.2) prob.SetConstant(para_td[0]);
        if (last_marg && last_marg->valid) prob.AddResidualBlock(std::make_shared<MargCost>(last_marg.get()), kNoLoss, last_marg->keep_addr);
        if (cfg.use_imu)
            for (int i = 0; i < frame; ++i) {
                int j = i + 1;
                if (pre[j]->sum_dt > 10.0) continue;
                prob.AddResidualBlock(std::make_shared<ImuCost>(pre[j].get(), g), kNoLoss, { para_pose[i], para_speed_bias[i], para_pose[j], para_speed_bias[j] });
            }
        int fi = -1;
        for (auto& lm : lms) {
            if (lm.feats.size() < 4) continue;
            ++fi;
            prob.AddParameterBlock(para_feature[fi], 1, kPlain, true);
            int imu_i = lm.start_frame, imu_j = imu_i - 1;
            for (auto& ft : lm.feats) {
                imu_j++;
                if (imu_i != imu_j) prob.AddResidualBlock(std::make_shared<ProjCost>(0, obs(lm.feats[0], ft, false)), kHuber1, { para_pose[imu_i], para_pose[imu_j], para_ex_pose[0], para_feature[fi], para_td[0] });
                if (cfg.stereo && ft.is_stereo) {
                    if (imu_i != imu_j) prob.AddResidualBlock(std::make_shared<ProjCost>(1, obs(lm.feats[0], ft, true)), kHuber1, { para_pose[imu_i], para_pose[imu_j], para_ex_pose[0], para_ex_pose[1], para_feature[fi], para_td[0] });
                    else prob.AddResidualBlock(std::make_shared<ProjCost>(2, obs(lm.feats[0], ft, true)), kHuber1, { para_ex_pose[0], para_ex_pose[1], para_feature[fi], para_td[0] });
                }
            }
        }
        add_line_residual_blocks(prob, kHuber1);      // estimator.cpp:283-286, same loss object as the points
        Solver solver(prob);
        SolveOptions so; so.max_num_iterations = cfg.max_iters;
        last_summary = solver.solve(so); n_solves++;
        if (cfg.dynamic) {       // im.AddInstanceParameterBlock (:272-276) registers the object blocks, no live residual touches them, im.GetOptimizationParameters (:321-323) reads them back
            im.InstExec([](unsigned, oim::Instance& inst) { inst.SetOptimizeParameters(); });
            im.InstExec([](unsigned, oim::Instance& inst) { inst.GetOptimizationParameters(); });
        }
        double2vector();
        if (frame < kWin) return;
        set_marginalization_info();
    }
    void set_marginalization_info() {       // estimator.cpp:403-619
        if (margin_old) {
            vector2double();
            Marginalizer mg;
            if (last_marg && last_marg->valid) {
                std::vector<int> drop;
                for (size_t i = 0; i < last_marg->keep_addr.size(); ++i) if (last_marg->keep_addr[i] == para_pose[0] || last_marg->keep_addr[i] == para_speed_bias[0]) drop.push_back((int)i);
                mg.add(RBInfo{ std::make_shared<MargCost>(last_marg.get()), kNoLoss, last_marg->keep_addr, drop });
            }
            if (cfg.use_imu && pre[1]->sum_dt < 10.0)
                mg.add(RBInfo{ std::make_shared<ImuCost>(pre[1].get(), g), kNoLoss, { para_pose[0], para_speed_bias[0], para_pose[1], para_speed_bias[1] }, { 0, 1 } });
            int fi = -1;
            for (auto& lm : lms) {
                if (lm.feats.size() < 4) continue;
                ++fi;
                int imu_i = lm.start_frame, imu_j = imu_i - 1;
                if (imu_i != 0) continue;
                for (auto& ft : lm.feats) {
                    imu_j++;
                    if (imu_i != imu_j) mg.add(RBInfo{ std::make_shared<ProjCost>(0, obs(lm.feats[0], ft, false)), kHuber1, { para_pose[imu_i], para_pose[imu_j], para_ex_pose[0], para_feature[fi], para_td[0] }, { 0, 3 } });
                    if (cfg.stereo && ft.is_stereo) {
                        if (imu_i != imu_j) mg.add(RBInfo{ std::make_shared<ProjCost>(1, obs(lm.feats[0], ft, true)), kHuber1, { para_pose[imu_i], para_pose[imu_j], para_ex_pose[0], para_ex_pose[1], para_feature[fi], para_td[0] }, { 0, 4 } });
                        else mg.add(RBInfo{ std::make_shared<ProjCost>(2, obs(lm.feats[0], ft, true)), kHuber1, { para_ex_pose[0], para_ex_pose[1], para_feature[fi], para_td[0] }, { 2 } });
                    }
                }
            }
            std::map<double*, double*> shift;
            for (int i = 1; i <= kWin; ++i) { shift[para_pose[i]] = para_pose[i - 1]; if (cfg.use_imu) shift[para_speed_bias[i]] = para_speed_bias[i - 1]; }
            for (int i = 0; i < 2; ++i) shift[para_ex_pose[i]] = para_ex_pose[i];
            shift[para_td[0]] = para_td[0];
            last_marg = mg.run(shift);
        } else {
            if (last_marg && std::count(last_marg->keep_addr.begin(), last_marg->keep_addr.end(), (double*)para_pose[kWin - 1])) {
                Marginalizer mg;
                vector2double();
                if (last_marg->valid) {
                    std::vector<int> drop;
                    for (size_t i = 0; i < last_marg->keep_addr.size(); ++i) if (last_marg->keep_addr[i] == para_pose[kWin - 1]) drop.push_back((int)i);
                    mg.add(RBInfo{ std::make_shared<MargCost>(last_marg.get()), kNoLoss, last_marg->keep_addr, drop });
                }
                std::map<double*, double*> shift;
                for (int i = 0; i <= kWin; ++i) {
                    if (i == kWin - 1) continue;
                    else if (i == kWin) { shift[para_pose[i]] = para_pose[i - 1]; if (cfg.use_imu) shift[para_speed_bias[i]] = para_speed_bias[i - 1]; }
                    else { shift[para_pose[i]] = para_pose[i]; if (cfg.use_imu) shift[para_speed_bias[i]] = para_speed_bias[i]; }
                }
                for (int i = 0; i < 2; ++i) shift[para_ex_pose[i]] = para_ex_pose[i];
                shift[para_td[0]] = para_td[0];
                auto keep_alive = std::move(last_marg);      // MargCost above points into the old prior until run() returns
                last_marg = mg.run(shift);
            }
        }
    }
    void outliers_rejection(std::set<int>& remove) {        // vio_util.cpp:381-430
        for (auto& lm : lms) {
            if (lm.feats.size() < 4) continue;
            double err = 0; int cnt = 0; int imu_i = lm.start_frame, imu_j = imu_i - 1;
            auto reproj = [&](int j, int cam, const V3& uvj) {
                V3 pw = Rs[imu_i] * (ric[0] * (lm.feats[0].point * lm.depth) + tic[0]) + Ps[imu_i];
                V3 pc = ric[cam].t() * (Rs[j].t() * (pw - Ps[j]) - tic[cam]);
                double rx = pc.x / pc.z - uvj.x, ry = pc.y / pc.z - uvj.y;
                return std::sqrt(rx * rx + ry * ry);
            };
            for (auto& ft : lm.feats) {
                imu_j++;
                if (imu_i != imu_j) { err += reproj(imu_j, 0, ft.point); cnt++; }
                if (cfg.stereo && ft.is_stereo) { err += reproj(imu_j, 1, ft.point_right); cnt++; }
            }
            if (err / cnt * kFocalLength > 3) remove.insert(lm.feature_id);
        }
    }
    void slide_window() {        // estimator.cpp:1201-1312
        if (margin_old) {
            back_R0 = Rs[0]; back_P0 = Ps[0];
            if (frame == kWin) {
                for (int i = 0; i < kWin; ++i) {
                    headers[i] = headers[i + 1]; std::swap(Rs[i], Rs[i + 1]); std::swap(Ps[i], Ps[i + 1]);
                    if (cfg.use_imu) { std::swap(pre[i], pre[i + 1]); dt_buf[i].swap(dt_buf[i + 1]); la_buf[i].swap(la_buf[i + 1]); av_buf[i].swap(av_buf[i + 1]); std::swap(Vs[i], Vs[i + 1]); std::swap(Bas[i], Bas[i + 1]); std::swap(Bgs[i], Bgs[i + 1]); }
                }
                headers[kWin] = headers[kWin - 1]; Ps[kWin] = Ps[kWin - 1]; Rs[kWin] = Rs[kWin - 1];
                if (cfg.use_imu) {
                    Vs[kWin] = Vs[kWin - 1]; Bas[kWin] = Bas[kWin - 1]; Bgs[kWin] = Bgs[kWin - 1];
                    pre[kWin] = std::make_unique<Integration>(acc_0, gyr_0, Bas[kWin], Bgs[kWin], cfg.noise);
                    dt_buf[kWin].clear(); la_buf[kWin].clear(); av_buf[kWin].clear();
                }
                if (nonlinear) {       // SlideWindowOld
                    M3 R0 = back_R0 * ric[0], R1 = Rs[0] * ric[0];
                    V3 P0 = back_P0 + back_R0 * tic[0], P1 = Ps[0] + Rs[0] * tic[0];
                    remove_back_shift_depth(R0, P0, R1, P1); lines_remove_back_shift(R0, P0, R1, P1);
                } else { remove_back(); lines_remove_back(); }
            }
        } else if (frame == kWin) {
            headers[frame - 1] = headers[frame]; Ps[frame - 1] = Ps[frame]; Rs[frame - 1] = Rs[frame];
            if (cfg.use_imu) {
                for (size_t i = 0; i < dt_buf[frame].size(); ++i) {
                    pre[frame - 1]->push_back(dt_buf[frame][i], la_buf[frame][i], av_buf[frame][i]);
                    dt_buf[frame - 1].push_back(dt_buf[frame][i]); la_buf[frame - 1].push_back(la_buf[frame][i]); av_buf[frame - 1].push_back(av_buf[frame][i]);
                }
                Vs[frame - 1] = Vs[frame]; Bas[frame - 1] = Bas[frame]; Bgs[frame - 1] = Bgs[frame];
                pre[kWin] = std::make_unique<Integration>(acc_0, gyr_0, Bas[kWin], Bgs[kWin], cfg.noise);
                dt_buf[kWin].clear(); la_buf[kWin].clear(); av_buf[kWin].clear();
            }
            remove_front(frame); lines_remove_front(frame);
        }
    }
    void solve_gyro_bias() {     // initial_aligment.cpp:29-61
        M3 A; V3 b;
        for (size_t k = 0; k + 1 < all_frames.size(); ++k) {
            const Integration& pj = *all_frames[k + 1].second;
            Q q_ij = Q::fromR(frame_R[k].t() * frame_R[k + 1]);
            M3 tA = pj.jb(3, 12);
            V3 tb = (pj.delta_q.inverse() * q_ij).vec() * 2.0;
            A = A + tA.t() * tA; b = b + tA.t() * tb;
        }
        // A.ldlt().solve(b): 3x3 SPD solve
        Mat Am(3, 3); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Am(i, j) = A(i, j);
        Mat L; std::vector<double> rhs = { b.x, b.y, b.z };
        if (cholesky(Am, L)) chol_solve(L, rhs); else rhs = { 0, 0, 0 };
        V3 dbg(rhs[0], rhs[1], rhs[2]);
        for (int i = 0; i <= kWin; ++i) Bgs[i] += dbg;
        for (size_t k = 0; k + 1 < all_frames.size(); ++k) all_frames[k + 1].second->repropagate(V3(), Bgs[0]);
    }
    std::vector<M3> frame_R;
    void init_estimator() {      // InitEstimator (estimator.cpp:1424-1508), stereo paths
        if (cfg.stereo && cfg.use_imu) {
            init_frame_pose_by_pnp(frame);
            triangulate_points();
            if (frame == kWin) {
                frame_R.clear(); for (size_t i = 0; i < all_frames.size(); ++i) frame_R.push_back(Rs[i]);
                solve_gyro_bias();
                for (int j = 0; j <= kWin; ++j) pre[j]->repropagate(V3(), Bgs[j]);
                optimization();
                nonlinear = true;
                slide_window();
            }
        } else if (cfg.stereo && !cfg.use_imu) {
            init_frame_pose_by_pnp(frame);
            triangulate_points();
            optimization();
            if (frame == kWin) { optimization(); nonlinear = true; slide_window(); }
        }
        if (frame < kWin) {
            frame++;
            int p = frame - 1;
            Ps[frame] = Ps[p]; Vs[frame] = Vs[p]; Rs[frame] = Rs[p]; Bas[frame] = Bas[p]; Bgs[frame] = Bgs[p];
        }
    }
    void process_image(const dvo_feat* feats, int n, double header, std::map<unsigned, oim::FeatureInstance>* instances = nullptr) {      // ProcessImage (estimator.cpp:1516-1696)
        if (cfg.use_line) add_line_features(frame);
        margin_old = add_feature_check_parallax(frame, feats, n, td);
        headers[frame] = header;
        all_frames.push_back({ header, std::shared_ptr<Integration>(tmp_pre.release()) });
        tmp_pre = std::make_unique<Integration>(acc_0, gyr_0, Bas[frame], Bgs[frame], cfg.noise);
        if (!nonlinear) { init_estimator(); return; }
        if (!cfg.use_imu) init_frame_pose_by_pnp(frame);
        triangulate_points();
        if (cfg.use_line) triangulate_line_mono();
        const bool dyn = cfg.dynamic && instances;
        if (dyn) {               // estimator.cpp:1562-1622
            const oim::Body b = body();
            im.PushBack(b, (unsigned)frame, *instances);
            im.SetOutputInstInfo();          // para::is_static_inst_as_background (estimator.cpp:1583-1586; default true, vio_parameters.h:86)
            im.PropagatePose(b);
            im.Triangulate(b);
            im.InitialInstance(b);
            im.InitialInstanceVelocity(b);
            im.SetDynamicOrStatic(b);
            im.Optimization(b);
            im.OutliersRejection(b);
        }
        if (cfg.use_line) optimization_with_only_line();
        optimization();
        std::set<int> rm; outliers_rejection(rm);
        for (auto it = lms.begin(); it != lms.end();) { auto cur = it++; if (rm.count(cur->feature_id)) lms.erase(cur); }
        remove_line_outlier();
        if (dyn) { const oim::Body b = body(); im.ManageTriangulatePoint(b); im.SlideWindow(b, margin_old); }      // :1653-1658
        slide_window();
        if (dyn) {               // :1663-1676
            im.OutliersRejection(body());
            im.DeleteBadLandmarks();
            for (auto& kv : im.instances) if (kv.second.landmarks.empty() && kv.second.GetPointsExtraFrames() == 0) kv.second.ClearState();
        }
        for (auto it = lms.begin(); it != lms.end();) { auto cur = it++; if (cur->solve_flag == 2) lms.erase(cur); }      // RemoveFailures
        if (all_frames.size() > 2 * kWin + 2) all_frames.erase(all_frames.begin(), all_frames.end() - (kWin + 1));       // only used during initialisation
    }
    // one iteration of ProcessMeasurements (estimator.cpp:1786-1863); returns false if the IMU data does not yet cover t
    bool process(const dvo_feat* feats, int n, double t, std::map<unsigned, oim::FeatureInstance>* instances = nullptr) {
        cur_time = t + td;
        if (cfg.use_imu) {
            if (!imu_available(cur_time)) return false;
            std::vector<std::pair<double, V3>> av, gv;
            get_imu_interval(prev_time, cur_time, av, gv);
            add_imu(av, gv);
        }
        process_image(feats, n, t, instances);
        prev_time = cur_time;
        return true;
    }
};

}  // namespace obe

using namespace obe;

struct dvo_estimator { Estimator* e; };

extern "C" {

void dvo_proj_eval(int kind, const double* obs12, const double* const* par, double* res, double** J) {
    ProjObs o; o.pts_i = V3(obs12[0], obs12[1], obs12[2]); o.pts_j = V3(obs12[3], obs12[4], obs12[5]);
    o.vel_i = V3(obs12[6], obs12[7], 0); o.vel_j = V3(obs12[8], obs12[9], 0); o.td_i = obs12[10]; o.td_j = obs12[11];
    proj_eval(kind, o, par, res, J);
}

dvo_preint* dvo_preint_create(const double* acc0, const double* gyr0, const double* ba, const double* bg, const double* noise4) {
    ImuNoise n{ noise4[0], noise4[1], noise4[2], noise4[3] };
    return reinterpret_cast<dvo_preint*>(new Integration(P3(acc0), P3(gyr0), P3(ba), P3(bg), n));
}
void dvo_preint_destroy(dvo_preint* p) { delete reinterpret_cast<Integration*>(p); }
void dvo_preint_push(dvo_preint* p, double dt, const double* acc, const double* gyr) { reinterpret_cast<Integration*>(p)->push_back(dt, P3(acc), P3(gyr)); }
void dvo_preint_repropagate(dvo_preint* p, const double* ba, const double* bg) { reinterpret_cast<Integration*>(p)->repropagate(P3(ba), P3(bg)); }
void dvo_preint_get(const dvo_preint* p, double* sum_dt, double* dp, double* dq_xyzw, double* dv, double* jac225, double* cov225) {
    const Integration& I = *reinterpret_cast<const Integration*>(p);
    *sum_dt = I.sum_dt;
    for (int k = 0; k < 3; ++k) { dp[k] = I.delta_p[k]; dv[k] = I.delta_v[k]; }
    dq_xyzw[0] = I.delta_q.x; dq_xyzw[1] = I.delta_q.y; dq_xyzw[2] = I.delta_q.z; dq_xyzw[3] = I.delta_q.w;
    std::memcpy(jac225, I.jacobian.d.data(), 225 * sizeof(double)); std::memcpy(cov225, I.covariance.d.data(), 225 * sizeof(double));
}
void dvo_preint_set(dvo_preint* p, double sum_dt, const double* dp, const double* dq_xyzw, const double* dv, const double* jac225, const double* cov225) {
    Integration& I = *reinterpret_cast<Integration*>(p);
    I.sum_dt = sum_dt; I.delta_p = P3(dp); I.delta_v = P3(dv); I.delta_q = Q(dq_xyzw[3], dq_xyzw[0], dq_xyzw[1], dq_xyzw[2]);
    std::memcpy(I.jacobian.d.data(), jac225, 225 * sizeof(double)); std::memcpy(I.covariance.d.data(), cov225, 225 * sizeof(double));
}
void dvo_imu_eval(const dvo_preint* p, double g_norm, const double* const* par, double* res15, double** J) {
    imu_eval(*reinterpret_cast<const Integration*>(p), V3(0, 0, g_norm), par, res15, J);
}

// standalone window problem (same flat description the product's dv_ba_solve takes) -> Problem -> Solver
int dvo_ba_solve(dvo_ba_problem* P, dvo_ba_summary* S) {
    Problem prob;
    const int pose_kind = P->plane_kind == 1 ? kPosePlaneImu : (P->plane_kind == 2 ? kPosePlaneVo : kPose);
    for (int i = 0; i < P->nframes; ++i) {
        prob.AddParameterBlock(P->pose + 7 * i, 7, pose_kind);
        if (P->use_imu) prob.AddParameterBlock(P->speed_bias + 9 * i, 9);
    }
    if (!P->use_imu) prob.SetConstant(P->pose);
    for (int i = 0; i < 2; ++i) { prob.AddParameterBlock(P->ex_pose + 7 * i, 7, kPose); if (!(P->free_blocks & 1)) prob.SetConstant(P->ex_pose + 7 * i); }
    prob.AddParameterBlock(P->td, 1); if (!(P->free_blocks & 2)) prob.SetConstant(P->td);
    MargInfo mi;
    if (P->prior && P->prior->valid) {
        const int n = P->prior->n;
        Mat A(n, n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A(i, j) = 0.5 * (P->prior_A[i * n + j] + P->prior_A[j * n + i]);
        std::vector<double> ev; Mat V; sym_eig(A, ev, V);
        mi.m = 0; mi.n = n; mi.J0 = Mat(n, n); mi.r0.assign(n, 0.0);
        for (int k = 0; k < n; ++k) {
            const double Sv = ev[k] > 1e-8 ? ev[k] : 0.0, Si = ev[k] > 1e-8 ? 1.0 / ev[k] : 0.0;
            double vb = 0; for (int i = 0; i < n; ++i) { mi.J0(k, i) = std::sqrt(Sv) * V(i, k); vb += V(i, k) * P->prior_b[i]; }
            mi.r0[k] = std::sqrt(Si) * vb;
        }
        for (int b = 0; b < P->prior->nblocks; ++b) {
            const dvo_ba_prior_block& pb = P->prior->blocks[b];
            const int gs = pb.type == 0 || pb.type == 2 ? 7 : (pb.type == 1 ? 9 : 1);
            mi.keep_size.push_back(gs); mi.keep_idx.push_back(pb.off);
            mi.keep_data.push_back(std::vector<double>(P->prior->x0[b], P->prior->x0[b] + gs));
            double* addr = pb.type == 0 ? P->pose + 7 * pb.idx : pb.type == 1 ? P->speed_bias + 9 * pb.idx : pb.type == 2 ? P->ex_pose + 7 * pb.idx : P->td;
            mi.keep_addr.push_back(addr);
        }
        prob.AddResidualBlock(std::make_shared<MargCost>(&mi), kNoLoss, mi.keep_addr);
    }
    std::vector<std::unique_ptr<Integration>> pres;
    const V3 G(0, 0, P->g_norm);
    for (int k = 0; k < P->nimu; ++k) {
        const dvo_ba_imu& m = P->imu[k];
        auto I = std::make_unique<Integration>(V3(), V3(), P3(m.lin_ba), P3(m.lin_bg), ImuNoise{ 0, 0, 0, 0 });
        I->sum_dt = m.sum_dt; I->delta_p = P3(m.dp); I->delta_v = P3(m.dv); I->delta_q = Q(m.dq[0], m.dq[1], m.dq[2], m.dq[3]);
        std::memcpy(I->jacobian.d.data(), m.jacobian, 225 * 8); std::memcpy(I->covariance.d.data(), m.covariance, 225 * 8);
        prob.AddResidualBlock(std::make_shared<ImuCost>(I.get(), G), kNoLoss, { P->pose + 7 * m.fi, P->speed_bias + 9 * m.fi, P->pose + 7 * m.fj, P->speed_bias + 9 * m.fj });
        pres.push_back(std::move(I));
    }
    for (int l = 0; l < P->nlm; ++l) {
        prob.AddParameterBlock(P->inv_depth + l, 1, kPlain, true);
        const dvo_ba_lm& L = P->landmarks[l];
        for (int k = 0; k < L.count; ++k) {
            const dvo_ba_factor& f = P->factors[L.first + k];
            ProjObs o; o.pts_i = V3(f.pix, f.piy, 1); o.pts_j = V3(f.pjx, f.pjy, 1); o.vel_i = V3(f.vix, f.viy, 0); o.vel_j = V3(f.vjx, f.vjy, 0); o.td_i = f.td_i; o.td_j = f.td_j;
            if (f.kind == 0) prob.AddResidualBlock(std::make_shared<ProjCost>(0, o), kHuber1, { P->pose + 7 * f.fi, P->pose + 7 * f.fj, P->ex_pose, P->inv_depth + l, P->td });
            else if (f.kind == 1) prob.AddResidualBlock(std::make_shared<ProjCost>(1, o), kHuber1, { P->pose + 7 * f.fi, P->pose + 7 * f.fj, P->ex_pose, P->ex_pose + 7, P->inv_depth + l, P->td });
            else prob.AddResidualBlock(std::make_shared<ProjCost>(2, o), kHuber1, { P->ex_pose, P->ex_pose + 7, P->inv_depth + l, P->td });
        }
    }
    Solver solver(prob);
    SolveOptions so; so.max_num_iterations = P->max_iters; so.xnorm2_extra = P->x_norm2_extra;
    SolveSummary sum = solver.solve(so);
    if (S) { S->iterations = sum.iterations; S->successful = sum.successful; S->termination = sum.termination; S->slots = 0; S->initial_cost = sum.initial_cost; S->final_cost = sum.final_cost; }
    return 0;
}

// SetMarginalizationInfo on a flat window description: mode 0 = kMarginOld (factors = residual blocks of the
// landmarks anchored in frame 0, imu[0] = factor (0,1), prior), mode 1 = kMarginSecondNew (prior only).
// Output in information form (A = J0^T J0, b = J0^T r0, c0 = r0^T r0) with the oracle's own block order.
int dvo_marginalize(const dvo_ba_problem* P, int mode, dvo_ba_prior* out, double* out_A, double* out_b) {
    std::memset(out, 0, sizeof(*out));
    MargInfo old;
    const bool has_prior = P->prior && P->prior->valid;
    auto addr_of = [&](int type, int idx) -> double* { return type == 0 ? P->pose + 7 * idx : type == 1 ? P->speed_bias + 9 * idx : type == 2 ? P->ex_pose + 7 * idx : P->td; };
    if (has_prior) {
        const int n = P->prior->n;
        Mat A(n, n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A(i, j) = 0.5 * (P->prior_A[i * n + j] + P->prior_A[j * n + i]);
        std::vector<double> ev; Mat V; sym_eig(A, ev, V);
        old.m = 0; old.n = n; old.J0 = Mat(n, n); old.r0.assign(n, 0.0);
        for (int k = 0; k < n; ++k) {
            const double Sv = ev[k] > 1e-8 ? ev[k] : 0.0, Si = ev[k] > 1e-8 ? 1.0 / ev[k] : 0.0;
            double vb = 0; for (int i = 0; i < n; ++i) { old.J0(k, i) = std::sqrt(Sv) * V(i, k); vb += V(i, k) * P->prior_b[i]; }
            old.r0[k] = std::sqrt(Si) * vb;
        }
        for (int b = 0; b < P->prior->nblocks; ++b) {
            const dvo_ba_prior_block& pb = P->prior->blocks[b];
            const int gs = pb.type == 0 || pb.type == 2 ? 7 : (pb.type == 1 ? 9 : 1);
            old.keep_size.push_back(gs); old.keep_idx.push_back(pb.off);
            old.keep_data.push_back(std::vector<double>(P->prior->x0[b], P->prior->x0[b] + gs));
            old.keep_addr.push_back(addr_of(pb.type, pb.idx));
        }
    }
    Marginalizer mg;
    std::map<double*, double*> shift;
    std::vector<std::unique_ptr<Integration>> pres;
    const V3 G(0, 0, P->g_norm);
    double* drop_pose = P->pose + 7 * (mode == 0 ? 0 : kWin - 1);
    if (has_prior) {
        std::vector<int> drop;
        for (size_t i = 0; i < old.keep_addr.size(); ++i) if (old.keep_addr[i] == drop_pose || (mode == 0 && old.keep_addr[i] == P->speed_bias)) drop.push_back((int)i);
        mg.add(RBInfo{ std::make_shared<MargCost>(&old), kNoLoss, old.keep_addr, drop });
    }
    if (mode == 0) {
        if (P->nimu > 0) {
            const dvo_ba_imu& m = P->imu[0];
            auto I = std::make_unique<Integration>(V3(), V3(), P3(m.lin_ba), P3(m.lin_bg), ImuNoise{ 0, 0, 0, 0 });
            I->sum_dt = m.sum_dt; I->delta_p = P3(m.dp); I->delta_v = P3(m.dv); I->delta_q = Q(m.dq[0], m.dq[1], m.dq[2], m.dq[3]);
            std::memcpy(I->jacobian.d.data(), m.jacobian, 225 * 8); std::memcpy(I->covariance.d.data(), m.covariance, 225 * 8);
            mg.add(RBInfo{ std::make_shared<ImuCost>(I.get(), G), kNoLoss, { P->pose, P->speed_bias, P->pose + 7, P->speed_bias + 9 }, { 0, 1 } });
            pres.push_back(std::move(I));
        }
        for (int l = 0; l < P->nlm; ++l) {
            const dvo_ba_lm& L = P->landmarks[l];
            for (int k = 0; k < L.count; ++k) {
                const dvo_ba_factor& f = P->factors[L.first + k];
                ProjObs o; o.pts_i = V3(f.pix, f.piy, 1); o.pts_j = V3(f.pjx, f.pjy, 1); o.vel_i = V3(f.vix, f.viy, 0); o.vel_j = V3(f.vjx, f.vjy, 0); o.td_i = f.td_i; o.td_j = f.td_j;
                double* lam = P->inv_depth + f.lm;
                if (f.kind == 0) mg.add(RBInfo{ std::make_shared<ProjCost>(0, o), kHuber1, { P->pose + 7 * f.fi, P->pose + 7 * f.fj, P->ex_pose, lam, P->td }, { 0, 3 } });
                else if (f.kind == 1) mg.add(RBInfo{ std::make_shared<ProjCost>(1, o), kHuber1, { P->pose + 7 * f.fi, P->pose + 7 * f.fj, P->ex_pose, P->ex_pose + 7, lam, P->td }, { 0, 4 } });
                else mg.add(RBInfo{ std::make_shared<ProjCost>(2, o), kHuber1, { P->ex_pose, P->ex_pose + 7, lam, P->td }, { 2 } });
            }
        }
        for (int i = 1; i <= kWin; ++i) { shift[P->pose + 7 * i] = P->pose + 7 * (i - 1); if (P->use_imu) shift[P->speed_bias + 9 * i] = P->speed_bias + 9 * (i - 1); }
    } else {
        for (int i = 0; i <= kWin; ++i) {
            if (i == kWin - 1) continue;
            const int t = i == kWin ? i - 1 : i;
            shift[P->pose + 7 * i] = P->pose + 7 * t; if (P->use_imu) shift[P->speed_bias + 9 * i] = P->speed_bias + 9 * t;
        }
    }
    for (int i = 0; i < 2; ++i) shift[P->ex_pose + 7 * i] = P->ex_pose + 7 * i;
    shift[P->td] = P->td;
    auto res = mg.run(shift);
    if (!res->valid) { out->valid = 0; return 0; }
    const int n = res->n;
    out->valid = 1; out->n = n; out->nblocks = (int)res->keep_size.size();
    double c0 = 0;
    for (int k = 0; k < n; ++k) c0 += res->r0[k] * res->r0[k];
    out->c0 = c0;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += res->J0(k, i) * res->J0(k, j); out_A[i * n + j] = s; }
        double s = 0; for (int k = 0; k < n; ++k) s += res->J0(k, i) * res->r0[k]; out_b[i] = s;
    }
    for (size_t b = 0; b < res->keep_size.size(); ++b) {
        double* a = res->keep_addr[b];
        dvo_ba_prior_block& pb = out->blocks[b];
        if (a >= P->pose && a < P->pose + 7 * (kWin + 1)) { pb.type = 0; pb.idx = (int)((a - P->pose) / 7); }
        else if (P->speed_bias && a >= P->speed_bias && a < P->speed_bias + 9 * (kWin + 1)) { pb.type = 1; pb.idx = (int)((a - P->speed_bias) / 9); }
        else if (a >= P->ex_pose && a < P->ex_pose + 14) { pb.type = 2; pb.idx = (int)((a - P->ex_pose) / 7); }
        else { pb.type = 3; pb.idx = 0; }
        pb.off = res->keep_idx[b] - res->m; pb.size_local = res->keep_size[b] == 7 ? 6 : res->keep_size[b];
        for (size_t k = 0; k < res->keep_data[b].size(); ++k) out->x0[b][k] = res->keep_data[b][k];
    }
    return 0;
}

// c0 = r0^T r0 of the eigen-clamped prior built from (A, b): sum_{lambda_k > 1e-8} (v_k . b)^2 / lambda_k
double dvo_prior_c0(const double* A_, const double* b, int n) {
    Mat A(n, n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A(i, j) = 0.5 * (A_[i * n + j] + A_[j * n + i]);
    std::vector<double> ev; Mat V; sym_eig(A, ev, V);
    double c0 = 0;
    for (int k = 0; k < n; ++k) if (ev[k] > 1e-8) { double vb = 0; for (int i = 0; i < n; ++i) vb += V(i, k) * b[i]; c0 += vb * vb / ev[k]; }
    return c0;
}

dvo_estimator* dvo_estimator_create(const dvo_be_config* c) {
    Config cfg;
    cfg.use_imu = c->use_imu; cfg.stereo = c->stereo; cfg.plane_constraint = c->plane_constraint; cfg.max_iters = c->max_iters;
    cfg.min_parallax = c->keyframe_parallax / kFocalLength; cfg.init_depth = c->init_depth; cfg.g_norm = c->g_norm; cfg.td = c->td;
    cfg.noise = ImuNoise{ c->acc_n, c->gyr_n, c->acc_w, c->gyr_w };
    for (int k = 0; k < 2; ++k) { for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) cfg.ric[k](i, j) = c->ric[k][i * 3 + j]; cfg.tic[k][i] = c->tic[k][i]; } }
    cfg.dynamic = c->dynamic; cfg.use_det3d = c->use_det3d; cfg.instance_init_min_num = c->instance_init_min_num; cfg.static_inst_threshold = c->static_inst_threshold; cfg.estimate = c->estimate;
    cfg.use_line = c->use_line; cfg.line_min_obs = c->line_min_obs; for (int k = 0; k < 4; ++k) cfg.line_sqrt_info[k] = c->line_sqrt_info[k];
    return new dvo_estimator{ new Estimator(cfg) };
}
/* FitBox3DWithRANSAC / FitBox3DFromCameraFrame (vio_util.cpp:209-332) on a flat point list: exported so that the product's host-side restatement
 * (csrc/inst_host.h) can be unit-tested on the CPU against this one */
int dvo_fit_box_ransac(const double* pts, int n, const double* dims3, unsigned long long seed, double* out3) {
    std::vector<V3> p(n); for (int i = 0; i < n; ++i) p[i] = V3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    V3 c; const bool ok = oim::FitBox3DWithRANSAC(p, V3(dims3[0], dims3[1], dims3[2]), seed, c);
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z; return ok ? 1 : 0;
}
int dvo_fit_box_camera(const double* pts, int n, const double* dims3, double* out3) {
    std::vector<V3> p(n); for (int i = 0; i < n; ++i) p[i] = V3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    V3 c; const bool ok = oim::FitBox3DFromCameraFrame(p, V3(dims3[0], dims3[1], dims3[2]), c);
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z; return ok ? 1 : 0;
}
int dvo_estimator_set_lines(dvo_estimator* e, const dvo_line_row* lines, int n) { e->e->pending_lines.assign(lines, lines + n); return 0; }
int dvo_estimator_get_lines(dvo_estimator* e, dvo_line_landmark* out, int cap, int* n_out) {
    int k = 0;
    for (auto& l : e->e->line_landmarks) {
        if (k >= cap) break;
        dvo_line_landmark& o = out[k++]; o.id = l.feature_id; o.start_frame = l.start_frame; o.n_obs = (int)l.feats.size(); o.is_triangulation = l.is_triangulation;
        std::memcpy(o.plucker, l.line_plucker, 48); for (int c = 0; c < 3; ++c) { o.ptw1[c] = l.ptw1[c]; o.ptw2[c] = l.ptw2[c]; }
    }
    *n_out = k; return 0;
}
void dvo_estimator_get_extrinsics(dvo_estimator* e, double* ric18, double* tic6, double* td) {
    for (int c = 0; c < 2; ++c) { for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) ric18[9 * c + 3 * i + j] = e->e->ric[c](i, j); tic6[3 * c + i] = e->e->tic[c][i]; } }
    *td = e->e->td;
}
void dvo_estimator_destroy(dvo_estimator* e) { if (e) { delete e->e; delete e; } }
void dvo_estimator_input_imu(dvo_estimator* e, double t, const double* acc, const double* gyr) { e->e->input_imu(t, P3(acc), P3(gyr)); }
static void fill_state(Estimator& E, dvo_be_state* out);
int dvo_estimator_process(dvo_estimator* e, const dvo_feat* feats, int n, double t, dvo_be_state* out) {
    Estimator& E = *e->e;
    if (!E.process(feats, n, t)) return 1;
    fill_state(E, out);
    return 0;
}
// FrontendFeature::instances from the flat arrays (what InstsFeatManager::Output() builds, front_end/dynamic_tracker.cpp:521-577)
int dvo_estimator_process_dynamic(dvo_estimator* e, const dvo_feat* feats, int n, double t, const dvo_inst_obs* insts, int n_insts, const dvo_feat* inst_feats,
                                  const double* points, dvo_be_state* out) {
    Estimator& E = *e->e;
    std::map<unsigned, oim::FeatureInstance> in;
    for (int i = 0; i < n_insts; ++i) {
        const dvo_inst_obs& io = insts[i];
        oim::FeatureInstance fi;
        if (io.has_box3d) {
            fi.box3d = std::make_shared<oim::Box3D>();
            for (int k = 0; k < 3; ++k) { fi.box3d->dims[k] = io.box3d.dims[k]; fi.box3d->center_pt[k] = io.box3d.center[k]; }
            fi.box3d->yaw = io.box3d.yaw;
        }
        for (int k = 0; k < io.n_feats; ++k) {
            const dvo_feat& f = inst_feats[io.first_feat + k];
            auto fp = std::make_shared<oim::FeaturePoint>();
            fp->point = V3(f.left[0], f.left[1], 1); fp->vel[0] = f.left[5]; fp->vel[1] = f.left[6];
            if (f.has_right) { fp->is_stereo = true; fp->point_right = V3(f.right[0], f.right[1], 1); fp->vel_right[0] = f.right[5]; fp->vel_right[1] = f.right[6]; }
            fi.features.insert({ f.id, fp });
        }
        for (int k = 0; k < io.n_points; ++k) { const double* p = points + 3 * (size_t)(io.first_point + k); fi.points.emplace_back(p[0], p[1], p[2]); }
        in.insert({ io.id, fi });
    }
    if (!E.process(feats, n, t, &in)) return 1;
    fill_state(E, out);
    return 0;
}
int dvo_estimator_get_static_instances(dvo_estimator* e, uint32_t* ids, int cap, int* n_out) {          // InstanceManager::GetOutputInstInfo, the is_static ids (system/main.cpp:194,217-245)
    const std::vector<uint32_t>& v = e->e->im.insts_output_static;
    const int n = std::min((int)v.size(), cap);
    for (int i = 0; i < n; ++i) ids[i] = v[i];
    *n_out = n;
    return 0;
}
int dvo_estimator_get_instances(dvo_estimator* e, dvo_inst_state* out, int cap, int* n_out, double* summary4) {
    Estimator& E = *e->e;
    int k = 0;
    for (auto& kv : E.im.instances) {
        if (k >= cap) break;
        oim::Instance& I = kv.second; dvo_inst_state& o = out[k++];
        std::memset(&o, 0, sizeof(o));
        o.id = I.id; o.is_initial = I.is_initial; o.is_tracking = I.is_tracking; o.is_curr_visible = I.is_curr_visible; o.is_static = I.is_static; o.is_init_velocity = I.is_init_velocity;
        o.age = I.age; o.lost_number = I.lost_number; o.static_frame = I.static_frame; o.n_landmarks = (int)I.landmarks.size(); o.n_valid = I.valid_size(); o.triangle_num = I.triangle_num;
        for (int c = 0; c < 3; ++c) { o.dims[c] = I.box3d ? I.box3d->dims[c] : 0.0; o.vel_v[c] = I.vel.v[c]; o.vel_a[c] = I.vel.a[c]; }
        for (int i = 0; i <= kWin; ++i) { const Q q = Q::fromR(I.state[i].R); double* p = o.window[i]; p[0] = I.state[i].P.x; p[1] = I.state[i].P.y; p[2] = I.state[i].P.z; p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w; o.time[i] = I.state[i].time; }
    }
    *n_out = k;
    if (summary4) { summary4[0] = E.im.last_summary.iterations; summary4[1] = E.im.last_summary.termination; summary4[2] = E.im.last_summary.initial_cost; summary4[3] = E.im.last_summary.final_cost; }
    return 0;
}
static void fill_state(Estimator& E, dvo_be_state* out) {
    if (out) {
        std::memset(out, 0, sizeof(*out));
        out->frame = E.frame; out->nonlinear = E.nonlinear; out->margin_old = E.margin_old; out->n_landmarks = (int)E.lms.size(); out->n_long = E.feature_count();
        out->iterations = E.last_summary.iterations; out->initial_cost = E.last_summary.initial_cost; out->final_cost = E.last_summary.final_cost;
        for (int i = 0; i <= kWin; ++i) {
            Q q = Q::fromR(E.Rs[i]);
            double* p = out->window[i];
            p[0] = E.Ps[i].x; p[1] = E.Ps[i].y; p[2] = E.Ps[i].z; p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
            for (int k = 0; k < 3; ++k) { p[7 + k] = E.Vs[i][k]; p[10 + k] = E.Bas[i][k]; p[13 + k] = E.Bgs[i][k]; }
        }
    }
}

}  // extern "C"
