// est_host.hip — host side of the back end (product code): the bookkeeping of dynamic_vins' Estimator around the
// device kernels.  Mirrors, with flat arrays instead of std::list / Eigen / Ceres objects:
//   Estimator::{InputIMU,ProcessMeasurements (one iteration),ProcessImage,InitEstimator,Optimization,
//               SetMarginalizationInfo,SlideWindow,InitFramePoseByPnP,ClearState}   estimator/estimator.cpp
//   FeatureManager::{AddFeatureCheckParallax,TriangulatePoints,RemoveBack*,RemoveFront,RemoveOutlier,RemoveFailures}
//                                                                                       estimator/feature_manager.cpp
//   IntegrationBase (midpoint pre-integration)                                          estimator/imu/integration_base.h
//   OutliersRejection / TriangulatePoint / SolvePoseByPnP / SolveGyroscopeBias          estimator/vio_util.cpp, initial/
// What runs where: IMU pre-integration, triangulation, PnP seeding and the window shuffling are O(100) flops per
// item and stay on the host; every residual/Jacobian evaluation, the Schur elimination, the trust-region loop and the
// marginalization run on the GPU through dv_ba_solve / dv_marginalize.
// Deviations from the reference that change no result: feature lookup by hash map instead of the O(N*L) find_if
// (feature_manager.cpp:80-83); IMUFactor's sqrt-information cached per solve (Q8).
#include <algorithm>
#include <deque>
#include <memory>
#include <unordered_map>
#include "dv_ctx.h"
#include "be_math.h"
#include "inst_host.h"
#include "line_host.h"

using namespace be;

namespace {

constexpr int kWin = BE_WIN;
constexpr double kFocal = 460.0;

struct Preint {       // IntegrationBase
    d3 acc_0, gyr_0, lin_acc, lin_gyr, lin_ba, lin_bg, dp, dv; quat dq;
    double J[225], P[225], noise[18], sum_dt = 0;
    mutable double U[225]; mutable bool U_ok = false;      // cached sqrt-information of P (IMUFactor recomputes it on every Evaluate, Q8)
    std::vector<double> dts; std::vector<d3> accs, gyrs;
    Preint(d3 a0, d3 g0, d3 ba, d3 bg, const double n4[4]) : acc_0(a0), gyr_0(g0), lin_acc(a0), lin_gyr(g0), lin_ba(ba), lin_bg(bg) {
        reset_state();
        const double v[6] = { n4[0] * n4[0], n4[1] * n4[1], n4[0] * n4[0], n4[1] * n4[1], n4[2] * n4[2], n4[3] * n4[3] };
        for (int b = 0; b < 6; ++b) for (int k = 0; k < 3; ++k) noise[3 * b + k] = v[b];
    }
    void reset_state() { dp = mk3(0, 0, 0); dv = mk3(0, 0, 0); dq = mkq(1, 0, 0, 0); sum_dt = 0; for (int i = 0; i < 225; ++i) { J[i] = (i % 16 == 0) ? 1.0 : 0.0; P[i] = 0.0; } }
    void push_back(double dt, d3 a, d3 g) { dts.push_back(dt); accs.push_back(a); gyrs.push_back(g); propagate(dt, a, g); }
    void repropagate(d3 ba, d3 bg) {
        acc_0 = lin_acc; gyr_0 = lin_gyr; lin_ba = ba; lin_bg = bg; reset_state();
        for (size_t i = 0; i < dts.size(); ++i) propagate(dts[i], accs[i], gyrs[i]);
    }
    static void blk(double* M, int ld, int r, int c, const m33& b) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[(r + i) * ld + c + j] = b.m[i * 3 + j]; }
    void propagate(double dt, d3 a1, d3 g1) {      // midPointIntegration + propagate (integration_base.h:70-173)
        const d3 un_acc_0 = qrot(dq, acc_0 - lin_ba);
        const d3 un_gyr = (gyr_0 + g1) * 0.5 - lin_bg;
        const quat rq = qmul(dq, mkq(1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2));
        const d3 un_acc_1 = qrot(rq, a1 - lin_ba);
        const d3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
        const d3 rp = dp + dv * dt + un_acc * (0.5 * dt * dt), rv = dv + un_acc * dt;
        const m33 Rw = skew(un_gyr), Ra0 = skew(acc_0 - lin_ba), Ra1 = skew(a1 - lin_ba), I = eye3(), Rq = qR(dq), Rr = qR(rq);
        const m33 IRw = sub(I, scale(Rw, dt));
        static thread_local double F[225], V[270], T1[225], T2[270];
        std::fill(F, F + 225, 0.0); std::fill(V, V + 270, 0.0);
        blk(F, 15, 0, 0, I);
        blk(F, 15, 0, 3, add(scale(mul(Rq, Ra0), -0.25 * dt * dt), scale(mul(mul(Rr, Ra1), IRw), -0.25 * dt * dt)));
        blk(F, 15, 0, 6, scale(I, dt));
        blk(F, 15, 0, 9, scale(add(Rq, Rr), -0.25 * dt * dt));
        blk(F, 15, 0, 12, scale(mul(Rr, Ra1), -0.25 * dt * dt * -dt));
        blk(F, 15, 3, 3, IRw); blk(F, 15, 3, 12, scale(I, -dt));
        blk(F, 15, 6, 3, add(scale(mul(Rq, Ra0), -0.5 * dt), scale(mul(mul(Rr, Ra1), IRw), -0.5 * dt)));
        blk(F, 15, 6, 6, I); blk(F, 15, 6, 9, scale(add(Rq, Rr), -0.5 * dt)); blk(F, 15, 6, 12, scale(mul(Rr, Ra1), -0.5 * dt * -dt));
        blk(F, 15, 9, 9, I); blk(F, 15, 12, 12, I);
        const m33 V03 = scale(mul(scale(Rr, -1.0), Ra1), 0.25 * dt * dt * 0.5 * dt), V63 = scale(mul(scale(Rr, -1.0), Ra1), 0.5 * dt * 0.5 * dt);
        blk(V, 18, 0, 0, scale(Rq, 0.25 * dt * dt)); blk(V, 18, 0, 3, V03); blk(V, 18, 0, 6, scale(Rr, 0.25 * dt * dt)); blk(V, 18, 0, 9, V03);
        blk(V, 18, 3, 3, scale(I, 0.5 * dt)); blk(V, 18, 3, 9, scale(I, 0.5 * dt));
        blk(V, 18, 6, 0, scale(Rq, 0.5 * dt)); blk(V, 18, 6, 3, V63); blk(V, 18, 6, 6, scale(Rr, 0.5 * dt)); blk(V, 18, 6, 9, V63);
        blk(V, 18, 9, 12, scale(I, dt)); blk(V, 18, 12, 15, scale(I, dt));
        // J <- F J, P <- F P F^T + V diag(noise) V^T over the STRUCTURAL non-zeros of F and V only (81 of 225 and 78 of 270 entries: identity / zero blocks).
        // The skipped products are exact zeros and the kept ones are added in the same (ascending k) order as the dense loops of
        // integration_base.h, so the sums are the same bits; a third of the multiply-adds (44 -> ~16 us per frame at 10 samples).
        static const struct Pattern {
            int fn[15], fk[15][11], vn[15], vk[15][12];
            Pattern() {
                for (int i = 0; i < 15; ++i) {
                    const int b = i / 3, r = i % 3; int n = 0;
                    auto blk3 = [&](int c0) { fk[i][n++] = c0; fk[i][n++] = c0 + 1; fk[i][n++] = c0 + 2; };
                    if (b == 0) { fk[i][n++] = r; blk3(3); fk[i][n++] = 6 + r; blk3(9); blk3(12); }
                    else if (b == 1) { blk3(3); fk[i][n++] = 12 + r; }
                    else if (b == 2) { blk3(3); fk[i][n++] = 6 + r; blk3(9); blk3(12); }
                    else fk[i][n++] = i;
                    fn[i] = n;
                    int m = 0;
                    if (b == 0 || b == 2) for (int k = 0; k < 12; ++k) vk[i][m++] = k;
                    else if (b == 1) { vk[i][m++] = 3 + r; vk[i][m++] = 9 + r; }
                    else if (b == 3) vk[i][m++] = 12 + r;
                    else vk[i][m++] = 15 + r;
                    vn[i] = m;
                }
            }
        } pat;
        // axpy form (row of the result += scalar * row of the right operand): every entry still receives its products in ascending k, one rounding each,
        // but the inner loop runs over 15 independent entries and vectorises
        auto fmul = [&](const double* B, double* C) {          // C = F B
            for (int i = 0; i < 15; ++i) {
                double acc[15] = { 0 };
                for (int q = 0; q < pat.fn[i]; ++q) { const int k = pat.fk[i][q]; const double f = F[i * 15 + k]; const double* b = B + k * 15; for (int j = 0; j < 15; ++j) acc[j] += f * b[j]; }
                for (int j = 0; j < 15; ++j) C[i * 15 + j] = acc[j];
            }
        };
        fmul(J, T1); std::copy(T1, T1 + 225, J);
        fmul(P, T1);                                          // F P
        double Ft[225], Vt[270], FPFt[225];
        for (int i = 0; i < 15; ++i) for (int k = 0; k < 15; ++k) Ft[k * 15 + i] = F[i * 15 + k];
        for (int i = 0; i < 15; ++i) for (int k = 0; k < 18; ++k) Vt[k * 15 + i] = V[i * 18 + k];
        for (int i = 0; i < 15; ++i) {                        // (F P) F^T: the structural zeros of F^T contribute exact zeros
            double acc[15] = { 0 };
            for (int k = 0; k < 15; ++k) { const double t = T1[i * 15 + k]; const double* b = Ft + k * 15; for (int j = 0; j < 15; ++j) acc[j] += t * b[j]; }
            for (int j = 0; j < 15; ++j) FPFt[i * 15 + j] = acc[j];
        }
        for (int i = 0; i < 15; ++i) {                        // + V diag(noise) V^T
            double acc[15] = { 0 };
            for (int q = 0; q < pat.vn[i]; ++q) { const int k = pat.vk[i][q]; const double t = V[i * 18 + k] * noise[k]; const double* b = Vt + k * 15; for (int j = 0; j < 15; ++j) acc[j] += t * b[j]; }
            for (int j = 0; j < 15; ++j) P[i * 15 + j] = FPFt[i * 15 + j] + acc[j];
        }
        dp = rp; dq = qnormalized(rq); dv = rv; sum_dt += dt; acc_0 = a1; gyr_0 = g1; U_ok = false;
    }
    const double* sqrt_info() const { if (!U_ok) U_ok = be_imu_sqrt_info(P, U); return U_ok ? U : nullptr; }
    m33 jb(int r, int c) const { m33 b; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) b.m[i * 3 + j] = J[(r + i) * 15 + c + j]; return b; }
    void fill(dv_ba_imu& o, int fi, int fj) const {
        o.sum_dt = sum_dt;
        o.dp[0] = dp.x; o.dp[1] = dp.y; o.dp[2] = dp.z; o.dv[0] = dv.x; o.dv[1] = dv.y; o.dv[2] = dv.z;
        o.dq[0] = dq.w; o.dq[1] = dq.x; o.dq[2] = dq.y; o.dq[3] = dq.z;
        o.lin_ba[0] = lin_ba.x; o.lin_ba[1] = lin_ba.y; o.lin_ba[2] = lin_ba.z; o.lin_bg[0] = lin_bg.x; o.lin_bg[1] = lin_bg.y; o.lin_bg[2] = lin_bg.z;
        std::memcpy(o.jacobian, J, sizeof(J)); std::memcpy(o.covariance, P, sizeof(P));
        o.fi = fi; o.fj = fj; o.pad0 = o.pad1 = 0;
    }
};

struct Obs { d3 pt, pt_r, vel, vel_r; double td; bool stereo; };
struct Lm { int id, start; std::vector<Obs> obs; double depth = -1.0; int solve_flag = 0; int end() const { return start + (int)obs.size() - 1; } };

// symmetric 4x4 eigen (cyclic Jacobi) -> eigenvector of the smallest eigenvalue
void smallest_eigvec4(double A[4][4], double v[4]) {
    double V[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0; for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; ++p) for (int q = p + 1; q < 4; ++q) {
            if (std::fabs(A[p][q]) < 1e-300) continue;
            const double th = (A[q][q] - A[p][p]) / (2 * A[p][q]);
            const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1)), c = 1 / std::sqrt(t * t + 1), s = t * c;
            for (int k = 0; k < 4; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s * b; A[k][q] = s * a + c * b; }
            for (int k = 0; k < 4; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s * b; A[q][k] = s * a + c * b; }
            for (int k = 0; k < 4; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
        }
    }
    int m = 0; for (int i = 1; i < 4; ++i) if (A[i][i] < A[m][m]) m = i;
    for (int i = 0; i < 4; ++i) v[i] = V[i][m];
}

quat from_two_vectors(d3 a, d3 b) {
    const d3 v0 = a / norm(a), v1 = b / norm(b);
    const double c = dot(v1, v0);
    if (c < -1.0 + 1e-12) { d3 ax = std::fabs(v0.x) < 0.9 ? cross(mk3(1, 0, 0), v0) : cross(mk3(0, 1, 0), v0); ax = ax / norm(ax); return mkq(0, ax.x, ax.y, ax.z); }
    const d3 ax = cross(v0, v1); const double s = sqrt((1.0 + c) * 2.0), inv = 1.0 / s;
    return mkq(s * 0.5, ax.x * inv, ax.y * inv, ax.z * inv);
}

}  // namespace

struct dv_estimator {
    dv_est_config cfg;
    m33 ric[2]; d3 tic[2]; d3 Ps[kWin + 1], Vs[kWin + 1], Bas[kWin + 1], Bgs[kWin + 1]; m33 Rs[kWin + 1];
    d3 g; double td = 0, headers[kWin + 1] = { 0 };
    int frame = 0;
    std::vector<Lm> lms; std::unordered_map<int, size_t> lm_index; bool index_dirty = true;
    std::deque<std::pair<double, std::pair<d3, d3>>> imu_buf;
    double prev_time = -1, cur_time = 0; bool first_imu = false, init_pose = false; d3 acc_0, gyr_0;
    std::unique_ptr<Preint> pre[kWin + 1], tmp_pre;
    std::vector<double> dt_buf[kWin + 1]; std::vector<d3> la_buf[kWin + 1], av_buf[kWin + 1];
    std::vector<std::shared_ptr<Preint>> frame_pre;        // all_image_frame pre-integrations (initialisation only)
    bool nonlinear = false, margin_old = true;
    bool open_ex = false;          // Estimator::openExEstimation (estimator.h:173): sticky until ClearState (estimator.cpp:632)
    dv_ba_prior prior{}; const double* prior_dev_A = nullptr; const double* prior_dev_b = nullptr;      // header on the host, A' / b' device-resident
    m33 back_R0; d3 back_P0;
    dv_ba_summary last{};
    // dynamic mode (cfg.dynamic): Estimator::im + what body.para_pose holds when InstanceManager::Optimization reads it (estimator_insts.cpp:1043-1045):
    // the values the PREVIOUS frame's Optimization left there — ceres' raw output, or the gauge-fixed states where SetMarginalizationInfo called
    // Vector2double again (estimator.cpp:409,562)
    dvi::InstMgr im; double para_pose_ref[kWin + 1][7]; dv_ba_summary obj_last{}; bool dyn_frame = false;
    // line mode (cfg.use_line): FeatureManager::line_landmarks + body.para_line_features as Vector2double fills them
    dvl::LineMgr lines; std::vector<dv_line_row> pending_lines; std::vector<double> para_line; dv_ba_summary line_last{};
    // Estimator::latest_* (estimator.h): the state of the newest frame propagated by every IMU sample that arrived since (FastPredictIMU, estimator.cpp:1376-1392)
    double latest_time = 0; d3 latest_P, latest_V, latest_Ba, latest_Bg, latest_acc_0, latest_gyr_0; quat latest_Q; bool latest_valid = false;
    void fast_predict_imu(double t, d3 la, d3 av) {
        const double dt = t - latest_time; latest_time = t;
        const d3 un_acc_0 = qrot(latest_Q, latest_acc_0 - latest_Ba) - g;
        const d3 un_gyr = (latest_gyr_0 + av) * 0.5 - latest_Bg;
        latest_Q = qmul(latest_Q, dq_half(un_gyr * dt));
        const d3 un_acc_1 = qrot(latest_Q, la - latest_Ba) - g;
        const d3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
        latest_P = latest_P + latest_V * dt + un_acc * (0.5 * dt * dt);
        latest_V = latest_V + un_acc * dt;
        latest_acc_0 = la; latest_gyr_0 = av;
    }
    void update_latest_states() {          // UpdateLatestStates (estimator.cpp:1395-1418)
        latest_time = headers[frame] + td; latest_P = Ps[frame]; latest_Q = qfromR(Rs[frame]); latest_V = Vs[frame]; latest_Ba = Bas[frame]; latest_Bg = Bgs[frame];
        latest_acc_0 = acc_0; latest_gyr_0 = gyr_0; latest_valid = true;
        for (auto& smp : imu_buf) fast_predict_imu(smp.first, smp.second.first, smp.second.second);
    }
    // flat problem buffers
    std::vector<dv_ba_factor> fac; std::vector<dv_ba_lm> lmt; std::vector<dv_ba_imu> imu; std::vector<double> invd;
    double pose[kWin + 1][7], sb[kWin + 1][9], ex[2][7], tdv[1];

    explicit dv_estimator(const dv_est_config& c) : cfg(c) { clear(); }
    void clear() {
        for (int i = 0; i <= kWin; ++i) { Rs[i] = eye3(); Ps[i] = Vs[i] = Bas[i] = Bgs[i] = mk3(0, 0, 0); pre[i].reset(); dt_buf[i].clear(); la_buf[i].clear(); av_buf[i].clear(); headers[i] = 0; }
        lms.clear(); lm_index.clear(); imu_buf.clear(); frame_pre.clear(); tmp_pre.reset();
        prior = dv_ba_prior{}; prior_dev_A = prior_dev_b = nullptr;
        prev_time = -1; cur_time = 0; first_imu = false; init_pose = false; frame = 0; nonlinear = false; open_ex = false; acc_0 = gyr_0 = mk3(0, 0, 0);
        for (int k = 0; k < 2; ++k) { for (int i = 0; i < 9; ++i) ric[k].m[i] = cfg.ric[k][i]; tic[k] = mk3(cfg.tic[k][0], cfg.tic[k][1], cfg.tic[k][2]); }
        td = cfg.td; g = mk3(0, 0, cfg.g_norm);
        lines.clear(); lines.min_obs = cfg.line_min_obs > 0 ? cfg.line_min_obs : 5; pending_lines.clear(); para_line.clear(); line_last = dv_ba_summary{};
        im.clear(); im.cfg.use_det3d = cfg.use_det3d; im.cfg.init_min_num = cfg.instance_init_min_num; im.cfg.static_threshold = cfg.static_inst_threshold;
        im.cfg.plane_kind = cfg.plane_constraint ? (cfg.use_imu ? 1 : 2) : 0; im.cfg.max_iters = cfg.max_iters;
        std::memset(para_pose_ref, 0, sizeof(para_pose_ref)); obj_last = dv_ba_summary{}; dyn_frame = false; dyn_tail_deferred = false;
        latest_valid = false; latest_time = 0; latest_P = latest_V = latest_Ba = latest_Bg = latest_acc_0 = latest_gyr_0 = mk3(0, 0, 0); latest_Q = mkq(1, 0, 0, 0);
    }
    dvi::BodyView body_view() const { return dvi::BodyView{ Rs, Ps, ric, tic, headers, td, frame }; }
    double noise4[4];
    const double* noise() { noise4[0] = cfg.acc_n; noise4[1] = cfg.gyr_n; noise4[2] = cfg.acc_w; noise4[3] = cfg.gyr_w; return noise4; }

    // ---------------- IMU ----------------
    void process_imu(double dt, d3 la, d3 av) {
        if (!first_imu) { first_imu = true; acc_0 = la; gyr_0 = av; }
        if (!pre[frame]) pre[frame] = std::make_unique<Preint>(acc_0, gyr_0, Bas[frame], Bgs[frame], noise());
        if (frame != 0) {
            pre[frame]->push_back(dt, la, av); tmp_pre->push_back(dt, la, av);
            dt_buf[frame].push_back(dt); la_buf[frame].push_back(la); av_buf[frame].push_back(av);
            const int j = frame;
            const d3 un_acc_0 = mul(Rs[j], acc_0 - Bas[j]) - g;
            const d3 un_gyr = (gyr_0 + av) * 0.5 - Bgs[j];
            Rs[j] = mul(Rs[j], qR(dq_half(un_gyr * dt)));
            const d3 un_acc_1 = mul(Rs[j], la - Bas[j]) - g;
            const d3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
            Ps[j] = Ps[j] + Vs[j] * dt + un_acc * (0.5 * dt * dt);
            Vs[j] = Vs[j] + un_acc * dt;
        }
        acc_0 = la; gyr_0 = av;
    }
    bool add_imu_until(double t1) {       // GetIMUInterval + AddIMU (estimator.cpp:752-778,1765-1779)
        if (imu_buf.empty() || !(t1 <= imu_buf.back().first)) return false;
        std::vector<std::pair<double, std::pair<d3, d3>>> v;
        while (imu_buf.front().first <= prev_time) imu_buf.pop_front();
        while (imu_buf.front().first < t1) { v.push_back(imu_buf.front()); imu_buf.pop_front(); }
        v.push_back(imu_buf.front());
        if (!init_pose) {                  // InitFirstIMUPose
            init_pose = true;
            d3 aver = mk3(0, 0, 0); for (auto& s : v) aver = aver + s.second.first; aver = aver / (double)v.size();
            m33 R0 = qR(from_two_vectors(aver, mk3(0, 0, 1)));
            R0 = mul(ypr2r(mk3(-r2ypr(R0).x, 0, 0)), R0);                 // Utility::g2R
            Rs[0] = mul(ypr2r(mk3(-r2ypr(R0).x, 0, 0)), R0);
        }
        for (size_t i = 0; i < v.size(); ++i) {
            double dt;
            if (i == 0) dt = v[i].first - prev_time; else if (i == v.size() - 1) dt = cur_time - v[i - 1].first; else dt = v[i].first - v[i - 1].first;
            process_imu(dt, v[i].second.first, v[i].second.second);
        }
        return true;
    }

    // ---------------- feature manager ----------------
    void reindex() { lm_index.clear(); for (size_t i = 0; i < lms.size(); ++i) lm_index[lms[i].id] = i; index_dirty = false; }
    int long_count() const { int c = 0; for (auto& l : lms) if (l.obs.size() >= 4) ++c; return c; }
    bool add_features(int fc, const dv_feat* feats, int n) {
        // feature ids are handed out by a monotone counter, and landmarks are appended in id order and only ever erased: `lms` stays sorted
        // by id, so matching the (sorted) frame against it is a merge join — no hash index to rebuild after every slide / rejection.
        // Should the invariant ever be broken by a caller (ids not monotone), fall back to the hash index.
        bool sorted_ids = true;
        for (size_t i = 1; i < lms.size(); ++i) if (lms[i - 1].id >= lms[i].id) { sorted_ids = false; break; }
        if (!sorted_ids && index_dirty) reindex();
        std::vector<const dv_feat*> order(n);
        for (int i = 0; i < n; ++i) order[i] = &feats[i];
        std::sort(order.begin(), order.end(), [](const dv_feat* a, const dv_feat* b) { return a->id < b->id; });     // the reference iterates a std::map keyed by id
        int last_track = 0, new_feat = 0, long_track = 0;
        const size_t n_old = lms.size(); size_t cur = 0;
        for (const dv_feat* f : order) {
            Obs o; o.pt = mk3(f->left[0], f->left[1], f->left[2]); o.vel = mk3(f->left[5], f->left[6], 0); o.td = td; o.stereo = f->has_right != 0;
            o.pt_r = o.stereo ? mk3(f->right[0], f->right[1], f->right[2]) : mk3(0, 0, 0); o.vel_r = o.stereo ? mk3(f->right[5], f->right[6], 0) : mk3(0, 0, 0);
            long found = -1;
            if (sorted_ids) {
                while (cur < n_old && lms[cur].id < (int)f->id) ++cur;
                if (cur < n_old && lms[cur].id == (int)f->id) found = (long)cur;
                else if (!lms.empty() && lms.back().id >= (int)f->id) {      // a new id below a known one: the order would break -> hash index from here on
                    sorted_ids = false; reindex();
                }
            }
            if (!sorted_ids) { auto it = lm_index.find((int)f->id); if (it != lm_index.end()) found = (long)it->second; }
            if (found < 0) { Lm l; l.id = (int)f->id; l.start = fc; l.obs.push_back(o); if (!sorted_ids) lm_index[l.id] = lms.size(); lms.push_back(std::move(l)); new_feat++; }
            else { Lm& l = lms[(size_t)found]; l.obs.push_back(o); last_track++; if (l.obs.size() >= 4) long_track++; }
        }
        if (fc < 2 || last_track < 20 || long_track < 40 || new_feat > 0.5 * last_track) return true;
        double psum = 0; int pnum = 0;
        for (auto& l : lms) if (l.start <= fc - 2 && l.end() >= fc - 1) {
            const Obs& a = l.obs[fc - 2 - l.start]; const Obs& b = l.obs[fc - 1 - l.start];
            const double du = a.pt.x / a.pt.z - b.pt.x, dv = a.pt.y / a.pt.z - b.pt.y;
            psum += sqrt(du * du + dv * dv); pnum++;
        }
        if (pnum == 0) return true;
        return psum / pnum >= cfg.keyframe_parallax / kFocal;
    }
    void cam34(int k, int cam, double P[3][4]) const {
        const d3 t0 = Ps[k] + mul(Rs[k], tic[cam]); const m33 Rt = tr(mul(Rs[k], ric[cam])); const d3 t = -mul(Rt, t0);
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P[i][j] = Rt.m[i * 3 + j]; P[i][3] = get(t, i); }
    }
    static double tri_depth(const double L[3][4], const double R[3][4], double x0, double y0, double x1, double y1) {
        double D[4][4], A[4][4], v[4];
        for (int c = 0; c < 4; ++c) { D[0][c] = x0 * L[2][c] - L[0][c]; D[1][c] = y0 * L[2][c] - L[1][c]; D[2][c] = x1 * R[2][c] - R[0][c]; D[3][c] = y1 * R[2][c] - R[1][c]; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += D[k][i] * D[k][j]; A[i][j] = s; }
        smallest_eigvec4(A, v);
        const double X = v[0] / v[3], Y = v[1] / v[3], Z = v[2] / v[3];
        return L[2][0] * X + L[2][1] * Y + L[2][2] * Z + L[2][3];
    }
    void triangulate() {
        for (auto& l : lms) {
            if (l.depth > 0) continue;
            double L[3][4], R[3][4];
            if (cfg.stereo && l.obs[0].stereo) { cam34(l.start, 0, L); cam34(l.start, 1, R); const double d = tri_depth(L, R, l.obs[0].pt.x, l.obs[0].pt.y, l.obs[0].pt_r.x, l.obs[0].pt_r.y); l.depth = d > 0 ? d : cfg.init_depth; }
            else if (l.obs.size() > 1) { cam34(l.start, 0, L); cam34(l.start + 1, 0, R); const double d = tri_depth(L, R, l.obs[0].pt.x, l.obs[0].pt.y, l.obs[1].pt.x, l.obs[1].pt.y); l.depth = d > 0 ? d : cfg.init_depth; }
        }
    }
    template <class Pred> void erase_if(Pred p) { lms.erase(std::remove_if(lms.begin(), lms.end(), p), lms.end()); index_dirty = true; }

    // ---------------- PnP seed (solvePnP iterative restated as LM on rvec/t) ----------------
    static m33 rodrigues(d3 r) { const double th = norm(r); if (th < 1e-12) return add(eye3(), skew(r)); const m33 K = skew(r / th); return add(add(eye3(), scale(K, sin(th))), scale(mul(K, K), 1 - cos(th))); }
    static d3 inv_rodrigues(const m33& R) {
        quat q = qnormalized(qfromR(R)); if (q.w < 0) q = mkq(-q.w, -q.x, -q.y, -q.z);
        const double s = norm(qvec(q)); if (s < 1e-12) return qvec(q) * 2.0;
        return qvec(q) / s * (2 * atan2(s, q.w));
    }
    void pnp_frame(int fc) {
        if (fc <= 0) return;
        std::vector<d3> p3; std::vector<std::pair<float, float>> p2;
        for (auto& l : lms) if (l.depth > 0) {
            const int idx = fc - l.start;
            if ((int)l.obs.size() >= idx + 1) {
                const d3 w = mul(Rs[l.start], mul(ric[0], l.obs[0].pt * l.depth) + tic[0]) + Ps[l.start];
                p3.push_back(mk3((float)w.x, (float)w.y, (float)w.z)); p2.push_back({ (float)l.obs[idx].pt.x, (float)l.obs[idx].pt.y });
            }
        }
        if ((int)p2.size() < 4) return;
        m33 RCam = mul(Rs[fc - 1], ric[0]); d3 PCam = mul(Rs[fc - 1], tic[0]) + Ps[fc - 1];
        const m33 Ri = tr(RCam); const d3 ti = -mul(Ri, PCam), rv = inv_rodrigues(Ri);
        double x[6] = { rv.x, rv.y, rv.z, ti.x, ti.y, ti.z };
        auto resid = [&](const double* p, std::vector<double>& r) {
            const m33 Rm = rodrigues(mk3(p[0], p[1], p[2])); const d3 t = mk3(p[3], p[4], p[5]); double c = 0; r.resize(p2.size() * 2);
            for (size_t i = 0; i < p2.size(); ++i) { const d3 q = mul(Rm, p3[i]) + t; r[2 * i] = q.x / q.z - p2[i].first; r[2 * i + 1] = q.y / q.z - p2[i].second; c += r[2 * i] * r[2 * i] + r[2 * i + 1] * r[2 * i + 1]; }
            return c;
        };
        std::vector<double> r0, r1; double lambda = 1e-3, c0 = resid(x, r0);
        for (int it = 0; it < 20; ++it) {
            double JtJ[6][6] = { { 0 } }, Jtr[6] = { 0 };
            std::vector<double> rp;
            std::vector<std::vector<double>> Jc(6);
            for (int k = 0; k < 6; ++k) { double xp[6]; std::memcpy(xp, x, sizeof(xp)); xp[k] += 1e-7; resid(xp, rp); Jc[k].resize(rp.size()); for (size_t q = 0; q < rp.size(); ++q) Jc[k][q] = (rp[q] - r0[q]) / 1e-7; }
            for (int a = 0; a < 6; ++a) { for (size_t q = 0; q < r0.size(); ++q) Jtr[a] += Jc[a][q] * r0[q]; for (int b = 0; b < 6; ++b) { double s = 0; for (size_t q = 0; q < r0.size(); ++q) s += Jc[a][q] * Jc[b][q]; JtJ[a][b] = s; } }
            bool improved = false;
            for (int tries = 0; tries < 10 && !improved; ++tries) {
                double A[6][6], L[6][6] = { { 0 } }, d[6]; bool ok = true;
                for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) A[a][b] = JtJ[a][b] * (a == b ? 1 + lambda : 1.0);
                for (int j = 0; j < 6 && ok; ++j) { double s = A[j][j]; for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k]; if (!(s > 0)) { ok = false; break; } L[j][j] = sqrt(s); for (int i = j + 1; i < 6; ++i) { double t = A[i][j]; for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k]; L[i][j] = t / L[j][j]; } }
                if (!ok) { lambda *= 10; continue; }
                for (int i = 0; i < 6; ++i) { double s = Jtr[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * d[k]; d[i] = s / L[i][i]; }
                for (int i = 5; i >= 0; --i) { double s = d[i]; for (int k = i + 1; k < 6; ++k) s -= L[k][i] * d[k]; d[i] = s / L[i][i]; }
                double xn[6]; for (int a = 0; a < 6; ++a) xn[a] = x[a] - d[a];
                const double c1 = resid(xn, r1);
                if (c1 < c0) { std::memcpy(x, xn, sizeof(xn)); r0 = r1; const double dc = c0 - c1; c0 = c1; lambda = std::max(lambda / 10, 1e-16); improved = true; if (dc < 1e-20) it = 100; }
                else lambda *= 10;
            }
            if (!improved) break;
        }
        const m33 Rp = rodrigues(mk3(x[0], x[1], x[2]));
        RCam = tr(Rp); PCam = mul(RCam, -mk3(x[3], x[4], x[5]));
        Rs[fc] = mul(RCam, tr(ric[0])); Ps[fc] = -mul(mul(RCam, tr(ric[0])), tic[0]) + PCam;
        if (cfg.plane_constraint) { if (cfg.use_imu) Ps[fc].z = 0; else Ps[fc].y = 0; }
    }

    // ---------------- optimisation (device) ----------------
    void states_to_arrays() {        // BodyState::SetOptimizeParameters
        for (int i = 0; i <= kWin; ++i) {
            const quat q = qfromR(Rs[i]);
            const double p[7] = { Ps[i].x, Ps[i].y, Ps[i].z, q.x, q.y, q.z, q.w }; std::memcpy(pose[i], p, sizeof(p));
            const double s[9] = { Vs[i].x, Vs[i].y, Vs[i].z, Bas[i].x, Bas[i].y, Bas[i].z, Bgs[i].x, Bgs[i].y, Bgs[i].z }; std::memcpy(sb[i], s, sizeof(s));
        }
        for (int c = 0; c < 2; ++c) { const quat q = qfromR(ric[c]); const double p[7] = { tic[c].x, tic[c].y, tic[c].z, q.x, q.y, q.z, q.w }; std::memcpy(ex[c], p, sizeof(p)); }
        tdv[0] = td;
    }
    static dv_ba_factor mkfac(const Obs& o0, const Obs& o, bool right, int kind, int lm, int fi, int fj) {
        dv_ba_factor f{}; f.pix = o0.pt.x; f.piy = o0.pt.y; f.pjx = right ? o.pt_r.x : o.pt.x; f.pjy = right ? o.pt_r.y : o.pt.y;
        f.vix = o0.vel.x; f.viy = o0.vel.y; f.vjx = right ? o.vel_r.x : o.vel.x; f.vjy = right ? o.vel_r.y : o.vel.y; f.td_i = o0.td; f.td_j = o.td;
        f.kind = kind; f.lm = lm; f.fi = fi; f.fj = fj; return f;
    }
    // builds the residual blocks of AddResidualBlock (estimator.cpp:130-178); only_anchor0: the marginalization subset (:440-493)
    // fac_out / fac_cap: optional destination for the factor records — the pinned staging mirror of the device's upload region
    // (be_staging_factors), so that the table is assembled where the upload reads it instead of being copied there (336 KB per frame)
    dv_ba_factor* fac_out = nullptr; int fac_cap = 0, nfac = 0;
    void put_factor(const dv_ba_factor& f) {
        if (fac_out && nfac < fac_cap) fac_out[nfac] = f;
        else { if (fac_out) { fac.assign(fac_out, fac_out + nfac); fac_out = nullptr; } fac.push_back(f); }      // over capacity: back to the vector
        ++nfac;
    }
    void build_factors(bool only_anchor0) {
        fac.clear(); nfac = 0; lmt.clear(); if (!only_anchor0) invd.clear();
        int fi = -1;
        for (auto& l : lms) {
            if (l.obs.size() < 4) continue;
            ++fi;
            if (!only_anchor0) invd.push_back(1.0 / l.depth);
            if (only_anchor0 && l.start != 0) continue;
            dv_ba_lm t{}; t.first = nfac; t.anchor = l.start; t.mask = 0;
            int j = l.start - 1;
            for (auto& o : l.obs) {
                ++j; t.mask |= 1 << j;
                if (j != l.start) put_factor(mkfac(l.obs[0], o, false, 0, fi, l.start, j));
                if (cfg.stereo && o.stereo) put_factor(mkfac(l.obs[0], o, true, j != l.start ? 1 : 2, fi, l.start, j));
            }
            t.count = nfac - t.first;
            lmt.push_back(t);
        }
    }
    dv_ba_problem make_problem(int nframes) {
        dv_ba_problem P{};
        P.nframes = nframes; P.nlm = (int)lmt.size(); P.nfac = nfac; P.nimu = (int)imu.size(); P.use_imu = cfg.use_imu;
        P.plane_kind = cfg.plane_constraint ? (cfg.use_imu ? 1 : 2) : 0; P.max_iters = cfg.max_iters; P.g_norm = cfg.g_norm;
        P.pose = &pose[0][0]; P.speed_bias = &sb[0][0]; P.ex_pose = &ex[0][0]; P.td = tdv; P.inv_depth = invd.data();
        P.factors = fac_out ? fac_out : fac.data(); P.landmarks = lmt.data(); P.imu = imu.data();
        P.prior = prior.valid ? &prior : nullptr; P.prior_A = prior_dev_A; P.prior_b = prior_dev_b;      // A', b' stay in HBM (written by the fused marginalization)
        return P;
    }
    dv_ba_problem P{}; BeFused fu;          // the solve in flight (optimization_begin .. optimization_end)
    // diagnostics (dv_debug_set "hash_log"; the open multi-sequence defect of round 4): per window solve [solve counter, hash of the states uploaded, hash of the tables
    // uploaded (factors, landmarks, IMU records, prior descriptor), hash of the states downloaded, hash of the outlier flags consumed, iterations]
    std::vector<unsigned long long> hash_log; unsigned long long solve_no = 0;
    static unsigned long long fnv(unsigned long long h, const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; }
    unsigned long long hash_states() const {
        unsigned long long h = 1469598103934665603ull;
        h = fnv(h, pose, sizeof(pose)); h = fnv(h, sb, sizeof(sb)); h = fnv(h, ex, sizeof(ex)); h = fnv(h, tdv, sizeof(tdv)); h = fnv(h, invd.data(), invd.size() * sizeof(double));
        return h;
    }
    int optimization_begin(dv_ctx* ctx) {        // Estimator::Optimization (estimator.cpp:261-339) incl. SetMarginalizationInfo (:403-619)
        { HostScope h(ctx, "h_build");
          states_to_arrays();
          fac_out = (dv_ba_factor*)be_staging_factors(ctx, &fac_cap);
          build_factors(false);
          imu.clear();
          ctx->be.sqrt_hint.clear();
          if (cfg.use_imu) for (int i = 0; i < frame; ++i) { if (pre[i + 1]->sum_dt > 10.0) continue; dv_ba_imu r; pre[i + 1]->fill(r, i, i + 1); imu.push_back(r); ctx->be.sqrt_hint.push_back(pre[i + 1]->sqrt_info()); }
          prior_dev_A = prior.valid ? ctx->be.priorA_buf[ctx->be.prior_cur] : nullptr; prior_dev_b = prior.valid ? ctx->be.priorb_buf[ctx->be.prior_cur] : nullptr;
          P = make_problem(frame + 1);
          {   // AddBodyParameterBlock (estimator.cpp:87-100): which of para_ex_pose / para_td this solve may move
              const double v0 = norm(Vs[0]);
              if (((cfg.estimate & 1) && frame == kWin && v0 > 0.2) || open_ex) { open_ex = true; P.free_blocks |= 1; }
              if ((cfg.estimate & 2) && !(v0 < 0.2)) P.free_blocks |= 2;
          }
          if (cfg.use_line) {          // Vector2double's line part + AddLineResidualBlock under zero weights: the blocks only count in |x| (see dv_ba_problem::x_norm2_extra)
              lines.get_orth(Rs, Ps, ric[0], tic[0], para_line);
              double sq = 0; for (double v : para_line) sq += v * v;
              P.x_norm2_extra = sq;
          }
          // gauge reference: yaw and position of frame 0 before the solve (Double2vector, estimator.cpp:1111-1128)
          const d3 y0 = r2ypr(Rs[0]);
          std::memcpy(fu.R0, Rs[0].m, sizeof(fu.R0)); fu.ypr0[0] = y0.x; fu.ypr0[1] = y0.y; fu.ypr0[2] = y0.z; fu.P0[0] = Ps[0].x; fu.P0[1] = Ps[0].y; fu.P0[2] = Ps[0].z;
          fu.marg_mode = -1; fu.want_raw_pose = cfg.dynamic != 0;
          {   // OutliersRejection runs on the device behind the gauge fix (be_reject_kernel): the extrinsics it must use are the ones arrays_to_states will leave
              fu.want_reject = nonlinear; fu.rej_focal = kFocal;      // the initialisation solves never consume the flags (InitEstimator has no OutliersRejection)
              for (int c = 0; c < 2; ++c) {
                  const m33 r = cfg.use_imu ? qR(qnormalized(mkq(ex[c][6], ex[c][3], ex[c][4], ex[c][5]))) : ric[c];
                  const d3 tt = cfg.use_imu ? mk3(ex[c][0], ex[c][1], ex[c][2]) : tic[c];
                  std::memcpy(fu.rej_ric[c], r.m, sizeof(r.m)); fu.rej_tic[c][0] = tt.x; fu.rej_tic[c][1] = tt.y; fu.rej_tic[c][2] = tt.z;
              }
          }
          if (frame == kWin) {
              if (margin_old) fu.marg_mode = 0;
              else {
                  bool has9 = false;          // MARGIN_SECOND_NEW touches the prior only if it holds pose kWin-1 (estimator.cpp:557-560)
                  if (prior.valid) for (int i = 0; i < prior.nblocks; ++i) if (prior.blocks[i].type == 0 && prior.blocks[i].idx == kWin - 1) has9 = true;
                  if (has9) fu.marg_mode = 1;
              }
          }
        }
        if (ctx->be.debug_hash_log) {
            unsigned long long ht = 1469598103934665603ull;
            ht = fnv(ht, P.factors, (size_t)P.nfac * sizeof(dv_ba_factor)); ht = fnv(ht, lmt.data(), lmt.size() * sizeof(lmt[0])); ht = fnv(ht, imu.data(), imu.size() * sizeof(dv_ba_imu));
            const int pv = prior.valid ? 1 : 0; ht = fnv(ht, &pv, sizeof(pv));
            if (prior.valid) { ht = fnv(ht, &prior.n, sizeof(prior.n)); ht = fnv(ht, &prior.nblocks, sizeof(prior.nblocks)); ht = fnv(ht, prior.blocks, (size_t)prior.nblocks * sizeof(prior.blocks[0])); ht = fnv(ht, &fu.marg_mode, sizeof(fu.marg_mode)); }
            // the device-resident prior this solve starts from (written by the previous frame's marginalization): synchronises the solve stream — diagnostics only
            unsigned long long hp = 0;
            if (prior.valid && prior_dev_A && prior_dev_b) {
                std::vector<double> tmp((size_t)prior.n * prior.n + prior.n);
                (void)hipStreamSynchronize(ctx->be_stream);
                (void)hipMemcpy(tmp.data(), prior_dev_A, (size_t)prior.n * prior.n * sizeof(double), hipMemcpyDeviceToHost);
                (void)hipMemcpy(tmp.data() + (size_t)prior.n * prior.n, prior_dev_b, (size_t)prior.n * sizeof(double), hipMemcpyDeviceToHost);
                hp = fnv(1469598103934665603ull, tmp.data(), tmp.size() * sizeof(double));
            }
            hash_log.push_back(solve_no++); hash_log.push_back(hash_states()); hash_log.push_back(ht); hash_log.push_back(0); hash_log.push_back(hp); hash_log.push_back(0);
        }
        { HostScope h(ctx, "h_solve_begin"); const int rc = be_solve_fused_begin(ctx, &P, &fu); ctx->be.sqrt_hint.clear(); if (rc) return -1; }
        return 0;
    }
    int optimization_end(dv_ctx* ctx) {
        { HostScope h(ctx, "h_solve_wait"); if (be_solve_fused_end(ctx, &P, &last, &fu)) return -1; }
        if (ctx->be.debug_hash_log && hash_log.size() >= 6) {
            unsigned long long* r = hash_log.data() + hash_log.size() - 6;
            r[3] = hash_states(); r[5] = (unsigned long long)last.iterations;
            if (fu.rej_flags) { int n4 = 0; for (auto& l : lms) if (l.obs.size() >= 4) ++n4; r[3] = fnv(r[3], fu.rej_flags, (size_t)n4); }      // (the outlier flags consumed count as downloaded state)
        }
        { HostScope h(ctx, "h_post"); arrays_to_states(); }
        if (cfg.use_line && !para_line.empty()) lines.set_orth(Rs, Ps, ric[0], tic[0], para_line.data());      // Double2vector: SetLineOrth with the solved poses
        if (fu.marg_mode >= 0) prior = fu.new_prior;
        if (cfg.dynamic) {
            if (fu.marg_mode >= 0) {          // SetMarginalizationInfo re-ran Vector2double: para_pose = the gauge-fixed window
                for (int i = 0; i <= kWin; ++i) { const quat q = qfromR(Rs[i]); const double p[7] = { Ps[i].x, Ps[i].y, Ps[i].z, q.x, q.y, q.z, q.w }; std::memcpy(para_pose_ref[i], p, sizeof(p)); }
            } else if (ctx->be.pend->trivial) std::memcpy(para_pose_ref, pose, sizeof(para_pose_ref));
            else std::memcpy(para_pose_ref, fu.raw_pose, sizeof(para_pose_ref));
        }
        return 0;
    }
    int optimization(dv_ctx* ctx) { if (optimization_begin(ctx)) return -1; return optimization_end(ctx); }
    void arrays_to_states() {         // Double2vector + BodyState::GetOptimizationParameters (body.cpp:61-132); the yaw-gauge fix ran on the device
        auto qp = [&](int i) { return mkq(pose[i][6], pose[i][3], pose[i][4], pose[i][5]); };
        for (int i = 0; i <= kWin; ++i) {
            Rs[i] = qR(qnormalized(qp(i))); Ps[i] = mk3(pose[i][0], pose[i][1], pose[i][2]);
            if (cfg.use_imu) { Vs[i] = mk3(sb[i][0], sb[i][1], sb[i][2]); Bas[i] = mk3(sb[i][3], sb[i][4], sb[i][5]); Bgs[i] = mk3(sb[i][6], sb[i][7], sb[i][8]); }
        }
        if (cfg.use_imu) {
            for (int c = 0; c < 2; ++c) { tic[c] = mk3(ex[c][0], ex[c][1], ex[c][2]); ric[c] = qR(qnormalized(mkq(ex[c][6], ex[c][3], ex[c][4], ex[c][5]))); }
            td = tdv[0];
        }
        int k = -1;
        for (auto& l : lms) if (l.obs.size() >= 4) { l.depth = 1.0 / invd[++k]; l.solve_flag = l.depth < 0 ? 2 : 1; }
    }
    void reject_outliers() {          // OutliersRejection + RemoveOutlier (vio_util.cpp:381-430)
        if (fu.rej_flags) {           // decided on the device (same expressions, same order: be_reject_kernel), one flag per landmark of the problem = per landmark with >= 4 observations, in order
            const uint8_t* fl = fu.rej_flags; int k = -1;
            erase_if([&](const Lm& l) { if (l.obs.size() < 4) return false; return fl[++k] != 0; });
            fu.rej_flags = nullptr;
            return;
        }
        m33 RsT[kWin + 1], ricT[2];          // the transposes, once per frame instead of twice per observation (same values, same products)
        for (int i = 0; i <= kWin; ++i) RsT[i] = tr(Rs[i]);
        ricT[0] = tr(ric[0]); ricT[1] = tr(ric[1]);
        erase_if([&](const Lm& l) {
            if (l.obs.size() < 4) return false;
            double err = 0; int cnt = 0; int j = l.start - 1;
            const d3 pw = mul(Rs[l.start], mul(ric[0], l.obs[0].pt * l.depth) + tic[0]) + Ps[l.start];
            auto rp = [&](int fj, int cam, d3 uv) { const d3 pc = mul(ricT[cam], mul(RsT[fj], pw - Ps[fj]) - tic[cam]); const double rx = pc.x / pc.z - uv.x, ry = pc.y / pc.z - uv.y; return sqrt(rx * rx + ry * ry); };
            for (auto& o : l.obs) { ++j; if (j != l.start) { err += rp(j, 0, o.pt); cnt++; } if (cfg.stereo && o.stereo) { err += rp(j, 1, o.pt_r); cnt++; } }
            return err / cnt * kFocal > 3;
        });
    }
    void slide_window() {             // Estimator::SlideWindow (estimator.cpp:1201-1312)
        if (margin_old) {
            back_R0 = Rs[0]; back_P0 = Ps[0];
            if (frame != kWin) return;
            for (int i = 0; i < kWin; ++i) {
                headers[i] = headers[i + 1]; std::swap(Rs[i], Rs[i + 1]); std::swap(Ps[i], Ps[i + 1]);
                if (cfg.use_imu) { std::swap(pre[i], pre[i + 1]); dt_buf[i].swap(dt_buf[i + 1]); la_buf[i].swap(la_buf[i + 1]); av_buf[i].swap(av_buf[i + 1]); std::swap(Vs[i], Vs[i + 1]); std::swap(Bas[i], Bas[i + 1]); std::swap(Bgs[i], Bgs[i + 1]); }
            }
            headers[kWin] = headers[kWin - 1]; Ps[kWin] = Ps[kWin - 1]; Rs[kWin] = Rs[kWin - 1];
            if (cfg.use_imu) { Vs[kWin] = Vs[kWin - 1]; Bas[kWin] = Bas[kWin - 1]; Bgs[kWin] = Bgs[kWin - 1]; pre[kWin] = std::make_unique<Preint>(acc_0, gyr_0, Bas[kWin], Bgs[kWin], noise()); dt_buf[kWin].clear(); la_buf[kWin].clear(); av_buf[kWin].clear(); }
            if (nonlinear) {          // SlideWindowOld + RemoveBackShiftDepth
                const m33 R0 = mul(back_R0, ric[0]), R1 = mul(Rs[0], ric[0]); const d3 P0 = back_P0 + mul(back_R0, tic[0]), P1 = Ps[0] + mul(Rs[0], tic[0]);
                for (auto& l : lms) {
                    if (l.start != 0) { l.start--; continue; }
                    const d3 uv = l.obs[0].pt; l.obs.erase(l.obs.begin());
                    if (l.obs.size() < 2) { l.id = -1; continue; }
                    const d3 pj = mul(tr(R1), mul(R0, uv * l.depth) + P0 - P1);
                    l.depth = pj.z > 0 ? pj.z : cfg.init_depth;
                }
                erase_if([](const Lm& l) { return l.id < 0; });
                if (cfg.use_line) lines.remove_back_shift(R0, P0, R1, P1);
            } else {                  // RemoveBack
                for (auto& l : lms) { if (l.start != 0) l.start--; else { l.obs.erase(l.obs.begin()); if (l.obs.empty()) l.id = -1; } }
                erase_if([](const Lm& l) { return l.id < 0; });
                if (cfg.use_line) lines.remove_back();
            }
        } else if (frame == kWin) {
            headers[frame - 1] = headers[frame]; Ps[frame - 1] = Ps[frame]; Rs[frame - 1] = Rs[frame];
            if (cfg.use_imu) {
                for (size_t i = 0; i < dt_buf[frame].size(); ++i) { pre[frame - 1]->push_back(dt_buf[frame][i], la_buf[frame][i], av_buf[frame][i]); dt_buf[frame - 1].push_back(dt_buf[frame][i]); la_buf[frame - 1].push_back(la_buf[frame][i]); av_buf[frame - 1].push_back(av_buf[frame][i]); }
                Vs[frame - 1] = Vs[frame]; Bas[frame - 1] = Bas[frame]; Bgs[frame - 1] = Bgs[frame];
                pre[kWin] = std::make_unique<Preint>(acc_0, gyr_0, Bas[kWin], Bgs[kWin], noise()); dt_buf[kWin].clear(); la_buf[kWin].clear(); av_buf[kWin].clear();
            }
            for (auto& l : lms) {     // RemoveFront
                if (l.start == frame) { l.start--; continue; }
                if (l.end() < frame - 1) continue;
                l.obs.erase(l.obs.begin() + (kWin - 1 - l.start));
                if (l.obs.empty()) l.id = -1;
            }
            erase_if([](const Lm& l) { return l.id < 0; });
            if (cfg.use_line) lines.remove_front(frame);
        }
    }
    void solve_gyro_bias() {          // SolveGyroscopeBias (initial_aligment.cpp:29-61)
        double A[3][3] = { { 0 } }, b[3] = { 0 };
        for (size_t k = 0; k + 1 < frame_pre.size(); ++k) {
            const Preint& pj = *frame_pre[k + 1];
            const quat qij = qfromR(mul(tr(Rs[k]), Rs[k + 1]));
            const m33 tA = pj.jb(3, 12); const d3 tb = qvec(qmul(qinv(pj.dq), qij)) * 2.0;
            const m33 AtA = mul(tr(tA), tA); const d3 Atb = mul(tr(tA), tb);
            for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A[i][j] += AtA.m[i * 3 + j]; b[i] += get(Atb, i); }
        }
        double L[3][3] = { { 0 } }, d[3] = { 0, 0, 0 }; bool ok = true;
        for (int j = 0; j < 3 && ok; ++j) { double s = A[j][j]; for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k]; if (!(s > 0)) { ok = false; break; } L[j][j] = sqrt(s); for (int i = j + 1; i < 3; ++i) { double t = A[i][j]; for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k]; L[i][j] = t / L[j][j]; } }
        if (ok) { for (int i = 0; i < 3; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * d[k]; d[i] = s / L[i][i]; } for (int i = 2; i >= 0; --i) { double s = d[i]; for (int k = i + 1; k < 3; ++k) s -= L[k][i] * d[k]; d[i] = s / L[i][i]; } }
        const d3 dbg = mk3(d[0], d[1], d[2]);
        for (int i = 0; i <= kWin; ++i) Bgs[i] = Bgs[i] + dbg;
    }
    int init_estimator(dv_ctx* ctx) {        // InitEstimator (estimator.cpp:1424-1508), stereo paths
        if (cfg.stereo && cfg.use_imu) {
            pnp_frame(frame); triangulate();
            if (frame == kWin) {
                solve_gyro_bias();
                for (int j = 0; j <= kWin; ++j) pre[j]->repropagate(mk3(0, 0, 0), Bgs[j]);
                if (optimization(ctx)) return -1;
                nonlinear = true; slide_window();
            }
        } else if (cfg.stereo && !cfg.use_imu) {
            pnp_frame(frame); triangulate();
            if (optimization(ctx)) return -1;
            if (frame == kWin) { if (optimization(ctx)) return -1; nonlinear = true; slide_window(); }
        }
        if (frame < kWin) { frame++; const int p = frame - 1; Ps[frame] = Ps[p]; Vs[frame] = Vs[p]; Rs[frame] = Rs[p]; Bas[frame] = Bas[p]; Bgs[frame] = Bgs[p]; }
        return 0;
    }
    // Estimator::OptimizationWithOnlyLine (estimator.cpp:345-395): Vector2double, the line-only dogleg solve on the GPU (dv_line_solve; with the reference's zero
    // sqrt_info every residual and Jacobian is zero and ceres returns before its first step, so nothing is launched), Double2vector, RemoveLineOutlier.
    // Deviation (only reachable with non-zero weights): the reference leaves para_pose[kWinSize] a free 7-wide block here (its loop fixes poses 0..kWinSize-1);
    // dv_line_solve keeps all poses fixed.
    int optimization_only_line(dv_ctx* ctx) {
        lines.get_orth(Rs, Ps, ric[0], tic[0], para_line);
        line_last = dv_ba_summary{};
        const bool zero_w = cfg.line_sqrt_info[0] == 0.0 && cfg.line_sqrt_info[1] == 0.0 && cfg.line_sqrt_info[2] == 0.0 && cfg.line_sqrt_info[3] == 0.0;
        if (!para_line.empty() && !zero_w) {
            states_to_arrays();
            std::vector<dv_line_obs> obs; lines.observations(obs);
            dv_line_problem LP{};
            LP.n_lines = (int)para_line.size() / 4; LP.n_obs = (int)obs.size(); LP.max_iters = cfg.max_iters;
            LP.orth = para_line.data(); LP.pose = &pose[0][0]; LP.ex_pose = &ex[0][0]; std::memcpy(LP.sqrt_info, cfg.line_sqrt_info, sizeof(LP.sqrt_info)); LP.obs = obs.data();
            if (dv_line_solve(ctx, &LP, &line_last)) return -1;
        }
        if (!para_line.empty()) lines.set_orth(Rs, Ps, ric[0], tic[0], para_line.data());      // Double2vector (the poses did not move: re-normalises the Pluecker vectors)
        lines.remove_outliers(Rs, Ps, ric[0], tic[0]);
        return 0;
    }
    // the object branch of ProcessImage between TriangulatePoints and Optimization (estimator.cpp:1562-1622).  It neither reads what the window
    // solve writes nor writes what it reads (the joint factors are dead code), so it runs on the host + a third stream while the window
    // solve enqueued just before is in flight on the BA stream.
    int dynamic_branch(dv_ctx* ctx, const dv_inst_obs* insts, int n_insts, const dv_feat* inst_feats, const double* points) {
        HostScope h(ctx, "h_dynamic");
        flush_dyn_tail(ctx);          // the previous frame's object tail (see BodySnap)
        const dvi::BodyView B = body_view();
        { HostScope h1(ctx, "h_dyn_push"); im.push_back(frame, B, insts, n_insts, inst_feats, points); im.set_output_inst_info(); }
        { HostScope h1(ctx, "h_dyn_propagate"); im.propagate_pose(B); }
        { HostScope h1(ctx, "h_dyn_triangulate"); im.triangulate(B); }
        { HostScope h1(ctx, "h_dyn_initial"); im.initial_instance(B); im.initial_velocity(B); im.set_dynamic_or_static(B); }
        obj_solved = false;
        obj_last = dv_ba_summary{};
        bool have;
        { HostScope h1(ctx, "h_dyn_build"); have = im.build_problem(OP, &para_pose_ref[0][0], ric[0]) && (OP.n_boxes > 0 || OP.n_points > 0); }
        if (have) {
            HostScope h1(ctx, "h_dyn_solve_begin");
            if (be_obj_solve_begin(ctx, &OP, ctx->obj_stream, ctx->obj_buf, ctx->obj_pend)) return -1;      // enqueued; collected in dynamic_branch_finish
            obj_solved = true;
        }
        return 0;
    }
    // second half of the object branch (InstanceManager::GetOptimizationParameters + OutliersRejection, estimator_insts.cpp:804, estimator.cpp:1615): needs the object
    // solve's result and still the PRE-optimisation body states, so it runs at the top of process_image_end, before the window solve's result is applied
    dv_obj_problem OP{}; bool obj_solved = false;
    // The object branch's tail of a frame — ManageTriangulatePoint + SlideWindow on the body window BEFORE it slides, the per-frame clean-up on the slid window
    // (estimator.cpp:1653-1676) — reads nothing the next frame's ego branch writes and writes nothing it reads: it is kept back, with snapshots of the two body views it needs,
    // and runs at the top of the next frame's object branch, i.e. beside that frame's window solve instead of in front of it (45 us of the host path the GPU waits for).
    // Anything that looks at the objects before that (dv_est_get_instances, a reset) runs it first: same calls on the same data in the same order.
    struct BodySnap {
        m33 Rs[kWin + 1]; d3 Ps[kWin + 1]; m33 ric[2]; d3 tic[2]; double headers[kWin + 1]; double td = 0; int frame = 0;
        dvi::BodyView view() const { return dvi::BodyView{ Rs, Ps, ric, tic, headers, td, frame }; }
    };
    BodySnap snap_pre, snap_post; bool dyn_tail_deferred = false, dyn_tail_margin_old = false;
    void snap_body(BodySnap& b) const {
        for (int i = 0; i <= kWin; ++i) { b.Rs[i] = Rs[i]; b.Ps[i] = Ps[i]; b.headers[i] = headers[i]; }
        for (int c = 0; c < 2; ++c) { b.ric[c] = ric[c]; b.tic[c] = tic[c]; }
        b.td = td; b.frame = frame;
    }
    void flush_dyn_tail(dv_ctx* ctx) {
        if (!dyn_tail_deferred) return;
        dyn_tail_deferred = false;
        { HostScope h1(ctx, "h_dyn_slide"); const dvi::BodyView B = snap_pre.view(); im.manage_triangulate_point(B); im.slide_window(B, dyn_tail_margin_old); }
        { HostScope h1(ctx, "h_dyn_finish_frame"); im.finish_frame(snap_post.view()); }
    }
    int dynamic_branch_finish(dv_ctx* ctx) {
        HostScope h(ctx, "h_dynamic_finish");
        { HostScope h1(ctx, "h_dyn_solve_wait"); if (obj_solved && be_obj_solve_end(ctx, &OP, &obj_last, ctx->obj_pend)) return -1; }
        im.read_back(obj_solved);
        { HostScope h1(ctx, "h_dyn_reject"); im.outliers_rejection(body_view()); }
        return 0;
    }
    bool dyn_deferred = false;      // dv_est_process_dynamic_begin_ego: the object branch of this frame comes with dv_est_process_dynamic_attach
    int process_image_begin(dv_ctx* ctx, const dv_feat* feats, int n, double header, const dv_inst_obs* insts = nullptr, int n_insts = 0, const dv_feat* inst_feats = nullptr,
                            const double* points = nullptr, bool defer_dynamic = false) {      // ProcessImage (estimator.cpp:1516-1696), up to and including the enqueue of the window solve
        { HostScope h(ctx, "h_add_features");
          if (cfg.use_line) { lines.add(frame, pending_lines.data(), (int)pending_lines.size()); pending_lines.clear(); }
          margin_old = add_features(frame, feats, n); }
        headers[frame] = header;
        frame_pre.push_back(std::shared_ptr<Preint>(tmp_pre.release()));
        if (frame_pre.size() > (size_t)kWin + 1 && nonlinear) frame_pre.erase(frame_pre.begin());
        tmp_pre = std::make_unique<Preint>(acc_0, gyr_0, Bas[frame], Bgs[frame], noise());
        in_flight = false; dyn_frame = false; dyn_deferred = false;
        if (!nonlinear) return init_estimator(ctx);      // initialisation: synchronous
        if (!cfg.use_imu) pnp_frame(frame);
        { HostScope h(ctx, "h_triangulate"); triangulate(); if (cfg.use_line) lines.triangulate(Rs, Ps, ric[0], tic[0]); }
        if (cfg.use_line) {
            const bool zero_w = cfg.line_sqrt_info[0] == 0.0 && cfg.line_sqrt_info[1] == 0.0 && cfg.line_sqrt_info[2] == 0.0 && cfg.line_sqrt_info[3] == 0.0;
            if (!zero_w && lines.count() > 0 && !line_weight_warned) { line_weight_warned = true; dv_set_error(ctx, "line mode: non-zero lineProjectionFactor::sqrt_info is honoured by OptimizationWithOnlyLine only; inside the window solve the line blocks stay inert"); }
            HostScope h(ctx, "h_line_only"); if (optimization_only_line(ctx)) return -1;
        }
        if (optimization_begin(ctx)) return -1;
        in_flight = true;
        dyn_frame = cfg.dynamic != 0;
        if (dyn_frame && defer_dynamic) { dyn_deferred = true; return 0; }
        if (dyn_frame && dynamic_branch(ctx, insts, n_insts, inst_feats, points)) return -1;
        return 0;
    }
    int process_image_end(dv_ctx* ctx) {
        if (!in_flight) return 0;
        in_flight = false;
        if (dyn_frame && dyn_deferred) { dyn_deferred = false; if (dynamic_branch(ctx, nullptr, 0, nullptr, nullptr)) return -1; }      // never attached: a frame without objects
        if (dyn_frame && dynamic_branch_finish(ctx)) return -1;      // before optimization_end: Rs / Ps are still the states the reference's object branch saw
        if (optimization_end(ctx)) return -1;
        if (dyn_frame) im.touch_in_main_optimization();      // AddInstanceParameterBlock / im.GetOptimizationParameters inside Estimator::Optimization
        { HostScope h(ctx, "h_reject"); reject_outliers(); if (cfg.use_line) lines.remove_outliers(Rs, Ps, ric[0], tic[0]); }
        if (dyn_frame) { flush_dyn_tail(ctx); snap_body(snap_pre); dyn_tail_margin_old = margin_old; }      // estimator.cpp:1653-1658 sees the body window BEFORE it slides ...
        { HostScope h(ctx, "h_slide"); slide_window(); }
        if (dyn_frame) { snap_body(snap_post); dyn_tail_deferred = true; }                                     // ... and :1663-1676 the slid one: both kept back (flush_dyn_tail)
        erase_if([](const Lm& l) { return l.solve_flag == 2; });      // RemoveFailures
        update_latest_states();          // unconditional in the reference (estimator.cpp:1688); the IMU replay inside is empty in vision-only mode
        return 0;
    }
    bool in_flight = false, begun = false, line_weight_warned = false;
};

extern "C" {

int dv_est_create(dv_ctx* ctx, const dv_est_config* cfg) {
    if (!ctx) return -1;
    if (!cfg) DV_FAIL("dv_est_create: null config");
    if (!cfg->stereo) DV_FAIL("dv_est_create: monocular initialisation is out of scope (every BASELINE config is stereo)");
    if (cfg->estimate & ~3) DV_FAIL("dv_est_create: estimate: bit 0 estimate_extrinsic 1, bit 1 estimate_td 1 (estimate_extrinsic 2, the from-scratch calibration, is not built)");
    if (cfg->estimate && cfg->dynamic)      // the reference's static-instance factors take para_ex_pose[0] / para_td (estimator.cpp:205); the object branch here builds its problem with constant extrinsics
        DV_FAIL("dv_est_create: estimate != 0 (free extrinsic / td blocks) is not built for dynamic = 1: the object branch's factors carry no extrinsic / td Jacobians");
    delete ctx->est;
    ctx->est = new dv_estimator(*cfg);
    return be_prepare(ctx, cfg->dynamic != 0);      // nothing is allocated or created lazily in the middle of a sequence
}
// diagnostics (dv_debug_set "hash_log"): rows of six uint64 per window solve, see dv_estimator::hash_log
int dv_est_debug_hash_log(dv_ctx* ctx, unsigned long long* rows6, int cap, int* n_rows) {
    if (!ctx || !ctx->est) return -1;
    const int n = (int)(ctx->est->hash_log.size() / 6);
    if (n_rows) *n_rows = n;
    if (rows6) std::memcpy(rows6, ctx->est->hash_log.data(), sizeof(unsigned long long) * 6 * (size_t)std::min(n, std::max(cap, 0)));
    return 0;
}
int dv_est_reset(dv_ctx* ctx) { if (!ctx || !ctx->est) return -1; if (ctx->est->begun) { ctx->est->begun = false; (void)ctx->est->process_image_end(ctx); } ctx->est->clear(); return 0; }
int dv_est_input_imu(dv_ctx* ctx, double t, const double* acc, const double* gyr) {
    if (!ctx || !ctx->est) return -1;
    dv_estimator& E = *ctx->est;
    E.imu_buf.push_back({ t, { mk3(acc[0], acc[1], acc[2]), mk3(gyr[0], gyr[1], gyr[2]) } });
    if (E.nonlinear && E.latest_valid) E.fast_predict_imu(t, mk3(acc[0], acc[1], acc[2]), mk3(gyr[0], gyr[1], gyr[2]));      // InputIMU: FastPredictIMU + PubLatestOdometry (estimator.cpp:734-741)
    return 0;
}
int dv_est_set_lines(dv_ctx* ctx, const dv_line_row* rows, int n) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_set_lines: call dv_est_create first");
    if (n < 0 || (n > 0 && !rows)) DV_FAIL("dv_est_set_lines: bad argument");
    if (!ctx->est->cfg.use_line) DV_FAIL("dv_est_set_lines: the estimator was created with use_line = 0");
    ctx->est->pending_lines.assign(rows, rows + n);
    return 0;
}
int dv_est_get_lines(dv_ctx* ctx, dv_line_landmark* out, int cap, int* n_out) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_get_lines: call dv_est_create first");
    if (!n_out || cap < 0 || (cap > 0 && !out)) DV_FAIL("dv_est_get_lines: bad argument");
    int k = 0;
    for (auto& L : ctx->est->lines.lms) {
        if (k >= cap) break;
        dv_line_landmark& o = out[k++];
        std::memset(&o, 0, sizeof(o));
        o.id = L.id; o.start_frame = L.start; o.n_obs = (int)L.obs.size(); o.is_triangulation = L.tri;
        if (L.tri) { const double pl[6] = { L.plk.n.x, L.plk.n.y, L.plk.n.z, L.plk.v.x, L.plk.v.y, L.plk.v.z }; std::memcpy(o.plucker, pl, sizeof(pl));
                     o.ptw1[0] = L.ptw1.x; o.ptw1[1] = L.ptw1.y; o.ptw1[2] = L.ptw1.z; o.ptw2[0] = L.ptw2.x; o.ptw2[1] = L.ptw2.y; o.ptw2[2] = L.ptw2.z; }
    }
    *n_out = k;
    return 0;
}
// Estimator::ChangeSensorType (estimator.cpp:697-726)
int dv_est_change_sensor_type(dv_ctx* ctx, int use_imu, int use_stereo) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_change_sensor_type: call dv_est_create first");
    dv_estimator& E = *ctx->est;
    if (E.begun) DV_FAIL("dv_est_change_sensor_type: a frame is in flight");
    if (!use_imu && !use_stereo) DV_FAIL("dv_est_change_sensor_type: at least two sensors are needed (reference: message only)");
    if (!use_stereo) DV_FAIL("dv_est_change_sensor_type: monocular operation is out of scope (every BASELINE config is stereo)");
    bool restart = false;
    if ((E.cfg.use_imu != 0) != (use_imu != 0)) {
        E.cfg.use_imu = use_imu ? 1 : 0;
        if (use_imu) restart = true;
        else { E.prior = dv_ba_prior{}; E.prior_dev_A = E.prior_dev_b = nullptr; ctx->be.prior_resident = false; E.tmp_pre.reset(); E.latest_valid = false; }      // last_marg_info, tmp_pre_integration dropped
    }
    E.cfg.stereo = 1;
    if (restart) E.clear();          // ClearState + SetParameter
    return 0;
}
// latest_P / latest_Q / latest_V (what Publisher::PubLatestOdometry gets, estimator.cpp:737-739)
int dv_est_get_latest(dv_ctx* ctx, double* t, double* P3o, double* Q4o, double* V3o) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_get_latest: call dv_est_create first");
    const dv_estimator& E = *ctx->est;
    if (!E.latest_valid) return 1;
    const quat q = qnormalized(E.latest_Q);
    if (t) *t = E.latest_time;
    if (P3o) { P3o[0] = E.latest_P.x; P3o[1] = E.latest_P.y; P3o[2] = E.latest_P.z; }
    if (Q4o) { Q4o[0] = q.x; Q4o[1] = q.y; Q4o[2] = q.z; Q4o[3] = q.w; }
    if (V3o) { V3o[0] = E.latest_V.x; V3o[1] = E.latest_V.y; V3o[2] = E.latest_V.z; }
    return 0;
}
// body.ric / body.tic / body.td as Double2vector left them (what pubOdometry writes to the extrinsic file when estimate_extrinsic is on, utils/io/visualization.cpp:94-118)
int dv_est_get_extrinsics(dv_ctx* ctx, double* ric18, double* tic6, double* td) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_get_extrinsics: call dv_est_create first");
    const dv_estimator& E = *ctx->est;
    for (int c = 0; c < 2; ++c) {
        if (ric18) std::memcpy(ric18 + 9 * c, E.ric[c].m, 72);
        if (tic6) { tic6[3 * c] = E.tic[c].x; tic6[3 * c + 1] = E.tic[c].y; tic6[3 * c + 2] = E.tic[c].z; }
    }
    if (td) *td = E.td;
    return 0;
}
// feat_manager.point_landmarks as the point-cloud publishers read them (utils/io/visualization.cpp:214-249)
int dv_est_get_landmarks(dv_ctx* ctx, dv_landmark* out, int cap, int* n_out) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_get_landmarks: call dv_est_create first");
    if (!n_out || cap < 0 || (cap > 0 && !out)) DV_FAIL("dv_est_get_landmarks: bad argument");
    const dv_estimator& E = *ctx->est;
    int k = 0;
    for (auto& l : E.lms) {
        if (k >= cap) break;
        dv_landmark& o = out[k++];
        std::memset(&o, 0, sizeof(o));
        o.id = l.id; o.start_frame = l.start; o.n_obs = (int)l.obs.size(); o.solve_flag = l.solve_flag; o.depth = l.depth;
        const d3 pw = mul(E.Rs[l.start], mul(E.ric[0], l.obs[0].pt * l.depth) + E.tic[0]) + E.Ps[l.start];      // body.CamToWorld(point * depth, start_frame)
        o.p_w[0] = pw.x; o.p_w[1] = pw.y; o.p_w[2] = pw.z;
        const bool base = o.n_obs >= 2 && l.start < kWin - 2;
        o.in_point_cloud = base && !(l.start > kWin * 3.0 / 4.0 || l.solve_flag != 1);
        o.in_margin_cloud = base && l.start == 0 && o.n_obs <= 2 && l.solve_flag == 1;
    }
    *n_out = k;
    return 0;
}
static int est_begin(dv_ctx* ctx, const dv_feat* feats, int n, double t, const dv_inst_obs* insts, int n_insts, const dv_feat* inst_feats, const double* points, bool defer_dynamic = false) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_process: call dv_est_create first");
    dv_estimator& E = *ctx->est;
    if (E.begun) DV_FAIL("dv_est_process_begin: previous frame not collected (dv_est_process_end)");
    E.cur_time = t + E.td;
    { HostScope h(ctx, "h_imu"); if (E.cfg.use_imu && !E.add_imu_until(E.cur_time)) return 1; }       // "wait for imu" (estimator.cpp:1801-1805)
    { HostScope h(ctx, "h_process_begin"); if (E.process_image_begin(ctx, feats, n, t, insts, n_insts, inst_feats, points, defer_dynamic)) return -1; }
    E.begun = true;
    return 0;
}
// Estimator::IMUAvailable (estimator.h:128-133) at cur_time = t + td, as ProcessMeasurements tests it before it pops the frame (estimator.cpp:1800-1805)
int dv_est_imu_available(dv_ctx* ctx, double t) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_imu_available: call dv_est_create first");
    const dv_estimator& E = *ctx->est;
    if (!E.cfg.use_imu) return 1;
    return (!E.imu_buf.empty() && t + E.td <= E.imu_buf.back().first) ? 1 : 0;
}
int dv_est_process_begin(dv_ctx* ctx, const dv_feat* feats, int n, double t) { return est_begin(ctx, feats, n, t, nullptr, 0, nullptr, nullptr); }
int dv_est_process_dynamic_begin(dv_ctx* ctx, const dv_feat* feats, int n, double t, const dv_inst_obs* insts, int n_insts, const dv_feat* inst_feats, const double* points) {
    if (!ctx) return -1;
    if (ctx->est && !ctx->est->cfg.dynamic) DV_FAIL("dv_est_process_dynamic: the estimator was created with dynamic = 0");
    if (n_insts < 0 || (n_insts > 0 && !insts)) DV_FAIL("dv_est_process_dynamic: bad instance list");
    for (int i = 0; i < n_insts; ++i) {
        if (insts[i].n_feats < 0 || insts[i].n_points < 0 || (insts[i].n_feats > 0 && !inst_feats) || (insts[i].n_points > 0 && !points)) DV_FAIL("dv_est_process_dynamic: bad instance record");
    }
    return est_begin(ctx, feats, n, t, insts, n_insts, inst_feats, points);
}
static int check_insts(dv_ctx* ctx, const dv_inst_obs* insts, int n_insts, const dv_feat* inst_feats, const double* points) {
    if (n_insts < 0 || (n_insts > 0 && !insts)) DV_FAIL("dv_est_process_dynamic: bad instance list");
    for (int i = 0; i < n_insts; ++i) {
        if (insts[i].n_feats < 0 || insts[i].n_points < 0 || (insts[i].n_feats > 0 && !inst_feats) || (insts[i].n_points > 0 && !points)) DV_FAIL("dv_est_process_dynamic: bad instance record");
    }
    return 0;
}
// three-phase form: the window solve is enqueued with the background features alone (_begin_ego); the frame's instances follow (_attach: the object branch, host
// bookkeeping + dv_obj_solve on the third stream) while it is in flight — whatever the caller does between the two calls (enqueueing the next frame's tracking,
// collecting the object tracker) no longer sits in front of the window solve
int dv_est_process_dynamic_begin_ego(dv_ctx* ctx, const dv_feat* feats, int n, double t) {
    if (!ctx) return -1;
    if (ctx->est && !ctx->est->cfg.dynamic) DV_FAIL("dv_est_process_dynamic: the estimator was created with dynamic = 0");
    return est_begin(ctx, feats, n, t, nullptr, 0, nullptr, nullptr, true);
}
int dv_est_process_dynamic_attach(dv_ctx* ctx, const dv_inst_obs* insts, int n_insts, const dv_feat* inst_feats, const double* points) {
    if (!ctx) return -1;
    if (!ctx->est || !ctx->est->begun) DV_FAIL("dv_est_process_dynamic_attach: no frame in flight (dv_est_process_dynamic_begin_ego)");
    if (check_insts(ctx, insts, n_insts, inst_feats, points)) return -1;
    dv_estimator& E = *ctx->est;
    if (!E.dyn_deferred) return 0;                      // initialisation frame (processed synchronously, no object branch) or already attached
    E.dyn_deferred = false;
    return E.dynamic_branch(ctx, insts, n_insts, inst_feats, points) ? -1 : 0;
}
int dv_est_process_dynamic(dv_ctx* ctx, const dv_feat* feats, int n, double t, const dv_inst_obs* insts, int n_insts, const dv_feat* inst_feats, const double* points, dv_est_state* out) {
    const int rc = dv_est_process_dynamic_begin(ctx, feats, n, t, insts, n_insts, inst_feats, points);
    if (rc) return rc;
    return dv_est_process_end(ctx, out);
}
int dv_est_get_instances(dv_ctx* ctx, dv_inst_state* out, int cap, int* n_out, double* summary4) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_get_instances: call dv_est_create first");
    if (!n_out || cap < 0 || (cap > 0 && !out)) DV_FAIL("dv_est_get_instances: bad argument");
    dv_estimator& E = *ctx->est;
    E.flush_dyn_tail(ctx);
    int k = 0;
    for (auto& kv : E.im.insts) {
        if (k >= cap) break;
        const dvi::Inst& I = kv.second; dv_inst_state& o = out[k++];
        std::memset(&o, 0, sizeof(o));
        o.id = I.id; o.is_initial = I.is_initial; o.is_tracking = I.is_tracking; o.is_curr_visible = I.is_curr_visible; o.is_static = I.is_static; o.is_init_velocity = I.is_init_velocity;
        o.age = I.age; o.lost_number = I.lost_number; o.static_frame = I.static_frame; o.n_landmarks = (int)I.lms.size(); o.n_valid = I.valid_size(); o.triangle_num = I.triangle_num;
        for (int c = 0; c < 3; ++c) { o.dims[c] = I.dims[c]; o.vel_v[c] = get(I.vel_v, c); o.vel_a[c] = get(I.vel_a, c); }
        for (int i = 0; i <= kWin; ++i) { const quat q = qfromR(I.R[i]); double* p = o.window[i]; p[0] = I.P[i].x; p[1] = I.P[i].y; p[2] = I.P[i].z; p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w; o.time[i] = I.time[i]; }
    }
    *n_out = k;
    if (summary4) { summary4[0] = E.obj_last.iterations; summary4[1] = E.obj_last.termination; summary4[2] = E.obj_last.initial_cost; summary4[3] = E.obj_last.final_cost; }
    return 0;
}
int dv_est_get_static_instances(dv_ctx* ctx, uint32_t* ids, int cap, int* n_out) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_get_static_instances: call dv_est_create first");
    if (!n_out || cap < 0 || (cap > 0 && !ids)) DV_FAIL("dv_est_get_static_instances: bad argument");
    const std::vector<uint32_t>& v = ctx->est->im.static_out;
    const int n = std::min((int)v.size(), cap);
    for (int i = 0; i < n; ++i) ids[i] = v[i];
    *n_out = n;
    return 0;
}
int dv_est_process_end(dv_ctx* ctx, dv_est_state* out) {
    if (!ctx) return -1;
    if (!ctx->est) DV_FAIL("dv_est_process: call dv_est_create first");
    dv_estimator& E = *ctx->est;
    if (!E.begun) DV_FAIL("dv_est_process_end: no frame in flight");
    E.begun = false;
    { HostScope h(ctx, "h_process_end"); if (E.process_image_end(ctx)) return -1; }
    E.prev_time = E.cur_time;
    if (out) {
        std::memset(out, 0, sizeof(*out));
        out->frame = E.frame; out->nonlinear = E.nonlinear; out->margin_old = E.margin_old; out->n_landmarks = (int)E.lms.size(); out->n_long = E.long_count();
        out->iterations = E.last.iterations; out->initial_cost = E.last.initial_cost; out->final_cost = E.last.final_cost;
        for (int i = 0; i <= kWin; ++i) {
            const quat q = qfromR(E.Rs[i]); double* p = out->window[i];
            p[0] = E.Ps[i].x; p[1] = E.Ps[i].y; p[2] = E.Ps[i].z; p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
            p[7] = E.Vs[i].x; p[8] = E.Vs[i].y; p[9] = E.Vs[i].z; p[10] = E.Bas[i].x; p[11] = E.Bas[i].y; p[12] = E.Bas[i].z; p[13] = E.Bgs[i].x; p[14] = E.Bgs[i].y; p[15] = E.Bgs[i].z;
        }
    }
    return 0;
}
int dv_est_process(dv_ctx* ctx, const dv_feat* feats, int n, double t, dv_est_state* out) {
    const int rc = dv_est_process_begin(ctx, feats, n, t);
    if (rc) return rc;
    return dv_est_process_end(ctx, out);
}

}  // extern "C"

void dv_est_destroy_internal(dv_estimator* e) { delete e; }
