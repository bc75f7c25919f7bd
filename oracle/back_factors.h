// back_factors.h — CPU ORACLE (test infrastructure, not the product): first-party cost functions of
// the reference's sliding-window BA, restated dependency-free.
//   ProjectionTwoFrameOneCamFactor::Evaluate  estimator/factor/projection_two_frame_one_cam_factor.cpp:48-155
//   ProjectionTwoFrameTwoCamFactor::Evaluate  estimator/factor/projection_two_frame_two_cam_factor.cpp:47-170
//   ProjectionOneFrameTwoCamFactor::Evaluate  estimator/factor/projection_one_frame_two_cam_factor.cpp:48-140
//   IntegrationBase                           estimator/imu/integration_base.h:22-201
//   IMUFactor::Evaluate                       estimator/imu/imu_factor.h:31-172
// Jacobians are row-major with the global block sizes (2x7 / 15x7 / 15x9 ...), 7th pose column zero,
// exactly like the ceres::SizedCostFunction layout the reference fills.
#pragma once
#include "la.h"

namespace obe {
using namespace ola;

constexpr double kFocalLength = 460.0;                       // utils/parameters.h:44
constexpr double kSqrtInfo = kFocalLength / 1.5;             // estimator.cpp:685-687 (sqrt_info = 460/1.5 * I2)

struct ProjObs {            // constructor arguments of the three projection factors
    V3 pts_i, pts_j;        // normalised observations (z = 1)
    V3 vel_i, vel_j;        // z = 0
    double td_i = 0, td_j = 0;
};

inline V3 P3(const double* p) { return { p[0], p[1], p[2] }; }
inline Q Q4(const double* p) { return { p[6], p[3], p[4], p[5] }; }   // pose block = [p, qx qy qz qw]

inline void set_2x6(double* J, int stride, const double red[2][3], const M3& left, const M3& right) {
    for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 3; ++c) {
            double a = 0, b = 0;
            for (int k = 0; k < 3; ++k) { a += red[r][k] * left(k, c); b += red[r][k] * right(k, c); }
            J[r * stride + c] = a; J[r * stride + 3 + c] = b;
        }
        J[r * stride + 6] = 0.0;
    }
}

// kind: 0 = two-frame one-cam (blocks pose_i,pose_j,ex0,lambda,td)
//       1 = two-frame two-cam (pose_i,pose_j,ex0,ex1,lambda,td)
//       2 = one-frame two-cam (ex0,ex1,lambda,td)
// J[k] may be null.  Layout per block as above.
inline void proj_eval(int kind, const ProjObs& o, const double* const* par, double* res, double** J) {
    const double *pi = nullptr, *pj = nullptr, *ex0, *ex1 = nullptr; double inv_dep, td;
    if (kind == 0) { pi = par[0]; pj = par[1]; ex0 = par[2]; inv_dep = par[3][0]; td = par[4][0]; }
    else if (kind == 1) { pi = par[0]; pj = par[1]; ex0 = par[2]; ex1 = par[3]; inv_dep = par[4][0]; td = par[5][0]; }
    else { ex0 = par[0]; ex1 = par[1]; inv_dep = par[2][0]; td = par[3][0]; }
    const V3 tic = P3(ex0); const Q qic = Q4(ex0);
    const V3 tic2 = ex1 ? P3(ex1) : tic; const Q qic2 = ex1 ? Q4(ex1) : qic;
    const V3 pts_i_td = o.pts_i - o.vel_i * (td - o.td_i);
    const V3 pts_j_td = o.pts_j - o.vel_j * (td - o.td_j);
    const V3 pts_camera_i = pts_i_td / inv_dep;
    const V3 pts_imu_i = qic * pts_camera_i + tic;
    V3 Pi, Pj; Q Qi, Qj; V3 pts_imu_j;
    if (kind != 2) {
        Pi = P3(pi); Qi = Q4(pi); Pj = P3(pj); Qj = Q4(pj);
        const V3 pts_w = Qi * pts_imu_i + Pi;
        pts_imu_j = Qj.inverse() * (pts_w - Pj);
    } else pts_imu_j = pts_imu_i;
    const V3 pts_camera_j = qic2.inverse() * (pts_imu_j - tic2);
    const double dep_j = pts_camera_j.z;
    res[0] = kSqrtInfo * (pts_camera_j.x / dep_j - pts_j_td.x);
    res[1] = kSqrtInfo * (pts_camera_j.y / dep_j - pts_j_td.y);
    if (!J) return;
    const M3 ric = qic.R(), ric2 = qic2.R();
    double red[2][3] = { { kSqrtInfo * (1. / dep_j), 0, kSqrtInfo * (-pts_camera_j.x / (dep_j * dep_j)) },
                         { 0, kSqrtInfo * (1. / dep_j), kSqrtInfo * (-pts_camera_j.y / (dep_j * dep_j)) } };
    auto red_vec = [&](const V3& v, double* out) { for (int r = 0; r < 2; ++r) out[r] = red[r][0] * v.x + red[r][1] * v.y + red[r][2] * v.z; };
    if (kind != 2) {
        const M3 Ri = Qi.R(), Rj = Qj.R();
        const M3 rcT = (kind == 0 ? ric : ric2).t();
        if (J[0]) set_2x6(J[0], 7, red, rcT * Rj.t(), rcT * Rj.t() * Ri * -skew(pts_imu_i));
        if (J[1]) set_2x6(J[1], 7, red, rcT * -Rj.t(), rcT * skew(pts_imu_j));
        const M3 T = rcT * Rj.t() * Ri * ric;    // ric(2)^T Rj^T Ri ric
        if (kind == 0) {
            if (J[2]) {
                M3 right = -T * skew(pts_camera_i) + skew(T * pts_camera_i) + skew(ric.t() * (Rj.t() * (Ri * tic + Pi - Pj) - tic));
                set_2x6(J[2], 7, red, ric.t() * (Rj.t() * Ri - M3::identity()), right);
            }
            if (J[3]) { V3 v = T * pts_i_td * (-1.0 / (inv_dep * inv_dep)); red_vec(v, J[3]); }
            if (J[4]) { V3 v = T * o.vel_i / inv_dep * -1.0; double t[2]; red_vec(v, t); J[4][0] = t[0] + kSqrtInfo * o.vel_j.x; J[4][1] = t[1] + kSqrtInfo * o.vel_j.y; }
        } else {
            if (J[2]) set_2x6(J[2], 7, red, ric2.t() * Rj.t() * Ri, ric2.t() * Rj.t() * Ri * ric * -skew(pts_camera_i));
            if (J[3]) set_2x6(J[3], 7, red, -ric2.t(), skew(pts_camera_j));
            if (J[4]) { V3 v = T * pts_i_td * (-1.0 / (inv_dep * inv_dep)); red_vec(v, J[4]); }
            if (J[5]) { V3 v = T * o.vel_i / inv_dep * -1.0; double t[2]; red_vec(v, t); J[5][0] = t[0] + kSqrtInfo * o.vel_j.x; J[5][1] = t[1] + kSqrtInfo * o.vel_j.y; }
        }
    } else {
        const M3 T = ric2.t() * ric;
        if (J[0]) set_2x6(J[0], 7, red, ric2.t(), ric2.t() * ric * -skew(pts_camera_i));
        if (J[1]) set_2x6(J[1], 7, red, -ric2.t(), skew(pts_camera_j));
        if (J[2]) { V3 v = T * o.pts_i * (-1.0 / (inv_dep * inv_dep)); red_vec(v, J[2]); }   // Q7: pts_i, not pts_i_td (:125)
        if (J[3]) { V3 v = T * o.vel_i / inv_dep * -1.0; double t[2]; red_vec(v, t); J[3][0] = t[0] + kSqrtInfo * o.vel_j.x; J[3][1] = t[1] + kSqrtInfo * o.vel_j.y; }
    }
}

// ---------------------------------------------------------------------------------------------
struct ImuNoise { double acc_n, gyr_n, acc_w, gyr_w; };

struct Integration {     // IntegrationBase
    V3 acc_0, gyr_0, linearized_acc, linearized_gyr, linearized_ba, linearized_bg;
    Mat jacobian{ 15, 15 }, covariance{ 15, 15 };
    double noise_diag[18];
    double sum_dt = 0;
    V3 delta_p, delta_v; Q delta_q;
    std::vector<double> dt_buf; std::vector<V3> acc_buf, gyr_buf;

    Integration(const V3& a0, const V3& g0, const V3& ba, const V3& bg, const ImuNoise& n)
        : acc_0(a0), gyr_0(g0), linearized_acc(a0), linearized_gyr(g0), linearized_ba(ba), linearized_bg(bg) {
        for (int i = 0; i < 15; ++i) jacobian(i, i) = 1;
        const double v[6] = { n.acc_n * n.acc_n, n.gyr_n * n.gyr_n, n.acc_n * n.acc_n, n.gyr_n * n.gyr_n, n.acc_w * n.acc_w, n.gyr_w * n.gyr_w };
        for (int b = 0; b < 6; ++b) for (int k = 0; k < 3; ++k) noise_diag[3 * b + k] = v[b];
    }
    void push_back(double dt, const V3& acc, const V3& gyr) { dt_buf.push_back(dt); acc_buf.push_back(acc); gyr_buf.push_back(gyr); propagate(dt, acc, gyr); }
    void repropagate(const V3& ba, const V3& bg) {
        sum_dt = 0; acc_0 = linearized_acc; gyr_0 = linearized_gyr;
        delta_p = V3(); delta_q = Q(); delta_v = V3(); linearized_ba = ba; linearized_bg = bg;
        jacobian.zero(); for (int i = 0; i < 15; ++i) jacobian(i, i) = 1;
        covariance.zero();
        for (size_t i = 0; i < dt_buf.size(); ++i) propagate(dt_buf[i], acc_buf[i], gyr_buf[i]);
    }
    static void setb(Mat& M, int r, int c, const M3& b) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M(r + i, c + j) = b(i, j); }
    void propagate(double dt, const V3& acc_1, const V3& gyr_1) {       // midPointIntegration (:70-140) + propagate (:142-173)
        const V3 un_acc_0 = delta_q * (acc_0 - linearized_ba);
        const V3 un_gyr = (gyr_0 + gyr_1) * 0.5 - linearized_bg;
        const Q rq = delta_q * Q(1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2);
        const V3 un_acc_1 = rq * (acc_1 - linearized_ba);
        const V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
        const V3 rp = delta_p + delta_v * dt + un_acc * (0.5 * dt * dt);
        const V3 rv = delta_v + un_acc * dt;
        {
            const V3 w_x = (gyr_0 + gyr_1) * 0.5 - linearized_bg, a_0_x = acc_0 - linearized_ba, a_1_x = acc_1 - linearized_ba;
            const M3 R_w_x = skew(w_x), R_a_0_x = skew(a_0_x), R_a_1_x = skew(a_1_x);
            const M3 I = M3::identity(), Rq = delta_q.R(), Rr = rq.R();
            Mat F(15, 15), V(15, 18);
            setb(F, 0, 0, I);
            setb(F, 0, 3, Rq * R_a_0_x * (-0.25 * dt * dt) + Rr * R_a_1_x * (I - R_w_x * dt) * (-0.25 * dt * dt));
            setb(F, 0, 6, I * dt);
            setb(F, 0, 9, (Rq + Rr) * (-0.25 * dt * dt));
            setb(F, 0, 12, Rr * R_a_1_x * (-0.25 * dt * dt * -dt));
            setb(F, 3, 3, I - R_w_x * dt);
            setb(F, 3, 12, I * (-1.0 * dt));
            setb(F, 6, 3, Rq * R_a_0_x * (-0.5 * dt) + Rr * R_a_1_x * (I - R_w_x * dt) * (-0.5 * dt));
            setb(F, 6, 6, I);
            setb(F, 6, 9, (Rq + Rr) * (-0.5 * dt));
            setb(F, 6, 12, Rr * R_a_1_x * (-0.5 * dt * -dt));
            setb(F, 9, 9, I); setb(F, 12, 12, I);
            const M3 V03 = (-Rr) * R_a_1_x * (0.25 * dt * dt * 0.5 * dt), V63 = (-Rr) * R_a_1_x * (0.5 * dt * 0.5 * dt);
            setb(V, 0, 0, Rq * (0.25 * dt * dt)); setb(V, 0, 3, V03); setb(V, 0, 6, Rr * (0.25 * dt * dt)); setb(V, 0, 9, V03);
            setb(V, 3, 3, I * (0.5 * dt)); setb(V, 3, 9, I * (0.5 * dt));
            setb(V, 6, 0, Rq * (0.5 * dt)); setb(V, 6, 3, V63); setb(V, 6, 6, Rr * (0.5 * dt)); setb(V, 6, 9, V63);
            setb(V, 9, 12, I * dt); setb(V, 12, 15, I * dt);
            jacobian = matmul(F, jacobian);
            Mat FP = matmul(matmul(F, covariance), transpose(F));
            Mat VN(15, 18);
            for (int i = 0; i < 15; ++i) for (int j = 0; j < 18; ++j) VN(i, j) = V(i, j) * noise_diag[j];
            Mat VNV = matmul(VN, transpose(V));
            for (size_t i = 0; i < FP.d.size(); ++i) FP.d[i] += VNV.d[i];
            covariance = FP;
        }
        delta_p = rp; delta_q = rq.normalized(); delta_v = rv;
        sum_dt += dt; acc_0 = acc_1; gyr_0 = gyr_1;
    }
    M3 jb(int r, int c) const { M3 b; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) b(i, j) = jacobian(r + i, c + j); return b; }
    // evaluate (:175-201): O_P 0, O_R 3, O_V 6, O_BA 9, O_BG 12
    void evaluate(const V3& G, const V3& Pi, const Q& Qi, const V3& Vi, const V3& Bai, const V3& Bgi, const V3& Pj, const Q& Qj, const V3& Vj,
                  const V3& Baj, const V3& Bgj, double* r) const {
        const M3 dp_dba = jb(0, 9), dp_dbg = jb(0, 12), dq_dbg = jb(3, 12), dv_dba = jb(6, 9), dv_dbg = jb(6, 12);
        const V3 dba = Bai - linearized_ba, dbg = Bgi - linearized_bg;
        const Q cq = delta_q * deltaQ(dq_dbg * dbg);
        const V3 cv = delta_v + dv_dba * dba + dv_dbg * dbg;
        const V3 cp = delta_p + dp_dba * dba + dp_dbg * dbg;
        const V3 rp = Qi.inverse() * (G * (0.5 * sum_dt * sum_dt) + Pj - Pi - Vi * sum_dt) - cp;
        const V3 rq = (cq.inverse() * (Qi.inverse() * Qj)).vec() * 2.0;
        const V3 rv = Qi.inverse() * (G * sum_dt + Vj - Vi) - cv;
        const V3 rba = Baj - Bai, rbg = Bgj - Bgi;
        for (int k = 0; k < 3; ++k) { r[k] = rp[k]; r[3 + k] = rq[k]; r[6 + k] = rv[k]; r[9 + k] = rba[k]; r[12 + k] = rbg[k]; }
    }
};

inline void Qleft_br(const Q& q, M3& out) { out = M3::identity() * q.w + skew(q.vec()); }    // bottomRightCorner<3,3> of Qleft
inline void Qright_br(const Q& p, M3& out) { out = M3::identity() * p.w - skew(p.vec()); }
// bottom-right 3x3 of Qleft(a) * Qright(b)
inline M3 QlQr_br(const Q& a, const Q& b) {
    double L[4][4], Rm[4][4];
    auto fill = [](double M[4][4], const Q& q, double sgn) {
        M[0][0] = q.w; M[0][1] = -q.x; M[0][2] = -q.y; M[0][3] = -q.z;
        M[1][0] = q.x; M[2][0] = q.y; M[3][0] = q.z;
        M3 s = skew(q.vec());
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[1 + i][1 + j] = (i == j ? q.w : 0.0) + sgn * s(i, j);
    };
    fill(L, a, 1.0); fill(Rm, b, -1.0);
    M3 out;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += L[1 + i][k] * Rm[k][1 + j]; out(i, j) = s; }
    return out;
}

// IMUFactor::Evaluate.  par = pose_i(7), sb_i(9), pose_j(7), sb_j(9); J = 15x7, 15x9, 15x7, 15x9 row-major (may be null)
inline void imu_eval(const Integration& pre, const V3& G, const double* const* par, double* res, double** J) {
    const V3 Pi = P3(par[0]); const Q Qi = Q4(par[0]);
    const V3 Vi = P3(par[1]), Bai = P3(par[1] + 3), Bgi = P3(par[1] + 6);
    const V3 Pj = P3(par[2]); const Q Qj = Q4(par[2]);
    const V3 Vj = P3(par[3]), Baj = P3(par[3] + 3), Bgj = P3(par[3] + 6);
    double r[15];
    pre.evaluate(G, Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj, r);
    // sqrt_info = LLT(cov^-1).matrixL().transpose()  (recomputed every Evaluate in the reference, Q8)
    Mat Lc;
    Mat cinv = inverse(pre.covariance);
    for (int i = 0; i < 15; ++i) for (int j = i + 1; j < 15; ++j) { double s = 0.5 * (cinv(i, j) + cinv(j, i)); cinv(i, j) = cinv(j, i) = s; }
    bool ok = cholesky(cinv, Lc); (void)ok;
    Mat S = transpose(Lc);   // upper
    for (int i = 0; i < 15; ++i) { double s = 0; for (int k = 0; k < 15; ++k) s += S(i, k) * r[k]; res[i] = s; }
    if (!J) return;
    const double sum_dt = pre.sum_dt;
    const M3 dp_dba = pre.jb(0, 9), dp_dbg = pre.jb(0, 12), dq_dbg = pre.jb(3, 12), dv_dba = pre.jb(6, 9), dv_dbg = pre.jb(6, 12);
    const M3 RiT = Qi.inverse().R();
    auto apply = [&](Mat& Jb, double* out, int cols) {
        for (int i = 0; i < 15; ++i) for (int c = 0; c < cols; ++c) { double s = 0; for (int k = 0; k < 15; ++k) s += S(i, k) * Jb(k, c); out[i * cols + c] = s; }
    };
    const Q cq = pre.delta_q * deltaQ(dq_dbg * (Bgi - pre.linearized_bg));
    if (J[0]) {
        Mat Jb(15, 7);
        Integration::setb(Jb, 0, 0, -RiT);
        Integration::setb(Jb, 0, 3, skew(Qi.inverse() * (G * (0.5 * sum_dt * sum_dt) + Pj - Pi - Vi * sum_dt)));
        Integration::setb(Jb, 3, 3, -QlQr_br(Qj.inverse() * Qi, cq));
        Integration::setb(Jb, 6, 3, skew(Qi.inverse() * (G * sum_dt + Vj - Vi)));
        apply(Jb, J[0], 7);
    }
    if (J[1]) {
        Mat Jb(15, 9);
        Integration::setb(Jb, 0, 0, -RiT * sum_dt); Integration::setb(Jb, 0, 3, -dp_dba); Integration::setb(Jb, 0, 6, -dp_dbg);
        M3 ql; Qleft_br(Qj.inverse() * Qi * pre.delta_q, ql);
        Integration::setb(Jb, 3, 6, -ql * dq_dbg);
        Integration::setb(Jb, 6, 0, -RiT); Integration::setb(Jb, 6, 3, -dv_dba); Integration::setb(Jb, 6, 6, -dv_dbg);
        Integration::setb(Jb, 9, 3, -M3::identity()); Integration::setb(Jb, 12, 6, -M3::identity());
        apply(Jb, J[1], 9);
    }
    if (J[2]) {
        Mat Jb(15, 7);
        Integration::setb(Jb, 0, 0, RiT);
        M3 ql; Qleft_br(cq.inverse() * Qi.inverse() * Qj, ql);
        Integration::setb(Jb, 3, 3, ql);
        apply(Jb, J[2], 7);
    }
    if (J[3]) {
        Mat Jb(15, 9);
        Integration::setb(Jb, 6, 0, RiT); Integration::setb(Jb, 9, 3, M3::identity()); Integration::setb(Jb, 12, 6, M3::identity());
        apply(Jb, J[3], 9);
    }
}

}  // namespace obe
