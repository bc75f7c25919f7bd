// obj_factors.cpp — CPU ORACLE (test infrastructure, not the product): line and dynamic-object factors of the path
// (SURVEY 8(a) rows L1, I1-I3), restated from
//   lineProjectionFactor::Evaluate             estimator/factor/line_projection_factor.cpp:24-159
//   LineOrthParameterization::Plus             estimator/factor/line_parameterization.cpp:9-72
//   orth_to_plk / plk_to_pose / plk_from_pose  line_detector/line_geometry.cpp:97-135,210-229
//   BoxEncloseStereoPointFactor::Evaluate      estimator/factor/box_factor.cpp:523-565   (N_p from R_ojw (p_obj - P_woj), sic)
//   BoxDimsFactor::Evaluate                    estimator/factor/box_factor.cpp:728-743   (J = 2 (box - dims)^T although r = |box-dims|^4 / 100, sic)
//   BoxOrientationFactor::Evaluate             estimator/factor/box_factor.cpp:752-806   (camera-pose J = 0; the reference's own J_r formula, sic)
// Sophus SO3::log (un-vendored dependency) is restated from its published algorithm (quaternion log with the
// small-angle series; so3.hpp logAndTheta).  PARITY UNPINNED (dvo.h).  Jacobians are returned in the reference's
// global sizes (pose blocks 7 wide with a zero last column).
#include <cmath>
#include "dvo.h"
#include "la.h"

using namespace ola;

namespace {

inline V3 P3(const double* p) { return V3(p[0], p[1], p[2]); }
inline Q Q4(const double* p) { return Q(p[6], p[3], p[4], p[5]); }      // pose block [p, qx qy qz qw]

struct Plk { V3 n, v; };

Plk orth_to_plk(const double* orth) {
    const double s1 = std::sin(orth[0]), c1 = std::cos(orth[0]), s2 = std::sin(orth[1]), c2 = std::cos(orth[1]), s3 = std::sin(orth[2]), c3 = std::cos(orth[2]);
    M3 R;
    R(0, 0) = c2 * c3; R(0, 1) = s1 * s2 * c3 - c1 * s3; R(0, 2) = c1 * s2 * c3 + s1 * s3;
    R(1, 0) = c2 * s3; R(1, 1) = s1 * s2 * s3 + c1 * c3; R(1, 2) = c1 * s2 * s3 - s1 * c3;
    R(2, 0) = -s2;     R(2, 1) = s1 * c2;                R(2, 2) = c1 * c2;
    const double w1 = std::cos(orth[3]), w2 = std::sin(orth[3]);
    return { R.col(0) * w1, R.col(1) * w2 };
}
Plk plk_to_pose(const Plk& w, const M3& Rcw, const V3& tcw) { return { Rcw * w.n + skew(tcw) * (Rcw * w.v), Rcw * w.v }; }
Plk plk_from_pose(const Plk& c, const M3& Rcw, const V3& tcw) { const M3 Rwc = Rcw.t(); return plk_to_pose(c, Rwc, -(Rwc * tcw)); }

// 2x6 * 6x6 with the 6x6 given as four 3x3 blocks [[A B] [C D]]
struct M26 { double m[2][6]; };
M26 mul26(const M26& a, const M3& A, const M3& B, const M3& C, const M3& D) {
    M26 r{};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            double s0 = 0, s1 = 0;
            for (int k = 0; k < 3; ++k) { s0 += a.m[i][k] * A(k, j) + a.m[i][3 + k] * C(k, j); s1 += a.m[i][k] * B(k, j) + a.m[i][3 + k] * D(k, j); }
            r.m[i][j] = s0; r.m[i][3 + j] = s1;
        }
    return r;
}

V3 so3_log(const M3& R) {        // Sophus::SO3d(R).log()
    Q q = Q::fromR(R);      // Eigen::Quaterniond(R)
    q = q.normalized();
    const double sq = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w;
    double two_atan;
    if (sq < 1e-10 * 1e-10) two_atan = 2.0 / w - 2.0 / 3.0 * sq / (w * w * w);
    else {
        const double n = std::sqrt(sq);
        if (std::fabs(w) < 1e-10) two_atan = (w > 0 ? M_PI : -M_PI) / n;
        else two_atan = 2.0 * std::atan(n / w) / n;
    }
    return V3(q.x, q.y, q.z) * two_atan;
}

}  // namespace

extern "C" {

// par: pose_i[7], ex[7], orth[4]; J: 2x7, 2x7, 2x4 row-major (any may be null)
void dvo_line_eval(const double* obs4, const double* sqrt_info4, const double* const* par, double* res2, double** J) {
    const V3 Pi = P3(par[0]); const Q Qi = Q4(par[0]);
    const V3 tic = P3(par[1]); const Q qic = Q4(par[1]);
    const Plk lw = orth_to_plk(par[2]);
    const M3 Rwb = Qi.R(); const V3 twb = Pi;
    const Plk lb = plk_from_pose(lw, Rwb, twb);
    const M3 Rbc = qic.R(); const V3 tbc = tic;
    const Plk lc = plk_from_pose(lb, Rbc, tbc);
    const V3 nc = lc.n;
    const double l_norm = nc.x * nc.x + nc.y * nc.y, l_sqrt = std::sqrt(l_norm), l_tri = l_norm * l_sqrt;
    const double e1 = obs4[0] * nc.x + obs4[1] * nc.y + nc.z, e2 = obs4[2] * nc.x + obs4[3] * nc.y + nc.z;
    const double r0 = e1 / l_sqrt, r1 = e2 / l_sqrt;
    res2[0] = sqrt_info4[0] * r0 + sqrt_info4[1] * r1;
    res2[1] = sqrt_info4[2] * r0 + sqrt_info4[3] * r1;
    if (!J) return;
    double jel[2][3] = { { obs4[0] / l_sqrt - nc.x * e1 / l_tri, obs4[1] / l_sqrt - nc.y * e1 / l_tri, 1.0 / l_sqrt },
                         { obs4[2] / l_sqrt - nc.x * e2 / l_tri, obs4[3] / l_sqrt - nc.y * e2 / l_tri, 1.0 / l_sqrt } };
    M26 jeLc{};
    for (int j = 0; j < 3; ++j) { jeLc.m[0][j] = sqrt_info4[0] * jel[0][j] + sqrt_info4[1] * jel[1][j]; jeLc.m[1][j] = sqrt_info4[2] * jel[0][j] + sqrt_info4[3] * jel[1][j]; }
    const M3 Z;
    if (J[0]) {
        const M3 RbcT = Rbc.t();
        const M26 a = mul26(jeLc, RbcT, -(RbcT * skew(tbc)), Z, RbcT);                    // jaco_e_Lc * invTbc
        const M3 RwbT = Rwb.t();
        const M26 r = mul26(a, RwbT * skew(lw.v), skew(RwbT * (lw.n + skew(lw.v) * twb)), Z, skew(RwbT * lw.v));
        for (int i = 0; i < 2; ++i) { for (int j = 0; j < 6; ++j) J[0][i * 7 + j] = r.m[i][j]; J[0][i * 7 + 6] = 0; }
    }
    if (J[1]) {
        const M3 RbcT = Rbc.t();
        const M26 r = mul26(jeLc, RbcT * skew(lb.v), skew(RbcT * (lb.n + skew(lb.v) * tbc)), Z, skew(RbcT * lb.v));
        for (int i = 0; i < 2; ++i) { for (int j = 0; j < 6; ++j) J[1][i * 7 + j] = r.m[i][j]; J[1][i * 7 + 6] = 0; }
    }
    if (J[2]) {
        const M3 Rwc = Rwb * Rbc; const V3 twc = Rwb * tbc + twb;
        const M3 RwcT = Rwc.t();
        const M26 a = mul26(jeLc, RwcT, -(RwcT * skew(twc)), Z, RwcT);                    // jaco_e_Lc * invTwc
        const double nn = lw.n.norm(), vn = lw.v.norm();
        const V3 u1 = lw.n / nn, u2 = lw.v / vn, u3 = u1.cross(u2);
        const double wn = std::sqrt(nn * nn + vn * vn), w0 = nn / wn, w1 = vn / wn;
        // jaco_Lw_orth (6x4): columns
        const V3 top[4] = { V3(0, 0, 0), u3 * (-w0), u2 * w0, u1 * (-w1) };
        const V3 bot[4] = { u3 * w1, V3(0, 0, 0), u1 * (-w1), u2 * w0 };
        for (int i = 0; i < 2; ++i)
            for (int c = 0; c < 4; ++c) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += a.m[i][k] * top[c][k] + a.m[i][3 + k] * bot[c][k];
                J[2][i * 4 + c] = s;
            }
    }
}

void dvo_line_plus(const double* x, const double* delta, double* out) {      // LineOrthParameterization::Plus
    const double s1 = std::sin(x[0]), c1 = std::cos(x[0]), s2 = std::sin(x[1]), c2 = std::cos(x[1]), s3 = std::sin(x[2]), c3 = std::cos(x[2]);
    M3 R;
    R(0, 0) = c2 * c3; R(0, 1) = s1 * s2 * c3 - c1 * s3; R(0, 2) = c1 * s2 * c3 + s1 * s3;
    R(1, 0) = c2 * s3; R(1, 1) = s1 * s2 * s3 + c1 * c3; R(1, 2) = c1 * s2 * s3 - s1 * c3;
    R(2, 0) = -s2;     R(2, 1) = s1 * c2;                R(2, 2) = c1 * c2;
    const double w1 = std::cos(x[3]), w2 = std::sin(x[3]);
    M3 Rz, Ry, Rx;
    Rz(0, 0) = std::cos(delta[2]); Rz(0, 1) = -std::sin(delta[2]); Rz(1, 0) = std::sin(delta[2]); Rz(1, 1) = std::cos(delta[2]); Rz(2, 2) = 1;
    Ry(0, 0) = std::cos(delta[1]); Ry(0, 2) = std::sin(delta[1]); Ry(1, 1) = 1; Ry(2, 0) = -std::sin(delta[1]); Ry(2, 2) = std::cos(delta[1]);
    Rx(0, 0) = 1; Rx(1, 1) = std::cos(delta[0]); Rx(1, 2) = -std::sin(delta[0]); Rx(2, 1) = std::sin(delta[0]); Rx(2, 2) = std::cos(delta[0]);
    R = R * Rx * Ry * Rz;
    const double cd = std::cos(delta[3]), sd = std::sin(delta[3]);
    const double W10 = w2 * cd + w1 * sd;                   // (W * delta_W)(1,0)
    const V3 u1 = R.col(0), u2 = R.col(1), u3 = R.col(2);
    out[0] = std::atan2(u2.z, u3.z); out[1] = std::asin(-u1.z); out[2] = std::atan2(u1.y, u1.x);
    out[3] = std::asin(W10);
}

// par: pose_obj[7]; J: 3x7
void dvo_box_enclose_eval(const double* pts_w, const double* dims, const double* const* par, double* res3, double** J) {
    const V3 P = P3(par[0]); const Q q = Q4(par[0]);
    const V3 po = q.inverse() * (P3(pts_w) - P);
    const V3 err = V3(std::fabs(po.x) - dims[0] / 2, std::fabs(po.y) - dims[1] / 2, std::fabs(po.z) - dims[2] / 2) * 10.0;
    res3[0] = std::max(0.0, err.x); res3[1] = std::max(0.0, err.y); res3[2] = std::max(0.0, err.z);
    if (J && J[0]) {
        const M3 Rojw = q.inverse().R();
        const V3 e = Rojw * (po - P);
        const double np[3] = { e.x / std::fabs(e.x), e.y / std::fabs(e.y), e.z / std::fabs(e.z) };
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) J[0][i * 7 + j] = np[i] * Rojw(i, j); for (int j = 3; j < 7; ++j) J[0][i * 7 + j] = 0; }
    }
}

// par: box[3]; J: 1x3
void dvo_box_dims_eval(const double* dims, const double* const* par, double* res1, double** J) {
    const V3 d = P3(par[0]) - P3(dims);
    const double err = d.dot(d);
    res1[0] = err * err / 100.0;
    if (J && J[0]) { J[0][0] = 2 * d.x; J[0][1] = 2 * d.y; J[0][2] = 2 * d.z; }
}

// par: pose_body[7], pose_obj[7]; J: 3x7, 3x7
void dvo_box_orientation_eval(const double* R_cioi9, const double* R_bc9, const double* const* par, double* res3, double** J) {
    M3 Rcioi, Rbc;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Rcioi(i, j) = R_cioi9[i * 3 + j]; Rbc(i, j) = R_bc9[i * 3 + j]; }
    const M3 Rwbi = Q4(par[0]).R(), Rwoi = Q4(par[1]).R();
    const M3 R = Rwoi.t() * Rwbi * Rbc * Rcioi;
    const V3 err = so3_log(R);
    res3[0] = err.x; res3[1] = err.y; res3[2] = err.z;
    if (!J) return;
    if (J[0]) for (int k = 0; k < 21; ++k) J[0][k] = 0;
    if (J[1]) {
        const V3 phi = err;
        const double theta = -phi.norm();
        const V3 a = phi.normalized();
        const double st = std::sin(theta) / theta, ct = (1 - std::cos(theta) / theta);
        M3 aat; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) aat(i, j) = a[i] * a[j];
        const M3 Jr = M3::identity() * st + aat * (1 - st) + skew(a) * ct;
        // inverse of a 3x3 (Eigen's closed form)
        const double det = Jr(0, 0) * (Jr(1, 1) * Jr(2, 2) - Jr(1, 2) * Jr(2, 1)) - Jr(0, 1) * (Jr(1, 0) * Jr(2, 2) - Jr(1, 2) * Jr(2, 0)) + Jr(0, 2) * (Jr(1, 0) * Jr(2, 1) - Jr(1, 1) * Jr(2, 0));
        M3 inv;
        inv(0, 0) = (Jr(1, 1) * Jr(2, 2) - Jr(1, 2) * Jr(2, 1)) / det; inv(0, 1) = (Jr(0, 2) * Jr(2, 1) - Jr(0, 1) * Jr(2, 2)) / det; inv(0, 2) = (Jr(0, 1) * Jr(1, 2) - Jr(0, 2) * Jr(1, 1)) / det;
        inv(1, 0) = (Jr(1, 2) * Jr(2, 0) - Jr(1, 0) * Jr(2, 2)) / det; inv(1, 1) = (Jr(0, 0) * Jr(2, 2) - Jr(0, 2) * Jr(2, 0)) / det; inv(1, 2) = (Jr(0, 2) * Jr(1, 0) - Jr(0, 0) * Jr(1, 2)) / det;
        inv(2, 0) = (Jr(1, 0) * Jr(2, 1) - Jr(1, 1) * Jr(2, 0)) / det; inv(2, 1) = (Jr(0, 1) * Jr(2, 0) - Jr(0, 0) * Jr(2, 1)) / det; inv(2, 2) = (Jr(0, 0) * Jr(1, 1) - Jr(0, 1) * Jr(1, 0)) / det;
        const M3 jac = -(inv * R.t());
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) { J[1][i * 7 + j] = 0; J[1][i * 7 + 3 + j] = jac(i, j); } J[1][i * 7 + 6] = 0; }
    }
}

}  // extern "C"
