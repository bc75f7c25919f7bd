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

// ---- line geometry (line_detector/line_geometry.cpp:75-296) and two-view line triangulation (estimator/vio_util.cpp:447-561) ----
void dvo_plk_to_orth(const double* plk6, double* orth4) {
    const V3 n(plk6[0], plk6[1], plk6[2]), v(plk6[3], plk6[4], plk6[5]);
    const V3 u1 = n / n.norm(), u2 = v / v.norm(), u3 = u1.cross(u2);
    const double nn = n.norm(), vn = v.norm(), wn = std::sqrt(nn * nn + vn * vn);
    orth4[0] = std::atan2(u2.z, u3.z); orth4[1] = std::asin(-u1.z); orth4[2] = std::atan2(u1.y, u1.x); orth4[3] = std::asin(vn / wn);
}
void dvo_orth_to_plk(const double* orth4, double* plk6) { const Plk p = orth_to_plk(orth4); for (int k = 0; k < 3; ++k) { plk6[k] = p.n[k]; plk6[3 + k] = p.v[k]; } }
static void pi_from_ppp(const V3& x1, const V3& x2, const V3& x3, double pi[4]) { const V3 n = (x1 - x3).cross(x2 - x3); pi[0] = n.x; pi[1] = n.y; pi[2] = n.z; pi[3] = -x3.dot(x1.cross(x2)); }
static void pipi_plk(const double a[4], const double b[4], double plk[6]) {
    auto dp = [&](int i, int j) { return a[i] * b[j] - b[i] * a[j]; };
    plk[0] = dp(0, 3); plk[1] = dp(1, 3); plk[2] = dp(2, 3); plk[3] = -dp(1, 2); plk[4] = dp(0, 2); plk[5] = -dp(0, 1);
}
int dvo_line_trimming(const double* plk6, const double* obs4, double* p1, double* p2) {
    const V3 nc(plk6[0], plk6[1], plk6[2]), vc(plk6[3], plk6[4], plk6[5]);
    const M3 S = skew(nc);
    auto Lc = [&](const double pi[4], double e[4]) { for (int r = 0; r < 3; ++r) e[r] = S(r, 0) * pi[0] + S(r, 1) * pi[1] + S(r, 2) * pi[2] + vc[r] * pi[3]; e[3] = -(vc.x * pi[0] + vc.y * pi[1] + vc.z * pi[2]); };
    const V3 p11(obs4[0], obs4[1], 1.0), p21(obs4[2], obs4[3], 1.0);
    const V3 c = p11.cross(p21);
    const double ln = std::sqrt(c.x * c.x + c.y * c.y), lx = c.x / ln, ly = c.y / ln;
    const V3 p12(p11.x + lx, p11.y + ly, 1.0), p22(p21.x + lx, p21.y + ly, 1.0), cam(0, 0, 0);
    double pi1[4], pi2[4], e1[4], e2[4];
    pi_from_ppp(cam, p11, p12, pi1); pi_from_ppp(cam, p21, p22, pi2);
    Lc(pi1, e1); Lc(pi2, e2);
    for (int k = 0; k < 3; ++k) { p1[k] = e1[k] / e1[3]; p2[k] = e2[k] / e2[3]; }
    return p1[2] >= 0 && p2[2] >= 0;
}
// obs: nobs x 4 (frame start_frame + k); Rs: 11 x 9 row-major, Ps: 11 x 3; returns 1 and fills plk6 / ptw1 / ptw2 on success
int dvo_triangulate_line(const double* obs, int nobs, int start_frame, const double* Rs, const double* Ps, const double* ric9, const double* tic3,
                         double* plk6, double* ptw1, double* ptw2) {
    auto M = [](const double* p) { M3 m; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = p[i * 3 + j]; return m; };
    const M3 ric = M(ric9); const V3 tic(tic3[0], tic3[1], tic3[2]);
    const int i = start_frame;
    const M3 Ri = M(Rs + 9 * i); const V3 Pi(Ps[3 * i], Ps[3 * i + 1], Ps[3 * i + 2]);
    const V3 t0 = Pi + Ri * tic; const M3 R0 = Ri * ric;
    double min_cos = 1.0, pii[4] = { 0, 0, 0, 0 }; V3 ni, tij; M3 Rij; const double* obsj = nullptr;
    for (int k = 0; k < nobs; ++k) {
        const double* o = obs + 4 * k; const int j = i + k;
        if (k == 0) { pi_from_ppp(V3(o[0], o[1], 1), V3(o[2], o[3], 1), V3(0, 0, 0), pii); ni = V3(pii[0], pii[1], pii[2]).normalized(); continue; }
        const M3 Rj = M(Rs + 9 * j); const V3 Pj(Ps[3 * j], Ps[3 * j + 1], Ps[3 * j + 2]);
        const V3 t1 = Pj + Rj * tic; const M3 R1 = Rj * ric;
        const V3 t = R0.t() * (t1 - t0); const M3 R = R0.t() * R1;
        const V3 p3 = R * V3(o[0], o[1], 1) + t, p4 = R * V3(o[2], o[3], 1) + t;
        double pij[4]; pi_from_ppp(p3, p4, t, pij);
        const V3 nj = V3(pij[0], pij[1], pij[2]).normalized();
        const double c = ni.dot(nj);
        if (c < min_cos) { min_cos = c; tij = t; Rij = R; obsj = o; }
    }
    if (min_cos > 0.998 || !obsj) return 0;
    const V3 p3 = Rij * V3(obsj[0], obsj[1], 1) + tij, p4 = Rij * V3(obsj[2], obsj[3], 1) + tij;
    double pij[4]; pi_from_ppp(p3, p4, tij, pij);
    pipi_plk(pii, pij, plk6);
    double e1[3], e2[3];
    if (!dvo_line_trimming(plk6, obs, e1, e2)) return 0;
    const V3 a(e1[0], e1[1], e1[2]), b(e2[0], e2[1], e2[2]);
    if ((a - b).norm() > 10.0) return 0;
    const V3 w1 = Ri * (ric * a + tic) + Pi, w2 = Ri * (ric * b + tic) + Pi;
    for (int k = 0; k < 3; ++k) { ptw1[k] = w1[k]; ptw2[k] = w2[k]; }
    return 1;
}


// ProjectionInstanceFactor::Evaluate (estimator/factor/project_instance_factor.cpp:27-172): reprojection of an object point from
// frame j into frame i THROUGH the object's two poses.  Dead in the reference (every AddResidualBlock that would use it is commented out,
// estimator_insts.cpp:1258-1419), restated because north_star names the "dynamic-InstanceFactor residual+Jacobian".
// obs12 = pts_j(3) pts_i(3) vel_j(2) vel_i(2) td_j td_i ; cur_td ; par = pose_bj, pose_bi, ex_bc, pose_oj, pose_oi (7 each), inv_dep_j (1).
// J[k] row-major 2x7 (2x1 for the depth).  Bug-for-bug: d r / d inv_dep_j has a PLUS sign and uses pts_j, not pts_j_td (:166).
void dvo_inst_proj_eval(const double* obs12, double cur_td, const double* const* par, double* res2, double** J) {
    const V3 pts_j(obs12[0], obs12[1], obs12[2]), pts_i(obs12[3], obs12[4], obs12[5]);
    const V3 vel_j(obs12[6], obs12[7], 0), vel_i(obs12[8], obs12[9], 0);
    const double td_j = obs12[10], td_i = obs12[11];
    const V3 P_wbj = P3(par[0]), P_wbi = P3(par[1]), P_bc = P3(par[2]), P_woj = P3(par[3]), P_woi = P3(par[4]);
    const Q Q_wbj = Q4(par[0]), Q_wbi = Q4(par[1]), Q_bc = Q4(par[2]), Q_woj = Q4(par[3]), Q_woi = Q4(par[4]);
    const double inv_dep_j = par[5][0];
    const V3 pts_i_td = pts_i - vel_i * (cur_td - td_i), pts_j_td = pts_j - vel_j * (cur_td - td_j);
    const V3 pts_cam_j = pts_j_td / inv_dep_j;
    const V3 pts_imu_j = Q_bc * pts_cam_j + P_bc;
    const V3 pts_w_j = Q_wbj * pts_imu_j + P_wbj;
    const V3 pts_obj_j = Q_woj.inverse() * (pts_w_j - P_woj);
    const V3 pts_w_i = Q_woi * pts_obj_j + P_woi;
    const V3 pts_imu_i = Q_wbi.inverse() * (pts_w_i - P_wbi);
    const V3 pts_cam_i = Q_bc.inverse() * (pts_imu_i - P_bc);
    const double dep_i = pts_cam_i.z, inv_dep_i = 1.0 / dep_i;
    const double si = 460.0 / 1.5;                       // sqrt_info = kFocalLength / 1.5 * I (project_instance_factor.h:49)
    res2[0] = si * (pts_cam_i.x / dep_i - pts_i_td.x); res2[1] = si * (pts_cam_i.y / dep_i - pts_i_td.y);
    if (!J) return;
    const M3 R_wbj = Q_wbj.R(), R_wbi = Q_wbi.R(), R_biw = R_wbi.t(), R_bc = Q_bc.R(), R_cb = R_bc.t(), R_woj = Q_woj.R(), R_ojw = R_woj.t(), R_woi = Q_woi.R();
    double red[2][3] = { { si * inv_dep_i, 0, si * (-pts_cam_i.x / (dep_i * dep_i)) }, { 0, si * inv_dep_i, si * (-pts_cam_i.y / (dep_i * dep_i)) } };
    auto put = [&](double* out, const M3& A, const M3& B) {           // out (2x7) = reduce * [A | B], last column zero
        for (int r = 0; r < 2; ++r) {
            for (int c = 0; c < 3; ++c) { double a = 0, b = 0; for (int k = 0; k < 3; ++k) { a += red[r][k] * A(k, c); b += red[r][k] * B(k, c); } out[r * 7 + c] = a; out[r * 7 + 3 + c] = b; }
            out[r * 7 + 6] = 0;
        }
    };
    if (J[0]) { const M3 temp = R_cb * R_biw * R_woi * R_ojw; put(J[0], temp, -(temp * R_wbj * skew(pts_imu_j))); }
    if (J[1]) put(J[1], -(R_cb * R_biw), R_cb * skew(R_biw * (pts_w_i - P_wbi)));
    if (J[2]) { const M3 temp = R_cb * R_biw * R_woi * R_ojw * R_wbj; put(J[2], temp - R_cb, -(temp * R_bc * skew(pts_cam_j)) + skew(R_cb * (pts_imu_i - P_bc))); }
    if (J[3]) { const M3 temp = R_cb * R_biw * R_woi; put(J[3], -(temp * R_ojw), temp * skew(R_ojw * (pts_w_j - P_woj))); }
    if (J[4]) put(J[4], R_cb * R_biw, -(R_cb * R_biw * R_woi * skew(pts_obj_j)));
    if (J[5]) {
        const V3 v = (R_cb * R_biw * R_woi * R_ojw * R_wbj * R_bc * pts_j) / (inv_dep_j * inv_dep_j);      // sic: + sign, pts_j
        for (int r = 0; r < 2; ++r) J[5][r] = red[r][0] * v.x + red[r][1] * v.y + red[r][2] * v.z;
    }
}

}  // extern "C"
