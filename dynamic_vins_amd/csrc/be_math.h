// be_math.h — fixed-size fp64 vector / matrix / quaternion helpers shared by the host estimator and the
// HIP back-end kernels (__host__ __device__).  Product code; independent of the oracle's la.h.
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BE_HD __host__ __device__ __forceinline__
#else
#define BE_HD inline
#endif

namespace be {

struct d3 { double x, y, z; };
BE_HD d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
BE_HD d3 operator+(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
BE_HD d3 operator-(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
BE_HD d3 operator-(d3 a) { return mk3(-a.x, -a.y, -a.z); }
BE_HD d3 operator*(d3 a, double s) { return mk3(a.x * s, a.y * s, a.z * s); }
BE_HD d3 operator*(double s, d3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
BE_HD d3 operator/(d3 a, double s) { return mk3(a.x / s, a.y / s, a.z / s); }
BE_HD double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
BE_HD d3 cross(d3 a, d3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
BE_HD double norm(d3 a) { return sqrt(dot(a, a)); }
BE_HD double get(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct m33 { double m[9]; };      // row-major
BE_HD m33 eye3() { m33 r; for (int i = 0; i < 9; ++i) r.m[i] = 0; r.m[0] = r.m[4] = r.m[8] = 1; return r; }
BE_HD m33 zero3() { m33 r; for (int i = 0; i < 9; ++i) r.m[i] = 0; return r; }
BE_HD m33 mul(const m33& a, const m33& b) {
    m33 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
    return r;
}
BE_HD d3 mul(const m33& a, d3 v) { return mk3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z); }
BE_HD m33 tr(const m33& a) { m33 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = a.m[j * 3 + i]; return r; }
BE_HD m33 scale(const m33& a, double s) { m33 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] * s; return r; }
BE_HD m33 add(const m33& a, const m33& b) { m33 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
BE_HD m33 sub(const m33& a, const m33& b) { m33 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] - b.m[i]; return r; }
BE_HD m33 skew(d3 q) { m33 r = zero3(); r.m[1] = -q.z; r.m[2] = q.y; r.m[3] = q.z; r.m[5] = -q.x; r.m[6] = -q.y; r.m[7] = q.x; return r; }

struct quat { double w, x, y, z; };
BE_HD quat mkq(double w, double x, double y, double z) { quat q; q.w = w; q.x = x; q.y = y; q.z = z; return q; }
BE_HD quat qmul(quat a, quat b) {
    return mkq(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
               a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}
BE_HD quat qinv(quat q) { double n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; return mkq(q.w / n, -q.x / n, -q.y / n, -q.z / n); }
BE_HD quat qnormalized(quat q) { double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z); return mkq(q.w / n, q.x / n, q.y / n, q.z / n); }
BE_HD d3 qvec(quat q) { return mk3(q.x, q.y, q.z); }
BE_HD d3 qrot(quat q, d3 v) { d3 u = qvec(q); d3 uv = cross(u, v); uv = uv + uv; return v + uv * q.w + cross(u, uv); }
BE_HD m33 qR(quat q) {
    m33 r;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    r.m[0] = 1 - (tyy + tzz); r.m[1] = txy - twz; r.m[2] = txz + twy;
    r.m[3] = txy + twz; r.m[4] = 1 - (txx + tzz); r.m[5] = tyz - twx;
    r.m[6] = txz - twy; r.m[7] = tyz + twx; r.m[8] = 1 - (txx + tyy);
    return r;
}
BE_HD quat qfromR(const m33& m) {
    quat q; double t = m.m[0] + m.m[4] + m.m[8];
    if (t > 0) { t = sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t; q.x = (m.m[7] - m.m[5]) * t; q.y = (m.m[2] - m.m[6]) * t; q.z = (m.m[3] - m.m[1]) * t; }
    else {
        int i = 0; if (m.m[4] > m.m[0]) i = 1; if (m.m[8] > m.m[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m.m[i * 4] - m.m[j * 4] - m.m[k * 4] + 1.0);
        double v[3]; v[i] = 0.5 * t; t = 0.5 / t;
        q.w = (m.m[k * 3 + j] - m.m[j * 3 + k]) * t; v[j] = (m.m[j * 3 + i] + m.m[i * 3 + j]) * t; v[k] = (m.m[k * 3 + i] + m.m[i * 3 + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
// Utility::R2ypr / ypr2R (estimator/utility.h:86-131), degrees
BE_HD d3 r2ypr(const m33& R) {
    const d3 n = mk3(R.m[0], R.m[3], R.m[6]), o = mk3(R.m[1], R.m[4], R.m[7]), a = mk3(R.m[2], R.m[5], R.m[8]);
    const double y = atan2(n.y, n.x), p = atan2(-n.z, n.x * cos(y) + n.y * sin(y)), r = atan2(a.x * sin(y) - a.y * cos(y), -o.x * sin(y) + o.y * cos(y));
    return mk3(y, p, r) / M_PI * 180.0;
}
BE_HD m33 ypr2r(d3 ypr) {
    const double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
    m33 Rz = zero3(), Ry = zero3(), Rx = zero3();
    Rz.m[0] = cos(y); Rz.m[1] = -sin(y); Rz.m[3] = sin(y); Rz.m[4] = cos(y); Rz.m[8] = 1;
    Ry.m[0] = cos(p); Ry.m[2] = sin(p); Ry.m[4] = 1; Ry.m[6] = -sin(p); Ry.m[8] = cos(p);
    Rx.m[0] = 1; Rx.m[4] = cos(r); Rx.m[5] = -sin(r); Rx.m[7] = sin(r); Rx.m[8] = cos(r);
    return mul(mul(Rz, Ry), Rx);
}
BE_HD quat dq_half(d3 theta) { return mkq(1.0, theta.x / 2.0, theta.y / 2.0, theta.z / 2.0); }      // Utility::deltaQ (un-normalised)
BE_HD d3 P3(const double* p) { return mk3(p[0], p[1], p[2]); }
BE_HD quat Q4(const double* p) { return mkq(p[6], p[3], p[4], p[5]); }        // pose block [p, qx qy qz qw]

// Sophus::SO3d(R).log() (un-vendored; so3.hpp logAndTheta): quaternion log with the small-angle series
BE_HD d3 so3_log(const m33& R) {
    quat q = qnormalized(qfromR(R));
    const double sq = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w;
    double two_atan;
    if (sq < 1e-20) two_atan = 2.0 / w - 2.0 / 3.0 * sq / (w * w * w);
    else {
        const double nq = sqrt(sq);
        if (fabs(w) < 1e-10) two_atan = (w > 0 ? M_PI : -M_PI) / nq;
        else two_atan = 2.0 * atan(nq / w) / nq;
    }
    return mk3(q.x, q.y, q.z) * two_atan;
}
// Sophus::SO3d::exp(omega).matrix() (so3.hpp expAndTheta): quaternion (cos(theta/2), sin(theta/2)/theta omega), Taylor series below 1e-10
BE_HD m33 so3_exp(d3 w) {
    const double th2 = dot(w, w);
    double imag, real;
    if (th2 < 1e-20) { const double th4 = th2 * th2; imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4; real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4; }
    else { const double th = sqrt(th2), half = 0.5 * th; imag = sin(half) / th; real = cos(half); }
    return qR(mkq(real, imag * w.x, imag * w.y, imag * w.z));
}
// Eigen::Matrix3d::inverse(): cofactors over the determinant
BE_HD m33 inv3(const m33& a) {
    const double* J = a.m; m33 r;
    const double c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
    const double det = J[0] * c00 + J[1] * c01 + J[2] * c02, id = 1.0 / det;
    r.m[0] = c00 * id; r.m[1] = (J[2] * J[7] - J[1] * J[8]) * id; r.m[2] = (J[1] * J[5] - J[2] * J[4]) * id;
    r.m[3] = c01 * id; r.m[4] = (J[0] * J[8] - J[2] * J[6]) * id; r.m[5] = (J[2] * J[3] - J[0] * J[5]) * id;
    r.m[6] = c02 * id; r.m[7] = (J[1] * J[6] - J[0] * J[7]) * id; r.m[8] = (J[0] * J[4] - J[1] * J[3]) * id;
    return r;
}

// x (+) delta for a pose block: p += dp (plane constraint drops z or y), q = normalise(q * [1, dth/2])
BE_HD void pose_plus(const double* x, const double* d, int plane_kind, double* out) {
    double dx = d[0], dy = d[1], dz = d[2];
    if (plane_kind == 1) dz = 0;          // PoseConstraintLocalParameterization with IMU
    if (plane_kind == 2) dy = 0;          // ... vision only
    out[0] = x[0] + dx; out[1] = x[1] + dy; out[2] = x[2] + dz;
    quat r = qnormalized(qmul(mkq(x[6], x[3], x[4], x[5]), dq_half(mk3(d[3], d[4], d[5]))));
    out[3] = r.x; out[4] = r.y; out[5] = r.z; out[6] = r.w;
}

// ceres::HuberLoss(1.0): rho0 (cost*2) and the residual/Jacobian scale sqrt(rho'); rho'' <= 0 so the corrector's
// alpha term vanishes (marginalization_factor.cpp:54-78)
BE_HD void huber1(double s, double& rho0, double& scale) {
    if (s > 1.0) { const double r = sqrt(s); rho0 = 2 * r - 1; scale = sqrt(1.0 / r); }
    else { rho0 = s; scale = 1.0; }
}

}  // namespace be
