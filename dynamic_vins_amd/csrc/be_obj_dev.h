// be_obj_dev.h — device bodies of the dynamic-object factors, shared by the operator-level
// evaluators (be_obj.hip) and the object solve (be_objsolve.hip).  Bug-for-bug with
//   BoxEncloseStereoPointFactor::Evaluate   estimator/factor/box_factor.cpp:523-565
//   BoxDimsFactor::Evaluate                 estimator/factor/box_factor.cpp:728-743
//   BoxOrientationFactor::Evaluate          estimator/factor/box_factor.cpp:752-806
// Only the non-zero 3x3 part of each pose Jacobian is produced: the point factor's rotation columns and the
// orientation factor's position columns are identically zero in the reference.
#pragma once
#include <hip/hip_runtime.h>
#include "be_math.h"

namespace be {

// r[3] = max(0, 10 (|R_oj^T (p_w - P_oj)| - dims / 2)); Jp (3x3, row-major) = d r / d P_oj as the reference writes it:
// N_p R_ojw with N_p = sign(R_ojw (p_obj - P_woj))
__device__ __forceinline__ void box_enclose_dev(d3 pw, const double* dims, d3 P, quat q, double r[3], double Jp[9]) {
    const quat qi = qinv(q);
    const d3 po = qrot(qi, pw - P);
    r[0] = fmax(0.0, (fabs(po.x) - dims[0] / 2) * 10.0); r[1] = fmax(0.0, (fabs(po.y) - dims[1] / 2) * 10.0); r[2] = fmax(0.0, (fabs(po.z) - dims[2] / 2) * 10.0);
    const m33 Rojw = qR(qi);
    const d3 e = mul(Rojw, po - P);
    const double np[3] = { e.x / fabs(e.x), e.y / fabs(e.y), e.z / fabs(e.z) };
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) Jp[i * 3 + c] = np[i] * Rojw.m[i * 3 + c];
}

// r = |box - dims|^4 / 100, J = 2 (box - dims)^T
__device__ __forceinline__ void box_dims_dev(d3 box, d3 dims, double& r, double J[3]) {
    const d3 d = box - dims;
    const double err = dot(d, d);
    r = err * err / 100.0; J[0] = 2 * d.x; J[1] = 2 * d.y; J[2] = 2 * d.z;
}

// r = Log(R_oiw R_wbi R_bc R_cioi); Jr (3x3) = d r / d theta_obj = -J_r(theta)^-1 R^T with the reference's own J_r
__device__ __forceinline__ void box_orientation_dev(const m33& Rc, const m33& Rb, quat q_body, quat q_obj, double r[3], double Jr_out[9]) {
    const m33 Rwbi = qR(q_body), Rwoi = qR(q_obj);
    const m33 R = mul(mul(mul(tr(Rwoi), Rwbi), Rb), Rc);
    const d3 phi = so3_log(R);
    r[0] = phi.x; r[1] = phi.y; r[2] = phi.z;
    const double theta = -norm(phi);
    const double pn = norm(phi);
    const d3 a = pn > 0 ? phi / pn : phi;
    const double st = sin(theta) / theta, ct = 1 - cos(theta) / theta;
    m33 Jr;
    for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c) Jr.m[i * 3 + c] = (i == c ? st : 0.0) + get(a, i) * get(a, c) * (1 - st);
    const m33 ha = skew(a);
    for (int k = 0; k < 9; ++k) Jr.m[k] += ha.m[k] * ct;
    const double* J = Jr.m;
    const double det = J[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * J[7] - J[4] * J[6]);
    m33 inv;
    inv.m[0] = (J[4] * J[8] - J[5] * J[7]) / det; inv.m[1] = (J[2] * J[7] - J[1] * J[8]) / det; inv.m[2] = (J[1] * J[5] - J[2] * J[4]) / det;
    inv.m[3] = (J[5] * J[6] - J[3] * J[8]) / det; inv.m[4] = (J[0] * J[8] - J[2] * J[6]) / det; inv.m[5] = (J[2] * J[3] - J[0] * J[5]) / det;
    inv.m[6] = (J[3] * J[7] - J[4] * J[6]) / det; inv.m[7] = (J[1] * J[6] - J[0] * J[7]) / det; inv.m[8] = (J[0] * J[4] - J[1] * J[3]) / det;
    const m33 jac = scale(mul(inv, tr(R)), -1.0);
    for (int k = 0; k < 9; ++k) Jr_out[k] = jac.m[k];
}


// ProjectionInstanceFactor (estimator/factor/project_instance_factor.cpp:27-172): a point of a moving object seen in frame j (inverse depth
// known there) is carried into the object frame with the object's pose at j, back to the world with its pose at i, and reprojected into
// camera i.  Output (64): r[2] | J wrt body pose j, body pose i, extrinsic, object pose j, object pose i (2x6 each, tangent) | J wrt inv_dep_j (2).
// Kept as the reference writes it: the inverse-depth Jacobian carries a + sign and the un-compensated pts_j (:166).
__device__ __forceinline__ void inst_proj_dev(const double* f /* pts_j3 pts_i3 vel_j2 vel_i2 td_j td_i cur_td */, const double* pbj, const double* pbi, const double* pex,
                                              const double* poj, const double* poi, double lam, double* out) {
    const double cur_td = f[12];
    const d3 pj = mk3(f[0], f[1], f[2]), pi = mk3(f[3], f[4], f[5]);
    const d3 pj_td = pj - mk3(f[6], f[7], 0.0) * (cur_td - f[10]), pi_td = pi - mk3(f[8], f[9], 0.0) * (cur_td - f[11]);
    const quat qbj = Q4(pbj), qbi = Q4(pbi), qbc = Q4(pex), qoj = Q4(poj), qoi = Q4(poi);
    const d3 Pbj = P3(pbj), Pbi = P3(pbi), Pbc = P3(pex), Poj = P3(poj), Poi = P3(poi);
    const d3 cam_j = pj_td / lam;
    const d3 imu_j = qrot(qbc, cam_j) + Pbc;
    const d3 w_j = qrot(qbj, imu_j) + Pbj;
    const d3 obj_j = qrot(qinv(qoj), w_j - Poj);
    const d3 w_i = qrot(qoi, obj_j) + Poi;
    const d3 imu_i = qrot(qinv(qbi), w_i - Pbi);
    const d3 cam_i = qrot(qinv(qbc), imu_i - Pbc);
    const double dep = cam_i.z, si = 460.0 / 1.5;
    out[0] = si * (cam_i.x / dep - pi_td.x); out[1] = si * (cam_i.y / dep - pi_td.y);
    const double red[6] = { si * (1.0 / dep), 0.0, si * (-cam_i.x / (dep * dep)), 0.0, si * (1.0 / dep), si * (-cam_i.y / (dep * dep)) };
    const m33 Rbj = qR(qbj), Rbi = qR(qbi), Rbit = tr(Rbi), Rbc = qR(qbc), Rcb = tr(Rbc), Roj = qR(qoj), Rojt = tr(Roj), Roi = qR(qoi);
    const m33 CbBi = mul(Rcb, Rbit);                         // R_cb R_biw
    const m33 T_oi = mul(CbBi, Roi);                         // ... R_woi
    const m33 T_oj = mul(T_oi, Rojt);                        // ... R_ojw
    const m33 T_bj = mul(T_oj, Rbj);                         // ... R_wbj
    auto store = [&](double* o, const m33& A, const m33& B) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                o[r * 6 + c] = red[r * 3] * A.m[c] + red[r * 3 + 1] * A.m[3 + c] + red[r * 3 + 2] * A.m[6 + c];
                o[r * 6 + 3 + c] = red[r * 3] * B.m[c] + red[r * 3 + 1] * B.m[3 + c] + red[r * 3 + 2] * B.m[6 + c];
            }
    };
    store(out + 2, T_oj, scale(mul(T_bj, skew(imu_j)), -1.0));
    store(out + 14, scale(CbBi, -1.0), mul(Rcb, skew(mul(Rbit, w_i - Pbi))));
    store(out + 26, sub(T_bj, Rcb), add(scale(mul(mul(T_bj, Rbc), skew(cam_j)), -1.0), skew(mul(Rcb, imu_i - Pbc))));
    store(out + 38, scale(T_oj, -1.0), mul(T_oi, skew(mul(Rojt, w_j - Poj))));
    store(out + 50, CbBi, scale(mul(T_oi, skew(obj_j)), -1.0));
    const d3 v = mul(mul(T_bj, Rbc), pj) / (lam * lam);
    out[62] = red[0] * v.x + red[1] * v.y + red[2] * v.z; out[63] = red[3] * v.x + red[4] * v.y + red[5] * v.z;
}

// ---- line factor pieces (lineProjectionFactor / LineOrthParameterization) ----
struct Plk { d3 n, v; };

__device__ __forceinline__ m33 orth_R(const double* o) {
    const double s1 = sin(o[0]), c1 = cos(o[0]), s2 = sin(o[1]), c2 = cos(o[1]), s3 = sin(o[2]), c3 = cos(o[2]);
    m33 R;
    R.m[0] = c2 * c3; R.m[1] = s1 * s2 * c3 - c1 * s3; R.m[2] = c1 * s2 * c3 + s1 * s3;
    R.m[3] = c2 * s3; R.m[4] = s1 * s2 * s3 + c1 * c3; R.m[5] = c1 * s2 * s3 - s1 * c3;
    R.m[6] = -s2;     R.m[7] = s1 * c2;                R.m[8] = c1 * c2;
    return R;
}
__device__ __forceinline__ d3 col(const m33& R, int j) { return mk3(R.m[j], R.m[3 + j], R.m[6 + j]); }
__device__ __forceinline__ Plk plk_to_pose(const Plk& w, const m33& Rcw, d3 tcw) { Plk r; r.v = mul(Rcw, w.v); r.n = mul(Rcw, w.n) + mul(skew(tcw), r.v); return r; }
__device__ __forceinline__ Plk plk_from_pose(const Plk& c, const m33& Rcw, d3 tcw) { const m33 Rwc = tr(Rcw); return plk_to_pose(c, Rwc, -mul(Rwc, tcw)); }

// (2x6) * [[A B] [0 D]]  (every 6x6 the line factor multiplies with has a zero lower-left block)
__device__ __forceinline__ void mul26(const double a[2][6], const m33& A, const m33& B, const m33& D, double r[2][6]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { s0 += a[i][k] * A.m[k * 3 + j]; s1 += a[i][k] * B.m[k * 3 + j] + a[i][3 + k] * D.m[k * 3 + j]; }
            r[i][j] = s0; r[i][3 + j] = s1;
        }
}


// LineOrthParameterization::Plus (estimator/factor/line_parameterization.cpp:9-72): out = orth (+) delta
__device__ __forceinline__ void line_plus_dev(const double* x, const double* d, double* o) {
    m33 R = orth_R(x);
    const double w1 = cos(x[3]), w2 = sin(x[3]);
    m33 Rz = zero3(), Ry = zero3(), Rx = zero3();
    Rz.m[0] = cos(d[2]); Rz.m[1] = -sin(d[2]); Rz.m[3] = sin(d[2]); Rz.m[4] = cos(d[2]); Rz.m[8] = 1;
    Ry.m[0] = cos(d[1]); Ry.m[2] = sin(d[1]); Ry.m[4] = 1; Ry.m[6] = -sin(d[1]); Ry.m[8] = cos(d[1]);
    Rx.m[0] = 1; Rx.m[4] = cos(d[0]); Rx.m[5] = -sin(d[0]); Rx.m[7] = sin(d[0]); Rx.m[8] = cos(d[0]);
    R = mul(mul(mul(R, Rx), Ry), Rz);
    const double W10 = w2 * cos(d[3]) + w1 * sin(d[3]);
    const d3 u1 = col(R, 0), u2 = col(R, 1), u3 = col(R, 2);
    o[0] = atan2(u2.z, u3.z); o[1] = asin(-u1.z); o[2] = atan2(u1.y, u1.x); o[3] = asin(W10);
}

// residual and d r / d orth (2x4, row-major) of lineProjectionFactor (line_projection_factor.cpp:24-159) for given body / extrinsic poses:
// the part the line-only solve needs (poses are constant there); the same arithmetic as line_eval_kernel in be_obj.hip
__device__ __forceinline__ void line_orth_dev(const double* obs, const double* si, const m33& Rwb, d3 twb, const m33& Rbc, d3 tbc, const double* orth, double r[2], double Jo[8]) {
    const m33 U = orth_R(orth);
    const double w1 = cos(orth[3]), w2 = sin(orth[3]);
    Plk lw; lw.n = col(U, 0) * w1; lw.v = col(U, 1) * w2;
    const Plk lb = plk_from_pose(lw, Rwb, twb);
    const Plk lc = plk_from_pose(lb, Rbc, tbc);
    const d3 nc = lc.n;
    const double l_norm = nc.x * nc.x + nc.y * nc.y, l_sqrt = sqrt(l_norm), l_tri = l_norm * l_sqrt;
    const double e1 = obs[0] * nc.x + obs[1] * nc.y + nc.z, e2 = obs[2] * nc.x + obs[3] * nc.y + nc.z;
    const double r0 = e1 / l_sqrt, r1 = e2 / l_sqrt;
    r[0] = si[0] * r0 + si[1] * r1; r[1] = si[2] * r0 + si[3] * r1;
    const double jel[2][3] = { { obs[0] / l_sqrt - nc.x * e1 / l_tri, obs[1] / l_sqrt - nc.y * e1 / l_tri, 1.0 / l_sqrt },
                               { obs[2] / l_sqrt - nc.x * e2 / l_tri, obs[3] / l_sqrt - nc.y * e2 / l_tri, 1.0 / l_sqrt } };
    double jeLc[2][6];
#pragma unroll
    for (int j = 0; j < 3; ++j) { jeLc[0][j] = si[0] * jel[0][j] + si[1] * jel[1][j]; jeLc[1][j] = si[2] * jel[0][j] + si[3] * jel[1][j]; jeLc[0][3 + j] = 0; jeLc[1][3 + j] = 0; }
    const m33 Rwc = mul(Rwb, Rbc); const d3 twc = mul(Rwb, tbc) + twb;
    const m33 RwcT = tr(Rwc);
    double a[2][6];
    mul26(jeLc, RwcT, scale(mul(RwcT, skew(twc)), -1.0), RwcT, a);
    const double nn = norm(lw.n), vn = norm(lw.v);
    const d3 u1 = lw.n / nn, u2 = lw.v / vn, u3 = cross(u1, u2);
    const double wn = sqrt(nn * nn + vn * vn), w0 = nn / wn, w1n = vn / wn;
    const d3 top[4] = { mk3(0, 0, 0), u3 * (-w0), u2 * w0, u1 * (-w1n) };
    const d3 bot[4] = { u3 * w1n, mk3(0, 0, 0), u1 * (-w1n), u2 * w0 };
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            Jo[q * 4 + c] = a[q][0] * top[c].x + a[q][1] * top[c].y + a[q][2] * top[c].z + a[q][3] * bot[c].x + a[q][4] * bot[c].y + a[q][5] * bot[c].z;
}

}  // namespace be
