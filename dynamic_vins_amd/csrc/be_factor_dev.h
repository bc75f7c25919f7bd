// be_factor_dev.h — residual + Jacobian of the three live reprojection factors and of the IMU factor as
// __host__ __device__ functions (product code).  Formulas follow
//   ProjectionTwoFrameOneCamFactor::Evaluate  dynamic_vins/src/estimator/factor/projection_two_frame_one_cam_factor.cpp:48-155
//   ProjectionTwoFrameTwoCamFactor::Evaluate  .../projection_two_frame_two_cam_factor.cpp:47-170
//   ProjectionOneFrameTwoCamFactor::Evaluate  .../projection_one_frame_two_cam_factor.cpp:48-140 (incl. quirk Q7)
//   IMUFactor::Evaluate / IntegrationBase::evaluate  .../imu/imu_factor.h:31-172, integration_base.h:175-201
// Jacobians are in the 6-dim tangent space of each pose block (the 7th column of the reference's 2x7 blocks is
// zero and PoseLocalParameterization::ComputeJacobian is [I6;0]).
#pragma once
#include "be_math.h"
#include "be_types.h"

namespace be {

#define BE_SQRT_INFO (460.0 / 1.5)     // kFocalLength / 1.5 (estimator.cpp:685-687)

struct FrameGeom { m33 R; d3 P; };

// out[2][3] = red[2][3] * M
BE_HD void red_mul(const double red[6], const m33& M, double* out, int stride, int col0) {
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
            out[r * stride + col0 + c] = red[r * 3] * M.m[c] + red[r * 3 + 1] * M.m[3 + c] + red[r * 3 + 2] * M.m[6 + c];
}
BE_HD void red_vec(const double red[6], d3 v, double* out) {
    out[0] = red[0] * v.x + red[1] * v.y + red[2] * v.z;
    out[1] = red[3] * v.x + red[4] * v.y + red[5] * v.z;
}

// r[2]; Ji,Jj: 2x6 row-major (may be null for kind 2); Jl[2]; Jex0,Jex1: 2x6; Jtd[2] (EXTD only)
template <bool JAC, bool EXTD>
BE_HD void proj_factor(const BeFactor& f, const FrameGeom& Fi, const FrameGeom& Fj, const m33& ric, d3 tic, const m33& ric2, d3 tic2,
                       double lambda, double td, double* r, double* Ji, double* Jj, double* Jl, double* Jex0, double* Jex1, double* Jtd) {
    const d3 pts_i = mk3(f.pix, f.piy, 1.0), pts_j = mk3(f.pjx, f.pjy, 1.0);
    const d3 vi = mk3(f.vix, f.viy, 0.0), vj = mk3(f.vjx, f.vjy, 0.0);
    const d3 pts_i_td = pts_i - vi * (td - f.td_i);
    const d3 pts_j_td = pts_j - vj * (td - f.td_j);
    const d3 pc_i = pts_i_td / lambda;
    const d3 p_imu_i = mul(ric, pc_i) + tic;
    const bool two_frame = f.kind != 2;
    const m33& rcj = (f.kind == 0) ? ric : ric2;
    const d3 tcj = (f.kind == 0) ? tic : tic2;
    d3 p_imu_j;
    if (two_frame) { const d3 pw = mul(Fi.R, p_imu_i) + Fi.P; p_imu_j = mul(tr(Fj.R), pw - Fj.P); }
    else p_imu_j = p_imu_i;
    const d3 pcj = mul(tr(rcj), p_imu_j - tcj);
    const double dep = pcj.z;
    r[0] = BE_SQRT_INFO * (pcj.x / dep - pts_j_td.x);
    r[1] = BE_SQRT_INFO * (pcj.y / dep - pts_j_td.y);
    if (!JAC) return;
    const double red[6] = { BE_SQRT_INFO / dep, 0.0, BE_SQRT_INFO * (-pcj.x / (dep * dep)), 0.0, BE_SQRT_INFO / dep, BE_SQRT_INFO * (-pcj.y / (dep * dep)) };
    const m33 rcT = tr(rcj);
    m33 T;
    if (two_frame) {
        const m33 A = mul(rcT, tr(Fj.R));               // rc^T Rj^T
        const m33 ARi = mul(A, Fi.R);
        red_mul(red, A, Ji, 6, 0);
        red_mul(red, scale(mul(ARi, skew(p_imu_i)), -1.0), Ji, 6, 3);
        red_mul(red, scale(A, -1.0), Jj, 6, 0);
        red_mul(red, mul(rcT, skew(p_imu_j)), Jj, 6, 3);
        T = mul(ARi, ric);
        if (EXTD) {
            if (f.kind == 0) {
                const m33 left = mul(rcT, sub(mul(tr(Fj.R), Fi.R), eye3()));
                const d3 v3 = mul(rcT, mul(tr(Fj.R), mul(Fi.R, tic) + Fi.P - Fj.P) - tic);
                const m33 right = add(add(scale(mul(T, skew(pc_i)), -1.0), skew(mul(T, pc_i))), skew(v3));
                red_mul(red, left, Jex0, 6, 0); red_mul(red, right, Jex0, 6, 3);
                for (int k = 0; k < 12; ++k) Jex1[k] = 0.0;
            } else {
                red_mul(red, ARi, Jex0, 6, 0);
                red_mul(red, scale(mul(T, skew(pc_i)), -1.0), Jex0, 6, 3);
                red_mul(red, scale(rcT, -1.0), Jex1, 6, 0);
                red_mul(red, skew(pcj), Jex1, 6, 3);
            }
        }
        red_vec(red, mul(T, pts_i_td) * (-1.0 / (lambda * lambda)), Jl);
    } else {
        T = mul(rcT, ric);
        if (Ji) for (int k = 0; k < 12; ++k) { Ji[k] = 0.0; Jj[k] = 0.0; }
        if (EXTD) {
            red_mul(red, rcT, Jex0, 6, 0);
            red_mul(red, scale(mul(T, skew(pc_i)), -1.0), Jex0, 6, 3);
            red_mul(red, scale(rcT, -1.0), Jex1, 6, 0);
            red_mul(red, skew(pcj), Jex1, 6, 3);
        }
        red_vec(red, mul(T, pts_i) * (-1.0 / (lambda * lambda)), Jl);       // Q7: pts_i, not pts_i_td
    }
    if (EXTD) {
        double t[2];
        red_vec(red, mul(T, vi) / lambda * -1.0, t);
        Jtd[0] = t[0] + BE_SQRT_INFO * vj.x; Jtd[1] = t[1] + BE_SQRT_INFO * vj.y;
    }
}

// ---- IMU factor: raw residual (15) and raw Jacobian (15 x 30, tangent space: pose_i 6, sb_i 9, pose_j 6, sb_j 9) ----
BE_HD void set33(double* J, int ld, int r0, int c0, const m33& b) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[(r0 + i) * ld + c0 + j] = b.m[i * 3 + j]; }
BE_HD m33 load33(const double* p) { m33 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
BE_HD m33 qleft_br(quat q) { return add(scale(eye3(), q.w), skew(qvec(q))); }
BE_HD m33 qlqr_br(quat a, quat b) {        // bottom-right 3x3 of Qleft(a) * Qright(b)
    double L[3][4], Rm[4][3];
    const m33 sa = skew(qvec(a)), sb = skew(qvec(b));
    const double av[3] = { a.x, a.y, a.z }, bv[3] = { b.x, b.y, b.z };
    for (int i = 0; i < 3; ++i) { L[i][0] = av[i]; for (int j = 0; j < 3; ++j) L[i][1 + j] = (i == j ? a.w : 0.0) + sa.m[i * 3 + j]; }
    for (int j = 0; j < 3; ++j) { Rm[0][j] = -bv[j]; for (int i = 0; i < 3; ++i) Rm[1 + i][j] = (i == j ? b.w : 0.0) - sb.m[i * 3 + j]; }
    m33 o;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += L[i][k] * Rm[k][j]; o.m[i * 3 + j] = s; }
    return o;
}

template <bool JAC>
BE_HD void imu_raw(const BeImu& m, double g_norm, const double* pose_i, const double* sb_i, const double* pose_j, const double* sb_j,
                   double* r15, double* J /* 15 x 30, pre-zeroed */) {
    const d3 G = mk3(0, 0, g_norm);
    const d3 Pi = P3(pose_i), Pj = P3(pose_j); const quat Qi = Q4(pose_i), Qj = Q4(pose_j);
    const d3 Vi = P3(sb_i), Bai = P3(sb_i + 3), Bgi = P3(sb_i + 6), Vj = P3(sb_j), Baj = P3(sb_j + 3), Bgj = P3(sb_j + 6);
    const m33 dp_dba = load33(m.dp_dba), dp_dbg = load33(m.dp_dbg), dq_dbg = load33(m.dq_dbg), dv_dba = load33(m.dv_dba), dv_dbg = load33(m.dv_dbg);
    const d3 dba = Bai - P3(m.lin_ba), dbg = Bgi - P3(m.lin_bg);
    const quat dq = mkq(m.dq[0], m.dq[1], m.dq[2], m.dq[3]);
    const quat cq = qmul(dq, dq_half(mul(dq_dbg, dbg)));
    const d3 cv = P3(m.dv) + mul(dv_dba, dba) + mul(dv_dbg, dbg);
    const d3 cp = P3(m.dp) + mul(dp_dba, dba) + mul(dp_dbg, dbg);
    const double dt = m.sum_dt;
    const quat Qi_inv = qinv(Qi);
    const d3 a = qrot(Qi_inv, G * (0.5 * dt * dt) + Pj - Pi - Vi * dt);
    const d3 b = qrot(Qi_inv, G * dt + Vj - Vi);
    const d3 rp = a - cp;
    const d3 rq = qvec(qmul(qinv(cq), qmul(Qi_inv, Qj))) * 2.0;
    const d3 rv = b - cv;
    const d3 rba = Baj - Bai, rbg = Bgj - Bgi;
    for (int k = 0; k < 3; ++k) { r15[k] = get(rp, k); r15[3 + k] = get(rq, k); r15[6 + k] = get(rv, k); r15[9 + k] = get(rba, k); r15[12 + k] = get(rbg, k); }
    if (!JAC) return;
    const m33 RiT = qR(Qi_inv);
    // pose_i: cols 0..5
    set33(J, 30, 0, 0, scale(RiT, -1.0));
    set33(J, 30, 0, 3, skew(a));
    set33(J, 30, 3, 3, scale(qlqr_br(qmul(qinv(Qj), Qi), cq), -1.0));
    set33(J, 30, 6, 3, skew(b));
    // sb_i: cols 6..14
    set33(J, 30, 0, 6, scale(RiT, -dt)); set33(J, 30, 0, 9, scale(dp_dba, -1.0)); set33(J, 30, 0, 12, scale(dp_dbg, -1.0));
    set33(J, 30, 3, 12, scale(mul(qleft_br(qmul(qmul(qinv(Qj), Qi), dq)), dq_dbg), -1.0));
    set33(J, 30, 6, 6, scale(RiT, -1.0)); set33(J, 30, 6, 9, scale(dv_dba, -1.0)); set33(J, 30, 6, 12, scale(dv_dbg, -1.0));
    set33(J, 30, 9, 9, scale(eye3(), -1.0)); set33(J, 30, 12, 12, scale(eye3(), -1.0));
    // pose_j: cols 15..20
    set33(J, 30, 0, 15, RiT);
    set33(J, 30, 3, 18, qleft_br(qmul(qinv(cq), qmul(Qi_inv, Qj))));
    // sb_j: cols 21..29
    set33(J, 30, 6, 21, RiT); set33(J, 30, 9, 24, eye3()); set33(J, 30, 12, 27, eye3());
}

}  // namespace be
