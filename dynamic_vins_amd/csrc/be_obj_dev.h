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

__device__ __forceinline__ d3 so3_log(const m33& R) {        // Sophus::SO3d(R).log(): quaternion log with the small-angle series
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

}  // namespace be
