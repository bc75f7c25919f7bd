// be_obj.hip — line and dynamic-object factors of the path on gfx950: residual + Jacobian
// evaluation, one thread per residual block, operator-level entry points (dv_line_eval, dv_line_plus,
// dv_box_enclose_eval, dv_box_dims_eval, dv_box_orientation_eval).  Replaces the Evaluate() bodies of
//   lineProjectionFactor            estimator/factor/line_projection_factor.cpp:24-159  (+ line_detector/line_geometry.cpp:97-135,210-229)
//   LineOrthParameterization::Plus  estimator/factor/line_parameterization.cpp:9-72
//   BoxEncloseStereoPointFactor     estimator/factor/box_factor.cpp:523-565
//   BoxDimsFactor                   estimator/factor/box_factor.cpp:728-743
//   BoxOrientationFactor            estimator/factor/box_factor.cpp:752-806
// bug-for-bug where the reference's Jacobians are not the derivative of its residual: N_p of the box
// factor is built from R_ojw (p_obj - P_woj); BoxDims returns 2 (box - dims)^T for r = |box - dims|^4 / 100; the
// orientation factor's camera-pose Jacobian is zero and its J_r uses (1 - cos(theta) / theta).
// Jacobians are written in LOCAL sizes (pose blocks 6 wide: the reference's 7th column is always zero).
// These factors are a few hundred blocks per frame at most (K objects x points): the kernels are latency-bound
// one-launch-per-batch evaluators; the HBM layout is struct-of-arrays per argument, coalesced per lane.
#include <hip/hip_runtime.h>
#include "dv_ctx.h"
#include "be_math.h"
#include "be_obj_dev.h"

using namespace be;

namespace {

// out per block (34): r[2] | J_pose 2x6 | J_ex 2x6 | J_orth 2x4
__global__ void line_eval_kernel(const dv_line_factor* f, int n, const double* pose, const double* ex, const double* orth, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* obs = f[i].obs; const double* si = f[i].sqrt_info;
    const m33 Rwb = qR(Q4(pose + 7 * i)); const d3 twb = P3(pose + 7 * i);
    const m33 Rbc = qR(Q4(ex + 7 * i)); const d3 tbc = P3(ex + 7 * i);
    const m33 U = orth_R(orth + 4 * i);
    const double w1 = cos(orth[4 * i + 3]), w2 = sin(orth[4 * i + 3]);
    Plk lw; lw.n = col(U, 0) * w1; lw.v = col(U, 1) * w2;
    const Plk lb = plk_from_pose(lw, Rwb, twb);
    const Plk lc = plk_from_pose(lb, Rbc, tbc);
    const d3 nc = lc.n;
    const double l_norm = nc.x * nc.x + nc.y * nc.y, l_sqrt = sqrt(l_norm), l_tri = l_norm * l_sqrt;
    const double e1 = obs[0] * nc.x + obs[1] * nc.y + nc.z, e2 = obs[2] * nc.x + obs[3] * nc.y + nc.z;
    const double r0 = e1 / l_sqrt, r1 = e2 / l_sqrt;
    double* o = out + (size_t)i * 34;
    o[0] = si[0] * r0 + si[1] * r1; o[1] = si[2] * r0 + si[3] * r1;
    const double jel[2][3] = { { obs[0] / l_sqrt - nc.x * e1 / l_tri, obs[1] / l_sqrt - nc.y * e1 / l_tri, 1.0 / l_sqrt },
                               { obs[2] / l_sqrt - nc.x * e2 / l_tri, obs[3] / l_sqrt - nc.y * e2 / l_tri, 1.0 / l_sqrt } };
    double jeLc[2][6];
#pragma unroll
    for (int j = 0; j < 3; ++j) { jeLc[0][j] = si[0] * jel[0][j] + si[1] * jel[1][j]; jeLc[1][j] = si[2] * jel[0][j] + si[3] * jel[1][j]; jeLc[0][3 + j] = 0; jeLc[1][3 + j] = 0; }
    const m33 RbcT = tr(Rbc), RwbT = tr(Rwb);
    double a[2][6], r[2][6];
    // pose: jaco_e_Lc * invTbc * jaco_Lc_pose
    mul26(jeLc, RbcT, scale(mul(RbcT, skew(tbc)), -1.0), RbcT, a);
    mul26(a, mul(RwbT, skew(lw.v)), skew(mul(RwbT, lw.n + mul(skew(lw.v), twb))), skew(mul(RwbT, lw.v)), r);
#pragma unroll
    for (int k = 0; k < 12; ++k) o[2 + k] = r[k / 6][k % 6];
    // extrinsic: jaco_e_Lc * jaco_Lc_ex
    mul26(jeLc, mul(RbcT, skew(lb.v)), skew(mul(RbcT, lb.n + mul(skew(lb.v), tbc))), skew(mul(RbcT, lb.v)), r);
#pragma unroll
    for (int k = 0; k < 12; ++k) o[14 + k] = r[k / 6][k % 6];
    // orthonormal representation: jaco_e_Lc * invTwc * jaco_Lw_orth
    const m33 Rwc = mul(Rwb, Rbc); const d3 twc = mul(Rwb, tbc) + twb;
    const m33 RwcT = tr(Rwc);
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
            o[26 + q * 4 + c] = a[q][0] * top[c].x + a[q][1] * top[c].y + a[q][2] * top[c].z + a[q][3] * bot[c].x + a[q][4] * bot[c].y + a[q][5] * bot[c].z;
}

__global__ void line_plus_kernel(const double* x, const double* delta, int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    line_plus_dev(x + 4 * i, delta + 4 * i, out + 4 * i);
}

// out per block (21): r[3] | J_pose_obj 3x6
__global__ void box_enclose_kernel(const dv_box_point* f, int n, const double* pose_obj, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r[3], Jp[9];
    box_enclose_dev(mk3(f[i].pts_w[0], f[i].pts_w[1], f[i].pts_w[2]), f[i].dims, P3(pose_obj + 7 * i), Q4(pose_obj + 7 * i), r, Jp);
    double* o = out + (size_t)i * 21;
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[3 + k * 6 + c] = Jp[k * 3 + c]; o[3 + k * 6 + 3 + c] = 0.0; }
    }
}

// out per block (4): r | J_box 1x3
__global__ void box_dims_kernel(const double* dims, const double* box, int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r, J[3];
    box_dims_dev(P3(box + 3 * i), P3(dims + 3 * i), r, J);
    double* o = out + 4 * i;
    o[0] = r; o[1] = J[0]; o[2] = J[1]; o[3] = J[2];
}

// out per block (39): r[3] | J_pose_body 3x6 (zero) | J_pose_obj 3x6
__global__ void box_orientation_kernel(const double* R_cioi, const double* R_bc, const double* pose_b, const double* pose_o, int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    m33 Rc, Rb;
    for (int k = 0; k < 9; ++k) { Rc.m[k] = R_cioi[9 * i + k]; Rb.m[k] = R_bc[9 * i + k]; }
    double r[3], Jr[9];
    box_orientation_dev(Rc, Rb, Q4(pose_b + 7 * i), Q4(pose_o + 7 * i), r, Jr);
    double* o = out + (size_t)i * 39;
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
    for (int k = 0; k < 18; ++k) o[3 + k] = 0.0;
    for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) { o[21 + k * 6 + c] = 0.0; o[21 + k * 6 + 3 + c] = Jr[k * 3 + c]; }
}

// out per block (64): see inst_proj_dev
__global__ void inst_proj_kernel(const dv_inst_proj_factor* f, int n, const double* pbj, const double* pbi, const double* pex, const double* poj, const double* poi,
                                 const double* lam, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double o[64];
    inst_proj_dev(reinterpret_cast<const double*>(f + i), pbj + 7 * i, pbi + 7 * i, pex + 7 * i, poj + 7 * i, poi + 7 * i, lam[i], o);
    double* dst = out + (size_t)i * 64;
#pragma unroll
    for (int k = 0; k < 64; ++k) dst[k] = o[k];
}

// stage host arrays behind each other in one device buffer, run, copy the result back
struct Stage {
    dv_ctx* ctx; hipStream_t s; uint8_t* base = nullptr; size_t off = 0;
    int reserve(size_t bytes) { return ctx->s0.ensure(bytes + 4096) == hipSuccess ? (base = (uint8_t*)ctx->s0.p, 0) : -1; }
    template <class T> T* put(const T* host, size_t count) {
        T* d = (T*)(base + off);
        off += (count * sizeof(T) + 255) / 256 * 256;
        if (host && hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, s) != hipSuccess) return nullptr;
        return d;
    }
};

}  // namespace

extern "C" {

#define OBJ_PROLOGUE(name, bytes)                                                                  \
    if (!ctx) return -1;                                                                           \
    if (n <= 0 || !out) DV_FAIL(name ": bad argument");                                            \
    DV_CHECK(hipSetDevice(ctx->cfg.device));                                                       \
    Stage st{ ctx, ctx->be_stream };                                                               \
    if (st.reserve(bytes)) DV_FAIL(name ": out of device memory");

int dv_line_eval(dv_ctx* ctx, const dv_line_factor* factors, int n, const double* pose, const double* ex_pose, const double* orth, double* out) {
    OBJ_PROLOGUE("dv_line_eval", (sizeof(dv_line_factor) + 8 * (7 + 7 + 4 + 34)) * (size_t)n + 2048)
    if (!factors || !pose || !ex_pose || !orth) DV_FAIL("dv_line_eval: null argument");
    const dv_line_factor* df = st.put(factors, n); const double* dp = st.put(pose, 7 * (size_t)n); const double* de = st.put(ex_pose, 7 * (size_t)n);
    const double* dorth = st.put(orth, 4 * (size_t)n); double* dout = st.put((const double*)nullptr, 34 * (size_t)n);
    if (!df || !dp || !de || !dorth) DV_FAIL("dv_line_eval: upload failed");
    hipLaunchKernelGGL(line_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, st.s, df, n, dp, de, dorth, dout);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpyAsync(out, dout, 8 * 34 * (size_t)n, hipMemcpyDeviceToHost, st.s));
    DV_CHECK(hipStreamSynchronize(st.s));
    return 0;
}

int dv_line_plus(dv_ctx* ctx, const double* orth, const double* delta, int n, double* out) {
    OBJ_PROLOGUE("dv_line_plus", 8 * 12 * (size_t)n + 2048)
    if (!orth || !delta) DV_FAIL("dv_line_plus: null argument");
    const double* dx = st.put(orth, 4 * (size_t)n); const double* dd = st.put(delta, 4 * (size_t)n); double* dout = st.put((const double*)nullptr, 4 * (size_t)n);
    if (!dx || !dd) DV_FAIL("dv_line_plus: upload failed");
    hipLaunchKernelGGL(line_plus_kernel, dim3((n + 63) / 64), dim3(64), 0, st.s, dx, dd, n, dout);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpyAsync(out, dout, 8 * 4 * (size_t)n, hipMemcpyDeviceToHost, st.s));
    DV_CHECK(hipStreamSynchronize(st.s));
    return 0;
}

int dv_box_enclose_eval(dv_ctx* ctx, const dv_box_point* points, int n, const double* pose_obj, double* out) {
    OBJ_PROLOGUE("dv_box_enclose_eval", (sizeof(dv_box_point) + 8 * (7 + 21)) * (size_t)n + 2048)
    if (!points || !pose_obj) DV_FAIL("dv_box_enclose_eval: null argument");
    const dv_box_point* df = st.put(points, n); const double* dp = st.put(pose_obj, 7 * (size_t)n); double* dout = st.put((const double*)nullptr, 21 * (size_t)n);
    if (!df || !dp) DV_FAIL("dv_box_enclose_eval: upload failed");
    hipLaunchKernelGGL(box_enclose_kernel, dim3((n + 63) / 64), dim3(64), 0, st.s, df, n, dp, dout);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpyAsync(out, dout, 8 * 21 * (size_t)n, hipMemcpyDeviceToHost, st.s));
    DV_CHECK(hipStreamSynchronize(st.s));
    return 0;
}

int dv_box_dims_eval(dv_ctx* ctx, const double* dims, const double* box, int n, double* out) {
    OBJ_PROLOGUE("dv_box_dims_eval", 8 * 10 * (size_t)n + 2048)
    if (!dims || !box) DV_FAIL("dv_box_dims_eval: null argument");
    const double* dd = st.put(dims, 3 * (size_t)n); const double* db = st.put(box, 3 * (size_t)n); double* dout = st.put((const double*)nullptr, 4 * (size_t)n);
    if (!dd || !db) DV_FAIL("dv_box_dims_eval: upload failed");
    hipLaunchKernelGGL(box_dims_kernel, dim3((n + 63) / 64), dim3(64), 0, st.s, dd, db, n, dout);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpyAsync(out, dout, 8 * 4 * (size_t)n, hipMemcpyDeviceToHost, st.s));
    DV_CHECK(hipStreamSynchronize(st.s));
    return 0;
}

int dv_box_orientation_eval(dv_ctx* ctx, const double* R_cioi, const double* R_bc, const double* pose_body, const double* pose_obj, int n, double* out) {
    OBJ_PROLOGUE("dv_box_orientation_eval", 8 * (9 + 9 + 7 + 7 + 39) * (size_t)n + 4096)
    if (!R_cioi || !R_bc || !pose_body || !pose_obj) DV_FAIL("dv_box_orientation_eval: null argument");
    const double* dc = st.put(R_cioi, 9 * (size_t)n); const double* db = st.put(R_bc, 9 * (size_t)n);
    const double* dpb = st.put(pose_body, 7 * (size_t)n); const double* dpo = st.put(pose_obj, 7 * (size_t)n); double* dout = st.put((const double*)nullptr, 39 * (size_t)n);
    if (!dc || !db || !dpb || !dpo) DV_FAIL("dv_box_orientation_eval: upload failed");
    hipLaunchKernelGGL(box_orientation_kernel, dim3((n + 63) / 64), dim3(64), 0, st.s, dc, db, dpb, dpo, n, dout);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpyAsync(out, dout, 8 * 39 * (size_t)n, hipMemcpyDeviceToHost, st.s));
    DV_CHECK(hipStreamSynchronize(st.s));
    return 0;
}

int dv_inst_proj_eval(dv_ctx* ctx, const dv_inst_proj_factor* factors, int n, const double* pose_bj, const double* pose_bi, const double* ex_pose,
                      const double* pose_oj, const double* pose_oi, const double* inv_dep_j, double* out) {
    OBJ_PROLOGUE("dv_inst_proj_eval", (sizeof(dv_inst_proj_factor) + 8 * (5 * 7 + 1 + 64)) * (size_t)n + 4096)
    if (!factors || !pose_bj || !pose_bi || !ex_pose || !pose_oj || !pose_oi || !inv_dep_j) DV_FAIL("dv_inst_proj_eval: null argument");
    const dv_inst_proj_factor* df = st.put(factors, n);
    const double* d0 = st.put(pose_bj, 7 * (size_t)n); const double* d1 = st.put(pose_bi, 7 * (size_t)n); const double* d2 = st.put(ex_pose, 7 * (size_t)n);
    const double* d3p = st.put(pose_oj, 7 * (size_t)n); const double* d4 = st.put(pose_oi, 7 * (size_t)n); const double* dl = st.put(inv_dep_j, (size_t)n);
    double* dout = st.put((const double*)nullptr, 64 * (size_t)n);
    if (!df || !d0 || !d1 || !d2 || !d3p || !d4 || !dl) DV_FAIL("dv_inst_proj_eval: upload failed");
    hipLaunchKernelGGL(inst_proj_kernel, dim3((n + 63) / 64), dim3(64), 0, st.s, df, n, d0, d1, d2, d3p, d4, dl, dout);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpyAsync(out, dout, 8 * 64 * (size_t)n, hipMemcpyDeviceToHost, st.s));
    DV_CHECK(hipStreamSynchronize(st.s));
    return 0;
}

}  // extern "C"
