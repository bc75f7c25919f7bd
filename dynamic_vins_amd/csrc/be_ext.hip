// be_ext.hip — free extrinsic / td blocks in the window solve (product code; gfx950).
//
// Estimator::AddBodyParameterBlock (estimator/estimator.cpp:87-100) leaves para_ex_pose[0..1] variable once openExEstimation is set (estimate_extrinsic 1) and para_td
// variable while |Vs[0]| >= 0.2 (estimate_td 1); every shipped YAML has both switched off, so the blocks live in two launches of their own beside the default path
// (be_kernels.h BeExt): be_eval_ext evaluates the reprojection factors again WITH their extrinsic / td Jacobians
// (projection_two_frame_one_cam_factor.cpp:118-150, projection_two_frame_two_cam_factor.cpp:120-165, projection_one_frame_two_cam_factor.cpp:95-135) and writes one ext packet
// per landmark; be_reduce_ext adds the landmark sums into the 13 extra rows / columns of Hd, Sc and gvec behind be_reduce.  Same control-block predicates as the kernels they
// follow, so a slot that skips the one skips the other.
#include <hip/hip_runtime.h>
#include "be_kernels.h"

using namespace be;

#define XE_THREADS 256
// one workgroup per landmark
__global__ __launch_bounds__(XE_THREADS) void be_eval_ext_kernel(BeEvalArgs a, BeExt xt, int mode) {
    const BeCtl c = *a.ctl;
    if (c.done) return;
    if (mode == BE_EVAL_X ? !c.need_eval : !c.pending) return;
    const BeState* st = mode == BE_EVAL_X ? a.x : a.cand;
    const int set = mode == BE_EVAL_X ? c.cur : (c.cur ^ 1);
    double* const o = xt.xpk[set];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= a.dims.nlm) return;
    __shared__ FrameGeom fg[BE_NF];
    __shared__ m33 ric[2];
    __shared__ d3 tic[2];
    __shared__ double Jb[BE_MAX_OBS_FACTORS][28];          // r2 | Ji 2x6 | Jj 2x6 | Jl2, scaled by the loss (be_eval's layout)
    __shared__ double Jx[BE_MAX_OBS_FACTORS][2][BE_NX];     // [factor][residual row][ext entry q]
    __shared__ int s_fj[BE_MAX_OBS_FACTORS], s_two[BE_MAX_OBS_FACTORS];
    const BeLm L = a.lm[b];
    be_frame_geom_dev(st, a.dims.nframes, fg, ric, tic, lane);
    __syncthreads();
    const double lambda = st->inv_depth[b];
    if (lane < L.count) {
        const BeFactor f = a.fac[L.first + lane];
        double r[2], Ji[12], Jj[12], Jl[2], Je0[12], Je1[12], Jt[2];
        proj_factor<true, true>(f, fg[f.fi], fg[f.fj], ric[0], tic[0], ric[1], tic[1], lambda, st->td, r, Ji, Jj, Jl, Je0, Je1, Jt);
        double rho0, sc;
        huber1(r[0] * r[0] + r[1] * r[1], rho0, sc);
        double* q = Jb[lane];
        q[0] = r[0] * sc; q[1] = r[1] * sc;
        for (int k = 0; k < 12; ++k) { q[2 + k] = Ji[k] * sc; q[14 + k] = Jj[k] * sc; }
        q[26] = Jl[0] * sc; q[27] = Jl[1] * sc;
        for (int rr = 0; rr < 2; ++rr) {
            for (int k = 0; k < 6; ++k) { Jx[lane][rr][k] = Je0[rr * 6 + k] * sc; Jx[lane][rr][6 + k] = Je1[rr * 6 + k] * sc; }
            Jx[lane][rr][12] = Jt[rr] * sc;
        }
        s_fj[lane] = f.fj; s_two[lane] = f.kind != 2;
    }
    __syncthreads();
    const int nf = L.count, anchor = L.anchor;
    for (int e = lane; e < BX_XP + BE_NX * 66; e += XE_THREADS) {
        double v = 0.0;
        if (e < BX_G) { const int q = e - BX_W; for (int f = 0; f < nf; ++f) v += Jx[f][0][q] * Jb[f][26] + Jx[f][1][q] * Jb[f][27]; }
        else if (e < BX_XX) { const int q = e - BX_G; for (int f = 0; f < nf; ++f) v += Jx[f][0][q] * Jb[f][0] + Jx[f][1][q] * Jb[f][1]; }
        else if (e < BX_XP) { const int t = e - BX_XX, q = t / BE_NX, p = t - q * BE_NX; for (int f = 0; f < nf; ++f) v += Jx[f][0][q] * Jx[f][0][p] + Jx[f][1][q] * Jx[f][1][p]; }
        else {
            const int t = e - BX_XP, q = t / 66, pc = t - q * 66, fa = pc / 6, rr = pc - fa * 6;
            for (int f = 0; f < nf; ++f) {
                if (!s_two[f]) continue;                     // a one-frame two-camera factor touches no body pose
                if (fa == anchor) v += Jx[f][0][q] * Jb[f][2 + rr] + Jx[f][1][q] * Jb[f][8 + rr];
                else if (fa == s_fj[f]) v += Jx[f][0][q] * Jb[f][14 + rr] + Jx[f][1][q] * Jb[f][20 + rr];
            }
        }
        BE_PK(o, e, b) = v;
    }
}
void be_launch_eval_ext(const BeEvalArgs& a, const BeExt& xt, int mode, hipStream_t s) {
    if (!xt.on || a.dims.nlm <= 0 || mode == BE_EVAL_CAND_COST) return;
    hipLaunchKernelGGL(be_eval_ext_kernel, dim3(a.dims.nlm), dim3(XE_THREADS), 0, s, a, xt, mode);
}

// one workgroup per ext entry q: row xcol[q] of the reduced system against the 66 pose entries and the 13 ext entries.  Wave w takes the targets j = w, w + 4, ...;
// its lanes stride over the landmarks, the wave tree of be_reduce sums them (fixed shape: deterministic).
#define XR_THREADS 256
__global__ __launch_bounds__(XR_THREADS) void be_reduce_ext_kernel(BeSolveArgs a, int spec) {
    const BeCtl c = *a.ctl;
    if (c.done) return;
    int set = c.cur; double mu = c.mu;
    if (spec && c.pending) { set ^= 1; mu = fmax(1e-8, 2.0 * c.mu / 10.0); }      // be_reduce_body's predicate, word for word
    else if (!c.need_eval && !c.chol_fail) return;
    const int q = blockIdx.x, i = a.xt.xcol[q];
    if (i < 0) return;
    const int n = a.dims.nstate, nlm = a.dims.nlm;
    const double* pk = a.packets[set]; const double* xp = a.xt.xpk[set];
    double* const Hd = a.Hd[set]; double* const Sc = a.Sc[set]; double* const gvec = a.gvec[set];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, NBR = (n + 3) >> 2;
    auto blk = [&](int r, int cc) { const int bi = r >> 2, bj = cc >> 2; return (bj * NBR - bj * (bj - 1) / 2 + bi - bj) * 16 + (r & 3) * 4 + (cc & 3); };      // blk_pos of be_solve.hip
    for (int j = w; j <= 66 + BE_NX; j += XR_THREADS / 64) {      // j == 66 + BE_NX: the gradient entries
        const bool grad = j == 66 + BE_NX, pose = j < 66;
        int jc = -1;
        if (!grad) { if (pose) { const int f = j / 6; const int c0 = f < a.dims.nframes ? a.dims.pose_col[f] : -1; jc = c0 >= 0 ? c0 + (j - f * 6) : -1; } else jc = a.xt.xcol[j - 66]; }
        if (!grad && jc < 0) continue;
        if (!grad && !pose && jc > i) continue;               // ext x ext: the lower triangle (and the diagonal); the mirror is stored with it
        double D = 0, S = 0;
        for (int l = lane; l < nlm; l += 64) {
            const double h = BE_PK(pk, BE_PK_H, l);
            const double s = c.first ? 1.0 / (1.0 + sqrt(h)) : a.scale_l[l];
            double d2 = h * s * s; d2 = fmin(fmax(d2, 1e-6), 1e32);
            const double rho = 1.0 / (h + mu * d2 / (s * s));
            const double ui = BE_PK(xp, BX_W + q, l);
            if (grad) { D += BE_PK(xp, BX_G + q, l); S += rho * (ui * BE_PK(pk, BE_PK_G, l)); }
            else if (pose) { D += BE_PK(xp, BX_XP + q * 66 + j, l); S += rho * (ui * BE_PK(pk, BE_PK_W + j, l)); }
            else { D += BE_PK(xp, BX_XX + q * BE_NX + (j - 66), l); S += rho * (ui * BE_PK(xp, BX_W + (j - 66), l)); }
        }
        D = wave_sum_f64(D); S = wave_sum_f64(S);
        if (lane != 0) continue;
        if (grad) { gvec[i] = gvec[i] + D; gvec[n + i] = S; continue; }      // be_reduce left the prior's part in gvec[i] and 0 in gvec[n + i]
        const double hd = Hd[(size_t)i * n + jc] + D;                        // ... and the prior's part in Hd (the IMU factors do not touch these blocks)
        Hd[(size_t)i * n + jc] = hd;
        if ((jc >> 2) <= (i >> 2)) Sc[blk(i, jc)] = hd - S;
        if (jc != i) {
            Hd[(size_t)jc * n + i] = hd;
            if ((i >> 2) <= (jc >> 2)) Sc[blk(jc, i)] = hd - S;
        }
    }
}
void be_launch_reduce_ext(const BeSolveArgs& a, int spec, hipStream_t s) {
    if (!a.xt.on) return;
    hipLaunchKernelGGL(be_reduce_ext_kernel, dim3(BE_NX), dim3(XR_THREADS), 0, s, a, spec);
}
