// be_linesolve.hip — the line-only refinement on gfx950: replaces ceres::Solve inside
// Estimator::OptimizationWithOnlyLine (estimator/estimator.cpp:345-395) for the problem AddLineResidualBlock builds (:222-253):
//   variables   one orthonormal line representation (4 parameters, LineOrthParameterization: Plus is the rotation / phase update of
//               factor/line_parameterization.cpp:9-72, ComputeJacobian the identity) per triangulated line landmark
//   residuals   lineProjectionFactor(obs) on (body.para_pose[frame], body.para_ex_pose[0], line) per observation, CauchyLoss(1.0);
//               poses and extrinsics are constant in this problem, so every residual block touches one variable block
//   options     DENSE_SCHUR + DOGLEG, max_num_iterations = max_iters (wall-clock budget disabled, as everywhere)
// J^T J is block diagonal (4x4 per line): the trust-region loop is the generic persistent-workgroup solver of bd_solve.h; this file
// supplies the evaluation (8 lanes per line stride over its observations, shuffle-reduce 10 + 4 + 1 partials) and the Plus.
// Note: the reference never assigns lineProjectionFactor::sqrt_info, so as shipped every residual is zero and the solve
// returns at once (gradient tolerance); the entry point takes sqrt_info as an argument and reproduces exactly that when it is zero.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "dv_ctx.h"
#include "be_math.h"
#include "be_obj_dev.h"
#include "bd_solve.h"

using namespace be;

namespace {
using namespace bd;

struct LineProb {
    int n_lines, n_obs;
    const double* obs;                      // SoA x1 | y1 | x2 | y2, n_obs each, sorted by line
    const int* obs_frame;                   // n_obs
    const int* obs_start;                   // n_lines + 1
    const double* pose;                     // 11 x 7
    const double* ex;                       // 7
    double si[4];                           // lineProjectionFactor::sqrt_info

    __device__ int xdim(int) const { return 4; }
    __device__ int plus(int, const double* x, const double* d, double* o) const { line_plus_dev(x, d, o); return 4; }

    template <bool BUILD>
    __device__ void eval(const BdArgs& ba, const double* __restrict__ xb, double* __restrict__ Hb, double& cost, double& gmax) const {
        cost = 0; gmax = 0;
        const int tid = threadIdx.x, grp = tid / BD_GROUP, j = tid % BD_GROUP;
        const m33 Rbc = qR(Q4(ex)); const d3 tbc = P3(ex);
        for (int v = grp; v < n_lines; v += BD_THREADS / BD_GROUP) {
            if (!ba.active[v]) continue;                          // uniform over the 8 lanes
            const int k0 = obs_start[v], k1 = obs_start[v + 1];
            const double* orth = xb + 7 * v;
            double acc[15];                                       // H (10, packed lower 4x4) | g (4) | cost
#pragma unroll
            for (int k = 0; k < 15; ++k) acc[k] = 0;
            for (int k = k0 + j; k < k1; k += BD_GROUP) {
                const double o4[4] = { obs[k], obs[n_obs + k], obs[2 * (size_t)n_obs + k], obs[3 * (size_t)n_obs + k] };
                const double* pp = pose + 7 * obs_frame[k];
                double r[2], J[8];
                line_orth_dev(o4, si, qR(Q4(pp)), P3(pp), Rbc, tbc, orth, r, J);
                // ceres::CauchyLoss(1.0): rho = log(1 + s), rho' = 1 / (1 + s), rho'' < 0 -> the corrector scales by sqrt(rho')
                const double sq = r[0] * r[0] + r[1] * r[1], rho1 = 1.0 / (1.0 + sq), sc = sqrt(rho1);
                acc[14] += 0.5 * log(1.0 + sq);
                if (BUILD) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) J[q] *= sc;
                    r[0] *= sc; r[1] *= sc;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[10 + c] += J[c] * r[0] + J[4 + c] * r[1];
#pragma unroll
                        for (int c2 = 0; c2 <= c; ++c2) acc[tri(c, c2)] += J[c] * J[c2] + J[4 + c] * J[4 + c2];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                double t = acc[k];
                t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4);
                acc[k] = t;
            }
            if (j == 0) {
                cost += acc[14];
                if (BUILD) {
                    double* h = Hb + (size_t)BD_HSTRIDE * v;
#pragma unroll
                    for (int k = 0; k < BD_HSTRIDE; ++k) h[k] = 0.0;
#pragma unroll
                    for (int k = 0; k < 10; ++k) h[k] = acc[k];               // tri(c, c2) of the 4x4 block is the same index inside the 6x6
#pragma unroll
                    for (int c = 0; c < 4; ++c) { h[21 + c] = acc[10 + c]; gmax = fmax(gmax, fabs(acc[10 + c])); }
                }
            }
        }
    }
};

}  // namespace

extern "C" int dv_line_solve(dv_ctx* ctx, dv_line_problem* P, dv_ba_summary* summary) {
    if (!ctx) return -1;
    if (!P || !summary) DV_FAIL("dv_line_solve: null argument");
    if (P->n_lines <= 0 || P->n_obs < 0 || P->max_iters < 0 || !P->orth || !P->pose || !P->ex_pose || (P->n_obs > 0 && !P->obs)) DV_FAIL("dv_line_solve: bad problem");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    const int V = P->n_lines, nobs = P->n_obs;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_x0 = carve(8 * 7 * (size_t)V), o_pose = carve(8 * 77), o_ex = carve(8 * 7), o_obs = carve(8 * 4 * (size_t)std::max(nobs, 1)), o_fr = carve(4 * (size_t)std::max(nobs, 1)),
                 o_start = carve(4 * ((size_t)V + 1)), o_act = carve((size_t)V);
    const size_t up_bytes = off;
    const size_t o_x1 = carve(8 * 7 * (size_t)V), o_H0 = carve(8 * BD_HSTRIDE * (size_t)V), o_H1 = carve(8 * BD_HSTRIDE * (size_t)V), o_vec = carve(8 * 30 * (size_t)V), o_out = carve(64);
    std::vector<uint8_t> host(up_bytes, 0);
    double* hx = (double*)(host.data() + o_x0); double* hobs = (double*)(host.data() + o_obs); int* hfr = (int*)(host.data() + o_fr);
    int* hstart = (int*)(host.data() + o_start); uint8_t* hact = host.data() + o_act;
    for (int v = 0; v < V; ++v) memcpy(hx + 7 * (size_t)v, P->orth + 4 * (size_t)v, 32);
    memcpy(host.data() + o_pose, P->pose, 8 * 77); memcpy(host.data() + o_ex, P->ex_pose, 56);
    for (int i = 0; i < nobs; ++i) {
        const dv_line_obs& ob = P->obs[i];
        if (ob.line < 0 || ob.line >= V || ob.frame < 0 || ob.frame >= 11) DV_FAIL("dv_line_solve: observation index out of range");
        hstart[ob.line + 1]++;
    }
    for (int v = 0; v < V; ++v) { if (hstart[v + 1]) hact[v] = 1; hstart[v + 1] += hstart[v]; }
    {
        std::vector<int> fill(hstart, hstart + V);
        for (int i = 0; i < nobs; ++i) {
            const dv_line_obs& ob = P->obs[i];
            const int k = fill[ob.line]++;
            for (int c = 0; c < 4; ++c) hobs[(size_t)c * nobs + k] = ob.obs[c];
            hfr[k] = ob.frame;
        }
    }
    if (ctx->s1.ensure(off) != hipSuccess) DV_FAIL("dv_line_solve: out of device memory");
    uint8_t* base = (uint8_t*)ctx->s1.p;
    hipStream_t s = ctx->be_stream;
    DV_CHECK(hipMemcpyAsync(base, host.data(), up_bytes, hipMemcpyHostToDevice, s));
    // (x1 <- x0 and the zeroing of H | vec: the kernel's own first phase, bd_solve.h)
    BdArgs a{};
    a.V = V; a.max_iters = P->max_iters;
    a.x0 = (double*)(base + o_x0); a.x1 = (double*)(base + o_x1); a.H0 = (double*)(base + o_H0); a.H1 = (double*)(base + o_H1); a.vec = (double*)(base + o_vec);
    a.active = base + o_act; a.xnorm2_const = 0.0; a.out = (double*)(base + o_out);      // the constant pose / extrinsic blocks are not part of ceres' reduced program
    LineProb pr{};
    pr.n_lines = V; pr.n_obs = nobs; pr.obs = (const double*)(base + o_obs); pr.obs_frame = (const int*)(base + o_fr); pr.obs_start = (const int*)(base + o_start);
    pr.pose = (const double*)(base + o_pose); pr.ex = (const double*)(base + o_ex);
    memcpy(pr.si, P->sqrt_info, sizeof(pr.si));
    {
        StageScope sc(ctx, "line_solve", s);
        hipLaunchKernelGGL(bd_solve_kernel<LineProb>, dim3(1), dim3(BD_THREADS), 0, s, pr, a);
    }
    DV_CHECK(hipGetLastError());
    std::vector<double> hxo(7 * (size_t)V); double hout[8];
    DV_CHECK(hipMemcpyAsync(hxo.data(), a.x0, 8 * 7 * (size_t)V, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipMemcpyAsync(hout, a.out, 64, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    if (ctx->timing) dv_harvest_timers(ctx, s);
    for (int v = 0; v < V; ++v) memcpy(P->orth + 4 * (size_t)v, hxo.data() + 7 * (size_t)v, 32);
    summary->iterations = (int)hout[0]; summary->successful = (int)hout[1]; summary->termination = (int)hout[2]; summary->slots = 0;
    summary->initial_cost = hout[3]; summary->final_cost = hout[4];
    return 0;
}
