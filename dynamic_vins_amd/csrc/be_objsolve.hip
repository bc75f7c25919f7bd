// be_objsolve.hip — the per-frame solve of the dynamic objects on gfx950 (numeric part): replaces
// ceres::Solve inside InstanceManager::Optimization (estimator/estimator_insts.cpp:772-807) for the problem built by
// AddInstanceParameterBlock / AddResidualBlockForInstOpt (estimator_insts.cpp:989-1245): per object 11 pose blocks and
// one dims block; BoxDimsFactor + BoxOrientationFactor per 3-D detection, BoxEncloseStereoPointFactor per triangulated
// point; DENSE_SCHUR + DOGLEG with HuberLoss(1.0), Jacobi scaling, monotonic steps.
//
// Structure the kernel is built around: every residual block touches exactly one variable block (the body pose enters
// the orientation factor with a zero Jacobian, sic), so J^T J is BLOCK DIAGONAL — 6x6 per (object, frame), 3x3 per
// object — while the dogleg trust region couples all blocks through a handful of scalars (|g|, |gn|, g.gn, the model
// decrease, the candidate cost).  The whole trust-region loop therefore runs inside ONE persistent workgroup launch:
//   eval      8 lanes per variable block stride over its points (coalesced SoA reads), lane-reduce the 3x3 + 3 + cost
//             partials with wave shuffles, lane 0 adds the orientation factor and writes the block's H | g
//   solve     one thread per variable block: Jacobi-scaled, mu-regularised 6x6 Cholesky in registers
//   scalars   workgroup reductions (wave shuffle + 8 LDS partials) that every thread reads back identically, so the
//             trust-region bookkeeping is replicated in registers and control flow stays uniform
// No host round trip inside the solve; one H2D of the packed problem, one launch, one D2H of the states.
// The problem is K objects x (11 + 1) blocks and a few thousand points: latency-bound, neither HBM nor MFMA matter.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "dv_ctx.h"
#include "be_math.h"
#include "be_obj_dev.h"

using namespace be;

namespace {

constexpr int OS_THREADS = 512;
constexpr int OS_NW = OS_THREADS / 64;
constexpr int OS_GROUP = 8;                 // lanes per variable block in the evaluation
constexpr int OS_NF = 11;                   // kWinSize + 1 pose blocks per object
constexpr int OS_HSTRIDE = 27;              // per block: 21 packed lower-triangular H entries | 6 gradient entries

struct ObjSolveArgs {
    int n_obj, nblk, V, npts, max_iters, plane_kind;
    double* x0; double* x1;                 // V x 7 (dims blocks use the first 3)
    double* H0; double* H1;                 // V x 27
    double* vec;                            // scale | diag | grad | gn | delta, each 6 V
    const double* dims0;                    // n_obj x 3: dims at entry (constants of the point factor)
    const double* body;                     // 11 x 7
    const double* rbc;                      // 9
    const double* pt;                       // SoA x | y | z, npts each, sorted by block
    const int* pt_start;                    // nblk + 1
    const double* box_R;                    // nblk x 9
    const double* box_dims;                 // nblk x 3
    const unsigned char* has_box;           // nblk
    const unsigned char* active;            // V: block has at least one residual
    double xnorm2_const;                    // sum |body_pose[f]|^2 over the frames that carry an orientation factor
    double* out;                            // iterations, successful, termination, initial_cost, final_cost
};

__device__ __forceinline__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }      // i >= j

template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* s_red) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = v[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
        v[k] = x;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) s_red[k * OS_NW + w] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { double s = 0; for (int i = 0; i < OS_NW; ++i) s += s_red[k * OS_NW + i]; v[k] = s; }
    __syncthreads();
}
__device__ __forceinline__ double block_max(double x, double* s_red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x = fmax(x, __shfl_xor(x, o));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = x;
    __syncthreads();
    double m = 0; for (int i = 0; i < OS_NW; ++i) m = fmax(m, s_red[i]);
    __syncthreads();
    return m;
}

// cost (and, if BUILD, H | g per block and max |g|) at the parameter values in xb
template <bool BUILD>
__device__ void os_eval(const ObjSolveArgs& a, const double* __restrict__ xb, double* __restrict__ Hb, double& cost, double& gmax) {
    cost = 0; gmax = 0;
    const int tid = threadIdx.x, grp = tid / OS_GROUP, j = tid % OS_GROUP;
    for (int b = grp; b < a.nblk; b += OS_THREADS / OS_GROUP) {
        if (!a.active[b]) continue;                           // uniform over the 8 lanes
        const int k0 = a.pt_start[b], k1 = a.pt_start[b + 1];
        const double* xp = xb + 7 * b;
        const d3 P = P3(xp); const quat q = Q4(xp);
        const double* d0 = a.dims0 + 3 * (b / OS_NF);
        double acc[10];                                       // H_pp (6, packed lower) | g_p (3) | cost
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] = 0;
        for (int k = k0 + j; k < k1; k += OS_GROUP) {
            double r[3], J[9];
            box_enclose_dev(mk3(a.pt[k], a.pt[a.npts + k], a.pt[2 * a.npts + k]), d0, P, q, r, J);
            double rho0, s;
            huber1(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho0, s);
            acc[9] += 0.5 * rho0;
            if (BUILD) {
#pragma unroll
                for (int k2 = 0; k2 < 9; ++k2) J[k2] *= s;
                r[0] *= s; r[1] *= s; r[2] *= s;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    acc[6 + c] += J[c] * r[0] + J[3 + c] * r[1] + J[6 + c] * r[2];
#pragma unroll
                    for (int c2 = 0; c2 <= c; ++c2) acc[tri(c, c2)] += J[c] * J[c2] + J[3 + c] * J[3 + c2] + J[6 + c] * J[6 + c2];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            double v = acc[k];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            acc[k] = v;
        }
        if (j == 0) {
            double hr[6] = { 0, 0, 0, 0, 0, 0 }, gr[3] = { 0, 0, 0 };
            if (a.has_box[b]) {
                m33 Rc, Rb;
#pragma unroll
                for (int k = 0; k < 9; ++k) { Rc.m[k] = a.box_R[9 * b + k]; Rb.m[k] = a.rbc[k]; }
                double r[3], J[9];
                box_orientation_dev(Rc, Rb, Q4(a.body + 7 * (b % OS_NF)), q, r, J);
                acc[9] += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);          // no loss function on this factor
                if (BUILD) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        gr[c] = J[c] * r[0] + J[3 + c] * r[1] + J[6 + c] * r[2];
#pragma unroll
                        for (int c2 = 0; c2 <= c; ++c2) hr[tri(c, c2)] = J[c] * J[c2] + J[3 + c] * J[3 + c2] + J[6 + c] * J[6 + c2];
                    }
                }
            }
            cost += acc[9];
            if (BUILD) {
                double* h = Hb + (size_t)OS_HSTRIDE * b;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int c2 = 0; c2 <= c; ++c2) { h[tri(c, c2)] = acc[tri(c, c2)]; h[tri(3 + c, 3 + c2)] = hr[tri(c, c2)]; }
#pragma unroll
                    for (int c2 = 0; c2 < 3; ++c2) h[tri(3 + c, c2)] = 0.0;
                    h[21 + c] = acc[6 + c]; h[24 + c] = gr[c];
                    gmax = fmax(gmax, fmax(fabs(acc[6 + c]), fabs(gr[c])));
                }
            }
        }
    }
    // dims blocks: at most 11 one-dimensional residuals each, one lane per object
    for (int o = tid; o < a.n_obj; o += OS_THREADS) {
        const int v = a.nblk + o;
        if (!a.active[v]) continue;
        const d3 box = P3(xb + 7 * v);
        double h6[6] = { 0, 0, 0, 0, 0, 0 }, g3[3] = { 0, 0, 0 }, c = 0;
        for (int f = 0; f < OS_NF; ++f) {
            const int b = o * OS_NF + f;
            if (!a.has_box[b]) continue;
            double r, J[3];
            box_dims_dev(box, P3(a.box_dims + 3 * b), r, J);
            double rho0, s;
            huber1(r * r, rho0, s);
            c += 0.5 * rho0;
            if (BUILD) {
                r *= s;
#pragma unroll
                for (int k = 0; k < 3; ++k) J[k] *= s;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    g3[k] += J[k] * r;
#pragma unroll
                    for (int k2 = 0; k2 <= k; ++k2) h6[tri(k, k2)] += J[k] * J[k2];
                }
            }
        }
        cost += c;
        if (BUILD) {
            double* h = Hb + (size_t)OS_HSTRIDE * v;
#pragma unroll
            for (int k = 0; k < OS_HSTRIDE; ++k) h[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) h[k] = h6[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) { h[21 + k] = g3[k]; gmax = fmax(gmax, fabs(g3[k])); }
        }
    }
}

__device__ __forceinline__ double quad6(const double* H, const double* t) {      // t^T H t, H packed lower
    double s = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double row = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) row += H[i >= c ? tri(i, c) : tri(c, i)] * t[c];
        s += t[i] * row;
    }
    return s;
}

// (S H S + mu diag^2) y = S g by a 6x6 Cholesky in registers; false on a non-positive pivot or a non-finite result
__device__ __forceinline__ bool gn6(const double* H, const double* g, const double* sc, const double* dg, double mu, double* y) {
    double L[21]; bool ok = true;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double s = H[tri(c, c)] * sc[c] * sc[c] + mu * dg[c] * dg[c];
#pragma unroll
        for (int k = 0; k < c; ++k) s -= L[tri(c, k)] * L[tri(c, k)];
        ok = ok && (s > 0);
        const double d = sqrt(s);
        L[tri(c, c)] = d;
#pragma unroll
        for (int i = c + 1; i < 6; ++i) {
            double t = H[tri(i, c)] * sc[i] * sc[c];
#pragma unroll
            for (int k = 0; k < c; ++k) t -= L[tri(i, k)] * L[tri(c, k)];
            L[tri(i, c)] = t / d;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = g[i] * sc[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[tri(i, k)] * y[k];
        y[i] = s / L[tri(i, i)];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[tri(k, i)] * y[k];
        y[i] = s / L[tri(i, i)];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) ok = ok && isfinite(y[i]);
    return ok;
}

__global__ __launch_bounds__(OS_THREADS) void obj_solve_kernel(ObjSolveArgs a) {
    __shared__ double s_red[4 * OS_NW];
    const int tid = threadIdx.x, V = a.V;
    double* xb[2] = { a.x0, a.x1 }; double* Hb[2] = { a.H0, a.H1 };
    double* scale = a.vec; double* diag = a.vec + 6 * (size_t)V; double* grad = a.vec + 12 * (size_t)V; double* gn = a.vec + 18 * (size_t)V; double* delta = a.vec + 24 * (size_t)V;
    int cur = 0;

    // initial evaluation, Jacobi scaling (fixed for the whole solve), |x|
    double cpart, gpart;
    os_eval<true>(a, xb[0], Hb[0], cpart, gpart);
    __syncthreads();
    double xn2 = 0;
    for (int v = tid; v < V; v += OS_THREADS) {
        if (!a.active[v]) continue;
        const double* h = Hb[0] + (size_t)OS_HSTRIDE * v;
#pragma unroll
        for (int i = 0; i < 6; ++i) scale[6 * v + i] = 1.0 / (1.0 + sqrt(h[tri(i, i)]));
        const int nx = v < a.nblk ? 7 : 3;
        for (int i = 0; i < nx; ++i) xn2 += xb[0][7 * v + i] * xb[0][7 * v + i];
    }
    double red2[2] = { cpart, xn2 };
    block_sum(red2, s_red);
    double x_cost = red2[0], x_norm = sqrt(red2[1] + a.xnorm2_const);
    double gmax = block_max(gpart, s_red);
    const double initial_cost = x_cost;

    double radius = 1e4, mu = 1e-8, alpha = 0, dogleg_norm = 0, gg = 0, gnn2 = 0, gdot = 0;
    bool reuse = false; int invalid = 0, iterations = 0, successful = 0, termination = 0;
    if (gmax <= 1e-10) termination = 1;
    else for (int it = 1;; ++it) {
        if (it > a.max_iters) { termination = 0; break; }
        iterations = it;
        bool step_valid = true;
        if (!reuse) {
            reuse = true;
            bool first = true, ok = false;
            while (true) {
                double part[5] = { 0, 0, 0, 0, 0 };              // fail | gg | JgJg | |gn|^2 | grad.gn
                for (int v = tid; v < V; v += OS_THREADS) {
                    if (!a.active[v]) continue;
                    const double* hp = Hb[cur] + (size_t)OS_HSTRIDE * v;
                    double H[21], g[6], sc[6], dg[6], gr[6], y[6];
#pragma unroll
                    for (int k = 0; k < 21; ++k) H[k] = hp[k];
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        g[i] = hp[21 + i]; sc[i] = scale[6 * v + i];
                        dg[i] = sqrt(fmin(fmax(H[tri(i, i)] * sc[i] * sc[i], 1e-6), 1e32));
                        gr[i] = g[i] * sc[i] / dg[i];
                    }
                    if (first) {
                        double t[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) { t[i] = gr[i] / dg[i] * sc[i]; part[1] += gr[i] * gr[i]; diag[6 * v + i] = dg[i]; grad[6 * v + i] = gr[i]; }
                        part[2] += quad6(H, t);
                    }
                    if (!gn6(H, g, sc, dg, mu, y)) part[0] += 1.0;
#pragma unroll
                    for (int i = 0; i < 6; ++i) { const double s = -dg[i] * y[i]; gn[6 * v + i] = s; part[3] += s * s; part[4] += gr[i] * s; }
                }
                block_sum(part, s_red);
                if (first) { gg = part[1]; alpha = gg / part[2]; first = false; }
                gnn2 = part[3]; gdot = part[4];
                ok = part[0] == 0.0;
                if (ok) break;
                mu *= 10.0;
                if (mu > 1.0) break;
            }
            if (!ok) step_valid = false;
        }
        double mcc = 0;
        if (step_valid) {
            const double gnorm = sqrt(gg), gnn = sqrt(gnn2);
            int kind; double ca = 0, cb = 0;                      // step = ca grad + cb gn
            if (gnn <= radius) { kind = 0; ca = 0; cb = 1; dogleg_norm = gnn; }
            else if (gnorm * alpha >= radius) { kind = 1; ca = -(radius / gnorm); cb = 0; dogleg_norm = radius; }
            else {
                kind = 2;
                const double b_dot_a = -alpha * gdot, a2 = pow(alpha * gnorm, 2.0), bma2 = a2 - 2 * b_dot_a + pow(gnn, 2.0);
                const double c = b_dot_a - a2, d = sqrt(c * c + bma2 * (pow(radius, 2.0) - a2));
                const double beta = (c <= 0) ? (d - c) / bma2 : (radius * radius - a2) / (d + c);
                ca = -alpha * (1.0 - beta); cb = beta;
            }
            double part[3] = { 0, 0, 0 };                         // |step|^2 | s.g | s^T H s
            for (int v = tid; v < V; v += OS_THREADS) {
                if (!a.active[v]) continue;
                const double* hp = Hb[cur] + (size_t)OS_HSTRIDE * v;
                double H[21], t[6];
#pragma unroll
                for (int k = 0; k < 21; ++k) H[k] = hp[k];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    double s = kind == 0 ? gn[6 * v + i] : (kind == 1 ? ca * grad[6 * v + i] : ca * grad[6 * v + i] + cb * gn[6 * v + i]);
                    part[0] += s * s;
                    s /= diag[6 * v + i];
                    t[i] = s * scale[6 * v + i];
                    part[1] += t[i] * hp[21 + i];
                    delta[6 * v + i] = t[i];
                }
                part[2] += quad6(H, t);
            }
            block_sum(part, s_red);
            if (kind == 2) dogleg_norm = sqrt(part[0]);
            mcc = -(part[1] + 0.5 * part[2]);
            step_valid = mcc > 0.0;
            if (step_valid) invalid = 0;
        }
        if (!step_valid) {
            if (++invalid >= 5) { termination = 2; break; }
            mu *= 10.0; reuse = false;
            continue;
        }
        // candidate = x (+) delta
        double part[3] = { 0, 0, 0 };                             // |x - cand|^2 | |cand|^2 | cost
        const double* x = xb[cur]; double* cand = xb[cur ^ 1];
        for (int v = tid; v < V; v += OS_THREADS) {
            if (!a.active[v]) continue;
            double o[7]; int nx;
            if (v < a.nblk) { pose_plus(x + 7 * v, delta + 6 * v, a.plane_kind, o); nx = 7; }
            else { for (int i = 0; i < 3; ++i) o[i] = x[7 * v + i] + delta[6 * v + i]; nx = 3; }
            for (int i = 0; i < nx; ++i) { const double d = x[7 * v + i] - o[i]; part[0] += d * d; part[1] += o[i] * o[i]; cand[7 * v + i] = o[i]; }
        }
        __syncthreads();
        double cg;
        os_eval<true>(a, cand, Hb[cur ^ 1], part[2], cg);
        block_sum(part, s_red);
        const double cand_gmax = block_max(cg, s_red);
        const double sn = sqrt(part[0]), cand_cost = part[2];
        if (sn <= 1e-8 * (x_norm + 1e-8)) { termination = 1; break; }
        if (fabs(x_cost - cand_cost) <= 1e-6 * x_cost) { termination = 1; break; }
        const double rel = (x_cost - cand_cost) / mcc;
        if (rel > 1e-3) {
            cur ^= 1; x_cost = cand_cost; x_norm = sqrt(part[1] + a.xnorm2_const);
            ++successful;
            if (rel < 0.25) radius *= 0.5;
            if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_norm);
            mu = fmax(1e-8, 2.0 * mu / 10.0);
            reuse = false;
            if (cand_gmax <= 1e-10) { termination = 1; break; }
        } else {
            radius *= 0.5; reuse = true;
            if (radius < 1e-32) { termination = 1; break; }
        }
    }
    __syncthreads();
    if (cur == 1) for (int i = tid; i < 7 * V; i += OS_THREADS) a.x0[i] = a.x1[i];
    if (tid == 0) { a.out[0] = iterations; a.out[1] = successful; a.out[2] = termination; a.out[3] = initial_cost; a.out[4] = x_cost; }
}

}  // namespace

extern "C" int dv_obj_solve(dv_ctx* ctx, dv_obj_problem* P, dv_ba_summary* summary) {
    if (!ctx) return -1;
    if (!P || !summary) DV_FAIL("dv_obj_solve: null argument");
    if (P->n_obj <= 0 || P->n_boxes < 0 || P->n_points < 0 || P->max_iters < 0 || !P->state || !P->dims || !P->body_pose) DV_FAIL("dv_obj_solve: bad problem");
    if ((P->n_boxes > 0 && !P->boxes) || (P->n_points > 0 && !P->points)) DV_FAIL("dv_obj_solve: null factor list");
    if (P->plane_kind < 0 || P->plane_kind > 2) DV_FAIL("dv_obj_solve: bad plane_kind");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    const int n_obj = P->n_obj, nblk = n_obj * OS_NF, V = nblk + n_obj, npts = P->n_points;

    // host side: bucket the factors by variable block (stable counting sort), pack one upload
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_x0 = carve(8 * 7 * (size_t)V), o_dims0 = carve(8 * 3 * (size_t)n_obj), o_body = carve(8 * 7 * OS_NF), o_rbc = carve(8 * 9),
                 o_pt = carve(8 * 3 * (size_t)std::max(npts, 1)), o_start = carve(4 * ((size_t)nblk + 1)), o_boxR = carve(8 * 9 * (size_t)nblk),
                 o_boxd = carve(8 * 3 * (size_t)nblk), o_hasb = carve((size_t)nblk), o_act = carve((size_t)V);
    const size_t up_bytes = off;
    const size_t o_x1 = carve(8 * 7 * (size_t)V), o_H0 = carve(8 * OS_HSTRIDE * (size_t)V), o_H1 = carve(8 * OS_HSTRIDE * (size_t)V), o_vec = carve(8 * 30 * (size_t)V), o_out = carve(64);
    std::vector<uint8_t> host(up_bytes, 0);
    double* hx = (double*)(host.data() + o_x0); double* hd0 = (double*)(host.data() + o_dims0); double* hpt = (double*)(host.data() + o_pt);
    int* hstart = (int*)(host.data() + o_start); double* hR = (double*)(host.data() + o_boxR); double* hbd = (double*)(host.data() + o_boxd);
    uint8_t* hhas = host.data() + o_hasb; uint8_t* hact = host.data() + o_act;
    for (int b = 0; b < nblk; ++b) memcpy(hx + 7 * b, P->state + 7 * (size_t)b, 56);
    for (int o = 0; o < n_obj; ++o) { memcpy(hx + 7 * (size_t)(nblk + o), P->dims + 3 * o, 24); memcpy(hd0 + 3 * o, P->dims + 3 * o, 24); }
    memcpy(host.data() + o_body, P->body_pose, 8 * 7 * OS_NF);
    memcpy(host.data() + o_rbc, P->R_bc, 72);
    bool frame_has_box[OS_NF] = {};
    for (int i = 0; i < P->n_boxes; ++i) {
        const dv_obj_box& bx = P->boxes[i];
        if (bx.obj < 0 || bx.obj >= n_obj || bx.frame < 0 || bx.frame >= OS_NF) DV_FAIL("dv_obj_solve: box index out of range");
        const int b = bx.obj * OS_NF + bx.frame;
        if (hhas[b]) DV_FAIL("dv_obj_solve: more than one box for an (object, frame)");
        hhas[b] = 1; hact[b] = 1; hact[nblk + bx.obj] = 1; frame_has_box[bx.frame] = true;
        memcpy(hR + 9 * (size_t)b, bx.R_cioi, 72); memcpy(hbd + 3 * (size_t)b, bx.dims, 24);
    }
    for (int i = 0; i < npts; ++i) {
        const dv_obj_point& p = P->points[i];
        if (p.obj < 0 || p.obj >= n_obj || p.frame < 0 || p.frame >= OS_NF) DV_FAIL("dv_obj_solve: point index out of range");
        hstart[p.obj * OS_NF + p.frame + 1]++;
    }
    for (int b = 0; b < nblk; ++b) { if (hstart[b + 1]) hact[b] = 1; hstart[b + 1] += hstart[b]; }
    {
        std::vector<int> fill(hstart, hstart + nblk);
        for (int i = 0; i < npts; ++i) {
            const dv_obj_point& p = P->points[i];
            const int k = fill[p.obj * OS_NF + p.frame]++;
            hpt[k] = p.p_w[0]; hpt[npts + k] = p.p_w[1]; hpt[2 * (size_t)npts + k] = p.p_w[2];
        }
    }
    double xc = 0;
    for (int f = 0; f < OS_NF; ++f) if (frame_has_box[f]) for (int k = 0; k < 7; ++k) xc += P->body_pose[7 * f + k] * P->body_pose[7 * f + k];

    if (ctx->s1.ensure(off) != hipSuccess) DV_FAIL("dv_obj_solve: out of device memory");
    uint8_t* base = (uint8_t*)ctx->s1.p;
    hipStream_t s = ctx->be_stream;
    DV_CHECK(hipMemcpyAsync(base, host.data(), up_bytes, hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemcpyAsync(base + o_x1, base + o_x0, 8 * 7 * (size_t)V, hipMemcpyDeviceToDevice, s));
    DV_CHECK(hipMemsetAsync(base + o_H0, 0, o_out - o_H0, s));
    ObjSolveArgs a{};
    a.n_obj = n_obj; a.nblk = nblk; a.V = V; a.npts = npts; a.max_iters = P->max_iters; a.plane_kind = P->plane_kind;
    a.x0 = (double*)(base + o_x0); a.x1 = (double*)(base + o_x1); a.H0 = (double*)(base + o_H0); a.H1 = (double*)(base + o_H1); a.vec = (double*)(base + o_vec);
    a.dims0 = (const double*)(base + o_dims0); a.body = (const double*)(base + o_body); a.rbc = (const double*)(base + o_rbc); a.pt = (const double*)(base + o_pt);
    a.pt_start = (const int*)(base + o_start); a.box_R = (const double*)(base + o_boxR); a.box_dims = (const double*)(base + o_boxd);
    a.has_box = base + o_hasb; a.active = base + o_act; a.xnorm2_const = xc; a.out = (double*)(base + o_out);
    {
        StageScope sc(ctx, "obj_solve", s);
        hipLaunchKernelGGL(obj_solve_kernel, dim3(1), dim3(OS_THREADS), 0, s, a);
    }
    DV_CHECK(hipGetLastError());
    std::vector<double> hxo(7 * (size_t)V); double hout[8];
    DV_CHECK(hipMemcpyAsync(hxo.data(), a.x0, 8 * 7 * (size_t)V, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipMemcpyAsync(hout, a.out, 64, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    if (ctx->timing) dv_harvest_timers(ctx, s);
    for (int b = 0; b < nblk; ++b) memcpy(P->state + 7 * (size_t)b, hxo.data() + 7 * (size_t)b, 56);
    for (int o = 0; o < n_obj; ++o) memcpy(P->dims + 3 * o, hxo.data() + 7 * (size_t)(nblk + o), 24);
    summary->iterations = (int)hout[0]; summary->successful = (int)hout[1]; summary->termination = (int)hout[2]; summary->slots = 0;
    summary->initial_cost = hout[3]; summary->final_cost = hout[4];
    return 0;
}
