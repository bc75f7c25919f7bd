// bd_solve.h — trust-region (traditional dogleg) solve of a problem whose J^T J is BLOCK DIAGONAL, in one persistent workgroup launch:
// the restatement of ceres::Solve (DENSE_SCHUR + DOGLEG, Jacobi scaling, monotonic steps; the rule set of be_solve.hip) for
// the two auxiliary solves of the reference in which every residual block touches a single variable block —
//   InstanceManager::Optimization        estimator/estimator_insts.cpp:772-807   (be_objsolve.hip: object poses / dims)
//   Estimator::OptimizationWithOnlyLine  estimator/estimator.cpp:345-395          (be_linesolve.hip: line orthonormal parameters)
// The dogleg couples the blocks only through a handful of scalars (|g|, |gn|, g.gn, the model decrease, the candidate cost), so:
//   evaluation  supplied by the problem (Prob::eval): writes H | g per variable block (<= 6 x 6, packed lower + gradient, stride 27)
//   solve       one thread per variable block: Jacobi-scaled, mu-regularised 6x6 Cholesky in registers (smaller blocks are zero padded:
//               their padding rows are pure regularisation and produce zero steps)
//   scalars     workgroup reductions that every thread reads back identically, so the trust-region bookkeeping is replicated in
//               registers and control flow stays uniform; no host round trip inside the solve
// Prob interface:  template <bool BUILD> void eval(const BdArgs&, const double* x, double* H, double& cost_part, double& gmax_part) const;
//                  int plus(int v, const double* x, const double* delta, double* out) const;   // returns the block's global size
//                  int xdim(int v) const;
#pragma once
#include <hip/hip_runtime.h>
#include "be_math.h"

namespace bd {
using namespace be;

constexpr int BD_THREADS = 512;
constexpr int BD_NW = BD_THREADS / 64;
constexpr int BD_GROUP = 8;                 // lanes per variable block in the evaluations
constexpr int BD_HSTRIDE = 27;              // per block: 21 packed lower-triangular H entries | 6 gradient entries

struct BdArgs {
    int V, max_iters;
    double* x0; double* x1;                 // V x 7 (blocks with fewer global parameters use the first ones)
    double* H0; double* H1;                 // V x 27
    double* vec;                            // scale | diag | grad | gn | delta, each 6 V
    const unsigned char* active;            // V: block has at least one residual (ceres drops the others from the program)
    double xnorm2_const;                    // squared norm of the parameter blocks that are in the ceres program but never move
    double* out;                            // iterations, successful, termination, initial_cost, final_cost
    // round 5: the kernel initialises its own working set (x1 <- x0, H / vec zeroed: two enqueued operations less in front of it) and, with lds != 0, keeps x0 | x1 | H0 | H1 | vec
    // in LDS for the whole solve (bd_lds_bytes(V) of dynamic shared memory): every phase of an iteration used to start with a round trip to L2 for what the previous phase
    // of the SAME workgroup had just stored.  Same arithmetic in the same order: same bits.  h_x / h_out: optional pinned-host destinations of the result (V x 7 states,
    // the 5 summary scalars) written by the kernel itself — the two download copies behind it go away.
    int lds, pad;
    double* h_x; double* h_out;
};
inline size_t bd_lds_bytes(int V) { return sizeof(double) * (size_t)(14 + 2 * BD_HSTRIDE + 30) * (size_t)V; }
constexpr size_t BD_LDS_MAX = 96 * 1024;

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
        for (int k = 0; k < K; ++k) s_red[k * BD_NW + w] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { double s = 0; for (int i = 0; i < BD_NW; ++i) s += s_red[k * BD_NW + i]; v[k] = s; }
    __syncthreads();
}
__device__ __forceinline__ double block_max(double x, double* s_red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x = fmax(x, __shfl_xor(x, o));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = x;
    __syncthreads();
    double m = 0; for (int i = 0; i < BD_NW; ++i) m = fmax(m, s_red[i]);
    __syncthreads();
    return m;
}

// cost (and, if BUILD, H | g per block and max |g|) at the parameter values in xb

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


template <class Prob>
__global__ __launch_bounds__(BD_THREADS) void bd_solve_kernel(Prob prob, BdArgs a) {
    __shared__ double s_red[5 * BD_NW];          // block_sum<5> is the widest exchange (it used to be 4 * BD_NW: the fifth value's partials were written past the array — harmless while nothing
                                                  // else lived in LDS, the first state block of the LDS-resident working set otherwise)
    extern __shared__ __attribute__((aligned(16))) double bd_lds[];
    const int tid = threadIdx.x, V = a.V;
    double* xb[2] = { a.x0, a.x1 }; double* Hb[2] = { a.H0, a.H1 }; double* vecp = a.vec;
    if (a.lds) { xb[0] = bd_lds; xb[1] = xb[0] + 7 * (size_t)V; Hb[0] = xb[1] + 7 * (size_t)V; Hb[1] = Hb[0] + (size_t)BD_HSTRIDE * V; vecp = Hb[1] + (size_t)BD_HSTRIDE * V; }
    for (int i = tid; i < 7 * V; i += BD_THREADS) { const double v = a.x0[i]; xb[0][i] = v; xb[1][i] = v; }
    for (int i = tid; i < BD_HSTRIDE * V; i += BD_THREADS) { Hb[0][i] = 0.0; Hb[1][i] = 0.0; }
    for (int i = tid; i < 30 * V; i += BD_THREADS) vecp[i] = 0.0;
    __syncthreads();
    double* scale = vecp; double* diag = vecp + 6 * (size_t)V; double* grad = vecp + 12 * (size_t)V; double* gn = vecp + 18 * (size_t)V; double* delta = vecp + 24 * (size_t)V;
    int cur = 0;

    // initial evaluation, Jacobi scaling (fixed for the whole solve), |x|
    double cpart, gpart;
    prob.template eval<true>(a, xb[0], Hb[0], cpart, gpart);
    __syncthreads();
    double xn2 = 0;
    for (int v = tid; v < V; v += BD_THREADS) {
        if (!a.active[v]) continue;
        const double* h = Hb[0] + (size_t)BD_HSTRIDE * v;
#pragma unroll
        for (int i = 0; i < 6; ++i) scale[6 * v + i] = 1.0 / (1.0 + sqrt(h[tri(i, i)]));
        const int nx = prob.xdim(v);
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
                for (int v = tid; v < V; v += BD_THREADS) {
                    if (!a.active[v]) continue;
                    const double* hp = Hb[cur] + (size_t)BD_HSTRIDE * v;
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
            for (int v = tid; v < V; v += BD_THREADS) {
                if (!a.active[v]) continue;
                const double* hp = Hb[cur] + (size_t)BD_HSTRIDE * v;
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
        for (int v = tid; v < V; v += BD_THREADS) {
            if (!a.active[v]) continue;
            double o[7];
            const int nx = prob.plus(v, x + 7 * v, delta + 6 * v, o);
            for (int i = 0; i < nx; ++i) { const double d = x[7 * v + i] - o[i]; part[0] += d * d; part[1] += o[i] * o[i]; cand[7 * v + i] = o[i]; }
        }
        __syncthreads();
        double cg;
        prob.template eval<true>(a, cand, Hb[cur ^ 1], part[2], cg);
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
    if (cur == 1 || a.lds || a.h_x) for (int i = tid; i < 7 * V; i += BD_THREADS) { const double v = xb[cur][i]; if (cur == 1 || a.lds) a.x0[i] = v; if (a.h_x) a.h_x[i] = v; }
    if (tid == 0) {
        const double o5[5] = { (double)iterations, (double)successful, (double)termination, initial_cost, x_cost };
        for (int k = 0; k < 5; ++k) { a.out[k] = o5[k]; if (a.h_out) a.h_out[k] = o5[k]; }
    }
}


}  // namespace bd
