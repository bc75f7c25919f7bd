// be_solve.hip — the trust-region step of the sliding-window BA on gfx950 (replaces the CPU loop inside
// ceres::Solve as configured at estimator/estimator.cpp:296-314: DENSE_SCHUR + traditional DOGLEG, Jacobi
// scaling, monotonic steps).  Three kernels per iteration slot, all predicated on the device-resident BeCtl so
// the whole solve is enqueued without a host round trip:
//
//   be_reduce_kernel : assembles the reduced camera system from the per-landmark packets in a FIXED order
//                      (one thread per matrix entry looping over landmarks -> bitwise reproducible):
//                        Hd = prior A' + sum IMU blocks + sum_l D_l        (everything but the Schur term)
//                        Sc = sum_l rho_l w_l w_l^T ,  rho_l = 1 / (h_l + mu d_l^2 / s_l^2)
//                      and the gradient parts.
//   be_solve_kernel  : ONE 1024-thread workgroup.  Jacobi scaling, dogleg diagonal, Cauchy point, Schur system in
//                      LDS (packed lower triangle, <= 127 KB), blocked-free right-looking Cholesky, back
//                      substitution of the inverse depths, dogleg interpolation, model cost change, candidate
//                      point x (+) delta.
//   be_accept_kernel : sums the candidate costs in a fixed order, applies Ceres' parameter/function tolerance
//                      tests and the step acceptance rule, updates radius / mu / flags.
// The dense reduced system is n <= 178 (165 with the shipped configs): far too small for MFMA to matter at this
// stage; the kernels are latency-bound by design and are measured as such.
#include <hip/hip_runtime.h>
#include <cfloat>
#include "be_kernels.h"

using namespace be;

#define RED_THREADS 256

__global__ __launch_bounds__(RED_THREADS) void be_reduce_kernel(BeSolveArgs a) {
    const BeCtl c = *a.ctl;
    if (c.done || !c.need_eval && !c.chol_fail) return;      // Hd/Sc are still valid when the last step was rejected
    const int n = a.dims.nstate, nlm = a.dims.nlm, nimu = a.dims.nimu;
    const int t = blockIdx.x * RED_THREADS + threadIdx.x;
    const bool is_mat = t < n * n, is_g = !is_mat && t < n * n + n;
    int i = 0, j = 0;
    if (is_mat) { i = t / n; j = t - i * n; } else if (is_g) { i = t - n * n; j = i; }
    const int ki = a.col_kind[i], fi = a.col_frame[i], ci = a.col_comp[i];
    const int kj = a.col_kind[j], fj = a.col_frame[j], cj = a.col_comp[j];
    double H = 0.0, S = 0.0, G = 0.0, GS = 0.0;
    const bool active = is_mat || is_g;
    if (active) {
        // ---- IMU blocks ----
        for (int k = 0; k < nimu; ++k) {
            const BeImu* m = &a.imu[k];
            int li = -1, lj = -1;
            if (fi == m->fi) li = (ki == 0 ? ci : 6 + ci); else if (fi == m->fj) li = (ki == 0 ? 15 + ci : 21 + ci);
            if (fj == m->fi) lj = (kj == 0 ? cj : 6 + cj); else if (fj == m->fj) lj = (kj == 0 ? 15 + cj : 21 + cj);
            const double* o = a.imu_out + (size_t)k * IMU_OUT_STRIDE;
            if (is_mat) { if (li >= 0 && lj >= 0) H += o[31 + li * 30 + lj]; }
            else if (li >= 0) G += o[1 + li];
        }
        // ---- prior ----
        if (a.prior->valid) {
            const int pi = a.prior_col[i], pj = a.prior_col[j];
            if (is_mat) { if (pi >= 0 && pj >= 0) H += a.priorA[(size_t)pi * a.prior->n + pj]; }
            else if (pi >= 0) G += a.prior_out[1 + pi];
        }
    }
    // ---- landmarks (only pose columns) ----
    __shared__ double s_rho[RED_THREADS];
    const bool pose_pair = active && ki == 0 && kj == 0;
    const double mu = c.mu;
    for (int base = 0; base < nlm; base += RED_THREADS) {
        const int l = base + threadIdx.x;
        if (l < nlm) {
            const double h = a.packets[(size_t)l * BE_PK_SIZE + BE_PK_H];
            double s = c.first ? 1.0 / (1.0 + sqrt(h)) : a.scale_l[l];
            if (c.first && blockIdx.x == 0) a.scale_l[l] = s;
            double d2 = h * s * s; d2 = fmin(fmax(d2, 1e-6), 1e32);
            s_rho[threadIdx.x] = 1.0 / (h + mu * d2 / (s * s));
        }
        __syncthreads();
        if (pose_pair) {
            const int cnt = min(RED_THREADS, nlm - base);
            for (int q = 0; q < cnt; ++q) {
                const int ll = base + q;
                const BeLm L = a.lm[ll];
                if (!((L.mask >> fi) & 1)) continue;
                const double* pk = a.packets + (size_t)ll * BE_PK_SIZE;
                if (is_mat) {
                    if (!((L.mask >> fj) & 1)) continue;
                    S += s_rho[q] * pk[BE_PK_W + fi * 6 + ci] * pk[BE_PK_W + fj * 6 + cj];
                    if (fi == fj) H += pk[BE_PK_DD + fi * 36 + ci * 6 + cj];
                    else if (fi == L.anchor) H += pk[BE_PK_DA + fj * 36 + ci * 6 + cj];
                    else if (fj == L.anchor) H += pk[BE_PK_DA + fi * 36 + cj * 6 + ci];
                } else {
                    G += pk[BE_PK_GP + fi * 6 + ci];
                    GS += s_rho[q] * pk[BE_PK_W + fi * 6 + ci] * pk[BE_PK_G];
                }
            }
        }
        __syncthreads();
    }
    if (is_mat) { a.Hd[t] = H; a.Sc[t] = S; }
    else if (is_g) { a.gvec[i] = G; a.gvec[n + i] = GS; }
}

void be_launch_reduce(const BeSolveArgs& a, hipStream_t s) {
    const int n = a.dims.nstate;
    const int total = n * n + n;
    hipLaunchKernelGGL(be_reduce_kernel, dim3((total + RED_THREADS - 1) / RED_THREADS), dim3(RED_THREADS), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
#define SOL_THREADS 1024

__device__ __forceinline__ double block_sum(double v, double* red) {     // fixed-shape tree: deterministic
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0;
    for (int k = 0; k < SOL_THREADS / 64; ++k) s += red[k];
    return s;
}

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }      // packed lower, j <= i

__global__ __launch_bounds__(SOL_THREADS) void be_solve_kernel(BeSolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    BeCtl* ctl = a.ctl;
    const BeCtl c = *ctl;
    if (c.done) return;
    const int n = a.dims.nstate, nlm = a.dims.nlm, tid = threadIdx.x;
    double* Sm = sm;                                   // n(n+1)/2
    double* v_rhs = Sm + n * (n + 1) / 2;              // n   (rhs -> y_p)
    double* v_s = v_rhs + n;                           // scale
    double* v_d = v_s + n;                             // diag
    double* v_grad = v_d + n;
    double* v_gn = v_grad + n;
    double* v_t = v_gn + n;                            // temp (u_p / delta_p)
    double* v_t2 = v_t + n;                            // H * temp
    double* red = v_t2 + n;                            // 16 + misc
    __shared__ int s_fail;
    const double mu = c.mu;

    if (!c.reuse) {
        // ---------------- scaling, diagonal, gradient ----------------
        for (int i = tid; i < n; i += SOL_THREADS) {
            const double hii = a.Hd[(size_t)i * n + i];
            const double s = c.first ? 1.0 / (1.0 + sqrt(hii)) : a.scale_p[i];
            double d2 = hii * s * s; d2 = fmin(fmax(d2, 1e-6), 1e32);
            const double d = sqrt(d2);
            v_s[i] = s; v_d[i] = d; v_grad[i] = a.gvec[i] * s / d;
            v_t[i] = s * s * a.gvec[i] / d2;           // u_p = S v, v = gradient_/diag
            if (c.first) a.scale_p[i] = s;
            a.diag_p[i] = d; a.grad_p[i] = v_grad[i];
        }
        __syncthreads();
        // x_cost on the first slot (sum of packet/imu/prior costs at x, fixed order)
        if (c.first) {
            double part = 0;
            for (int l = tid; l < nlm; l += SOL_THREADS) part += a.packets[(size_t)l * BE_PK_SIZE + BE_PK_COST];
            for (int k = tid; k < a.dims.nimu; k += SOL_THREADS) part += a.imu_out[(size_t)k * IMU_OUT_STRIDE];
            if (tid == 0) part += a.prior_out[0];
            const double xc = block_sum(part, red);
            if (tid == 0) { ctl->x_cost = xc; ctl->initial_cost = xc; }
        }
        // gradient tolerance (trust_region_minimizer.cc: gradient_max_norm <= gradient_tolerance = 1e-10), checked on every new gradient
        {
            double gm = 0;
            for (int i = tid; i < n; i += SOL_THREADS) gm = fmax(gm, fabs(a.gvec[i]));
            for (int l = tid; l < nlm; l += SOL_THREADS) gm = fmax(gm, fabs(a.packets[(size_t)l * BE_PK_SIZE + BE_PK_G]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_xor(gm, o));
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = gm;
            __syncthreads();
            gm = 0; for (int k = 0; k < SOL_THREADS / 64; ++k) gm = fmax(gm, red[k]);
            __syncthreads();
            if (gm <= 1e-10) { if (tid == 0) { ctl->done = 1; ctl->termination = 1; ctl->first = 0; } return; }
        }
        // H u (pose part) + landmark terms of u^T H u, gradient norm
        for (int i = tid; i < n; i += SOL_THREADS) { double s = 0; const double* row = a.Hd + (size_t)i * n; for (int j = 0; j < n; ++j) s += row[j] * v_t[j]; v_t2[i] = s; }
        __syncthreads();
        double uHu = 0, gg = 0;
        for (int i = tid; i < n; i += SOL_THREADS) { uHu += v_t[i] * v_t2[i]; gg += v_grad[i] * v_grad[i]; }
        for (int l = tid; l < nlm; l += SOL_THREADS) {
            const double* pk = a.packets + (size_t)l * BE_PK_SIZE;
            const BeLm L = a.lm[l];
            const double h = pk[BE_PK_H], gl = pk[BE_PK_G];
            const double s = a.scale_l[l];
            double d2 = h * s * s; d2 = fmin(fmax(d2, 1e-6), 1e32);
            const double d = sqrt(d2);
            const double grad = gl * s / d, u = s * s * gl / d2;
            a.diag_l[l] = d; a.grad_l[l] = grad;
            double wu = 0;
            for (int f = 0; f < a.dims.nframes; ++f) if ((L.mask >> f) & 1) { const int col = a.dims.pose_col[f]; if (col >= 0) for (int r = 0; r < 6; ++r) wu += pk[BE_PK_W + f * 6 + r] * v_t[col + r]; }
            uHu += 2.0 * u * wu + h * u * u;
            gg += grad * grad;
        }
        uHu = block_sum(uHu, red);
        gg = block_sum(gg, red);
        const double alpha = gg / uHu;
        // ---------------- Schur system in LDS ----------------
        for (int e = tid; e < n * (n + 1) / 2; e += SOL_THREADS) {
            int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while (tri(i + 1, 0) <= e) ++i;
            while (tri(i, 0) > e) --i;
            const int j = e - tri(i, 0);
            double v = v_s[i] * v_s[j] * (a.Hd[(size_t)i * n + j] - a.Sc[(size_t)i * n + j]);
            if (i == j) v += mu * v_d[i] * v_d[i];
            Sm[e] = v;
        }
        for (int i = tid; i < n; i += SOL_THREADS) v_rhs[i] = v_s[i] * (a.gvec[i] - a.gvec[n + i]);
        if (tid == 0) s_fail = 0;
        __syncthreads();
        // right-looking Cholesky, packed lower
        for (int k = 0; k < n; ++k) {
            const double pivot = Sm[tri(k, k)];
            if (!(pivot > 0.0) || !isfinite(pivot)) { if (tid == 0) s_fail = 1; break; }      // uniform: every thread reads the same LDS word
            const double inv = 1.0 / sqrt(pivot);
            __syncthreads();
            for (int i = k + tid; i < n; i += SOL_THREADS) Sm[tri(i, k)] *= inv;
            __syncthreads();
            const int m = n - k - 1;
            for (int e = tid; e < m * (m + 1) / 2; e += SOL_THREADS) {
                int ii = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
                while (tri(ii + 1, 0) <= e) ++ii;
                while (tri(ii, 0) > e) --ii;
                const int jj = e - tri(ii, 0);
                const int i = k + 1 + ii, j = k + 1 + jj;
                Sm[tri(i, j)] -= Sm[tri(i, k)] * Sm[tri(j, k)];
            }
            __syncthreads();
        }
        __syncthreads();
        if (s_fail) {     // Ceres: LINEAR_SOLVER_FAILURE -> mu *= 10 and retry (dogleg_strategy.cc ComputeGaussNewtonStep)
            if (tid == 0) {
                ctl->mu = mu * 10.0; ctl->chol_fail = 1; ctl->first = 0; ctl->alpha = alpha;
                if (mu * 10.0 > 1.0) { ctl->done = 1; ctl->termination = 2; }
            }
            return;
        }
        // forward / backward substitution (n <= 178: one wave, LDS resident)
        if (tid < 64) {
            for (int i = 0; i < n; ++i) {
                double part = 0;
                for (int k = tid; k < i; k += 64) part += Sm[tri(i, k)] * v_rhs[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
                if (tid == 0) v_rhs[i] = (v_rhs[i] - part) / Sm[tri(i, i)];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
            for (int i = n - 1; i >= 0; --i) {
                double part = 0;
                for (int k = i + 1 + tid; k < n; k += 64) part += Sm[tri(k, i)] * v_rhs[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
                if (tid == 0) v_rhs[i] = (v_rhs[i] - part) / Sm[tri(i, i)];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int i = tid; i < n; i += SOL_THREADS) { v_gn[i] = -v_d[i] * v_rhs[i]; a.gn_p[i] = v_gn[i]; v_t[i] = v_s[i] * v_rhs[i]; }   // v_t = s_p . y_p
        __syncthreads();
        for (int l = tid; l < nlm; l += SOL_THREADS) {
            const double* pk = a.packets + (size_t)l * BE_PK_SIZE;
            const BeLm L = a.lm[l];
            const double h = pk[BE_PK_H], s = a.scale_l[l], d = a.diag_l[l];
            const double rho = 1.0 / (h + mu * d * d / (s * s));
            double wy = 0;
            for (int f = 0; f < a.dims.nframes; ++f) if ((L.mask >> f) & 1) { const int col = a.dims.pose_col[f]; if (col >= 0) for (int r = 0; r < 6; ++r) wy += pk[BE_PK_W + f * 6 + r] * v_t[col + r]; }
            const double yl = rho / s * (pk[BE_PK_G] - wy);
            a.gn_l[l] = -d * yl;
        }
        if (tid == 0) { ctl->alpha = alpha; }
        __syncthreads();
        red[20] = alpha;
    } else {
        for (int i = tid; i < n; i += SOL_THREADS) { v_s[i] = a.scale_p[i]; v_d[i] = a.diag_p[i]; v_grad[i] = a.grad_p[i]; v_gn[i] = a.gn_p[i]; }
        if (tid == 0) red[20] = c.alpha;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    const double alpha = red[20];
    // ---------------- traditional dogleg (dogleg_strategy.cc ComputeTraditionalDoglegStep) ----------------
    double p_gg = 0, p_nn = 0, p_gn = 0;
    for (int i = tid; i < n; i += SOL_THREADS) { p_gg += v_grad[i] * v_grad[i]; p_nn += v_gn[i] * v_gn[i]; p_gn += v_grad[i] * v_gn[i]; }
    for (int l = tid; l < nlm; l += SOL_THREADS) { const double g = a.grad_l[l], q = a.gn_l[l]; p_gg += g * g; p_nn += q * q; p_gn += g * q; }
    const double gnorm = sqrt(block_sum(p_gg, red)), gnn = sqrt(block_sum(p_nn, red)), gdot = block_sum(p_gn, red);
    const double radius = c.radius;
    double cg, cn, dnorm;
    if (gnn <= radius) { cg = 0; cn = 1; dnorm = gnn; }
    else if (gnorm * alpha >= radius) { cg = -(radius / gnorm); cn = 0; dnorm = radius; }
    else {
        const double b_dot_a = -alpha * gdot, a2 = (alpha * gnorm) * (alpha * gnorm), bma2 = a2 - 2 * b_dot_a + gnn * gnn;
        const double cc = b_dot_a - a2, dd = sqrt(cc * cc + bma2 * (radius * radius - a2));
        const double beta = (cc <= 0) ? (dd - cc) / bma2 : (radius * radius - a2) / (dd + cc);
        cg = -alpha * (1.0 - beta); cn = beta; dnorm = -1.0;
    }
    // delta = (cg*grad + cn*gn) / diag * scale   (v_t = delta_p)
    double p_dn = 0;
    for (int i = tid; i < n; i += SOL_THREADS) { const double st = cg * v_grad[i] + cn * v_gn[i]; p_dn += st * st; v_t[i] = st / v_d[i] * v_s[i]; }
    __syncthreads();
    for (int i = tid; i < n; i += SOL_THREADS) { double s = 0; const double* row = a.Hd + (size_t)i * n; for (int j = 0; j < n; ++j) s += row[j] * v_t[j]; v_t2[i] = s; }
    __syncthreads();
    double p_sg = 0, p_sHs = 0, p_step = 0, p_xn = 0;
    for (int i = tid; i < n; i += SOL_THREADS) { p_sg += v_t[i] * a.gvec[i]; p_sHs += v_t[i] * v_t2[i]; }
    for (int l = tid; l < nlm; l += SOL_THREADS) {
        const double* pk = a.packets + (size_t)l * BE_PK_SIZE;
        const BeLm L = a.lm[l];
        const double st = cg * a.grad_l[l] + cn * a.gn_l[l];
        p_dn += st * st;
        const double dl = st / a.diag_l[l] * a.scale_l[l];
        double wd = 0;
        for (int f = 0; f < a.dims.nframes; ++f) if ((L.mask >> f) & 1) { const int col = a.dims.pose_col[f]; if (col >= 0) for (int r = 0; r < 6; ++r) wd += pk[BE_PK_W + f * 6 + r] * v_t[col + r]; }
        p_sg += dl * pk[BE_PK_G];
        p_sHs += 2.0 * dl * wd + pk[BE_PK_H] * dl * dl;
        const double x0 = a.x->inv_depth[l];
        a.cand->inv_depth[l] = x0 + dl;
        p_step += dl * dl; p_xn += x0 * x0;
    }
    // candidate poses / speed-bias
    for (int f = tid; f < BE_NF; f += SOL_THREADS) {
        const int pc = f < a.dims.nframes ? a.dims.pose_col[f] : -1, sc = f < a.dims.nframes ? a.dims.sb_col[f] : -1;
        if (pc >= 0) {
            double out[7];
            pose_plus(a.x->pose[f], &v_t[pc], a.dims.plane_kind, out);
            for (int k = 0; k < 7; ++k) { const double d = out[k] - a.x->pose[f][k]; p_step += d * d; p_xn += a.x->pose[f][k] * a.x->pose[f][k]; a.cand->pose[f][k] = out[k]; }
        } else for (int k = 0; k < 7; ++k) a.cand->pose[f][k] = a.x->pose[f][k];
        if (sc >= 0) for (int k = 0; k < 9; ++k) { const double d = v_t[sc + k]; p_step += d * d; p_xn += a.x->sb[f][k] * a.x->sb[f][k]; a.cand->sb[f][k] = a.x->sb[f][k] + d; }
        else for (int k = 0; k < 9; ++k) a.cand->sb[f][k] = a.x->sb[f][k];
    }
    if (tid == 0) { for (int k = 0; k < 14; ++k) a.cand->ex[k / 7][k % 7] = a.x->ex[k / 7][k % 7]; a.cand->td = a.x->td; }
    const double dn2 = block_sum(p_dn, red), sg = block_sum(p_sg, red), sHs = block_sum(p_sHs, red);
    const double step2 = block_sum(p_step, red), xn2 = block_sum(p_xn, red);
    if (tid == 0) {
        const double mcc = -(sg + 0.5 * sHs);
        ctl->model_cost_change = mcc;
        ctl->dogleg_norm = dnorm >= 0 ? dnorm : sqrt(dn2);
        ctl->step_valid = mcc > 0.0 ? 1 : 0;
        ctl->step_norm = sqrt(step2);
        ctl->x_norm = sqrt(xn2);
        ctl->chol_fail = 0;
        ctl->first = 0;
    }
}

static size_t solve_smem(int n) { return ((size_t)n * (n + 1) / 2 + 7 * (size_t)n + 64) * sizeof(double); }

int be_launch_solve(const BeSolveArgs& a, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_smem(BE_MAX_STATE)) != hipSuccess) return -1;
        attr = true;
    }
    hipLaunchKernelGGL(be_solve_kernel, dim3(1), dim3(SOL_THREADS), solve_smem(a.dims.nstate), s, a);
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void be_accept_kernel(BeSolveArgs a) {
    BeCtl* ctl = a.ctl;
    const BeCtl c = *ctl;
    __shared__ double red[4];
    __shared__ int s_accept;
    const int tid = threadIdx.x;
    if (c.done) return;
    if (tid == 0) { ctl->slots = c.slots + 1; s_accept = 0; }
    if (c.chol_fail) { if (tid == 0) { ctl->reuse = 0; ctl->need_eval = 0; } return; }     // retry slot with the larger mu
    const int iter = c.iter + 1;
    if (!c.step_valid) {       // HandleInvalidStep
        if (tid == 0) {
            ctl->iter = iter; ctl->invalid = c.invalid + 1; ctl->mu = c.mu * 10.0; ctl->reuse = 0; ctl->need_eval = 0; ctl->chol_fail = 1;   // chol_fail=1 forces the reduce to rebuild Sc with the new mu
            if (c.invalid + 1 >= 5) { ctl->done = 1; ctl->termination = 2; }
            else if (iter >= c.max_iters) { ctl->done = 1; ctl->termination = 0; }
        }
        return;
    }
    // candidate cost: fixed-order sum
    const int ncost = a.dims.nlm + a.dims.nimu + 1;
    double part = 0;
    for (int k = tid; k < ncost; k += 256) part += a.cand_cost[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const double cand_cost = red[0] + red[1] + red[2] + red[3];
    if (tid == 0) {
        ctl->iter = iter; ctl->cand_cost = cand_cost; ctl->invalid = 0;
        bool done = false; int term = 0;
        if (c.step_norm <= 1e-8 * (c.x_norm + 1e-8)) { done = true; term = 1; }                       // parameter tolerance
        else if (fabs(c.x_cost - cand_cost) <= 1e-6 * c.x_cost) { done = true; term = 1; }           // function tolerance
        else {
            const double rel = (c.x_cost - cand_cost) / c.model_cost_change;
            if (rel > 1e-3) {
                s_accept = 1;
                double radius = c.radius;
                if (rel < 0.25) radius *= 0.5;
                if (rel > 0.75) radius = fmax(radius, 3.0 * c.dogleg_norm);
                ctl->radius = radius; ctl->mu = fmax(1e-8, 2.0 * c.mu / 10.0);
                ctl->x_cost = cand_cost; ctl->successful = c.successful + 1; ctl->reuse = 0; ctl->need_eval = 1;
            } else {
                ctl->radius = c.radius * 0.5; ctl->reuse = 1; ctl->need_eval = 0;
                if (c.radius * 0.5 < 1e-32) { done = true; term = 1; }
            }
            if (!done && iter >= c.max_iters) { done = true; term = 0; }
        }
        if (done) { ctl->done = 1; ctl->termination = term; }
    }
    __syncthreads();
    if (s_accept) {
        const int nd = sizeof(BeState) / sizeof(double);
        double* dst = reinterpret_cast<double*>(a.x);
        const double* src = reinterpret_cast<const double*>(a.cand);
        const int used = (int)(offsetof(BeState, inv_depth) / sizeof(double)) + a.dims.nlm;
        for (int k = tid; k < used && k < nd; k += 256) dst[k] = src[k];
    }
}

void be_launch_accept(const BeSolveArgs& a, hipStream_t s) { hipLaunchKernelGGL(be_accept_kernel, dim3(1), dim3(256), 0, s, a); }
