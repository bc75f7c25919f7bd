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
//                      LDS (packed lower triangle, <= 127 KB), right-looking LDL^T with thread-owned entries and ONE
//                      barrier per column (the right-hand side rides along as an extra row), back
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

#define RED_THREADS 384          // 6 waves: wave = row of a 6x6 pose block, lane = landmark
#define RED_PAIRS (BE_NF * BE_NF)

// IMU + prior part of Hd(i, j)  (everything that is not a landmark sum)
__device__ __forceinline__ double red_dense_h(const BeSolveArgs& a, int i, int j) {
    const int ki = a.col_kind[i], fi = a.col_frame[i], ci = a.col_comp[i];
    const int kj = a.col_kind[j], fj = a.col_frame[j], cj = a.col_comp[j];
    double H = 0.0;
    for (int k = 0; k < a.dims.nimu; ++k) {
        const BeImu* m = &a.imu[k];
        int li = -1, lj = -1;
        if (fi == m->fi) li = (ki == 0 ? ci : 6 + ci); else if (fi == m->fj) li = (ki == 0 ? 15 + ci : 21 + ci);
        if (fj == m->fi) lj = (kj == 0 ? cj : 6 + cj); else if (fj == m->fj) lj = (kj == 0 ? 15 + cj : 21 + cj);
        if (li >= 0 && lj >= 0) H += a.imu_out[(size_t)k * IMU_OUT_STRIDE + 31 + li * 30 + lj];
    }
    if (a.prior->valid) {
        const int pi = a.prior_col[i], pj = a.prior_col[j];
        if (pi >= 0 && pj >= 0) H += a.priorA[(size_t)pi * a.prior->n + pj];
    }
    return H;
}
__device__ __forceinline__ double red_dense_g(const BeSolveArgs& a, int i) {
    const int ki = a.col_kind[i], fi = a.col_frame[i], ci = a.col_comp[i];
    double G = 0.0;
    for (int k = 0; k < a.dims.nimu; ++k) {
        const BeImu* m = &a.imu[k];
        int li = -1;
        if (fi == m->fi) li = (ki == 0 ? ci : 6 + ci); else if (fi == m->fj) li = (ki == 0 ? 15 + ci : 21 + ci);
        if (li >= 0) G += a.imu_out[(size_t)k * IMU_OUT_STRIDE + 1 + li];
    }
    if (a.prior->valid) { const int pi = a.prior_col[i]; if (pi >= 0) G += a.prior_out[1 + pi]; }
    return G;
}
__device__ __forceinline__ double wave_sum(double v) {      // fixed xor tree: deterministic, every lane gets the sum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// blocks [0, 121): pose block (fi, fj) of the reduced system — landmark sums read the transposed packets coalesced
//                  (lane = landmark), wave-tree reduced; blocks [121, ..): every entry that has no landmark term.
__global__ __launch_bounds__(RED_THREADS) void be_reduce_kernel(BeSolveArgs a) {
    const BeCtl c = *a.ctl;
    if (c.done || (!c.need_eval && !c.chol_fail)) return;      // Hd/Sc are still valid when the last step was rejected
    const int n = a.dims.nstate, nlm = a.dims.nlm;
    const double* pk = a.packets;
    if (blockIdx.x < RED_PAIRS) {
        const int fi = blockIdx.x / BE_NF, fj = blockIdx.x - fi * BE_NF;
        if (fi >= a.dims.nframes || fj >= a.dims.nframes) return;
        const int ci0 = a.dims.pose_col[fi], cj0 = a.dims.pose_col[fj];
        if (ci0 < 0 || cj0 < 0) return;
        const int ci = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const bool diag = fi == fj;
        double S[6] = {0, 0, 0, 0, 0, 0}, H[6] = {0, 0, 0, 0, 0, 0}, G = 0, GS = 0;
        const int e_wi = BE_PK_W + fi * 6 + ci, e_wj = BE_PK_W + fj * 6;
        const int e_dd = BE_PK_DD + fi * 36 + ci * 6;
        const int e_da_i = BE_PK_DA + fj * 36 + ci * 6;                 // anchor == fi : row ci of block (anchor, fj)
        const int e_da_j = BE_PK_DA + fi * 36 + ci;                     // anchor == fj : column ci of the transposed block
        for (int l = lane; l < nlm; l += 64) {
            const double h = BE_PK(pk, BE_PK_H, l);
            const double s = c.first ? 1.0 / (1.0 + sqrt(h)) : a.scale_l[l];
            double d2 = h * s * s; d2 = fmin(fmax(d2, 1e-6), 1e32);
            const double rho = 1.0 / (h + c.mu * d2 / (s * s));
            const double wi = BE_PK(pk, e_wi, l);
#pragma unroll
            for (int q = 0; q < 6; ++q) S[q] += rho * (wi * BE_PK(pk, e_wj + q, l));      // rho * (wi * wj): bitwise symmetric
            if (diag) {
#pragma unroll
                for (int q = 0; q < 6; ++q) H[q] += BE_PK(pk, e_dd + q, l);
                G += BE_PK(pk, BE_PK_GP + fi * 6 + ci, l);
                GS += rho * (wi * BE_PK(pk, BE_PK_G, l));
            } else {
                const int anc = a.lm[l].anchor;
                if (anc == fi) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) H[q] += BE_PK(pk, e_da_i + q, l);
                } else if (anc == fj) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) H[q] += BE_PK(pk, e_da_j + q * 6, l);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) { S[q] = wave_sum(S[q]); H[q] = wave_sum(H[q]); }
        if (diag) { G = wave_sum(G); GS = wave_sum(GS); }
        if (lane < 6) {
            double sv = S[0], hv = H[0];
#pragma unroll
            for (int q = 1; q < 6; ++q) if (lane == q) { sv = S[q]; hv = H[q]; }
            const int i = ci0 + ci, j = cj0 + lane;
            a.Hd[(size_t)i * n + j] = hv + red_dense_h(a, i, j);
            a.Sc[(size_t)i * n + j] = sv;
        } else if (diag && lane == 6) {
            const int i = ci0 + ci;
            a.gvec[i] = G + red_dense_g(a, i);
            a.gvec[n + i] = GS;
        }
        return;
    }
    const int eb = blockIdx.x - RED_PAIRS;
    if (eb == 0 && c.first)
        for (int l = threadIdx.x; l < nlm; l += RED_THREADS) a.scale_l[l] = 1.0 / (1.0 + sqrt(BE_PK(pk, BE_PK_H, l)));
    const int t = eb * RED_THREADS + threadIdx.x;
    if (t < n * n) {
        const int i = t / n, j = t - i * n;
        if (a.col_kind[i] == 0 && a.col_kind[j] == 0) return;          // pose x pose: written by the pair blocks
        a.Hd[t] = red_dense_h(a, i, j); a.Sc[t] = 0.0;
    } else if (t < n * n + n) {
        const int i = t - n * n;
        if (a.col_kind[i] == 0) return;
        a.gvec[i] = red_dense_g(a, i); a.gvec[n + i] = 0.0;
    }
}

void be_launch_reduce(const BeSolveArgs& a, hipStream_t s) {
    const int n = a.dims.nstate;
    const int total = n * n + n;
    hipLaunchKernelGGL(be_reduce_kernel, dim3(RED_PAIRS + (total + RED_THREADS - 1) / RED_THREADS), dim3(RED_THREADS), 0, s, a);
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

// y = Hd * x (Hd symmetric: thread = output entry reading column-wise, i.e. coalesced rows), 4 partial sums per entry
__device__ __forceinline__ void gemv_hd(const double* Hd, int n, const double* x, double* y, double* scratch, int tid) {
    const int col = tid & 255, part = tid >> 8;
    if (col < n) {
        const int seg = (n + 3) / 4, j0 = part * seg, j1 = min(n, j0 + seg);
        double s = 0;
        for (int j = j0; j < j1; ++j) s += Hd[(size_t)j * n + col] * x[j];
        scratch[tid] = s;
    }
    __syncthreads();
    if (part == 0 && col < n) y[col] = (scratch[col] + scratch[256 + col]) + (scratch[512 + col] + scratch[768 + col]);
    __syncthreads();
}

// Register-blocked right-looking LDL^T of the (scaled, damped) reduced camera system.  Every thread owns NSLOT 4x4
// blocks of the lower triangle in registers (block-column-major, so whole waves retire as the factorisation
// proceeds); the pivot column travels through a double-buffered LDS vector -> ONE barrier per column and 8 LDS reads
// per 16 FMAs.  The right-hand side is forward-substituted alongside (z lives in a register of thread i).
// On success: Lm = unit-lower L (packed), dvec = D, zfin = L^-1 rhs.   Returns false on a non-positive pivot.
template <int NSLOT>
__device__ __forceinline__ bool ldlt_blocked(const BeSolveArgs& a, int n, double mu, const double* v_s, const double* v_d,
                                             double* Lm, double* colbuf, double* zfin, double* dvec, int* s_fail) {
    const int tid = threadIdx.x;
    const int NBR = (n + 3) >> 2, CB = NBR * 4, nblk = NBR * (NBR + 1) / 2;
    int bi[NSLOT], bj[NSLOT];
    double A[NSLOT][4][4];
#pragma unroll
    for (int b = 0; b < NSLOT; ++b) {
        const int idx = tid + b * SOL_THREADS;
        bi[b] = -1; bj[b] = -1;
        if (idx < nblk) { int c0 = 0, rem = idx; while (rem >= NBR - c0) { rem -= NBR - c0; ++c0; } bj[b] = c0; bi[b] = c0 + rem; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int i = bi[b] * 4 + r, j = bj[b] * 4 + cc;
                double v = 0.0;
                if (bi[b] >= 0 && i < n && j < n) {
                    v = v_s[i] * v_s[j] * (a.Hd[(size_t)i * n + j] - a.Sc[(size_t)i * n + j]);
                    if (i == j) v += mu * v_d[i] * v_d[i];
                }
                A[b][r][cc] = v;
            }
    }
    double z = 0.0;
    if (tid < n) z = v_s[tid] * (a.gvec[tid] - a.gvec[n + tid]);
    if (tid == 0) { zfin[0] = z; *s_fail = 0; }
#pragma unroll
    for (int b = 0; b < NSLOT; ++b)
        if (bj[b] == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) colbuf[bi[b] * 4 + r] = A[b][r][0];
        }
    __syncthreads();
    bool bad = false;
    for (int k0 = 0; k0 < n && !bad; k0 += 4) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {           // kk is a compile-time constant: every index into A[][][] stays static
            const int k = k0 + kk;
            if (k >= n || bad) break;
            const double* cb = colbuf + (kk & 1) * CB;
            double* cbn = colbuf + ((kk + 1) & 1) * CB;
            const double dk = cb[k];
            if (!(dk > 0.0) || !isfinite(dk)) { if (tid == 0) *s_fail = 1; bad = true; break; }      // uniform: every thread reads the same LDS word
            const double inv = 1.0 / dk;
            if (tid > k && tid < n) { z -= cb[tid] * inv * zfin[k]; if (tid == k + 1) zfin[k + 1] = z; }
#pragma unroll
            for (int b = 0; b < NSLOT; ++b) {
                const int j0 = bj[b] * 4;
                if (kk < 3 ? j0 >= k0 : j0 > k0) {      // block still has a column right of the pivot (bj = -1 never passes)
                    const int i0 = bi[b] * 4;
                    const bool pivot_blockcol = j0 == k0;
                    double ci[4], cj[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ci[r] = cb[i0 + r]; const double t = cb[j0 + r] * inv; cj[r] = (!pivot_blockcol || r > kk) ? t : 0.0; }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) A[b][r][cc] -= ci[r] * cj[cc];
                    if (kk < 3) {
                        if (pivot_blockcol) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (i0 + r > k) cbn[i0 + r] = A[b][r][(kk + 1) & 3];
                        }
                    } else if (j0 == k0 + 4) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) cbn[i0 + r] = A[b][r][0];
                    }
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (*s_fail) return false;
    // D, then the unit-lower factor in packed row-major form for the back substitution
#pragma unroll
    for (int b = 0; b < NSLOT; ++b)
        if (bi[b] >= 0 && bi[b] == bj[b]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (bi[b] * 4 + r < n) dvec[bi[b] * 4 + r] = A[b][r][r];
        }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NSLOT; ++b)
        if (bi[b] >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int i = bi[b] * 4 + r, j = bj[b] * 4 + cc;
                    if (i < n && j < i) Lm[tri(i, j)] = A[b][r][cc] / dvec[j];
                }
        }
    __syncthreads();
    return true;
}

template <int NSLOT>
__global__ __launch_bounds__(SOL_THREADS) void be_solve_kernel(BeSolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    BeCtl* ctl = a.ctl;
    const BeCtl c = *ctl;
    if (c.done) return;
    const int n = a.dims.nstate, nlm = a.dims.nlm, tid = threadIdx.x;
    const int npk = n * (n + 1) / 2;
    double* Lm = sm;                                   // npk: unit-lower factor, packed row-major
    double* v_s = Lm + npk;                            // scale
    double* v_d = v_s + n;                             // diag
    double* v_grad = v_d + n;
    double* v_gn = v_grad + n;
    double* v_t = v_gn + n;                            // u_p / s.y / delta_p
    double* v_t2 = v_t + n;                            // H * v_t
    double* v_x = v_t2 + n;                            // back-substitution vector
    double* zfin = v_x + n;                            // L^-1 rhs
    double* dvec = zfin + n;                           // D
    double* q66 = dvec + n;                            // 66 (+6 pad): v_t gathered into packet (frame, comp) order
    double* red = q66 + 72;                            // 32
    double* scratch = red + 32;                        // 1024
    double* colbuf = scratch + 1024;                   // 2 x 184 pivot-column buffers
    __shared__ int s_fail;
    const double mu = c.mu;
    const double* pk = a.packets;
    auto gather66 = [&](const double* v) {             // q66[a*6+r] = v[pose_col[a] + r] (0 for constant / absent poses)
        if (tid < 66) { const int fa = tid / 6, r = tid - fa * 6; const int col = fa < a.dims.nframes ? a.dims.pose_col[fa] : -1; q66[tid] = col >= 0 ? v[col + r] : 0.0; }
        __syncthreads();
    };
    auto wdot = [&](int l) { double s = 0;
#pragma unroll 6
        for (int q = 0; q < 66; ++q) s += BE_PK(pk, BE_PK_W + q, l) * q66[q];
        return s; };

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
        if (c.first) {        // x_cost: fixed-order sum of the per-block costs at x
            double part = 0;
            for (int l = tid; l < nlm; l += SOL_THREADS) part += BE_PK(pk, BE_PK_COST, l);
            for (int k = tid; k < a.dims.nimu; k += SOL_THREADS) part += a.imu_out[(size_t)k * IMU_OUT_STRIDE];
            if (tid == 0) part += a.prior_out[0];
            const double xc = block_sum(part, red);
            if (tid == 0) { ctl->x_cost = xc; ctl->initial_cost = xc; }
        }
        // gradient tolerance (trust_region_minimizer.cc: gradient_max_norm <= gradient_tolerance = 1e-10), checked on every new gradient
        {
            double gm = 0;
            for (int i = tid; i < n; i += SOL_THREADS) gm = fmax(gm, fabs(a.gvec[i]));
            for (int l = tid; l < nlm; l += SOL_THREADS) gm = fmax(gm, fabs(BE_PK(pk, BE_PK_G, l)));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_xor(gm, o));
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = gm;
            __syncthreads();
            gm = 0; for (int k = 0; k < SOL_THREADS / 64; ++k) gm = fmax(gm, red[k]);
            __syncthreads();
            if (gm <= 1e-10) { if (tid == 0) { ctl->done = 1; ctl->termination = 1; ctl->first = 0; } return; }
        }
        // Cauchy point: alpha = |gradient_|^2 / (u^T H u)
        gemv_hd(a.Hd, n, v_t, v_t2, scratch, tid);
        gather66(v_t);
        double uHu = 0, gg = 0;
        for (int i = tid; i < n; i += SOL_THREADS) { uHu += v_t[i] * v_t2[i]; gg += v_grad[i] * v_grad[i]; }
        for (int l = tid; l < nlm; l += SOL_THREADS) {
            const double h = BE_PK(pk, BE_PK_H, l), gl = BE_PK(pk, BE_PK_G, l), s = a.scale_l[l];
            double d2 = h * s * s; d2 = fmin(fmax(d2, 1e-6), 1e32);
            const double d = sqrt(d2), grad = gl * s / d, u = s * s * gl / d2;
            a.diag_l[l] = d; a.grad_l[l] = grad;
            uHu += 2.0 * u * wdot(l) + h * u * u;
            gg += grad * grad;
        }
        uHu = block_sum(uHu, red);
        gg = block_sum(gg, red);
        const double alpha = gg / uHu;
        // ---------------- Gauss-Newton step: LDL^T of the Schur complement ----------------
        if (!ldlt_blocked<NSLOT>(a, n, mu, v_s, v_d, Lm, colbuf, zfin, dvec, &s_fail)) {
            // Ceres: LINEAR_SOLVER_FAILURE -> mu *= 10 and retry (dogleg_strategy.cc ComputeGaussNewtonStep)
            if (tid == 0) {
                ctl->mu = mu * 10.0; ctl->chol_fail = 1; ctl->first = 0; ctl->alpha = alpha;
                if (mu * 10.0 > 1.0) { ctl->done = 1; ctl->termination = 2; }
            }
            return;
        }
        // back substitution L^T x = D^-1 z on one wave (column sweep, wave-level sync only)
        for (int i = tid; i < n; i += SOL_THREADS) v_x[i] = zfin[i] / dvec[i];
        __syncthreads();
        if (tid < 64) {
            for (int k = n - 1; k > 0; --k) {
                const double xk = v_x[k];
                const double* row = Lm + tri(k, 0);
                for (int j = tid; j < k; j += 64) v_x[j] -= row[j] * xk;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        __syncthreads();
        for (int i = tid; i < n; i += SOL_THREADS) { v_gn[i] = -v_d[i] * v_x[i]; a.gn_p[i] = v_gn[i]; v_t[i] = v_s[i] * v_x[i]; }   // v_t = s_p . y_p
        __syncthreads();
        gather66(v_t);
        for (int l = tid; l < nlm; l += SOL_THREADS) {
            const double h = BE_PK(pk, BE_PK_H, l), s = a.scale_l[l], d = a.diag_l[l];
            const double rho = 1.0 / (h + mu * d * d / (s * s));
            const double yl = rho / s * (BE_PK(pk, BE_PK_G, l) - wdot(l));
            a.gn_l[l] = -d * yl;
        }
        if (tid == 0) { ctl->alpha = alpha; red[20] = alpha; }
        __syncthreads();
    } else {
        for (int i = tid; i < n; i += SOL_THREADS) { v_s[i] = a.scale_p[i]; v_d[i] = a.diag_p[i]; v_grad[i] = a.grad_p[i]; v_gn[i] = a.gn_p[i]; }
        if (tid == 0) red[20] = c.alpha;
        __syncthreads();
    }
    const double alpha = red[20];
    __syncthreads();
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
    gemv_hd(a.Hd, n, v_t, v_t2, scratch, tid);
    gather66(v_t);
    double p_sg = 0, p_sHs = 0, p_step = 0, p_xn = 0;
    for (int i = tid; i < n; i += SOL_THREADS) { p_sg += v_t[i] * a.gvec[i]; p_sHs += v_t[i] * v_t2[i]; }
    for (int l = tid; l < nlm; l += SOL_THREADS) {
        const double st = cg * a.grad_l[l] + cn * a.gn_l[l];
        p_dn += st * st;
        const double dl = st / a.diag_l[l] * a.scale_l[l];
        p_sg += dl * BE_PK(pk, BE_PK_G, l);
        p_sHs += 2.0 * dl * wdot(l) + BE_PK(pk, BE_PK_H, l) * dl * dl;
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

static size_t solve_smem(int n) { return ((size_t)n * (n + 1) / 2 + 9 * (size_t)n + 72 + 32 + 1024 + 2 * 184) * sizeof(double); }

int be_launch_solve(const BeSolveArgs& a, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_solve_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_smem(BE_MAX_STATE)) != hipSuccess) return -1;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_solve_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_smem(BE_MAX_STATE)) != hipSuccess) return -1;
        attr = true;
    }
    const int nbr = (a.dims.nstate + 3) / 4;
    if (nbr * (nbr + 1) / 2 <= SOL_THREADS) hipLaunchKernelGGL(be_solve_kernel<1>, dim3(1), dim3(SOL_THREADS), solve_smem(a.dims.nstate), s, a);
    else hipLaunchKernelGGL(be_solve_kernel<2>, dim3(1), dim3(SOL_THREADS), solve_smem(a.dims.nstate), s, a);
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
