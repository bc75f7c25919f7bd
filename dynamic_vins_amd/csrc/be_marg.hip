// be_marg.hip — Schur-complement marginalization on gfx950 (replaces MarginalizationInfo::preMarginalize +
// marginalize, estimator/factor/marginalization_factor.cpp:124-309, driven by Estimator::SetMarginalizationInfo,
// estimator/estimator.cpp:403-619).
//
// Three launches on the BA stream:
//   be_marg_lm_kernel     one 256-thread workgroup per landmark anchored in the dropped frame: every residual block is
//                         re-evaluated at the linearisation point with ALL its Jacobians (pose_i, pose_j, ex0, ex1, td,
//                         inverse depth) and the Huber corrector (ResidualBlockInfo::Evaluate), expanded to dense rows
//                         in LDS, and the landmark's own inverse depth is eliminated on the spot (A_ll is 1x1):
//                         M_l = J^T J - w w^T / h,  b_l = J^T r - w g / h   written as a dense (D x D + D) slab.
//   be_marg_sum_kernel    sums the slabs over landmarks in a fixed order (bitwise reproducible).
//   be_marg_finish_kernel ONE 1024-thread workgroup, system resident in LDS (<= 97 x 97 fp64): adds the IMU factor (0,1)
//                         and the previous prior (as A', b' + A' dx), eliminates the dropped pose / speed-bias block
//                         (m' <= 15) with a dense Cholesky, writes the new prior in INFORMATION FORM (A', b', c0).
// The reference factors A' = Q S Q^T and stores J0 = S^1/2 Q^T, r0 = S^-1/2 Q^T b', but every consumer only ever
// forms J0^T J0 = A', J0^T r0 = b' and r0^T r0 = c0.  c0 = b'^T A'^+ b' comes from an LDL^T factorization that skips
// pivots <= the reference's eigenvalue threshold (1e-8); it reproduces the eigen-clamped value to the accuracy the
// quantity has at all (its 1/lambda-weighted rounding noise is ~1e-4 relative) — two dense
// eigendecompositions per frame are replaced by two small factorizations.
// The reference's pseudo-inverse of A_mm equals the inverse whenever lambda_min(A_mm) > 1e-8; the kernel reports the
// smallest pivot it met so the host can verify that (it is ~1e4 on every sequence measured).
#include <hip/hip_runtime.h>
#include <cfloat>
#include "be_kernels.h"
#include "dev_once.h"

using namespace be;

#ifdef BE_MARG_TS
__device__ long long be_marg_ts[32];
#define MTS(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) be_marg_ts[k] = wall_clock64(); } while (0)
extern "C" int dv_debug_marg_ts(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(be_marg_ts), sizeof(long long) * 32) == hipSuccess ? 0 : -1; }
#else
#define MTS(k) do {} while (0)
#endif
#define MG_THREADS 1024
__device__ __forceinline__ double mg_rcp(double d) {      // v_rcp_f64 + two Newton steps: full precision at a fraction of a division's latency (on the pivot chain)
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
}
#define LM_THREADS 256

// offsets of the two Jacobian entries (row 0, row 1) of column `comp` of slot `slot` inside a factor record
// Jf layout (54): r2 | Ji 12 | Jj 12 | Jex0 12 | Jex1 12 | Jl 2 | Jtd 2 ; returns false if the factor does not touch the slot
// (offsets, not pointers: selecting between LDS pointers and nullptr miscompiled on gfx950 / ROCm 7.2)
__device__ __forceinline__ bool mg_joff(int slot, int comp, int anchor, int fj, int two_frame, int& o0, int& o1) {
    int base;
    if (slot < BE_NF) { if (!two_frame) return false; if (slot == anchor) base = 2; else if (slot == fj) base = 14; else return false; }
    else if (slot == BE_NF) base = 26;
    else if (slot == BE_NF + 1) base = 38;
    else { o0 = 52; o1 = 53; return true; }
    o0 = base + comp; o1 = base + 6 + comp;
    return true;
}

__global__ __launch_bounds__(LM_THREADS) void be_marg_lm_kernel(BeMargArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = a.D, tid = threadIdx.x, l = blockIdx.x;
    double* Jb = sm;                                   // BE_MAX_OBS_FACTORS x 54
    double* Jd = Jb + BE_MAX_OBS_FACTORS * 54;         // (2 nf) x D dense rows
    double* wv = Jd + 2 * BE_MAX_OBS_FACTORS * D;      // D
    double* gpv = wv + D;                              // D
    __shared__ FrameGeom fg[BE_NF];
    __shared__ m33 ric[2];
    __shared__ d3 tic[2];
    __shared__ int s_fj[BE_MAX_OBS_FACTORS], s_two[BE_MAX_OBS_FACTORS];
    __shared__ double s_hg[2];
    const BeState* st = a.x;
    if (l == a.nlm) {
        // extra block (present when the window has an IMU factor to marginalize): factor (0,1) evaluated at the solved state and whitened, beside the landmark
        // blocks.  be_marg_finish used to do this itself: a serial raw evaluation on one lane in the middle of a single-workgroup kernel.
        double* Jraw = sm; double* Jw = sm + 450; double* rr = sm + 900;
        __shared__ BeImu s_m;
        for (int i = tid; i < 450; i += LM_THREADS) Jraw[i] = 0.0;
        {
            const double* src = reinterpret_cast<const double*>(a.imu); double* dst = reinterpret_cast<double*>(&s_m);
            for (int i = tid; i < (int)(sizeof(BeImu) / 8); i += LM_THREADS) dst[i] = src[i];
        }
        __syncthreads();
        const BeImu* mi = &s_m;
        if (tid == 0) imu_raw<true>(*mi, a.g_norm, st->pose[mi->fi], st->sb[mi->fi], st->pose[mi->fj], st->sb[mi->fj], rr, Jraw);
        __syncthreads();
        if (tid < 15) { double s = 0; for (int q = tid; q < 15; ++q) s += mi->sqrt_info[tid * 15 + q] * rr[q]; a.imu_w[450 + tid] = s; }
        for (int e = tid; e < 450; e += LM_THREADS) {
            const int i = e / 30, cc = e - i * 30; double s = 0;
            for (int q = i; q < 15; ++q) s += mi->sqrt_info[i * 15 + q] * Jraw[q * 30 + cc];
            a.imu_w[e] = s;
        }
        return;
    }
    MTS(0);
    __shared__ short s_slot[BE_MAX_PRIOR], s_comp[BE_MAX_PRIOR];      // dim -> (slot, component): read once, coalesced (they were two dependent global loads per dense-row entry)
    for (int i = tid; i < D; i += LM_THREADS) { s_slot[i] = (short)a.dim_slot[i]; s_comp[i] = (short)a.dim_comp[i]; }
    if (tid < 64) be_frame_geom_dev(st, a.nframes, fg, ric, tic, tid);
    __syncthreads();
    MTS(1);
    const BeLm L = a.lm[a.lm_sel ? a.lm_sel[l] : l];
    const int nf = L.count, anchor = L.anchor;
    if (tid < nf) {
        const BeFactor f = a.fac[L.first + tid];
        double* o = Jb + tid * 54;
        proj_factor<true, true>(f, fg[f.fi], fg[f.fj], ric[0], tic[0], ric[1], tic[1], st->inv_depth[f.lm], st->td, o, o + 2, o + 14, o + 50, o + 26, o + 38, o + 52);
        if (f.kind == 0) for (int k = 0; k < 12; ++k) o[38 + k] = 0.0;
        double rho0, sc;
        huber1(o[0] * o[0] + o[1] * o[1], rho0, sc);
        for (int k = 0; k < 54; ++k) o[k] *= sc;
        s_fj[tid] = f.fj; s_two[tid] = f.kind != 2;
    }
    __syncthreads();
    MTS(2);
    for (int e = tid; e < 2 * nf * D; e += LM_THREADS) {
        const int fr = e / D, i = e - fr * D, f = fr >> 1, r = fr & 1;
        const int slot = s_slot[i];
        double v = 0.0;
        int o0, o1;
        if (slot >= 0 && mg_joff(slot, s_comp[i], anchor, s_fj[f], s_two[f], o0, o1)) v = Jb[f * 54 + (r ? o1 : o0)];
        Jd[e] = v;
    }
    if (tid == 0) {
        double h = 0, g = 0;
        for (int f = 0; f < nf; ++f) { const double* o = Jb + f * 54; h += o[50] * o[50] + o[51] * o[51]; g += o[50] * o[0] + o[51] * o[1]; }
        s_hg[0] = h; s_hg[1] = g;
        a.lm_h[l] = h;
    }
    __syncthreads();
    MTS(3);
    // a landmark touches its anchor pose, the poses of the frames that observe it, the extrinsics and td: every other column of its
    // dense rows is identically zero, and so is every slab entry with such a row or column.  The slab is therefore written in two
    // passes: exact zeros for the structurally empty entries, and 2x2 register tiles over the lower triangle of the ACTIVE columns
    // (each loaded row value feeds two products; the mirror entry is the same sum of commuting products, bit for bit).
    __shared__ unsigned char s_act[BE_MAX_PRIOR];
    __shared__ short s_acols[BE_MAX_PRIOR + 2];
    __shared__ int s_na;
    for (int i = tid; i < D; i += LM_THREADS) {
        double w = 0, gp = 0; bool act = false;
        for (int fr = 0; fr < 2 * nf; ++fr) { const double j = Jd[fr * D + i]; const double* o = Jb + (fr >> 1) * 54; w += j * o[50 + (fr & 1)]; gp += j * o[fr & 1]; act = act || j != 0.0; }
        wv[i] = w; gpv[i] = gp; s_act[i] = act;
    }
    __syncthreads();
    {   // ordered compaction of the active columns: ballot + popcount per wave, wave offsets through LDS (D <= 192 < 256 threads)
        __shared__ int s_wcnt[LM_THREADS / 64];
        const bool actf = tid < D && s_act[tid];
        const unsigned long long m = __ballot(actf);
        const int lane = tid & 63, wv_id = tid >> 6;
        if (lane == 0) s_wcnt[wv_id] = __popcll(m);
        __syncthreads();
        int off = 0, tot = 0;
        for (int q = 0; q < LM_THREADS / 64; ++q) { if (q < wv_id) off += s_wcnt[q]; tot += s_wcnt[q]; }
        if (actf) s_acols[off + __popcll(m & ((1ull << lane) - 1ull))] = (short)tid;
        if (tid == 0) { s_acols[tot] = s_acols[tot + 1] = -1; s_na = tot; }
    }
    MTS(4);
    const double hinv = s_hg[0] > 1e-8 ? 1.0 / s_hg[0] : 0.0, g = s_hg[1];      // 1x1 pivot of the landmark's inverse depth, clamped like the reference's pseudo-inverse
    double* out = a.slabs + (size_t)l * (D * D + D);
    for (int i = tid >> 6; i < D; i += LM_THREADS / 64) {             // wave = row, lane = column: no integer division
        const bool ai = s_act[i];
        for (int j = tid & 63; j < D; j += 64) if (!(ai && s_act[j])) out[i * D + j] = 0.0;
    }
    for (int i = tid; i < D; i += LM_THREADS) out[D * D + i] = gpv[i] - wv[i] * g * hinv;
    __syncthreads();
    const int na = s_na, T = (na + 1) >> 1;
    for (int t = tid; t < T * (T + 1) / 2; t += LM_THREADS) {                 // lower-triangular tile index: t = ti (ti + 1) / 2 + tj
        int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (ti * (ti + 1) / 2 > t) --ti;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        const int tj = t - ti * (ti + 1) / 2;
        const int i0 = s_acols[2 * ti], i1 = s_acols[2 * ti + 1], j0 = s_acols[2 * tj], j1 = s_acols[2 * tj + 1];      // -1 past the end
        const int ri1 = i1 >= 0 ? i1 : i0, rj1 = j1 >= 0 ? j1 : j0;
        double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
        for (int fr = 0; fr < 2 * nf; ++fr) {
            const double* row = Jd + fr * D;
            const double a0 = row[i0], a1 = row[ri1], b0 = row[j0], b1 = row[rj1];
            s00 += a0 * b0; s01 += a0 * b1; s10 += a1 * b0; s11 += a1 * b1;
        }
        auto put = [&](int i, int j, double sv) {
            const double v = sv - wv[i] * wv[j] * hinv;
            out[i * D + j] = v;
            if (i != j) out[j * D + i] = v;
        };
        put(i0, j0, s00);
        if (j1 >= 0 && !(ti == tj)) put(i0, j1, s01);      // on a diagonal tile (i0, j1) is the mirror of (i1, j0)
        if (i1 >= 0) put(i1, j0, s10);
        if (i1 >= 0 && j1 >= 0) put(i1, j1, s11);
    }
    MTS(5);
}

#define MG_SUM_CHUNKS 4
// sums the per-landmark slabs: grid.y = MG_SUM_CHUNKS landmark ranges (parallelism: 37 x 4 workgroups instead of 37; 8 ranges doubled what the finish
// kernel has to read back through one CU: be_marg 98.8 -> 97.7 us; 2 ranges: 102 us), each
// summed in landmark order; the finish kernel adds the chunk sums in chunk order -> still one fixed, reproducible order
__global__ __launch_bounds__(256) void be_marg_sum_kernel(BeMargArgs a) {
    const int D = a.D, total = D * D + D;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int per = (a.nlm + MG_SUM_CHUNKS - 1) / MG_SUM_CHUNKS, l0 = blockIdx.y * per, l1 = min(a.nlm, l0 + per);
    double s = 0;
    // eight slabs requested per trip (a load + wait per landmark made the kernel a chain of dependent round trips: ~1 us each, 9 per chunk); added in landmark order
    for (int l = l0; l < l1; l += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = a.slabs[(size_t)(l + u < l1 ? l + u : l0) * total + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (l + u < l1) s += v[u];
    }
    a.sum[(size_t)blockIdx.y * total + e] = s;
}

__global__ __launch_bounds__(MG_THREADS) void be_marg_finish_kernel(BeMargArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = a.D, m = a.m, n = D - m, tid = threadIdx.x;
    double* A = sm;                                       // D x D, then b (D)
    double* bv = A + D * D;
    double* W2 = bv + D;                                  // n x n (also scratch for prior dx / IMU)
    double* yv = W2 + (n * n > 1024 ? n * n : 1024);      // n
    double* misc = yv + n;                                // 16
    double* imu_ws = misc + 16;                           // 960: raw / whitened Jacobian and residual of the IMU factor (its own region: prepared beside the phases below)
    const BeState* st = a.x;
    MTS(8);
    // The last wave takes no share of the assembly below: it fetches the whitened IMU factor (0,1) that the extra block of be_marg_lm has prepared (the raw
    // evaluation on ONE lane and the whitening used to be a 9 us phase of this kernel, behind the prior).
    const int MGW = MG_THREADS - 64;                      // worker threads of the assembly
    const bool imu_wave = tid >= MGW;
    double* Jw = imu_ws + 450; double* rr = imu_ws + 900;
    if (imu_wave && a.nimu > 0) {
        const int lane = tid - MGW;
        for (int i = lane; i < 465; i += 64) { const double v = a.imu_w[i]; if (i < 450) Jw[i] = v; else rr[15 + i - 450] = v; }
    }
    {   // 4 entries x 8 chunk sums per batch: 32 independent global loads in flight per thread (they were 8 at a time behind a loop-carried wait)
        const int total = D * D + D;
        for (int e0 = imu_wave ? total : tid; e0 < total; e0 += 4 * MGW) {
            double v[4] = { 0.0, 0.0, 0.0, 0.0 }, ld[4][MG_SUM_CHUNKS];
            if (a.nlm > 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * MGW;
#pragma unroll
                    for (int c = 0; c < MG_SUM_CHUNKS; ++c) ld[u][c] = e < total ? a.sum[(size_t)c * total + e] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < MG_SUM_CHUNKS; ++c) v[u] += ld[u][c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = e0 + u * MGW; if (e < total) A[e] = v[u]; }
        }
    }
    if (tid < 64) {                                     // smallest landmark pivot: strided over one wave, xor-tree
        double hmin = DBL_MAX;
        for (int l = tid; l < a.nlm; l += 64) hmin = fmin(hmin, a.lm_h[l]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hmin = fmin(hmin, __shfl_xor(hmin, o));
        if (tid == 0) { misc[0] = hmin; misc[1] = !(hmin > 1e-8) ? 1.0 : 0.0; }      // be_marg_lm dropped a landmark pivot <= 1e-8 (hinv := 0): reported like the dense pivots below
    }
    __syncthreads();
    MTS(9);
    // ---------------- previous prior: A += A_old (mapped), b += b_old + A_old dx ----------------
    if (a.prior->valid) {
        const int no = a.prior->n;
        double* dx = W2;
        __shared__ short s_pmap[BE_MAX_PRIOR];
        if (!imu_wave) { for (int i = tid; i < no; i += MGW) s_pmap[i] = (short)a.prior_map[i]; be_prior_dx_dev(a.prior, st, dx, tid, MGW); }
        __syncthreads();
        // b += b_old + A_old dx: A_old is symmetric, so thread = output entry reads DOWN its column (coalesced across threads), the column
        // split over the thread groups; partial sums meet in LDS in a fixed order
        {
            double* part = W2 + BE_MAX_PRIOR;                       // groups x no_pad partial sums (W2 holds >= 1024 doubles)
            const int no_pad = (no + 63) & ~63, groups = (1024 - BE_MAX_PRIOR) / no_pad > 0 ? min(MGW / no_pad, (1024 - BE_MAX_PRIOR) / no_pad) : 1;
            const int i = tid % no_pad, gI = tid / no_pad;
            const int seg = (no + groups - 1) / groups, j0 = gI * seg, j1 = min(no, j0 + seg);
            double sp = 0;
            if (gI < groups && i < no)
                for (int j = j0; j < j1; j += 16) {      // 16 rows requested per trip (one load + wait per row before: 14 dependent round trips for an 82-dimensional prior); added in row order
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = a.priorA[(size_t)(j + u < j1 ? j + u : j0) * no + i];
#pragma unroll
                    for (int u = 0; u < 16; ++u) if (j + u < j1) sp += v[u] * dx[j + u];
                }
            if (gI < groups && i < no) part[gI * no_pad + i] = sp;
            __syncthreads();
            if (tid < no) {
                double t = a.priorb[tid];
                for (int q = 0; q < groups; ++q) t += part[q * no_pad + tid];
                const int di = s_pmap[tid];
                if (di >= 0) bv[di] += t;
            }
        }
        for (int e0 = imu_wave ? no * no : tid; e0 < no * no; e0 += 4 * MGW) {          // 4 independent loads of A_old in flight per thread
            double pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = e0 + u * MGW; pv[u] = e < no * no ? a.priorA[e] : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * MGW;
                if (e >= no * no) continue;
                const int i = e / no, j = e - i * no;
                const int di = s_pmap[i], dj = s_pmap[j];
                if (di >= 0 && dj >= 0) A[di * D + dj] += pv[u];
            }
        }
        __syncthreads();
    }
    MTS(10);
    // ---------------- IMU factor (0,1) ----------------
    if (a.nimu > 0) {          // the products of the whitened Jacobian the last wave has prepared
        __syncthreads();
        for (int e = tid; e < 900 + 30; e += MG_THREADS) {
            if (e < 900) {
                const int r0 = e / 30, c0 = e - r0 * 30;
                const int di = a.imu_map[r0], dj = a.imu_map[c0];
                if (di < 0 || dj < 0) continue;
                double s = 0; for (int i = 0; i < 15; ++i) s += Jw[i * 30 + r0] * Jw[i * 30 + c0];
                A[di * D + dj] += s;
            } else {
                const int c0 = e - 900, di = a.imu_map[c0];
                if (di < 0) continue;
                double s = 0; for (int i = 0; i < 15; ++i) s += Jw[i * 30 + c0] * rr[15 + i];
                bv[di] += s;
            }
        }
        __syncthreads();
    }
    MTS(11);
    // ---------------- eliminate the dropped pose / speed-bias dims [0, m): right-looking LDL^T in panels of <= 4 columns on the LOWER
    // triangle of the whole system (A is bitwise symmetric by construction: commuting products summed in one order), with b as an extra
    // row.  (A) thread 0 factors the pivot block, (B) one thread per row below forms its panel entries and updates b, (C) all threads
    // apply the rank-<=4 update.  After the last panel the kept block IS the Schur complement A' and b the reduced b' — the separate
    // Cholesky, forward substitution and X^T X product (24 us) are gone ----------------
    {
        double* PLm = W2; double* PPm = W2 + 4 * D; double* dbm = W2 + 8 * D;      // W2 is free until A' is copied into it
        for (int k0 = 0; k0 < m; k0 += 4) {
            const int nb = min(4, m - k0);
            if (tid == 0) {
                double am[4][4], yk[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int cc = 0; cc <= r; ++cc) am[r][cc] = (r < nb) ? A[(k0 + r) * D + k0 + cc] : (r == cc ? 1.0 : 0.0);
                    yk[r] = r < nb ? bv[k0 + r] : 0.0;
                }
                // pivots <= 1e-8 are skipped (their column drops out: 1/d := 0), the LDL^T counterpart of the reference's eigen-clamped
                // pseudo-inverse of A_mm (marginalization_factor.cpp:286-289, eps = 1e-8): the result is finite for ANY input and equals the
                // plain Schur complement whenever A_mm is positive definite (every measured sequence: smallest pivot ~1e4).  misc[1] only reports it.
                const double d0 = am[0][0], i0 = d0 > 1e-8 ? mg_rcp(d0) : 0.0;
                const double l10 = am[1][0] * i0, l20 = am[2][0] * i0, l30 = am[3][0] * i0;
                const double d1 = am[1][1] - l10 * am[1][0], i1 = d1 > 1e-8 ? mg_rcp(d1) : 0.0;
                const double t21 = am[2][1] - l20 * am[1][0], t31 = am[3][1] - l30 * am[1][0];
                const double l21 = t21 * i1, l31 = t31 * i1;
                const double d2 = am[2][2] - l20 * am[2][0] - l21 * t21, i2 = d2 > 1e-8 ? mg_rcp(d2) : 0.0;
                const double t32 = am[3][2] - l30 * am[2][0] - l31 * t21;
                const double l32 = t32 * i2;
                const double d3 = am[3][3] - l30 * am[3][0] - l31 * t31 - l32 * t32, i3 = d3 > 1e-8 ? mg_rcp(d3) : 0.0;
                const double dd[4] = { d0, d1, d2, d3 };
                for (int r = 0; r < nb; ++r) { if (dd[r] < misc[0]) misc[0] = dd[r]; if (!(dd[r] > 1e-8)) misc[1] = 1.0; }
                const double y0 = yk[0], y1 = yk[1] - l10 * y0, y2 = yk[2] - l20 * y0 - l21 * y1, y3 = yk[3] - l30 * y0 - l31 * y1 - l32 * y2;
                dbm[0] = l10; dbm[1] = l20; dbm[2] = l30; dbm[3] = l21; dbm[4] = l31; dbm[5] = l32;
                dbm[6] = i0; dbm[7] = nb > 1 ? i1 : 0.0; dbm[8] = nb > 2 ? i2 : 0.0; dbm[9] = nb > 3 ? i3 : 0.0;      // padded columns contribute nothing
                dbm[10] = y0; dbm[11] = y1; dbm[12] = y2; dbm[13] = y3;
            }
            __syncthreads();
            const double l10 = dbm[0], l20 = dbm[1], l30 = dbm[2], l21 = dbm[3], l31 = dbm[4], l32 = dbm[5];
            const double i0 = dbm[6], i1 = dbm[7], i2 = dbm[8], i3 = dbm[9];
            const int r0 = k0 + nb;                                   // first row below the pivot block
            for (int i = r0 + tid; i < D; i += MG_THREADS) {
                const double* row = A + i * D + k0;
                const double p0 = row[0], p1 = nb > 1 ? row[1] - p0 * l10 : 0.0, p2 = nb > 2 ? row[2] - p0 * l20 - p1 * l21 : 0.0, p3 = nb > 3 ? row[3] - p0 * l30 - p1 * l31 - p2 * l32 : 0.0;
                const double x0 = p0 * i0, x1 = p1 * i1, x2 = p2 * i2, x3 = p3 * i3;
                PLm[i] = x0; PLm[D + i] = x1; PLm[2 * D + i] = x2; PLm[3 * D + i] = x3;
                PPm[i] = p0; PPm[D + i] = p1; PPm[2 * D + i] = p2; PPm[3 * D + i] = p3;
                bv[i] -= x0 * dbm[10] + x1 * dbm[11] + x2 * dbm[12] + x3 * dbm[13];
            }
            __syncthreads();
            for (int i = r0 + (tid >> 6); i < D; i += MG_THREADS / 64) {          // wave = row, lane = column: no integer division, the row's panel entries are wave-uniform
                const double x0 = PLm[i], x1 = PLm[D + i], x2 = PLm[2 * D + i], x3 = PLm[3 * D + i];
                for (int j = r0 + (tid & 63); j <= i; j += 64) A[i * D + j] -= x0 * PPm[j] + x1 * PPm[D + j] + x2 * PPm[2 * D + j] + x3 * PPm[3 * D + j];
            }
            __syncthreads();
        }
    }
    MTS(13);
    // A' = the kept block (mirrored to a full symmetric matrix), b' = the kept part of b
    for (int e = tid; e < n * n + n; e += MG_THREADS) {
        if (e >= n * n) { const int i = e - n * n; yv[i] = bv[m + i]; continue; }
        const int i = e / n, j = e - i * n;
        W2[e] = i >= j ? A[(m + i) * D + m + j] : A[(m + j) * D + m + i];
    }
    __syncthreads();
    MTS(14);
    for (int e = tid; e < n * n; e += MG_THREADS) a.outA[e] = W2[e];
    for (int i = tid; i < n; i += MG_THREADS) a.outb[i] = yv[i];
    __syncthreads();
    MTS(15);
    // ---------------- c0 = b'^T A'^+ b': LDL^T on the lower triangle, pivots <= 1e-8 skipped, one barrier per step ----------------
    // 4-column panels (the matrix stays in LDS): (A) thread 0 factors the 4x4 pivot block, skipping pivots <= 1e-8 (their columns drop out),
    // forward-substitutes its part of y and accumulates c0; (B) one thread per row below forms the panel L (and L D) and updates its y;
    // (C) all threads apply the rank-4 trailing update.  Three barriers per FOUR pivots instead of one per pivot with a rank-1 update.
    double c0 = 0.0; int rank = 0;
    double* PLc = A; double* PPc = A + 4 * n; double* dblk = A + 8 * n;      // the D x D buffer is free by now
    for (int k0 = 0; k0 < n; k0 += 4) {
        if (tid == 0) {
            double am[4][4], yk[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool real = k0 + r < n;
#pragma unroll
                for (int cc = 0; cc <= r; ++cc) am[r][cc] = (real && k0 + cc < n) ? W2[(k0 + r) * n + k0 + cc] : (r == cc ? 1.0 : 0.0);
                yk[r] = real ? yv[k0 + r] : 0.0;
            }
            const double d0 = am[0][0], i0 = d0 > 1e-8 ? mg_rcp(d0) : 0.0;
            const double l10 = am[1][0] * i0, l20 = am[2][0] * i0, l30 = am[3][0] * i0;
            const double d1 = am[1][1] - l10 * am[1][0], i1 = d1 > 1e-8 ? mg_rcp(d1) : 0.0;
            const double t21 = am[2][1] - l20 * am[1][0], t31 = am[3][1] - l30 * am[1][0];
            const double l21 = t21 * i1, l31 = t31 * i1;
            const double d2 = am[2][2] - l20 * am[2][0] - l21 * t21, i2 = d2 > 1e-8 ? mg_rcp(d2) : 0.0;
            const double t32 = am[3][2] - l30 * am[2][0] - l31 * t21;
            const double l32 = t32 * i2;
            const double d3 = am[3][3] - l30 * am[3][0] - l31 * t31 - l32 * t32, i3 = d3 > 1e-8 ? mg_rcp(d3) : 0.0;
            const double y0 = yk[0], y1 = yk[1] - l10 * y0, y2 = yk[2] - l20 * y0 - l21 * y1, y3 = yk[3] - l30 * y0 - l31 * y1 - l32 * y2;
            c0 += y0 * y0 * i0;
            if (k0 + 1 < n) c0 += y1 * y1 * i1;
            if (k0 + 2 < n) c0 += y2 * y2 * i2;
            if (k0 + 3 < n) c0 += y3 * y3 * i3;
            rank += (i0 != 0.0) + (k0 + 1 < n && i1 != 0.0) + (k0 + 2 < n && i2 != 0.0) + (k0 + 3 < n && i3 != 0.0);
            dblk[0] = l10; dblk[1] = l20; dblk[2] = l30; dblk[3] = l21; dblk[4] = l31; dblk[5] = l32;
            dblk[6] = i0; dblk[7] = i1; dblk[8] = i2; dblk[9] = i3; dblk[10] = y0; dblk[11] = y1; dblk[12] = y2; dblk[13] = y3;
        }
        __syncthreads();
        if (k0 + 4 >= n) break;                                   // uniform: nothing below the last pivot block
        const double l10 = dblk[0], l20 = dblk[1], l30 = dblk[2], l21 = dblk[3], l31 = dblk[4], l32 = dblk[5];
        const double i0 = dblk[6], i1 = dblk[7], i2 = dblk[8], i3 = dblk[9];
        for (int i = k0 + 4 + tid; i < n; i += MG_THREADS) {
            const double* row = W2 + i * n + k0;
            const double p0 = row[0], p1 = row[1] - p0 * l10, p2 = row[2] - p0 * l20 - p1 * l21, p3 = row[3] - p0 * l30 - p1 * l31 - p2 * l32;
            const double x0 = p0 * i0, x1 = p1 * i1, x2 = p2 * i2, x3 = p3 * i3;
            PLc[i] = x0; PLc[n + i] = x1; PLc[2 * n + i] = x2; PLc[3 * n + i] = x3;
            PPc[i] = p0; PPc[n + i] = p1; PPc[2 * n + i] = p2; PPc[3 * n + i] = p3;
            yv[i] -= x0 * dblk[10] + x1 * dblk[11] + x2 * dblk[12] + x3 * dblk[13];
        }
        __syncthreads();
        for (int i = k0 + 4 + (tid >> 6); i < n; i += MG_THREADS / 64) {          // wave = row, lane = column
            const double x0 = PLc[i], x1 = PLc[n + i], x2 = PLc[2 * n + i], x3 = PLc[3 * n + i];
            for (int j = k0 + 4 + (tid & 63); j <= i; j += 64) W2[i * n + j] -= x0 * PPc[j] + x1 * PPc[n + j] + x2 * PPc[2 * n + j] + x3 * PPc[3 * n + j];
        }
        __syncthreads();
    }
    if (tid == 0) { misc[2] = c0; misc[3] = (double)rank; }
    __syncthreads();
    c0 = misc[2]; rank = (int)misc[3];
    MTS(16);
    if (tid == 0) { a.out_scalars[0] = c0; a.out_scalars[1] = misc[0]; a.out_scalars[2] = misc[1]; a.out_scalars[3] = (double)rank; if (a.c0_out) a.c0_out[0] = c0; }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void be_gauge_body(const BeGaugeArgs& a) {
    __shared__ m33 rot;
    __shared__ d3 p0;
    const int i = threadIdx.x;
    const BeState* st = a.x;
    BeState* o = a.out;
    if (i == 0) {
        p0 = P3(st->pose[0]);
        if (a.use_imu) {
            m33 R0; for (int k = 0; k < 9; ++k) R0.m[k] = a.R0[k];
            const m33 R00 = qR(Q4(st->pose[0]));
            const d3 y00 = r2ypr(R00);
            m33 rd = ypr2r(mk3(a.ypr0[0] - y00.x, 0, 0));
            if (fabs(fabs(a.ypr0[1]) - 90) < 1.0 || fabs(fabs(y00.y) - 90) < 1.0) rd = mul(R0, tr(R00));      // near the Euler singularity (estimator.cpp:1121-1128)
            rot = rd;
        }
    }
    __syncthreads();
    for (int l = i; l < a.nlm; l += 256) o->inv_depth[l] = st->inv_depth[l];
    if (i < 14) o->ex[i / 7][i % 7] = st->ex[i / 7][i % 7];
    if (i == 14) o->td = st->td;
    if (i < BE_NF) {
        if (i >= a.nframes) { for (int k = 0; k < 7; ++k) o->pose[i][k] = st->pose[i][k]; for (int k = 0; k < 9; ++k) o->sb[i][k] = st->sb[i][k]; }
        else {
            const quat qn = qnormalized(Q4(st->pose[i]));
            for (int k = 0; k < 9; ++k) o->sb[i][k] = st->sb[i][k];
            if (a.use_imu) {
                const m33 R = mul(rot, qR(qn));
                const d3 P = mul(rot, P3(st->pose[i]) - p0) + mk3(a.P0[0], a.P0[1], a.P0[2]);
                const d3 V = mul(rot, mk3(st->sb[i][0], st->sb[i][1], st->sb[i][2]));
                const quat q = qfromR(R);
                o->pose[i][0] = P.x; o->pose[i][1] = P.y; o->pose[i][2] = P.z; o->pose[i][3] = q.x; o->pose[i][4] = q.y; o->pose[i][5] = q.z; o->pose[i][6] = q.w;
                o->sb[i][0] = V.x; o->sb[i][1] = V.y; o->sb[i][2] = V.z;
            } else {
                const quat q = qfromR(qR(qn));
                o->pose[i][0] = st->pose[i][0]; o->pose[i][1] = st->pose[i][1]; o->pose[i][2] = st->pose[i][2];
                o->pose[i][3] = q.x; o->pose[i][4] = q.y; o->pose[i][5] = q.z; o->pose[i][6] = q.w;
            }
        }
    }
    if (!a.h_out) return;
    __syncthreads();          // the workgroup's stores to `o` are visible to all of its threads
    {
        const double* src = reinterpret_cast<const double*>(o);
        double* dst = reinterpret_cast<double*>(a.h_out);
        for (int k = i; k < a.state_doubles; k += 256) dst[k] = src[k];
        const int nc = (int)(sizeof(BeCtl) / sizeof(double));
        if (i < nc) reinterpret_cast<double*>(a.h_ctl)[i] = reinterpret_cast<const double*>(a.ctl)[i];
        if (a.h_raw_pose) for (int k = i; k < 7 * BE_NF; k += 256) a.h_raw_pose[k] = reinterpret_cast<const double*>(st->pose)[k];
    }
}
__global__ __launch_bounds__(256) void be_gauge_kernel(BeGaugeArgs a) { be_gauge_body(a); }
// accept / reject decision of the last slot + gauge fix + download in ONE launch (the decision's copy x <- candidate is ordered before the gauge fix, which
// reads x and overwrites the candidate buffer, by the workgroup barrier): one launch less on the path the host's wake-up and the marginalization both wait for
__global__ __launch_bounds__(256) void be_accept_gauge_kernel(BeSolveArgs sa, BeGaugeArgs ga) {
    be_accept_body(sa);
    __syncthreads();
    be_gauge_body(ga);
}
void be_launch_gauge(const BeGaugeArgs& a, hipStream_t s) { hipLaunchKernelGGL(be_gauge_kernel, dim3(1), dim3(256), 0, s, a); }
void be_launch_accept_gauge(const BeSolveArgs& sa, const BeGaugeArgs& ga, hipStream_t s) { hipLaunchKernelGGL(be_accept_gauge_kernel, dim3(1), dim3(256), 0, s, sa, ga); }

static size_t finish_smem(int D, int n) { return ((size_t)D * D + D + std::max((size_t)n * n, (size_t)1024) + n + 16 + 960) * sizeof(double); }
static size_t lm_smem(int D) { return ((size_t)BE_MAX_OBS_FACTORS * 54 + 2 * (size_t)BE_MAX_OBS_FACTORS * D + 2 * D) * sizeof(double); }

int be_launch_marg(const BeMargArgs& a, hipStream_t s) {
    static DevOnce once;
    if (once.run([] {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_marg_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return 1;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_marg_lm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return 1;
            return 0; })) return -1;
    const size_t bytes = finish_smem(a.D, a.D - a.m);
    if (bytes > 156 * 1024 || lm_smem(a.D) > 64 * 1024) return -2;
    if (a.nlm > 0 || a.nimu > 0) hipLaunchKernelGGL(be_marg_lm_kernel, dim3(a.nlm + (a.nimu > 0 ? 1 : 0)), dim3(LM_THREADS), lm_smem(a.D), s, a);
    if (a.nlm > 0) {
        const int total = a.D * a.D + a.D;
        hipLaunchKernelGGL(be_marg_sum_kernel, dim3((total + 255) / 256, MG_SUM_CHUNKS), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(be_marg_finish_kernel, dim3(1), dim3(MG_THREADS), bytes, s, a);
    return 0;
}
