// be_eval.hip — residual/Jacobian evaluation of one sliding-window BA problem on gfx950: replaces the
// per-residual-block ceres::CostFunction::Evaluate() calls of the reference's ceres::Solve
// (estimator/estimator.cpp:109-214, 314) with one launch per evaluation.
//
// Grid: one 64-lane workgroup per landmark, then one per IMU factor, then one for the prior.
//   landmark block : lane = residual block.  Each lane evaluates its ProjectionFactor (fp64 residual, 2x6
//                    Jacobians wrt pose_i / pose_j, 2x1 wrt inverse depth), applies the Huber(1.0)
//                    corrector and parks the 28 numbers in LDS; then lane = output entry: the landmark's
//                    Hessian pieces (h, g, w[a], gp[a], Ddiag[a], Danch[a]) are summed over its residual
//                    blocks in a fixed order and written as one 7.4 KB packet.  Nothing is accumulated
//                    with atomics, so the reduced system is bitwise reproducible.
//   IMU block      : lane 0 evaluates the raw 15-residual / 15x30 Jacobian, all lanes whiten with the cached
//                    sqrt-information and form the factor's 30x30 Hessian block and gradient.
//   prior block    : g = b' + A' dx and cost = c0/2 + b'.dx + dx.A'dx/2 (information form of the
//                    MarginalizationFactor, factor/marginalization_factor.cpp:350-396).
// COST-only variant (candidate points of the trust-region loop): residuals only.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "be_kernels.h"
#include "wave_dpp.h"

using namespace be;

#ifdef BE_EVAL_TS
__device__ long long be_ev_ts[64];
#define ETS(k) do { if (FULL && threadIdx.x == 0) be_ev_ts[k] = wall_clock64(); } while (0)
extern "C" int dv_debug_ev_ts(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(be_ev_ts), sizeof(long long) * 64) == hipSuccess ? 0 : -1; }
#else
#define ETS(k) do {} while (0)
#endif
// Workgroups are dealt to the 8 XCDs round-robin, and a packet row holds 16 consecutive landmarks per 128-byte line: with block b evaluating landmark b the
// 16 writers of one line sit on 8 different L2s and every one of them writes its 8 bytes back as a 32-byte sector (PMC: 7.9 MB written for 1.95 MB of
// packets).  The landmark blocks are therefore renumbered so that each XCD evaluates a CONTIGUOUS range of landmarks and its L2 merges the lines.
__device__ __forceinline__ int be_eval_block_of(int bx, int nlm) {
    if (bx >= nlm) return bx;                                  // IMU / prior blocks keep their numbers
    const int x = bx & 7, k = bx >> 3;
    int start = 0;                                             // landmarks of the residue classes before x: class y holds (nlm - y + 7) / 8 blocks
    for (int y = 0; y < x; ++y) start += nlm > y ? (nlm - y + 7) >> 3 : 0;
    return start + k;                                          // bijection [0, nlm) -> [0, nlm): class x (the blocks one XCD receives), in order, onto a contiguous range
}
#define EV_THREADS 256
// ARGS: BeEvalArgs (kernel arguments by value: invariant scalar loads) or the same struct in the constant address space (batched form: the argument table in
// HBM read through the scalar / invariant path instead of re-loading every field after each store)
#define DV_CONSTANT __attribute__((address_space(4)))
// LPB = landmarks per workgroup: 1 (the single-window launch: 256 threads sum one landmark's packet, lowest latency) or 4 (the batched launch: one WAVE per landmark
// — only L.count <= 22 lanes of a workgroup do the factor arithmetic and the kernel's 214 VGPRs allow two workgroups per CU, so with one landmark per workgroup a CU
// worked on 2 landmarks at a time; with four it works on 8 and the frame geometry is formed once for all four).  Same arithmetic, same order per landmark.
// PART: 0 every block kind; 1 the landmark blocks alone, 2 the IMU / prior blocks alone (the batched path from 12 windows per launch on, be_launch_eval_batch: the IMU factor's
// single-thread evaluation is what holds 222 VGPRs — the landmark branch alone compiles to 176 — so a launch of its own for the handful of IMU / prior blocks lets the landmark
// launch keep a third workgroup per CU)
template <bool FULL, int LPB, class ARGS, int PART = 0>
__device__ __forceinline__ void be_eval_body(const ARGS& a, int mode, int bg) {
    const BeCtl c = *a.ctl;
    if (c.done) return;
    if (mode == BE_EVAL_X ? !c.need_eval : !c.pending) return;
    const BeState* st = mode == BE_EVAL_X ? a.x : a.cand;
    const int set = mode == BE_EVAL_X ? c.cur : (c.cur ^ 1);
    double* const o_packets = a.packets[set]; double* const o_imu = a.imu_out[set]; double* const o_prior = a.prior_out[set];
    const bool want_cost = mode != BE_EVAL_X;
    const int nlm = a.dims.nlm, nimu = a.dims.nimu;
    constexpr int TPL = EV_THREADS / LPB;                      // threads per landmark
    const int ngrp = (nlm + LPB - 1) / LPB;                    // landmark workgroups of this window
    const int wsel = LPB == 1 ? 0 : (int)threadIdx.x / TPL;    // which of the workgroup's landmarks this thread belongs to
    const int lane = LPB == 1 ? (int)threadIdx.x : (int)threadIdx.x % TPL;
    const int b = bg < ngrp ? bg * LPB + wsel : nlm + (bg - ngrp);      // landmark index, or nlm + k for the IMU / prior blocks
    __shared__ FrameGeom fg[BE_NF];
    __shared__ m33 ric[2];
    __shared__ d3 tic[2];
    __shared__ double Jb_all[LPB][BE_MAX_OBS_FACTORS][28];
    __shared__ int s_fj_all[LPB][BE_MAX_OBS_FACTORS], s_two_all[LPB][BE_MAX_OBS_FACTORS];
    __shared__ int s_flist_all[LPB][BE_NF][2];      // per observing frame: the (<= 2: left cam, right cam) two-frame factors whose frame j it is
    __shared__ double s_cost_all[LPB][BE_MAX_OBS_FACTORS];
    __shared__ double s_imu[PART == 1 ? 1 : 450 + 450 + 32];
    if (b == 0) ETS(0);
    if (b == nlm) ETS(8);
    if (b == nlm + nimu) ETS(16);
    if (bg < ngrp) {
      if constexpr (PART != 2) {
        // ------------------------------- landmark(s) -------------------------------
        double (*Jb)[28] = Jb_all[wsel]; int* s_fj = s_fj_all[wsel]; int* s_two = s_two_all[wsel]; int (*s_flist)[2] = s_flist_all[wsel]; double* s_cost = s_cost_all[wsel];
        const bool valid = b < nlm && !(b < a.lm_lo || b >= a.lm_hi);      // (sharded window: another rank's landmark)
        if (LPB == 1 && !valid) return;
        BeLm L{}; if (valid) L = a.lm[b];
        be_frame_geom_dev(st, a.dims.nframes, fg, ric, tic, (int)threadIdx.x);
        if (LPB == 1) { if (lane >= 64 && lane < 64 + 2 * BE_NF) s_flist[(lane - 64) >> 1][(lane - 64) & 1] = -1; }
        else if (lane >= TPL - 2 * BE_NF) s_flist[(lane - (TPL - 2 * BE_NF)) >> 1][(lane - (TPL - 2 * BE_NF)) & 1] = -1;      // (the last 22 lanes of the landmark's group: TPL >= 32)
        __syncthreads();
        if (b == 0) ETS(1);
        const double lambda = valid ? st->inv_depth[b] : 1.0;
        if (valid && lane < L.count) {
            const BeFactor f = a.fac[L.first + lane];
            double r[2], Ji[12], Jj[12], Jl[2];
            proj_factor<FULL, false>(f, fg[f.fi], fg[f.fj], ric[0], tic[0], ric[1], tic[1], lambda, st->td, r, Ji, Jj, Jl, nullptr, nullptr, nullptr);
            double rho0, sc;
            huber1(r[0] * r[0] + r[1] * r[1], rho0, sc);
            s_cost[lane] = 0.5 * rho0;
            if (FULL) {
                double* o = Jb[lane];
                o[0] = r[0] * sc; o[1] = r[1] * sc;
#pragma unroll
                for (int k = 0; k < 12; ++k) { o[2 + k] = Ji[k] * sc; o[14 + k] = Jj[k] * sc; }
                o[26] = Jl[0] * sc; o[27] = Jl[1] * sc;
                s_fj[lane] = f.fj; s_two[lane] = f.kind != 2;
                if (f.kind != 2) s_flist[f.fj][f.kind == 1 ? 1 : 0] = lane;
            }
        }
        __syncthreads();
        if (!valid) return;                                        // (no workgroup barrier below)
        if (b == 0) ETS(2);
        double cost = 0;
        for (int f = 0; f < L.count; ++f) cost += s_cost[f];      // fixed order
        if (want_cost && lane == 0) a.cand_cost[b] = cost;
        if (!FULL) return;
        const int anchor = L.anchor, nf = L.count;
        // observation mask: pose f is touched by this landmark iff f is its anchor or a frame with a two-frame factor.  Only those frames' GP / DD / DA slots are
        // written (the rest were 8-byte stores of zeros: 40 % of the packet's 7.4 KB and of the 4x write amplification PMC showed) — be_reduce reads them under
        // the same mask.
        // every wave forms the mask by itself: lane f (of the landmark's group inside the wave) votes for frame f.  TPL = 32: a wave holds TWO landmarks, each reads its half
        constexpr int GL = TPL < 64 ? TPL : 64;                       // lanes of one landmark inside a wave
        const int wl = lane & (GL - 1);
        const unsigned long long votes = __ballot(wl < a.dims.nframes && (wl == anchor || s_flist[wl < BE_NF ? wl : 0][0] >= 0 || s_flist[wl < BE_NF ? wl : 0][1] >= 0));
        const int obs = (int)((votes >> (((int)threadIdx.x & 63) & ~(GL - 1))) & ((1ull << BE_NF) - 1));
        if (lane == 0 && mode != BE_EVAL_CAND_COST) a.lm_obs[b] = obs;
        for (int e = lane; e < BE_PK_SIZE; e += TPL) {
            if (e >= BE_PK_GP && e < BE_PK_DA + BE_NF * 36) {
                const int fa = e < BE_PK_DD ? (e - BE_PK_GP) / 6 : (e < BE_PK_DA ? (e - BE_PK_DD) / 36 : (e - BE_PK_DA) / 36);
                if (!((obs >> fa) & 1)) continue;
            }
            double v = 0.0;
            if (e == BE_PK_H) { for (int f = 0; f < nf; ++f) v += Jb[f][26] * Jb[f][26] + Jb[f][27] * Jb[f][27]; }
            else if (e == BE_PK_G) { for (int f = 0; f < nf; ++f) v += Jb[f][26] * Jb[f][0] + Jb[f][27] * Jb[f][1]; }
            else if (e == BE_PK_COST) v = cost;
            else if (e < BE_PK_DD) {
                const bool is_w = e < BE_PK_GP;
                const int q = e - (is_w ? BE_PK_W : BE_PK_GP), fa = q / 6, rr = q - fa * 6;
                if (fa == anchor) {
                    for (int f = 0; f < nf; ++f) {
                        if (!s_two[f]) continue;
                        const double m0 = is_w ? Jb[f][26] : Jb[f][0], m1 = is_w ? Jb[f][27] : Jb[f][1];
                        v += Jb[f][2 + rr] * m0 + Jb[f][8 + rr] * m1;
                    }
                } else {
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        const int f = s_flist[fa][sl];
                        if (f < 0) continue;
                        const double m0 = is_w ? Jb[f][26] : Jb[f][0], m1 = is_w ? Jb[f][27] : Jb[f][1];
                        v += Jb[f][14 + rr] * m0 + Jb[f][20 + rr] * m1;
                    }
                }
            } else if (e < BE_PK_DA) {
                const int q = e - BE_PK_DD, fa = q / 36, rc = q - fa * 36, rr = rc / 6, cc = rc - rr * 6;
                if (fa == anchor) {
                    for (int f = 0; f < nf; ++f) {
                        if (!s_two[f]) continue;
                        v += Jb[f][2 + rr] * Jb[f][2 + cc] + Jb[f][8 + rr] * Jb[f][8 + cc];
                    }
                } else {
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        const int f = s_flist[fa][sl];
                        if (f < 0) continue;
                        v += Jb[f][14 + rr] * Jb[f][14 + cc] + Jb[f][20 + rr] * Jb[f][20 + cc];
                    }
                }
            } else if (e < BE_PK_DA + BE_NF * 36) {
                const int q = e - BE_PK_DA, fa = q / 36, rc = q - fa * 36, rr = rc / 6, cc = rc - rr * 6;
                if (fa != anchor) {
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        const int f = s_flist[fa][sl];
                        if (f < 0) continue;
                        v += Jb[f][2 + rr] * Jb[f][14 + cc] + Jb[f][8 + rr] * Jb[f][20 + cc];
                    }
                }
            }
            BE_PK(o_packets, e, b) = v;
        }
        if (b == 0) ETS(3);
      }
    } else if (b < nlm + nimu) {
      if constexpr (PART != 1) {
        // ------------------------------- IMU factor -------------------------------
        const int lane = threadIdx.x;                         // (all EV_THREADS threads of the workgroup)
        const int k = b - nlm;
        // the factor record (pre-integrated deltas, bias Jacobians, sqrt-information: 2.4 KB) is staged in LDS by all threads:
        // lane 0's raw evaluation and the whitening loops then read LDS instead of chasing dependent global loads
        __shared__ BeImu s_m;
        {
            const double* src = reinterpret_cast<const double*>(&a.imu[k]); double* dst = reinterpret_cast<double*>(&s_m);
            for (int i = lane; i < (int)(sizeof(BeImu) / 8); i += EV_THREADS) dst[i] = src[i];
        }
        const BeImu* m = &s_m;
        double* Jraw = s_imu; double* Jw = s_imu + 450; double* rr = s_imu + 900;      // rr[0..14] raw, rr[15..29] whitened
        for (int i = lane; i < 450; i += EV_THREADS) Jraw[i] = 0.0;
        __syncthreads();
        if (b == nlm) ETS(9);
        if (lane == 0) imu_raw<FULL>(*m, a.g_norm, st->pose[m->fi], st->sb[m->fi], st->pose[m->fj], st->sb[m->fj], rr, Jraw);
        __syncthreads();
        if (b == nlm) ETS(10);
        if (lane < 15) { double s = 0; for (int q = lane; q < 15; ++q) s += m->sqrt_info[lane * 15 + q] * rr[q]; rr[15 + lane] = s; }
        if (FULL)
            for (int e = lane; e < 450; e += EV_THREADS) {
                const int i = e / 30, cc = e - i * 30; double s = 0;
                for (int q = i; q < 15; ++q) s += m->sqrt_info[i * 15 + q] * Jraw[q * 30 + cc];
                Jw[e] = s;
            }
        __syncthreads();
        if (b == nlm) ETS(11);
        double cost = 0;
        for (int i = 0; i < 15; ++i) cost += rr[15 + i] * rr[15 + i];
        cost *= 0.5;
        if (want_cost && lane == 0) a.cand_cost[b] = cost;
        if (!FULL) return;
        double* o = o_imu + (size_t)k * IMU_OUT_STRIDE;
        if (lane == 0) o[0] = cost;
        if (lane < 30) { double s = 0; for (int i = 0; i < 15; ++i) s += Jw[i * 30 + lane] * rr[15 + i]; o[1 + lane] = s; }
        for (int e = lane; e < 900; e += EV_THREADS) {
            const int r0 = e / 30, c0 = e - r0 * 30; double s = 0;
            for (int i = 0; i < 15; ++i) s += Jw[i * 30 + r0] * Jw[i * 30 + c0];
            o[31 + e] = s;
        }
        if (b == nlm) ETS(12);
      }
    } else {
      if constexpr (PART != 1) {
        // ------------------------------- prior -------------------------------
        const int lane = threadIdx.x;
        const BePriorHdr* p = a.prior;
        if (!p->valid) { if (lane == 0) { if (FULL) o_prior[0] = 0.0; if (want_cost) a.cand_cost[b] = 0.0; } return; }
        __shared__ double dx[BE_MAX_PRIOR], Adx[BE_MAX_PRIOR];
        const int n = p->n;
        be_prior_dx_dev(p, st, dx, lane, EV_THREADS);
        __syncthreads();
        ETS(17);
        {   // A' dx with A' symmetric: thread = output entry reading DOWN its column (coalesced across threads), the column split
            // over EV_THREADS / n_pad thread groups whose partial sums are added in a fixed order
            __shared__ double part[EV_THREADS];
            const int n_pad = (n + 63) & ~63, groups = n_pad <= EV_THREADS ? EV_THREADS / n_pad : 1;
            const int i = lane % n_pad, gI = lane / n_pad;
            const int seg = (n + groups - 1) / groups, j0 = gI * seg, j1 = min(n, j0 + seg);
            double s = 0;
            if (gI < groups && i < n) {
                // 8 rows per batch: the loads are address-independent (A' was written by the previous frame's marginalization on other XCDs,
                // every one is a ~0.5 us round trip), the sum keeps its ascending-j order
                int j = j0;
                for (; j + 8 <= j1; j += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = a.priorA[(size_t)(j + u) * n + i];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v[u] * dx[j + u];
                }
                for (; j < j1; ++j) s += a.priorA[(size_t)j * n + i] * dx[j];
            }
            part[lane] = s;
            __syncthreads();
            if (n_pad <= EV_THREADS) { if (lane < n) { double t = part[lane]; for (int q = 1; q < groups; ++q) t += part[q * n_pad + lane]; Adx[lane] = t; } }
            else for (int r = lane; r < n; r += EV_THREADS) { double t = 0; for (int j = 0; j < n; ++j) t += a.priorA[(size_t)j * n + r] * dx[j]; Adx[r] = t; }
        }
        __syncthreads();
        ETS(18);
        if (lane < 64) {          // b'.dx and dx.A'dx: strided over one wave + xor tree (was a serial loop of n dependent global loads on lane 0)
            double bd = 0, dAd = 0;
            for (int i = lane; i < n; i += 64) { bd += a.priorb[i] * dx[i]; dAd += dx[i] * Adx[i]; }
            bd = wave_sum_f64(bd); dAd = wave_sum_f64(dAd);          // (lane here is the thread index: the first wave, all 64 lanes active)
        if (lane == 0) {
            const double cost = 0.5 * a.prior_c0[0] + bd + 0.5 * dAd;
            if (FULL) o_prior[0] = cost;
            if (want_cost) a.cand_cost[b] = cost;
        }
        }
        if (FULL) for (int i = lane; i < n; i += EV_THREADS) o_prior[1 + i] = a.priorb[i] + Adx[i];
        ETS(19);
      }
    }
}

template <bool FULL>
__global__ __launch_bounds__(EV_THREADS) void be_eval_kernel(BeEvalArgs a, int mode) { be_eval_body<FULL, 1, BeEvalArgs>(a, mode, be_eval_block_of(blockIdx.x, a.dims.nlm)); }
// batched form (several independent windows in one launch: blockIdx.y = window, argument table in HBM): LPB landmarks per workgroup.
// LPB 4 (rounds 4-5): one wave per landmark — at most 22 of its 64 lanes evaluate a factor, and 16 windows are 1232 workgroups of ~40 us against 512 - 768 resident
// (the kernel holds 168 - 222 VGPRs: two or three workgroups per CU): 1.6 - 2.4 occupancy rounds, which is what its 73 - 86 us per 16 windows were (round 5's counters:
// not bandwidth).  LPB 8 (round 6, default): 32 threads per landmark, TWO landmarks per wave — the factor phase runs on twice the lanes, the launch is HALF the
// workgroups (616 for 16 windows: one round) and every packet entry is still summed by ONE thread over the same factors in the same order: the same bits.
// DVINS_EVAL_LPB=4 selects the old form (A/B).
// (Round 5: capping the registers with amdgpu_waves_per_eu(3 / 4) on the whole evaluation buys the occupancy with 109 / 203 spilled VGPRs and loses:
//  16 sequences 5969 -> 5880 / 5852 frames/s, 64 sequences 8285 -> 8121 / 7772; profiles/r05_experiments/eval_waves_ab.txt.)
static int ev_lpb() { static const int v = [] { const char* e = std::getenv("DVINS_EVAL_LPB"); return (e && std::atoi(e) == 4) ? 4 : 8; }(); return v; }
template <bool FULL, int LPB>
__global__ __launch_bounds__(EV_THREADS) void be_eval_batch_kernel(const BeEvalArgs* __restrict__ tab, int mode) {
    const DV_CONSTANT BeEvalArgs& a = *reinterpret_cast<const DV_CONSTANT BeEvalArgs*>(reinterpret_cast<uintptr_t>(tab + blockIdx.y));
    const int ngrp = (a.dims.nlm + LPB - 1) / LPB;
    if ((int)blockIdx.x >= ngrp + a.dims.nimu + 1) return;
    be_eval_body<FULL, LPB, DV_CONSTANT BeEvalArgs>(a, mode, be_eval_block_of(blockIdx.x, ngrp));      // (grid x is a multiple of 8: the XCD of a block is blockIdx.x % 8 as in the single-window launch)
}
// the two launches of the split form: grid x = landmark workgroups resp. IMU + prior blocks of the largest window
template <int LPB>
__global__ __launch_bounds__(EV_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void be_eval_batch_lm_kernel(const BeEvalArgs* __restrict__ tab, int mode) {
    const DV_CONSTANT BeEvalArgs& a = *reinterpret_cast<const DV_CONSTANT BeEvalArgs*>(reinterpret_cast<uintptr_t>(tab + blockIdx.y));
    const int ngrp = (a.dims.nlm + LPB - 1) / LPB;
    if ((int)blockIdx.x >= ngrp) return;
    be_eval_body<true, LPB, DV_CONSTANT BeEvalArgs, 1>(a, mode, be_eval_block_of(blockIdx.x, ngrp));
}
template <int LPB>
__global__ __launch_bounds__(EV_THREADS) void be_eval_batch_rest_kernel(const BeEvalArgs* __restrict__ tab, int mode) {
    const DV_CONSTANT BeEvalArgs& a = *reinterpret_cast<const DV_CONSTANT BeEvalArgs*>(reinterpret_cast<uintptr_t>(tab + blockIdx.y));
    const int ngrp = (a.dims.nlm + LPB - 1) / LPB;
    if ((int)blockIdx.x >= a.dims.nimu + 1) return;
    be_eval_body<true, LPB, DV_CONSTANT BeEvalArgs, 2>(a, mode, ngrp + (int)blockIdx.x);
}
int be_eval_batch_blocks(int nlm, int nimu) { const int lpb = ev_lpb(); return (nlm + lpb - 1) / lpb + nimu + 1; }
template <int LPB>
static void launch_eval_batch(const BeEvalArgs* tab_dev, int n_win, int max_grid, int mode, bool split, hipStream_t s) {
    if (split && mode != BE_EVAL_CAND_COST) {          // (max_grid bounds both parts)
        hipLaunchKernelGGL(be_eval_batch_rest_kernel<LPB>, dim3(BE_WIN + 1, n_win), dim3(EV_THREADS), 0, s, tab_dev, mode);
        hipLaunchKernelGGL(be_eval_batch_lm_kernel<LPB>, dim3((max_grid + 7) & ~7, n_win), dim3(EV_THREADS), 0, s, tab_dev, mode);
        return;
    }
    if (mode != BE_EVAL_CAND_COST) hipLaunchKernelGGL((be_eval_batch_kernel<true, LPB>), dim3((max_grid + 7) & ~7, n_win), dim3(EV_THREADS), 0, s, tab_dev, mode);
    else hipLaunchKernelGGL((be_eval_batch_kernel<false, LPB>), dim3((max_grid + 7) & ~7, n_win), dim3(EV_THREADS), 0, s, tab_dev, mode);
}
void be_launch_eval_batch(const BeEvalArgs* tab_dev, int n_win, int max_grid, int mode, hipStream_t s) {      // max_grid = max over the windows of be_eval_batch_blocks
    // DVINS_EVAL_SPLIT: 1 always, 0 never; default: from 12 windows per launch on — the IMU / prior blocks in a launch of their own let the landmark launch hold fewer registers
    // (64 sequences: 9.6 -> 10.0 k frames/s in round 5; a group of 4 windows only pays for the extra launch: 6.75 -> 6.3 k).  profiles/r05_experiments/solve_cus_and_eval_split_ab.txt
    static const int split_env = [] { const char* e = std::getenv("DVINS_EVAL_SPLIT"); return e ? std::atoi(e) : -1; }();
    const bool split = split_env < 0 ? n_win >= 12 : split_env != 0;
    if (ev_lpb() == 4) launch_eval_batch<4>(tab_dev, n_win, max_grid, mode, split, s); else launch_eval_batch<8>(tab_dev, n_win, max_grid, mode, split, s);
}
void be_launch_eval(const BeEvalArgs& a, int mode, hipStream_t s) {
    const int grid = a.dims.nlm + a.dims.nimu + 1;
    if (mode != BE_EVAL_CAND_COST) hipLaunchKernelGGL(be_eval_kernel<true>, dim3(grid), dim3(EV_THREADS), 0, s, a, mode);
    else hipLaunchKernelGGL(be_eval_kernel<false>, dim3(grid), dim3(EV_THREADS), 0, s, a, mode);
}

// ---- operator-level factor evaluation (parity tests against the oracle's dvo_proj_eval / dvo_imu_eval) ----
__global__ void be_proj_op_kernel(const BeFactor* fac, int n, const double* pose_i, const double* pose_j, const double* ex0, const double* ex1,
                                  const double* lambda, const double* td, double* out /* n x 54: r2 Ji12 Jj12 Jex0_12 Jex1_12 Jl2 Jtd2 */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FrameGeom Fi, Fj;
    Fi.R = qR(Q4(pose_i + 7 * i)); Fi.P = P3(pose_i + 7 * i); Fj.R = qR(Q4(pose_j + 7 * i)); Fj.P = P3(pose_j + 7 * i);
    double* o = out + (size_t)i * 54;
    for (int k = 0; k < 54; ++k) o[k] = 0.0;
    proj_factor<true, true>(fac[i], Fi, Fj, qR(Q4(ex0 + 7 * i)), P3(ex0 + 7 * i), qR(Q4(ex1 + 7 * i)), P3(ex1 + 7 * i), lambda[i], td[i],
                            o, o + 2, o + 14, o + 50, o + 26, o + 38, o + 52);
}
void be_launch_proj_op(const BeFactor* fac, int n, const double* pose_i, const double* pose_j, const double* ex0, const double* ex1,
                       const double* lambda, const double* td, double* out, hipStream_t s) {
    hipLaunchKernelGGL(be_proj_op_kernel, dim3((n + 63) / 64), dim3(64), 0, s, fac, n, pose_i, pose_j, ex0, ex1, lambda, td, out);
}

__global__ void be_imu_op_kernel(const BeImu* m, double g_norm, const double* par /* pose_i7 sb_i9 pose_j7 sb_j9 */, double* out /* r15, J 15x30 whitened */) {
    __shared__ double Jraw[450], rr[15];
    const int lane = threadIdx.x;
    for (int i = lane; i < 450; i += 64) Jraw[i] = 0.0;
    __syncthreads();
    if (lane == 0) imu_raw<true>(*m, g_norm, par, par + 7, par + 16, par + 23, rr, Jraw);
    __syncthreads();
    if (lane < 15) { double s = 0; for (int q = lane; q < 15; ++q) s += m->sqrt_info[lane * 15 + q] * rr[q]; out[lane] = s; }
    for (int e = lane; e < 450; e += 64) {
        const int i = e / 30, cc = e - i * 30; double s = 0;
        for (int q = i; q < 15; ++q) s += m->sqrt_info[i * 15 + q] * Jraw[q * 30 + cc];
        out[15 + e] = s;
    }
}
void be_launch_imu_op(const BeImu* m, double g_norm, const double* par, double* out, hipStream_t s) {
    hipLaunchKernelGGL(be_imu_op_kernel, dim3(1), dim3(64), 0, s, m, g_norm, par, out);
}

// be_prepare (be_api.hip): the runtime loads a code object on the first launch of one of its kernels — asking for a kernel's attributes loads it now, at create time
int be_eval_prepare() { hipFuncAttributes fa; return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(be_eval_kernel<true>)) == hipSuccess ? 0 : -1; }
