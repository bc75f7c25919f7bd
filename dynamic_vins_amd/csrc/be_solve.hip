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
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include "be_kernels.h"
#include "dev_once.h"
#include "wave_dpp.h"
#include "be_mf16.h"

using namespace be;

#ifdef BE_RED_TS
__device__ long long be_red_ts[64];
#define RTS(k) do { if (threadIdx.x == 0) be_red_ts[k] = wall_clock64(); } while (0)
extern "C" int dv_debug_red_ts(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(be_red_ts), sizeof(long long) * 64) == hipSuccess ? 0 : -1; }
#else
#define RTS(k) do {} while (0)
#endif
#define RED_THREADS 448          // 7 waves: waves 0-5 = rows of a 6x6 pose block (lane = landmark), wave 6 = the block's IMU / prior terms
#define RED_PAIRS (BE_NF * (BE_NF + 1) / 2)          // pose block pairs (fi >= fj): the landmark sums are bitwise symmetric (rho * (wi * wj); the direct terms are one stored value read from
                                                      // either side), so the upper blocks are mirrored stores instead of a second pass over the same packet rows (round 5: 121 -> 66 pair blocks)
// Which pair a block takes decides what its XCD's L2 must hold: blocks are dealt to the 8 XCDs round-robin (bx % 8), and pair (fi, fj) streams the coupling rows W of both
// frames and the direct-term rows of one of them.  With the pairs in triangular order an XCD met 10 of the 11 frames (81 frame row-sets fetched into the eight L2s for 11
// that exist); the table below is a partition of the 66 pairs into eight classes of 8 - 9 pairs over 4 - 6 frames each (37 row-sets; found by annealing,
// scripts/dbg/red_pair_partition.py).  Which block computes a pair changes no bit of the result.  Entry = fi * 16 + fj.
__device__ const unsigned char RED_PAIR_TAB[RED_PAIRS] = {
    150, 152, 51, 132, 98, 166, 170, 112, 147, 85, 68, 65, 32, 168, 49, 167, 118, 153, 66, 129, 96, 130, 165, 144, 33, 148, 67, 48, 102, 162, 161, 0, 97,
    133, 50, 16, 101, 134, 17, 119, 99, 84, 114, 64, 80, 100, 163, 160, 145, 149, 115, 131, 82, 164, 81, 151, 113, 117, 116, 128, 34, 136, 83, 169, 146, 135 };
__device__ __forceinline__ void red_pair(int bx, int& fi, int& fj) { const int t = RED_PAIR_TAB[bx]; fi = t >> 4; fj = t & 15; }      // fj <= fi

// IMU + prior part of Hd(i, j)  (everything that is not a landmark sum).  rc: the frames of every IMU factor and the prior's header
// fields, staged in LDS by the caller (they were dependent global loads per factor and entry)
struct RedCtx { const int* ifi; const int* ifj; int prior_valid, prior_n; };
template <class ARGS>
__device__ __forceinline__ double red_dense_h(const ARGS& a, const RedCtx& rc, const double* imu_out, int i, int j) {
    const int ki = a.col_kind[i], fi = a.col_frame[i], ci = a.col_comp[i];
    const int kj = a.col_kind[j], fj = a.col_frame[j], cj = a.col_comp[j];
    double H = 0.0;
    for (int k = 0; k < a.dims.nimu; ++k) {
        const int mfi = rc.ifi[k], mfj = rc.ifj[k];
        int li = -1, lj = -1;
        if (fi == mfi) li = (ki == 0 ? ci : 6 + ci); else if (fi == mfj) li = (ki == 0 ? 15 + ci : 21 + ci);
        if (fj == mfi) lj = (kj == 0 ? cj : 6 + cj); else if (fj == mfj) lj = (kj == 0 ? 15 + cj : 21 + cj);
        if (li >= 0 && lj >= 0) H += imu_out[(size_t)k * IMU_OUT_STRIDE + 31 + li * 30 + lj];
    }
    if (rc.prior_valid) {
        const int pi = a.prior_col[i], pj = a.prior_col[j];
        if (pi >= 0 && pj >= 0) H += a.priorA[(size_t)pi * rc.prior_n + pj];
    }
    return H;
}
// the same sums for a POSE column pair: kind / frame / component are known from the block (no column-map loads), the prior indices come from LDS
template <class ARGS>
__device__ __forceinline__ double red_dense_h_pose(const ARGS& a, const RedCtx& rc, const double* imu_out, int fi, int ci, int fj, int cj, int pi, int pj) {
    // every load unconditional (a dummy address where the term is absent) and the loop unrolled: the <= 11 values travel in ONE round trip instead of one
    // per contributing factor behind its branch; added in the same order
    double v[BE_WIN]; bool ok[BE_WIN];
#pragma unroll
    for (int k = 0; k < BE_WIN; ++k) {
        const bool live = k < a.dims.nimu;
        const int mfi = live ? rc.ifi[k] : -1, mfj = live ? rc.ifj[k] : -1;
        int li = -1, lj = -1;
        if (fi == mfi) li = ci; else if (fi == mfj) li = 15 + ci;
        if (fj == mfi) lj = cj; else if (fj == mfj) lj = 15 + cj;
        ok[k] = live && li >= 0 && lj >= 0;
        v[k] = imu_out[ok[k] ? (size_t)k * IMU_OUT_STRIDE + 31 + li * 30 + lj : 0];
    }
    const bool okp = rc.prior_valid && pi >= 0 && pj >= 0;
    const double vp = a.priorA[okp ? (size_t)pi * rc.prior_n + pj : 0];
    double H = 0.0;
#pragma unroll
    for (int k = 0; k < BE_WIN; ++k) if (ok[k]) H += v[k];
    if (okp) H += vp;
    return H;
}
template <class ARGS>
__device__ __forceinline__ double red_dense_g_pose(const ARGS& a, const RedCtx& rc, const double* imu_out, const double* prior_out, int fi, int ci, int pi) {
    double v[BE_WIN]; bool ok[BE_WIN];
#pragma unroll
    for (int k = 0; k < BE_WIN; ++k) {
        const bool live = k < a.dims.nimu;
        const int mfi = live ? rc.ifi[k] : -1, mfj = live ? rc.ifj[k] : -1;
        int li = -1;
        if (fi == mfi) li = ci; else if (fi == mfj) li = 15 + ci;
        ok[k] = live && li >= 0;
        v[k] = imu_out[ok[k] ? (size_t)k * IMU_OUT_STRIDE + 1 + li : 0];
    }
    const bool okp = rc.prior_valid && pi >= 0;
    const double vp = prior_out[okp ? 1 + pi : 0];
    double G = 0.0;
#pragma unroll
    for (int k = 0; k < BE_WIN; ++k) if (ok[k]) G += v[k];
    if (okp) G += vp;
    return G;
}
template <class ARGS>
__device__ __forceinline__ double red_dense_g(const ARGS& a, const RedCtx& rc, const double* imu_out, const double* prior_out, int i) {
    const int ki = a.col_kind[i], fi = a.col_frame[i], ci = a.col_comp[i];
    double G = 0.0;
    for (int k = 0; k < a.dims.nimu; ++k) {
        const int mfi = rc.ifi[k], mfj = rc.ifj[k];
        int li = -1;
        if (fi == mfi) li = (ki == 0 ? ci : 6 + ci); else if (fi == mfj) li = (ki == 0 ? 15 + ci : 21 + ci);
        if (li >= 0) G += imu_out[(size_t)k * IMU_OUT_STRIDE + 1 + li];
    }
    if (rc.prior_valid) { const int pi = a.prior_col[i]; if (pi >= 0) G += prior_out[1 + pi]; }
    return G;
}
// Position of entry (i, j), j's block <= i's block, in the block-packed lower triangle consumed by the factorisation:
// 4x4 blocks, block-column-major (block (bi, bj) at index bj*NBR - bj(bj-1)/2 + bi - bj), row-major inside a block.
__device__ __forceinline__ int blk_pos(int i, int j, int NBR) {
    const int bi = i >> 2, bj = j >> 2;
    return (bj * NBR - bj * (bj - 1) / 2 + bi - bj) * 16 + (i & 3) * 4 + (j & 3);
}
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_f64(v); }      // fixed tree on the DPP path (wave_dpp.h): deterministic, every lane gets the sum

// blocks [0, 121): pose block (fi, fj) of the reduced system — landmark sums read the transposed packets coalesced
//                  (lane = landmark), wave-tree reduced; blocks [121, ..): every entry that has no landmark term.
#define DV_CONSTANT __attribute__((address_space(4)))
template <bool SHARD, class ARGS>
__device__ __forceinline__ void be_reduce_body(const ARGS& a, int spec, int bx) {
    const BeCtl c = *a.ctl;
    if (c.done) return;
    int set = c.cur; double mu = c.mu;
    if (spec && c.pending) { set ^= 1; mu = fmax(1e-8, 2.0 * c.mu / 10.0); }      // the candidate's set, with the mu be_accept leaves behind an accepted step
    else if (!c.need_eval && !c.chol_fail) return;                                // Hd/Sc are still valid when the last step was rejected
    const int n = a.dims.nstate, nlm = a.dims.nlm;
    const double* pk = a.packets[set];
    double* const Hd = a.Hd[set]; double* const Sc = a.Sc[set]; double* const gvec = a.gvec[set];
    const double* const imu_out = a.imu_out[set]; const double* const prior_out = a.prior_out[set];
    __shared__ int s_ifi[BE_WIN + 1], s_ifj[BE_WIN + 1], s_pr[2];
    __shared__ unsigned short s_list[BE_MAX_LM]; __shared__ int s_cnt;
    if ((int)threadIdx.x < a.dims.nimu) { s_ifi[threadIdx.x] = a.imu[threadIdx.x].fi; s_ifj[threadIdx.x] = a.imu[threadIdx.x].fj; }
    if (threadIdx.x == 64) { s_pr[0] = a.prior->valid; s_pr[1] = a.prior->n; }
    __shared__ int s_pci[6], s_pcj[6];      // pair blocks: prior index of the block's six row / column pose entries (fetched with the tables above: one round trip)
    if (bx < RED_PAIRS && threadIdx.x >= 65 && threadIdx.x < 77) {
        int fi, fj; red_pair(bx, fi, fj);
        const int t = threadIdx.x - 65;
        const int f = t < 6 ? fi : fj, c0 = f < a.dims.nframes ? a.dims.pose_col[f] : -1;
        const int v = c0 >= 0 ? a.prior_col[c0 + (t < 6 ? t : t - 6)] : -1;
        if (t < 6) s_pci[t] = v; else s_pcj[t - 6] = v;
    }
    if (bx < RED_PAIRS && (threadIdx.x >> 6) == 2) {
        // pair block (fi, fj): the landmarks whose factors touch BOTH poses, compacted in ascending order by one wave (ballot + prefix count) — for two
        // frames of an 11-frame window that is a fraction of the landmarks (observed in ~6 frames each); every other packet row would contribute zeros
        int fi, fj; red_pair(bx, fi, fj);
        const int lane = threadIdx.x & 63;
        const int lo = SHARD ? a.sh.lo : 0, hi = SHARD ? a.sh.hi : nlm;
        int base = 0;
        // all observation masks requested up front (16 x 64 lanes covers BE_MAX_LM): with the load inside the ballot loop every 64 landmarks cost one
        // dependent round trip (five for a 300-landmark window) on the path every other wave of the block waits for
        int obv[(BE_MAX_LM + 63) / 64];
#pragma unroll
        for (int u = 0; u < (BE_MAX_LM + 63) / 64; ++u) { const int l = lo + 64 * u + lane; obv[u] = a.lm_obs[l < hi ? l : lo]; }
#pragma unroll
        for (int u = 0; u < (BE_MAX_LM + 63) / 64; ++u) {
            const int c0 = lo + 64 * u;
            if (c0 >= hi) break;
            const int l = c0 + lane;
            const int ob = l < hi ? obv[u] : 0;
            const bool m = ((ob >> fi) & 1) && ((ob >> fj) & 1);
            const unsigned long long bal = __ballot(m);
            if (m) s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)l;
            base += __popcll(bal);
        }
        if (lane == 0) s_cnt = base;
    }
    __syncthreads();
    const RedCtx rc{ s_ifi, s_ifj, s_pr[0], s_pr[1] };
    if (bx == 0) RTS(0);
    if (bx == RED_PAIRS + 3) RTS(8);
    if (bx < RED_PAIRS) {
        int fi, fj; red_pair(bx, fi, fj);
        if (fi >= a.dims.nframes || fj >= a.dims.nframes) return;
        const int ci0 = a.dims.pose_col[fi], cj0 = a.dims.pose_col[fj];
        if (ci0 < 0 || cj0 < 0) return;
        const int ci = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const bool diag = fi == fj;
        // the IMU / prior part of the block's 36 entries (and 6 gradient entries) is a chain of dependent global loads (column maps ->
        // factor / prior indices -> values): the seventh wave walks it WHILE the other six stream the landmark packets
        __shared__ double s_dh[36], s_dg[6];
        if (ci == 6 && !SHARD) {
#ifdef BE_RED_TS
            if (bx == 0 && lane == 0) be_red_ts[16] = wall_clock64();
#endif
            if (lane < 36) s_dh[lane] = red_dense_h_pose(a, rc, imu_out, fi, lane / 6, fj, lane % 6, s_pci[lane / 6], s_pcj[lane % 6]);
            else if (diag && lane < 42) s_dg[lane - 36] = red_dense_g_pose(a, rc, imu_out, prior_out, fi, lane - 36, s_pci[lane - 36]);
#ifdef BE_RED_TS
            if (bx == 0 && lane == 0) be_red_ts[17] = wall_clock64();
#endif
        }
        double S[6] = {0, 0, 0, 0, 0, 0}, H[6] = {0, 0, 0, 0, 0, 0}, G = 0, GS = 0;
        // sharded window: the weighted images of w w^T and g w behind the quadratic forms of be_solve_shard_kernel (be_kernels.h)
        double T2[6] = {0, 0, 0, 0, 0, 0}, T6[6] = {0, 0, 0, 0, 0, 0}, T9[6] = {0, 0, 0, 0, 0, 0}, B2 = 0, B4 = 0, B5 = 0, B6 = 0, B8 = 0, B9 = 0;
        const int e_wi = BE_PK_W + fi * 6 + ci, e_wj = BE_PK_W + fj * 6;
        const int e_dd = BE_PK_DD + fi * 36 + ci * 6;
        const int e_da_i = BE_PK_DA + fj * 36 + ci * 6;                 // anchor == fi : row ci of block (anchor, fj)
        const int e_da_j = BE_PK_DA + fi * 36 + ci;                     // anchor == fj : column ci of the transposed block
        const int cnt = s_cnt;
        for (int idx = ci < 6 ? lane : cnt; idx < cnt; idx += 64) {
            const int l = s_list[idx];
            const double h = BE_PK(pk, BE_PK_H, l);
            const double s = c.first ? 1.0 / (1.0 + sqrt(h)) : a.scale_l[l];
            double d2 = h * s * s; d2 = fmin(fmax(d2, 1e-6), 1e32);
            const double rho = 1.0 / (h + mu * d2 / (s * s));
            const double wi = BE_PK(pk, e_wi, l);
#pragma unroll
            for (int q = 0; q < 6; ++q) S[q] += rho * (wi * BE_PK(pk, e_wj + q, l));      // rho * (wi * wj): bitwise symmetric
            const double al = SHARD ? (s * s) / d2 : 0.0, w2 = SHARD ? rho * rho / al : 0.0, w6 = h * (rho * rho), w9 = rho * rho;
            if (SHARD) {
#pragma unroll
                for (int q = 0; q < 6; ++q) { const double wp = wi * BE_PK(pk, e_wj + q, l); T2[q] += w2 * wp; T6[q] += w6 * wp; T9[q] += w9 * wp; }
            }
            if (diag) {
#pragma unroll
                for (int q = 0; q < 6; ++q) H[q] += BE_PK(pk, e_dd + q, l);
                G += BE_PK(pk, BE_PK_GP + fi * 6 + ci, l);
                GS += rho * (wi * BE_PK(pk, BE_PK_G, l));
                if (SHARD) {
                    const double gw = BE_PK(pk, BE_PK_G, l) * wi;
                    B2 += w2 * gw; B4 += al * gw; B5 += (h * al * rho) * gw; B6 += w6 * gw; B8 += (al * rho) * gw; B9 += w9 * gw;
                }
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
        if (bx == 0) RTS(1);
#pragma unroll
        for (int q = 0; q < 6; ++q) { S[q] = wave_sum(S[q]); H[q] = wave_sum(H[q]); }
        if (diag) { G = wave_sum(G); GS = wave_sum(GS); }
        if (SHARD) {
#pragma unroll
            for (int q = 0; q < 6; ++q) { T2[q] = wave_sum(T2[q]); T6[q] = wave_sum(T6[q]); T9[q] = wave_sum(T9[q]); }
            if (diag) { B2 = wave_sum(B2); B4 = wave_sum(B4); B5 = wave_sum(B5); B6 = wave_sum(B6); B8 = wave_sum(B8); B9 = wave_sum(B9); }
        }
        __syncthreads();
        if (bx == 0) RTS(2);
        if (ci == 6) return;
        if (SHARD) {              // sharded window: the partial landmark sums go to the exchange vector (lower block pair bx, row ci, column lane); be_shard_finalize_kernel adds the ranks up and the dense part in
            if (lane < 6) {
                double v0 = S[0], v1 = H[0], v2 = T2[0], v3 = T6[0], v4 = T9[0];
#pragma unroll
                for (int q = 1; q < 6; ++q) if (lane == q) { v0 = S[q]; v1 = H[q]; v2 = T2[q]; v3 = T6[q]; v4 = T9[q]; }
                const int e = bx * 36 + ci * 6 + lane;
                a.sh.xsend[BE_XS_M(0) + e] = v0; a.sh.xsend[BE_XS_M(1) + e] = v1; a.sh.xsend[BE_XS_M(2) + e] = v2; a.sh.xsend[BE_XS_M(3) + e] = v3; a.sh.xsend[BE_XS_M(4) + e] = v4;
            } else if (diag && lane == 6) {
                const int e = fi * 6 + ci;
                a.sh.xsend[BE_XS_V(0) + e] = G; a.sh.xsend[BE_XS_V(1) + e] = GS; a.sh.xsend[BE_XS_V(2) + e] = B2; a.sh.xsend[BE_XS_V(3) + e] = B4;
                a.sh.xsend[BE_XS_V(4) + e] = B5; a.sh.xsend[BE_XS_V(5) + e] = B6; a.sh.xsend[BE_XS_V(6) + e] = B8; a.sh.xsend[BE_XS_V(7) + e] = B9;
            }
            return;
        }
        if (lane < 6) {
            double sv = S[0], hv = H[0];
#pragma unroll
            for (int q = 1; q < 6; ++q) if (lane == q) { sv = S[q]; hv = H[q]; }
            const int i = ci0 + ci, j = cj0 + lane;
            const double hd = hv + s_dh[ci * 6 + lane];
            Hd[(size_t)i * n + j] = hd;
            if ((j >> 2) <= (i >> 2)) Sc[blk_pos(i, j, (n + 3) >> 2)] = hd - sv;
            if (!diag) {                                  // the mirrored entry (the IMU / prior part is symmetric as well: the same sums in the same order with the roles swapped)
                Hd[(size_t)j * n + i] = hd;
                if ((i >> 2) <= (j >> 2)) Sc[blk_pos(j, i, (n + 3) >> 2)] = hd - sv;
            }
        } else if (diag && lane == 6) {
            const int i = ci0 + ci;
            gvec[i] = G + s_dg[ci];
            gvec[n + i] = GS;
        }
        if (bx == 0) RTS(3);
        return;
    }
    int eb = bx - RED_PAIRS;
    const int n_dense = (n * n + n + RED_THREADS - 1) / RED_THREADS;
    if (eb < n_dense) {
        // the dense blocks walk the matrix row by row: with block b on XCD b % 8 every L2 fetched nearly all of the IMU blocks' output and of the prior (8 x 129 KB for one
        // window).  Renumbered so that the blocks one XCD receives cover a CONTIGUOUS range of rows: the frames of that range, their two IMU factors, their rows of A'
        const int cls = bx & 7;
        int start = 0;
#pragma unroll
        for (int y = 0; y < 8; ++y) { const int e0 = (y - RED_PAIRS) & 7; if (y < cls && n_dense > e0) start += (n_dense - e0 + 7) >> 3; }
        eb = start + ((eb - ((cls - RED_PAIRS) & 7)) >> 3);          // bijection on [0, n_dense)
    }
    if (eb >= n_dense) {          // sharded window only (one block): the scalar sums over the owned landmarks, fixed order (strided partial sums, block tree), and the owned Jacobi scales
        if (!SHARD) return;
        __shared__ double s_red[7][13];
        const double* xst = reinterpret_cast<const double*>((spec && c.pending) ? a.cand : a.x) + offsetof(BeState, inv_depth) / sizeof(double);
        double acc[13];          // [12]: the owned candidate costs (a speculative linearisation carries the costs its accept decision will sum: the cost-only exchange belongs to the last slot alone)
#pragma unroll
        for (int k = 0; k < 13; ++k) acc[k] = 0.0;
        for (int l = a.sh.lo + (int)threadIdx.x; l < a.sh.hi; l += RED_THREADS) {
            const double h = BE_PK(pk, BE_PK_H, l), g = BE_PK(pk, BE_PK_G, l);
            const double s = c.first ? 1.0 / (1.0 + sqrt(h)) : a.scale_l[l];
            if (c.first) a.scale_l[l] = s;
            double d2 = h * s * s; d2 = fmin(fmax(d2, 1e-6), 1e32);
            const double rho = 1.0 / (h + mu * d2 / (s * s)), al = (s * s) / d2, g2 = g * g, x0 = xst[l];
            acc[0] += al * g2; acc[1] += (rho * rho / al) * g2; acc[2] += rho * g2; acc[3] += (h * al * al) * g2; acc[4] += (h * al * rho) * g2; acc[5] += (h * (rho * rho)) * g2;
            acc[6] += (al * al) * g2; acc[7] += (al * rho) * g2; acc[8] += (rho * rho) * g2; acc[9] += x0 * x0; acc[10] += BE_PK(pk, BE_PK_COST, l); acc[11] = fmax(acc[11], fabs(g)); acc[12] += a.cand_cost[l];
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) acc[k] = wave_sum(acc[k]);
        acc[12] = wave_sum(acc[12]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[11] = fmax(acc[11], __shfl_xor(acc[11], o));
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 13; ++k) s_red[threadIdx.x >> 6][k] = acc[k];
        }
        __syncthreads();
        if (threadIdx.x < 16) {
            double v = 0.0;
            if (threadIdx.x < 11 || threadIdx.x == 12) for (int w = 0; w < RED_THREADS / 64; ++w) v += s_red[w][threadIdx.x];
            else if (threadIdx.x == 11) for (int w = 0; w < RED_THREADS / 64; ++w) v = fmax(v, s_red[w][11]);
            a.sh.xsend[BE_XS_A + threadIdx.x] = v;
        }
        return;
    }
    if (eb == 0 && c.first && !SHARD)
        for (int l = threadIdx.x; l < nlm; l += RED_THREADS) a.scale_l[l] = 1.0 / (1.0 + sqrt(BE_PK(pk, BE_PK_H, l)));
    const int t = eb * RED_THREADS + threadIdx.x;
    if (t < n * n) {
        const int i = t / n, j = t - i * n;
        if (a.col_kind[i] == 0 && a.col_kind[j] == 0) return;          // pose x pose: written by the pair blocks
        const double hd = red_dense_h(a, rc, imu_out, i, j);
        Hd[t] = hd;
        if ((j >> 2) <= (i >> 2)) Sc[blk_pos(i, j, (n + 3) >> 2)] = hd;
    } else if (t < n * n + n) {
        const int i = t - n * n;
        if (a.col_kind[i] == 0) return;
        gvec[i] = red_dense_g(a, rc, imu_out, prior_out, i); gvec[n + i] = 0.0;
    }
    if (bx == RED_PAIRS + 3) RTS(9);
}

__global__ __launch_bounds__(RED_THREADS) void be_reduce_kernel(BeSolveArgs a, int spec) { be_reduce_body<false>(a, spec, blockIdx.x); }
__global__ __launch_bounds__(RED_THREADS) void be_reduce_shard_kernel(BeSolveArgs a, int spec) { be_reduce_body<true>(a, spec, blockIdx.x); }
__global__ __launch_bounds__(RED_THREADS) void be_reduce_batch_kernel(const BeSolveArgs* __restrict__ tab, int spec) {      // blockIdx.y = window
    const DV_CONSTANT BeSolveArgs& a = *reinterpret_cast<const DV_CONSTANT BeSolveArgs*>(reinterpret_cast<uintptr_t>(tab + blockIdx.y));      // the table through the constant (scalar, invariant) path
    const int n = a.dims.nstate;
    if ((int)blockIdx.x >= RED_PAIRS + (n * n + n + RED_THREADS - 1) / RED_THREADS) return;
    be_reduce_body<false>(a, spec, blockIdx.x);
}
// the same body capped at 80 VGPRs (5 of them spilled, 24 B of scratch per lane): THREE 7-wave workgroups per CU instead of two — 16 windows are 2048 workgroups against 512
// resident at 96 VGPRs (four occupancy rounds of ~15 us: what the launch's 66 us were in round 5), 768 with the cap.  Used from 12 windows per launch on (be_launch_reduce_batch).
__global__ __launch_bounds__(RED_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8))) void be_reduce_batch_occ_kernel(const BeSolveArgs* __restrict__ tab, int spec) {
    const DV_CONSTANT BeSolveArgs& a = *reinterpret_cast<const DV_CONSTANT BeSolveArgs*>(reinterpret_cast<uintptr_t>(tab + blockIdx.y));
    const int n = a.dims.nstate;
    if ((int)blockIdx.x >= RED_PAIRS + (n * n + n + RED_THREADS - 1) / RED_THREADS) return;
    be_reduce_body<false>(a, spec, blockIdx.x);
}
void be_launch_reduce_batch(const BeSolveArgs* tab_dev, int n_win, int max_n, int spec, hipStream_t s) {
    // measured (round 6, one box): 64 sequences (groups of 16 windows) 10167 / 10547 -> 10458 / 11121 frames/s with the cap; 16 sequences (groups of 4: 512 workgroups, one round either
    // way) 7245 -> 7187: the cap only pays when a launch exceeds the resident set — from 12 windows per launch on, like the evaluation's split.  DVINS_REDUCE_OCC=0 / 1 forces it.
    static const int occ_env = [] { const char* e = std::getenv("DVINS_REDUCE_OCC"); return e ? (e[0] != '0' ? 1 : 0) : -1; }();
    const bool occ = occ_env < 0 ? n_win >= 12 : occ_env != 0;
    const dim3 grid((RED_PAIRS + (max_n * max_n + max_n + RED_THREADS - 1) / RED_THREADS + 7) & ~7, n_win);      // (x a multiple of 8: a block's XCD is blockIdx.x % 8 in every window, as RED_PAIR_TAB assumes)
    if (occ) hipLaunchKernelGGL(be_reduce_batch_occ_kernel, grid, dim3(RED_THREADS), 0, s, tab_dev, spec);
    else hipLaunchKernelGGL(be_reduce_batch_kernel, grid, dim3(RED_THREADS), 0, s, tab_dev, spec);
}
void be_launch_reduce(const BeSolveArgs& a, int spec, hipStream_t s) {
    const int n = a.dims.nstate;
    const int total = n * n + n;
    if (a.sh.on) hipLaunchKernelGGL(be_reduce_shard_kernel, dim3(RED_PAIRS + (total + RED_THREADS - 1) / RED_THREADS + 1), dim3(RED_THREADS), 0, s, a, spec);
    else hipLaunchKernelGGL(be_reduce_kernel, dim3(RED_PAIRS + (total + RED_THREADS - 1) / RED_THREADS), dim3(RED_THREADS), 0, s, a, spec);
}

// Sharded window, after the all-gather of the exchange vectors: sums the ranks' partial sums in RANK ORDER (identical bits on every rank), adds the IMU / prior part, writes
// Hd / Sc / gvec of the set be_reduce worked on (upper blocks mirrored) and the summed form coefficients (qf of that set) for be_solve_shard_kernel.  Same predicate as
// be_reduce_kernel, so an idle slot leaves the previous system untouched.
#define FIN_THREADS 256
__global__ __launch_bounds__(FIN_THREADS) void be_shard_finalize_kernel(BeSolveArgs a, int spec) {
    const BeCtl c = *a.ctl;
    if (c.done) return;
    int set = c.cur;
    if (spec && c.pending) set ^= 1;
    else if (!c.need_eval && !c.chol_fail) return;
    const int n = a.dims.nstate, W = a.sh.world, len = a.sh.len, t = threadIdx.x;
    const double* xr = a.sh.xrecv;
    double* qf = a.sh.qf[set];
    auto rsum = [&](int e) { double v = 0.0; for (int r = 0; r < W; ++r) v += xr[(size_t)r * len + e]; return v; };
    if (blockIdx.x < RED_PAIRS) {
        int fi, fj; red_pair(blockIdx.x, fi, fj);
        const bool in_win = fi < a.dims.nframes && fj < a.dims.nframes;
        const int ci0 = in_win ? a.dims.pose_col[fi] : -1, cj0 = in_win ? a.dims.pose_col[fj] : -1;
        const bool valid = ci0 >= 0 && cj0 >= 0;      // (be_reduce writes the exchange vector for such pairs only: the rest of the 66-space is zero)
        if (t < 36) {          // the form matrices in the 66-space (frame * 6 + component), mirrored
            const int ci = t / 6, q = t - ci * 6, e = blockIdx.x * 36 + t, i66 = fi * 6 + ci, j66 = fj * 6 + q;
            const double v0 = valid ? rsum(BE_XS_M(0) + e) : 0.0, v2 = valid ? rsum(BE_XS_M(2) + e) : 0.0, v3 = valid ? rsum(BE_XS_M(3) + e) : 0.0, v4 = valid ? rsum(BE_XS_M(4) + e) : 0.0;
            qf[BE_QF_M(0) + i66 * 66 + j66] = v0; qf[BE_QF_M(1) + i66 * 66 + j66] = v2; qf[BE_QF_M(2) + i66 * 66 + j66] = v3; qf[BE_QF_M(3) + i66 * 66 + j66] = v4;
            if (fi != fj) { qf[BE_QF_M(0) + j66 * 66 + i66] = v0; qf[BE_QF_M(1) + j66 * 66 + i66] = v2; qf[BE_QF_M(2) + j66 * 66 + i66] = v3; qf[BE_QF_M(3) + j66 * 66 + i66] = v4; }
        } else if (fi == fj && t >= 64 && t < 70) {
            const int e = fi * 6 + (t - 64);
#pragma unroll
            for (int k = 0; k < 7; ++k) qf[BE_QF_V(k) + e] = valid ? rsum(BE_XS_V(k + 1) + e) : 0.0;
        }
        if (!valid) return;
        __shared__ int s_ifi[BE_WIN + 1], s_ifj[BE_WIN + 1], s_pr[2];
        if (t < a.dims.nimu) { s_ifi[t] = a.imu[t].fi; s_ifj[t] = a.imu[t].fj; }
        if (t == 64) { s_pr[0] = a.prior->valid; s_pr[1] = a.prior->n; }
        __syncthreads();
        const RedCtx rc{ s_ifi, s_ifj, s_pr[0], s_pr[1] };
        if (t < 36) {
            const int ci = t / 6, q = t - ci * 6, e = blockIdx.x * 36 + t;
            const double sv = rsum(BE_XS_M(0) + e), hv = rsum(BE_XS_M(1) + e);
            const int i = ci0 + ci, j = cj0 + q;
            const double hd = hv + red_dense_h(a, rc, a.imu_out[set], i, j);
            a.Hd[set][(size_t)i * n + j] = hd;
            if ((j >> 2) <= (i >> 2)) a.Sc[set][blk_pos(i, j, (n + 3) >> 2)] = hd - sv;
            if (fi != fj) {
                a.Hd[set][(size_t)j * n + i] = hd;
                if ((i >> 2) <= (j >> 2)) a.Sc[set][blk_pos(j, i, (n + 3) >> 2)] = hd - sv;
            }
        } else if (fi == fj && t >= 64 && t < 70) {
            const int ci = t - 64, i = ci0 + ci;
            a.gvec[set][i] = rsum(BE_XS_V(0) + fi * 6 + ci) + red_dense_g(a, rc, a.imu_out[set], a.prior_out[set], i);
            a.gvec[set][n + i] = rsum(BE_XS_V(1) + fi * 6 + ci);
        }
        return;
    }
    if (t < 16) {              // the scalars (entry 11: a maximum)
        double v = 0.0;
        if (t == 11) for (int r = 0; r < W; ++r) v = fmax(v, xr[(size_t)r * len + BE_XS_A + 11]);
        else v = rsum(BE_XS_A + t);
        qf[BE_QF_A + t] = v;
    }
    if (spec && c.pending) {   // the candidate's costs for the decision the next solve takes: the rank-ordered total in the first landmark slot, zeros in the others (be_shard_cost_kernel's convention)
        double* cc = const_cast<double*>(a.cand_cost);
        for (int l = t; l < a.dims.nlm; l += FIN_THREADS) cc[l] = l == 0 ? rsum(BE_XS_A + 12) : 0.0;
    }
}
void be_launch_shard_finalize(const BeSolveArgs& a, int spec, hipStream_t s) {
    hipLaunchKernelGGL(be_shard_finalize_kernel, dim3(RED_PAIRS + 1), dim3(FIN_THREADS), 0, s, a, spec);
}
// cost-only exchange (the slot that ends with be_accept): phase 0 sums the owned candidate costs in a fixed order into ONE double; phase 1 adds the ranks' partial sums in
// rank order and leaves the total in the first landmark slot of cand_cost, zeros in the others — the accept decision's own summation tree then yields the same bits on every rank
__global__ __launch_bounds__(FIN_THREADS) void be_shard_cost_kernel(BeSolveArgs a, int phase) {
    const BeCtl c = *a.ctl;
    if (c.done || !c.pending) return;
    const int t = threadIdx.x;
    if (phase == 0) {
        __shared__ double s_w[FIN_THREADS / 64];
        double v = 0.0;
        for (int l = a.sh.lo + t; l < a.sh.hi; l += FIN_THREADS) v += a.cand_cost[l];
        v = wave_sum(v);
        if ((t & 63) == 0) s_w[t >> 6] = v;
        __syncthreads();
        if (t == 0) { double tot = 0.0; for (int w = 0; w < FIN_THREADS / 64; ++w) tot += s_w[w]; a.sh.xsend[0] = tot; }
        else if (t < 8) a.sh.xsend[t] = 0.0;
    } else {
        double* cc = const_cast<double*>(a.cand_cost);
        for (int l = t; l < a.dims.nlm; l += FIN_THREADS) {
            double v = 0.0;
            if (l == 0) for (int r = 0; r < a.sh.world; ++r) v += a.sh.xrecv[(size_t)r * 8];
            cc[l] = v;
        }
    }
}
void be_launch_shard_cost(const BeSolveArgs& a, int phase, hipStream_t s) { hipLaunchKernelGGL(be_shard_cost_kernel, dim3(1), dim3(FIN_THREADS), 0, s, a, phase); }
// behind the last iteration slot: every rank holds the solved inverse depths of its own landmarks only; phase 0 packs them, phase 1 scatters the gathered ranges into x
__global__ __launch_bounds__(FIN_THREADS) void be_shard_depth_kernel(BeSolveArgs a, int phase) {
    const int cap = a.sh.cap;
    if (phase == 0) { for (int k = threadIdx.x; k < cap; k += FIN_THREADS) { const int l = a.sh.lo + k; a.sh.xsend[k] = l < a.sh.hi ? a.x->inv_depth[l] : 0.0; } }
    else for (int l = threadIdx.x; l < a.dims.nlm; l += FIN_THREADS) { const int r = l / cap; if (r != a.sh.rank) a.x->inv_depth[l] = a.sh.xrecv[(size_t)r * cap + (l - r * cap)]; }
}
void be_launch_shard_depth(const BeSolveArgs& a, int phase, hipStream_t s) { hipLaunchKernelGGL(be_shard_depth_kernel, dim3(1), dim3(FIN_THREADS), 0, s, a, phase); }

// ---------------------------------------------------------------------------------------------
#define SOL_THREADS 1024
#ifndef BS_R
#define BS_R 8
#endif
#ifdef BE_SOLVE_TS
__device__ long long be_dbg_ts[48];
#define TS(k) do { if (threadIdx.x == 0) { be_dbg_ts[k] = wall_clock64(); if ((k) == 4) be_dbg_ts[30] = clock64(); if ((k) == 5) be_dbg_ts[31] = clock64(); } } while (0)
#define TSW(k) do { if (threadIdx.x == SOL_THREADS - 64) be_dbg_ts[k] = wall_clock64(); } while (0)
extern "C" int dv_debug_solve_ts(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(be_dbg_ts), sizeof(long long) * 48) == hipSuccess ? 0 : -1; }
#else
#define TS(k) do {} while (0)
#define TSW(k) do {} while (0)
#endif

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

// N sums with ONE LDS exchange (two barriers) instead of N: red must hold 16 * N doubles
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[i * 16 + w] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) { double s = 0; for (int k = 0; k < SOL_THREADS / 64; ++k) s += red[i * 16 + k]; v[i] = s; }
}
// The same N sums on the cross-lane data path instead of LDS shuffles (__shfl_xor of a double is two ds_bpermute round trips per step, and the second
// stage above has every thread read all 16 * N partials).  Stage 1: inclusive row scan (row_shr 1, 2, 4, 8), row_bcast15, row_bcast31: lane 63 holds the
// wave's total, v_readlane hands it to lane 0 for red[].  Stage 2: lane 16 * i + k of EVERY wave loads partial k of value i (a second register for
// i >= 4), one row scan, the totals come back through v_readlane (lane-uniform).  Fixed tree: deterministic; LDS-only barriers.  N <= 8.
template <int N>
__device__ __forceinline__ void block_sum_n_dpp(double (&v)[N], double* red) {
    static_assert(N <= 8, "two registers of 4 x 16 partials");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        v[i] = wave_sum_f64(v[i]);
    }
    lds_barrier();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[i * 16 + w] = v[i];
    }
    lds_barrier();
    double r0 = lane < (N < 4 ? N : 4) * 16 ? red[lane] : 0.0;
    double r1 = (N > 4 && lane < (N - 4) * 16) ? red[64 + lane] : 0.0;
    r0 = row_scan_f64(r0);
    if (N > 4) r1 = row_scan_f64(r1);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = i < 4 ? lane_bcast(r0, 16 * i + 15) : lane_bcast(r1, 16 * (i - 4) + 15);
}

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }      // packed lower, j <= i

// y = Hd * x (Hd symmetric: thread = output entry reading column-wise, i.e. coalesced rows).  5 partial sums per
// entry; the (up to 36) loads of a thread are address-independent and issued in batches (the matrix was written by
// other XCDs, so every load is a ~1 us round trip: memory-level parallelism is what matters here).
__device__ __forceinline__ void gemv_hd(const double* Hd, int n, const double* x, double* y, double* scratch, int tid) {
    const int col = tid % 192, part = tid / 192;       // parts 0..4 (tid >= 960 idle)
    const int seg = (n + 4) / 5, j0 = part * seg;
    const bool live = part < 5 && col < n;
    double s = 0;
#pragma unroll 12
    for (int q = 0; q < 36; ++q) {
        const int j = j0 + q;
        const bool ok = live && q < seg && j < n;
        const double h = Hd[ok ? (size_t)j * n + col : 0];
        const double xv = x[ok ? j : 0];
        s += ok ? h * xv : 0.0;
    }
    if (part < 5) scratch[part * 192 + col] = s;
    __syncthreads();
    if (tid < n) y[tid] = (scratch[tid] + scratch[192 + tid]) + (scratch[384 + tid] + scratch[576 + tid]) + scratch[768 + tid];
    __syncthreads();
}

// Panel-blocked right-looking LDL^T of the (scaled, damped) reduced camera system.  Every thread owns NSLOT 4x4
// blocks of the lower triangle in registers (block-column-major, so whole waves retire as the factorisation
// proceeds).  Per block column kb:  (a) the owner of the diagonal block factors it in registers and publishes
// L_kk, D_kk;  (b) the owners of the blocks below turn theirs into the panel of L (and L D) and publish it in LDS;
// (c) every block to the right subtracts  L_i D L_j^T  (64 FMAs against 32 LDS reads).  Two barriers per FOUR
// columns.  The right-hand side is forward-substituted alongside in the registers of the diagonal-block owners.
// On success: Lm = unit-lower L (packed row-major), dvec = D, zfin = L^-1 rhs.   Returns false on a non-positive pivot.
#define PSTR 184            // row stride of the panel buffers (>= 4 * ceil(178 / 4), even)
template <int NSLOT>
__device__ __forceinline__ bool ldlt_blocked(const double* __restrict__ Sc, const double* __restrict__ gvec, int n, double mu, const double* v_s, const double* v_d,
                                             double* Lm, double* PL, double* PD, double* dinfo, double* zfin, double* dvec, int* s_fail) {
    const int tid = threadIdx.x;
    const int NBR = (n + 3) >> 2, nblk = NBR * (NBR + 1) / 2;
    int bi[NSLOT], bj[NSLOT];
    double A[NSLOT][4][4], zr[NSLOT][4];
#pragma unroll
    for (int b = 0; b < NSLOT; ++b) {
        const int idx = tid + b * SOL_THREADS;
        bi[b] = -1; bj[b] = -1;
        if (idx < nblk) { int c0 = 0, rem = idx; while (rem >= NBR - c0) { rem -= NBR - c0; ++c0; } bj[b] = c0; bi[b] = c0 + rem; }
        const bool have = bi[b] >= 0;
        double dv[4][4];
        {   // one 128-byte block per thread, consecutive threads -> consecutive blocks: 4 coalesced 32-byte loads
            const double4* src = reinterpret_cast<const double4*>(Sc + (size_t)(have ? idx : 0) * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) { const double4 q = src[r]; dv[r][0] = q.x; dv[r][1] = q.y; dv[r][2] = q.z; dv[r][3] = q.w; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = bi[b] * 4 + r;
            const bool iok = have && i < n;
            const double si = v_s[iok ? i : 0];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int j = bj[b] * 4 + cc;
                const bool ok = iok && j < n;
                double v = ok ? si * v_s[ok ? j : 0] * dv[r][cc] : 0.0;
                if (i == j) v = ok ? v + mu * v_d[ok ? i : 0] * v_d[ok ? i : 0] : 1.0;      // padding rows: identity
                A[b][r][cc] = v;
            }
            zr[b][r] = (iok && bi[b] == bj[b]) ? si * (gvec[i] - gvec[n + i]) : 0.0;
        }
    }
    if (tid == 0) *s_fail = 0;
    __syncthreads();
    TS(4);
#ifdef BE_SOLVE_TS
    long long t_prev = wall_clock64(), acc_a = 0, acc_b = 0;
#endif
    for (int kb = 0; kb < NBR; ++kb) {
        double* di = dinfo + (kb & 1) * 16;       // double-buffered: (a) of the next block column may overwrite while (c) still reads
        // ---- (a) diagonal block ----
#pragma unroll
        for (int b = 0; b < NSLOT; ++b)
            if (bi[b] == kb && bj[b] == kb) {
                const double a10 = A[b][1][0], a20 = A[b][2][0], a30 = A[b][3][0];
                const double d0 = A[b][0][0], i0 = fast_rcp(d0);
                const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
                const double d1 = __builtin_fma(-l10, a10, A[b][1][1]), i1 = fast_rcp(d1);
                const double t21 = __builtin_fma(-l20, a10, A[b][2][1]), t31 = __builtin_fma(-l30, a10, A[b][3][1]);
                const double l21 = t21 * i1, l31 = t31 * i1;
                const double d2 = __builtin_fma(-l21, t21, __builtin_fma(-l20, a20, A[b][2][2])), i2 = fast_rcp(d2);
                const double t32 = __builtin_fma(-l31, t21, __builtin_fma(-l30, a20, A[b][3][2]));
                const double l32 = t32 * i2;
                const double d3 = __builtin_fma(-l32, t32, __builtin_fma(-l31, t31, __builtin_fma(-l30, a30, A[b][3][3]))), i3 = fast_rcp(d3);
                if (!(d0 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0) || !(d3 > 0.0) || !isfinite(d0 + d1 + d2 + d3)) *s_fail = 1;
                const double z0 = zr[b][0], z1 = __builtin_fma(-l10, z0, zr[b][1]);
                const double z2 = __builtin_fma(-l21, z1, __builtin_fma(-l20, z0, zr[b][2]));
                const double z3 = __builtin_fma(-l32, z2, __builtin_fma(-l31, z1, __builtin_fma(-l30, z0, zr[b][3])));
                di[0] = l10; di[1] = l20; di[2] = l30; di[3] = l21; di[4] = l31; di[5] = l32;
                di[6] = i0; di[7] = i1; di[8] = i2; di[9] = i3;
                di[10] = z0; di[11] = z1; di[12] = z2; di[13] = z3;
                const int r0 = kb * 4;
                if (r0 < n) { dvec[r0] = d0; zfin[r0] = z0; }
                if (r0 + 1 < n) { dvec[r0 + 1] = d1; zfin[r0 + 1] = z1; }
                if (r0 + 2 < n) { dvec[r0 + 2] = d2; zfin[r0 + 2] = z2; }
                if (r0 + 3 < n) { dvec[r0 + 3] = d3; zfin[r0 + 3] = z3; }
                A[b][1][0] = l10; A[b][2][0] = l20; A[b][3][0] = l30; A[b][2][1] = l21; A[b][3][1] = l31; A[b][3][2] = l32;
            }
        __syncthreads();
#ifdef BE_SOLVE_TS
        { long long t = wall_clock64(); acc_a += t - t_prev; t_prev = t; }
#endif
        if (*s_fail) break;                          // uniform (LDS word)
        // ---- (b) panel below the diagonal block ----
#pragma unroll
        for (int b = 0; b < NSLOT; ++b)
            if (bj[b] == kb && bi[b] > kb) {
                const double l10 = di[0], l20 = di[1], l30 = di[2], l21 = di[3], l31 = di[4], l32 = di[5];
                const double i0 = di[6], i1 = di[7], i2 = di[8], i3 = di[9];
                const int i0r = bi[b] * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double p0 = A[b][r][0];
                    const double p1 = __builtin_fma(-p0, l10, A[b][r][1]);
                    const double p2 = __builtin_fma(-p1, l21, __builtin_fma(-p0, l20, A[b][r][2]));
                    const double p3 = __builtin_fma(-p2, l32, __builtin_fma(-p1, l31, __builtin_fma(-p0, l30, A[b][r][3])));
                    const double x0 = p0 * i0, x1 = p1 * i1, x2 = p2 * i2, x3 = p3 * i3;
                    double* pl = PL + i0r + r; double* pd = PD + i0r + r;          // [m][row]: lanes (consecutive block rows) are 32 B apart -> conflict-free
                    pl[0] = x0; pl[PSTR] = x1; pl[2 * PSTR] = x2; pl[3 * PSTR] = x3;
                    pd[0] = p0; pd[PSTR] = p1; pd[2 * PSTR] = p2; pd[3 * PSTR] = p3;
                    A[b][r][0] = x0; A[b][r][1] = x1; A[b][r][2] = x2; A[b][r][3] = x3;
                }
            }
        __syncthreads();
#ifdef BE_SOLVE_TS
        { long long t = wall_clock64(); acc_b += t - t_prev; t_prev = t; }
#endif
        // ---- (c) trailing update ----
#pragma unroll
        for (int b = 0; b < NSLOT; ++b)
            if (bj[b] > kb) {
                const double* pl = PL + bi[b] * 4; const double* pd = PD + bj[b] * 4;
                double li[4][4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double2 u0 = *reinterpret_cast<const double2*>(pl + m * PSTR), u1 = *reinterpret_cast<const double2*>(pl + m * PSTR + 2);
                    li[0][m] = u0.x; li[1][m] = u0.y; li[2][m] = u1.x; li[3][m] = u1.y;
                }
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const double q0 = pd[cc], q1 = pd[PSTR + cc], q2 = pd[2 * PSTR + cc], q3 = pd[3 * PSTR + cc];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        A[b][r][cc] = __builtin_fma(-li[r][3], q3, __builtin_fma(-li[r][2], q2, __builtin_fma(-li[r][1], q1, __builtin_fma(-li[r][0], q0, A[b][r][cc]))));
                }
                if (bi[b] == bj[b]) {
                    const double y0 = di[10], y1 = di[11], y2 = di[12], y3 = di[13];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        zr[b][r] = __builtin_fma(-li[r][3], y3, __builtin_fma(-li[r][2], y2, __builtin_fma(-li[r][1], y1, __builtin_fma(-li[r][0], y0, zr[b][r]))));
                }
            }
    }
    __syncthreads();
    TS(5);
#ifdef BE_SOLVE_TS
    if (tid == 0) { be_dbg_ts[16] = acc_a; be_dbg_ts[17] = acc_b; }
#endif
    if (*s_fail) return false;
    // the unit-lower factor in packed row-major form for the back substitution
#pragma unroll
    for (int b = 0; b < NSLOT; ++b)
        if (bi[b] >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int i = bi[b] * 4 + r, j = bj[b] * 4 + cc;
                    if (i < n && j < i) Lm[tri(i, j)] = A[b][r][cc];
                }
        }
    __syncthreads();
    return true;
}


// tile entries: the raw Schur-complement entry is requested early (mf_prefetch, before the scaling phase: the loads' round trip hides behind it) and finished
// (scaled, damped; right-hand side as row / column n; identity padding behind it) when the factorisation starts
__device__ __forceinline__ double mf_raw(const double* __restrict__ Sc, int n, int i, int j) {
    if (i < j) { const int t = i; i = j; j = t; }
    const int NBR = (n + 3) >> 2, bi = i >> 2, bj = j >> 2;
    const bool in = i < n;
    return Sc[in ? (size_t)(bj * NBR - bj * (bj - 1) / 2 + bi - bj) * 16 + (i & 3) * 4 + (j & 3) : 0];      // (the value is ignored for i >= n)
}
__device__ __forceinline__ double mf_finish(double raw, const double* v_s, const double* v_d, const double* rhs, int n, double mu, int i, int j) {
    // by selection, not by branch (the lanes of a diagonal or last-row tile fall into all the cases at once)
    const int hi = i > j ? i : j, lo = i > j ? j : i, hc = hi < n ? hi : n - 1, lc = lo < n ? lo : n - 1;
    const double s_hi = v_s[hc], s_lo = v_s[lc], d_hi = v_d[hc];
    const double body = s_hi * s_lo * raw + (i == j ? mu * d_hi * d_hi : 0.0);
    const double row_n = lo == n ? MF_RHO : s_lo * rhs[lc];
    return hi < n ? body : (hi == n ? row_n : (hi == lo ? 1.0 : 0.0));
}
__device__ __forceinline__ void mf_prefetch(const double* __restrict__ Sc, const uint8_t* plan, int n, mf_d4 (&U)[5]) {
    const int lane = threadIdx.x & 63, wave = mf_wave(), c = lane & 15, rho = lane >> 4;
    int sI[5], sJ[5];
    mf_slots(plan, wave, (n + 16) >> 4, sI, sJ);
    const int NBR = (n + 3) >> 2;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        if (sI[s] > sJ[s]) {
            // a tile below the diagonal: i > j for every entry, so the 4x4 block of entry (i, j = 16 J + rho + 4 r) is (bi, bj) = (i >> 2, 4 J + r) and its address is
            // a wave-uniform block-column base (scalar arithmetic) plus one lane offset: two integer operations per load instead of the general index arithmetic
            // (transposition test, two divisions by the block size) that cost ~2 us of issue for the 20 loads of a wave.  Rows behind the system (last block row) are clamped.
            const int i = min(16 * sI[s] + c, n - 1);
            const int lane_off = ((i >> 2) << 4) + ((i & 3) << 2) + rho;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int bj = 4 * sJ[s] + r;
                const int bjc = bj < NBR ? bj : 0;                        // (a column behind the system is never used)
                U[s][r] = Sc[(size_t)((bjc * NBR - bjc * (bjc - 1) / 2 - bjc) << 4) + lane_off];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) U[s][r] = sI[s] >= 0 ? mf_raw(Sc, n, 16 * sI[s] + c, 16 * sJ[s] + rho + 4 * r) : 0.0;
        }
    }
}
// side(excl): work of the caller that does not feed the factorisation (cost at x, gradient tolerance, landmark diagonal, L2 warm-up), run by every wave but
// `excl` while that wave factors a diagonal tile — they would wait at the barrier otherwise; side0(): the excluded wave's share, one step later.
// Schedule per block column k (two workgroup barriers A, B per 16 pivots):
//   wave k+1 (owner of diagonal tile k+1 AND of tile (k+1, k)):  panel of (k+1, k)  ->  its diagonal tile -= V D V^T straight from its registers  ->  B  ->
//                                                                 factor diagonal tile k+1, publish W_k+1  ->  A
//   every other wave:                                             its panel tile of column k  ->  B  ->  trailing update of its tiles with panel k  ->  A
// so the diagonal chain (the critical path) never waits for the other waves' panels or updates: they run beside it.
template <class Side, class Side0>
__device__ __forceinline__ bool ldlt_mf16(mf_d4 (&U)[MF_SLOTS], const uint8_t* plan, int n, double mu, const double* v_s, const double* v_d, const double* rhs, const MfLds& m, int* s_fail, Side side, Side0 side0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = mf_wave(), c = lane & 15, rho = lane >> 4;      // wave in an SGPR: the slot tests below are scalar branches
    const int NB = (n + 16) >> 4, IB = n >> 4, c0 = n & 15;
    int sI[MF_SLOTS], sJ[MF_SLOTS];
    mf_slots(plan, wave, NB, sI, sJ);
    TS(21);
#pragma unroll
    for (int s = 0; s < MF_SLOTS; ++s) {
        if (sI[s] > sJ[s]) {          // a tile below the diagonal (column block J < IB: every column belongs to the system): s_i s_j a_ij; in the last block row the
            const int i = 16 * sI[s] + c;      // right-hand-side row n reads s_j rhs_j and the padding rows behind it are zero — by selection, not by branch
            const double si = v_s[min(i, n - 1)];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * sJ[s] + rho + 4 * r;
                const double sj = v_s[j];
                const double body = si * sj * U[s][r];
                U[s][r] = i < n ? body : (i == n ? sj * rhs[j] : 0.0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) U[s][r] = sI[s] >= 0 ? mf_finish(U[s][r], v_s, v_d, rhs, n, mu, 16 * sI[s] + c, 16 * sJ[s] + rho + 4 * r) : 0.0;
        }
    }
    TS(22);
    for (int i = tid; i < 16 * NB; i += SOL_THREADS) m.yv[i] = 0.0;
    if (tid == 0) *s_fail = 0;
#ifdef BE_SOLVE_TS
    if (tid == 0) { be_dbg_ts[19] = 0; be_dbg_ts[20] = 0; be_dbg_ts[32] = 0; be_dbg_ts[33] = 0; be_dbg_ts[34] = 0; be_dbg_ts[35] = 0; }
#endif
    lds_barrier();
    TS(4);
    const int k_side = -1; // (NB >= 9 ? 6 : -1: the side work inside the loop costs 70 spilled VGPRs)                 // the step whose slack takes the side work (small systems: beside the first diagonal tile)
    if (wave == 0) { mf_diag(U[0], m, 0, NB, n, s_fail); if (k_side < 0) side0(); }
    else if (k_side < 0) side(0);
    lds_barrier();                                       // A: W_0 published
#ifdef BE_SOLVE_TS
    long long t_prev = wall_clock64(), acc_panel = 0, acc_update = 0;
    if (tid == 0) be_dbg_ts[16] = t_prev - be_dbg_ts[4];      // load + first diagonal tile
#endif
    // one panel tile: V = D^-1 W_k U, kept in the registers, stored as the factor's fragment, row n of L (the last block row) copied out as y
    auto panel = [&](mf_d4& T, int I, int k) -> mf_d4 {      // returns W_k U (the panel BEFORE the division by D): -(W U) is what d V is, without waiting for D
        const mf_d4 wf = *reinterpret_cast<const mf_d4*>(m.gat + 768 + lane * 4);      // W_k[c][rho + 4 q], q = 0..3: the panel-order image (mf_diag_factor)
        double ivr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ivr[q] = m.iv[16 * k + rho + 4 * q];
        mf_d4 Y = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int q = 0; q < 4; ++q) Y = mf_mfma(wf[q], T[q], Y);
        const mf_d4 Y0 = Y;
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[r] *= ivr[r];
        T = Y;
        *reinterpret_cast<mf_d4*>(m.Tl + (size_t)mf_tix(I, k, NB) * 256 + lane * 4) = Y;
        if (I == IB && c == c0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) m.yv[16 * k + rho + 4 * r] = Y[r];      // row n of L: y
        }
        return Y0;
    };
    for (int k = 0; k + 1 < NB; ++k) {
        if (wave == k + 1) {
            // ---- the diagonal chain: slot 1 of this wave is tile (k+1, k) ----
            __builtin_amdgcn_s_setprio(3);
#ifdef BE_SOLVE_TS
            const long long t_c0 = wall_clock64();
#endif
            const mf_d4 Y0 = panel(U[1], k + 1, k);
#ifdef BE_SOLVE_TS
            __builtin_amdgcn_s_waitcnt(0xc07f); const long long t_c1 = wall_clock64();
#endif
            // own diagonal tile -= V D V^T with D V = W U = Y0 straight from the panel's accumulator: the chain neither reads D (an LDS round trip behind barrier A, 0.19 us
            // per step by the stamps) nor multiplies by it
#pragma unroll
            for (int q = 0; q < 4; ++q) U[0] = mf_mfma(U[1][q], -Y0[q], U[0]);
#ifdef BE_SOLVE_TS
            asm volatile("s_nop 15\n\ts_nop 3\n\tv_mov_b64 %0, %0" : "+v"(U[0][0])); const long long t_c2 = wall_clock64();      // the last MFMA's result has landed
#endif
            if (k + 2 < NB) lds_barrier();               // B (the last column has one tile below the diagonal, this wave's: nobody waits for a panel there)
#ifdef BE_SOLVE_TS
            const long long t_u = wall_clock64();
            if (lane == 0) { be_dbg_ts[32] += t_c0 - t_prev; be_dbg_ts[33] += t_c1 - t_c0; be_dbg_ts[34] += t_c2 - t_c1; be_dbg_ts[35] += t_u - t_c2; }
#endif
            mf_diag(U[0], m, k + 1, NB, n, s_fail);
#ifdef BE_SOLVE_TS
            if (lane == 0) { be_dbg_ts[19] += wall_clock64() - t_u; be_dbg_ts[20] += t_u - t_prev; }      // diagonal tiles; the owner's panel + update before them
#endif
            lds_barrier();                               // A
        } else {
            // ---- everybody else: panel tile of column k, then the trailing update ----
#pragma unroll
            for (int s = 1; s < MF_SLOTS; ++s) if (sJ[s] == k) panel(U[s], sI[s], k);
            if (k + 2 < NB) lds_barrier();               // B
#ifdef BE_SOLVE_TS
            { const long long t = wall_clock64(); acc_panel += t - t_prev; t_prev = t; }
#endif
            double dk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) dk[q] = -m.dv[16 * k + rho + 4 * q];
#pragma unroll
            for (int s = 0; s < MF_SLOTS; ++s) if (sJ[s] > k) {
                const mf_d4 a = *reinterpret_cast<const mf_d4*>(m.Tl + (size_t)mf_tix(sJ[s], k, NB) * 256 + lane * 4);
                const mf_d4 b = *reinterpret_cast<const mf_d4*>(m.Tl + (size_t)mf_tix(sI[s], k, NB) * 256 + lane * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) U[s] = mf_mfma(a[q], dk[q] * b[q], U[s]);
            }
            if (k == k_side) side(k + 1);
            if (k == k_side + 1 && k_side >= 0 && wave == k_side + 1) side0();
            lds_barrier();                               // A
#ifdef BE_SOLVE_TS
            { const long long t = wall_clock64(); acc_update += t - t_prev; t_prev = t; }
#endif
        }
    }
    // pivots of the n x n system must be positive and finite: one look at D behind the last tile (it was an LDS round trip on every diagonal tile's path)
    if (tid < n) { const double d = m.dv[tid]; if (!(d > 0.0) || !isfinite(d)) *s_fail = 1; }
    lds_barrier();
    TS(5);
#ifdef BE_SOLVE_TS
    if (tid == 0) { be_dbg_ts[17] = acc_panel; be_dbg_ts[18] = acc_update; }
#endif
    return *s_fail == 0;
}
// sum over the four 16-lane rows of a wave (lanes c, 16 + c, 32 + c, 48 + c), the result in every lane, on the VALU: v_permlane32_swap folds the upper half onto the lower,
// v_permlane16_swap the odd rows onto the even ones (gfx950) — (r0 + r2) + (r1 + r3) — instead of a round trip through the wave's gather buffer (write, wave sync, four reads,
// wave sync).  Round 5: the two such sums per tile and step were the back substitution's "known yield" of the earlier rounds.
#ifndef MF_BS_PERMLANE
#define MF_BS_PERMLANE 1
#endif
typedef unsigned mf_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double mf_rows4_sum(double v) {
    mf_u2 a = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    mf_u2 b = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    const double t = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    a = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(t), (unsigned)__double2loint(t), false, false);
    b = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(t), (unsigned)__double2hiint(t), false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// block back substitution L^T x = y on the tiles: per block row i (last to first) x_i = W_i^T y_i (every wave that needs it forms it itself: 4 FMAs per lane and
// one exchange through its gather buffer), then every tile (i, k) of that block row takes its share out of y_k.  One barrier per 16 unknowns.
__device__ __forceinline__ void bs_mf16(const uint8_t* plan, int n, const MfLds& m, double* v_x) {
    const int tid = threadIdx.x, lane = tid & 63, wave = mf_wave(), c = lane & 15, rho = lane >> 4;
    const int NB = (n + 16) >> 4, IB = n >> 4, c0 = n & 15;
    int sI[MF_SLOTS], sJ[MF_SLOTS];
    mf_slots(plan, wave, NB, sI, sJ);
    double* gat = m.gat + wave * 64;
    for (int i = NB - 1; i >= 0; --i) {
        bool need = wave == 0;
#pragma unroll
        for (int s = 1; s < MF_SLOTS; ++s) need = need || sI[s] == i;
        if (need) {
            const mf_d4 Wi = *reinterpret_cast<const mf_d4*>(m.Tl + (size_t)mf_tix(i, i, NB) * 256 + lane * 4);
            double part = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) part = __builtin_fma(Wi[r], m.yv[16 * i + rho + 4 * r], part);
            double xi = MF_BS_PERMLANE ? mf_rows4_sum(part) : 0.0;                 // (W_i^T y_i)[c], in every 16-lane row
            if (!MF_BS_PERMLANE) {
                gat[lane] = part;
                wave_lds_sync();
                xi = (gat[c] + gat[16 + c]) + (gat[32 + c] + gat[48 + c]);
                wave_lds_sync();
            }
            if (i == IB && c >= c0) xi = 0.0;                                     // the right-hand-side row and the padding are not unknowns
            if (wave == 0 && rho == 0 && 16 * i + c < n) v_x[16 * i + c] = xi;
#pragma unroll
            for (int s = 1; s < MF_SLOTS; ++s) if (sI[s] == i) {
                // y_k -= L_ik^T x_i = V x_i with V = the stored fragment (V[R][C], R = k-local row): lane (R = c, rho) sums its columns C = rho + 4 q
                const double* V = m.Tl + (size_t)mf_tix(i, sJ[s], NB) * 256;
                gat[lane] = xi;                                                   // x_i[c] -> readable by index
                wave_lds_sync();
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int C = rho + 4 * q; acc = __builtin_fma(V[((((c & 3) << 4) + C) << 2) + (c >> 2)], gat[C], acc); }
                wave_lds_sync();
                double t = MF_BS_PERMLANE ? mf_rows4_sum(acc) : 0.0;
                if (!MF_BS_PERMLANE) {
                    gat[lane] = acc;
                    wave_lds_sync();
                    t = (gat[c] + gat[16 + c]) + (gat[32 + c] + gat[48 + c]);
                    wave_lds_sync();
                }
                if (rho == 0) m.yv[16 * sJ[s] + c] -= t;
            }
        }
        lds_barrier();
    }
}

// (Round 5: the same back substitution with every matrix-vector product on the matrix cores — x_i replicated over the B operand's columns, results chained register to
//  register, one producer wave per x_i — was built and measured: 15.4 us against 8.4 us for bs_mf16.  A product is four chained MFMAs whose result must reach a VALU /
//  LDS consumer before the next stage can start; that hand-over costs more than the two LDS reductions it replaces.  Removed; scripts/dbg notes in DESIGN.md 4.)
// back-substitution helper: applies ROWS consecutive pivots (kt, kt-1, ...) that all lie in 64-lane segment SEG of x
template <int SEG, int ROWS>
__device__ __forceinline__ void bs_chunk(const double* Lm, int kt, int lane, double& x0, double& x1, double& x2) {
    double cr[ROWS][3];
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
        const int k = kt - u;
        const double* row = Lm + tri(k, 0) + lane;      // reads past the end of row k are masked below (they stay inside the LDS allocation)
        cr[u][0] = row[0];
        if (SEG >= 1) cr[u][1] = row[64];
        if (SEG >= 2) cr[u][2] = row[128];
        cr[u][SEG] = lane < k - 64 * SEG ? cr[u][SEG] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
        const int k = kt - u;
        const double xs = SEG == 0 ? x0 : (SEG == 1 ? x1 : x2);
        const int lo = __builtin_amdgcn_readlane(__double2loint(xs), k - 64 * SEG), hi = __builtin_amdgcn_readlane(__double2hiint(xs), k - 64 * SEG);
        const double xk = __hiloint2double(hi, lo);
        x0 = __builtin_fma(-cr[u][0], xk, x0);
        if (SEG >= 1) x1 = __builtin_fma(-cr[u][1], xk, x1);
        if (SEG >= 2) x2 = __builtin_fma(-cr[u][2], xk, x2);
    }
}
// Software-pipelined form: the rows of stage c+1 (R rows) are requested from LDS before the R dependent pivots of stage c run, so the LDS round trip
// (~460 cycles per 24 loads, measured: scripts/dbg/bs_bench.hip) passes behind the pivot chain (~41 cycles per pivot).  SEG compile-time, two explicit
// buffers.  Same operations per entry as bs_chunk.
template <int SEG, int R>
__device__ __forceinline__ void bs_ld(const double* Lm, int kt, int lane, double (&cr)[R][3]) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int k = kt - u;
        const double* row = Lm + tri(k, 0) + lane;
        cr[u][0] = row[0];
        if (SEG >= 1) cr[u][1] = row[64];
        if (SEG >= 2) cr[u][2] = row[128];              // raw: the mask of the pivot's own segment is applied in bs_ap (a use here would wait for the load)
    }
}
template <int SEG, int R>
__device__ __forceinline__ void bs_ap(const double (&cr)[R][3], int kt, int lane, double& x0, double& x1, double& x2) {
    double m[R];
#pragma unroll
    for (int u = 0; u < R; ++u) m[u] = lane < kt - u - 64 * SEG ? cr[u][SEG] : 0.0;      // entries at / right of the pivot are not part of row k
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const double xs = SEG == 0 ? x0 : (SEG == 1 ? x1 : x2);
        const double xk = lane_bcast(xs, kt - u - 64 * SEG);
        x0 = __builtin_fma(-(SEG == 0 ? m[u] : cr[u][0]), xk, x0);
        if (SEG >= 1) x1 = __builtin_fma(-(SEG == 1 ? m[u] : cr[u][1]), xk, x1);
        if (SEG >= 2) x2 = __builtin_fma(-m[u], xk, x2);
    }
}
// rows k .. 64 * SEG of segment SEG (k + 1 and 64 multiples of R); returns the top row of the next segment
template <int SEG, int R>
__device__ __forceinline__ int bs_segment(const double* Lm, int k, int lane, double& x0, double& x1, double& x2) {
    const int kend = 64 * SEG + R - 1;                   // top row of the segment's last stage
    double ca[R][3], cb[R][3];
    bs_ld<SEG, R>(Lm, k, lane, ca);
    for (; k >= kend; k -= 2 * R) {
        if (k - R >= kend) bs_ld<SEG, R>(Lm, k - R, lane, cb);
        __builtin_amdgcn_sched_barrier(0);               // the scheduler otherwise sinks the requests to their uses (it minimises live ranges under the 128-VGPR cap)
        bs_ap<SEG, R>(ca, k, lane, x0, x1, x2);
        __builtin_amdgcn_sched_barrier(0);
        if (k - R >= kend) {
            if (k - 2 * R >= kend) bs_ld<SEG, R>(Lm, k - 2 * R, lane, ca);
            __builtin_amdgcn_sched_barrier(0);
            bs_ap<SEG, R>(cb, k - R, lane, x0, x1, x2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return 64 * SEG - 1;
}
template <int ROWS>
__device__ __forceinline__ void bs_rows(const double* Lm, int kt, int lane, double& x0, double& x1, double& x2) {
    const int seg = kt >> 6;                             // uniform
    if (seg == 0) bs_chunk<0, ROWS>(Lm, kt, lane, x0, x1, x2);
    else if (seg == 1) bs_chunk<1, ROWS>(Lm, kt, lane, x0, x1, x2);
    else bs_chunk<2, ROWS>(Lm, kt, lane, x0, x1, x2);
}

// The body lives in be_solve_body.inc and is included textually: called through a function taking `const BeSolveArgs&` the single-window kernel lost 4 %
// (99.2 vs 95.2 us; the by-value kernel arguments stopped being treated as invariant scalar loads).
template <int NSLOT, bool MF16>
__global__ __launch_bounds__(SOL_THREADS) void be_solve_kernel(BeSolveArgs a, int spec) {
    constexpr bool SHARD = false;
#include "be_solve_body.inc"
}
template <int NSLOT, bool MF16>
__global__ __launch_bounds__(SOL_THREADS) void be_solve_batch_kernel(const BeSolveArgs* __restrict__ tab, int spec) {      // one workgroup per window
    const BeSolveArgs& a = tab[blockIdx.x];
    constexpr bool SHARD = false;
#include "be_solve_body.inc"
}
// one window sharded by landmark (be_kernels.h BeShard): the same step on every rank, landmark sums from the exchanged forms, the rank's own landmarks moved
__global__ __launch_bounds__(SOL_THREADS) void be_solve_shard_kernel(BeSolveArgs a, int spec) {
    constexpr int NSLOT = 1; constexpr bool MF16 = true, SHARD = true;
#include "be_solve_body.inc"
}

static size_t solve_smem(int n) {      // the generic form: packed factor | vectors | panel buffers
    const size_t tri = (size_t)n * (n + 1) / 2;
    return (tri + 9 * (size_t)n + 72 + 80 + 8 + 1536) * sizeof(double);
}
// MF16: the tiles (which double as the later phases' scratch: >= 3072 doubles) | vectors | per-wave gather buffers | D, 1 / D, y
static size_t solve_smem_mf16(int n) {
    const size_t NB = ((size_t)n + 16) >> 4, nt = NB * (NB + 1) / 2;
    return (std::max<size_t>(nt * 256, 3072) + 9 * (size_t)n + 72 + 80 + 8 + 1024 + 3 * 16 * NB) * sizeof(double);
}
// MF16 tile plan (ldlt_mf16): wave j < NB owns diagonal tile j (slot 0) and the tile left of it, (j, j-1) (slot 1).  Any other off-diagonal tile (I, J) may go to wave 0 (its diagonal tile is factored before the
// loop), to a wave >= NB, or to a wave w with J < w: then the wave that factors diagonal tile k+1 behind the update of step k has no other active tile (its
// off-diagonal tiles lie in columns <= k).  Every column's tiles go to DIFFERENT waves (one panel tile per wave and step), the waves whose eligibility ends
// first are used first (earliest deadline first: wave w < NB is useless from column w on).  Feasible for every n <= 175 (checked exhaustively).
bool be_mf16_plan(int n, uint8_t* plan /* [16][4] */, bool check_solve_lds) {
    const int NB = (n + 16) >> 4, W = SOL_THREADS / 64;
    if (NB > MF_MAXNB || (check_solve_lds && solve_smem_mf16(n) > 160 * 1024 - 512)) return false;
    int load[SOL_THREADS / 64] = { 0 };
    std::memset(plan, 0xFF, (size_t)W * 4);
    for (int j = 1; j < NB; ++j) plan[j * 4 + load[j]++] = (uint8_t)((j << 4) | (j - 1));      // slot 1 of wave j: tile (j, j-1), the one its diagonal tile waits for
    for (int J = 0; J < NB - 1; ++J) {
        int el[SOL_THREADS / 64], ne = 0;
        for (int w = 0; w < W; ++w) if ((w == 0 || w >= NB || J < w) && w != J + 1 && load[w] < 4) el[ne++] = w;      // (wave J+1 has its tile of this column)
        auto key = [&](int w) { return ((1 <= w && w < NB) ? w - 1 : NB) * 1024 + load[w] * 32 + w; };
        std::sort(el, el + ne, [&](int x, int y) { return key(x) < key(y); });
        if (ne < NB - 2 - J) return false;
        for (int t = 0, I = J + 2; I < NB; ++I, ++t) { const int w = el[t]; plan[w * 4 + load[w]++] = (uint8_t)((I << 4) | J); }
    }
    return true;
}

// Two forms of the factorisation ship: MF16 (every system whose tiles fit: n <= 175, i.e. every window the estimator builds — 11 frames x 15 with constant
// extrinsics and td) and the generic 4-wide panel form (any n <= BE_MAX_STATE; dv_debug_set "ldl_generic" selects it for A/B runs and agreement tests).
int be_launch_solve(const BeSolveArgs& a, int spec, hipStream_t s) {
    static DevOnce once;
    if (once.run([] {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_solve_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_smem(BE_MAX_STATE)) != hipSuccess) return 1;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_solve_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_smem(BE_MAX_STATE)) != hipSuccess) return 1;
            return hipFuncSetAttribute(reinterpret_cast<const void*>(be_solve_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess ? 1 : 0; })) return -1;      // 160 KB per workgroup less the kernel's static arrays
    const int nbr = (a.dims.nstate + 3) / 4;
    if (a.sh.on) {
        if (!a.ldl_mf16 || solve_smem_mf16(a.dims.nstate) > 160 * 1024 - 1024) return -2;          // (the sharded step exists on the MF16 factorisation only: every window the estimator builds; its static LDS is 0.7 KB)
        static DevOnce once_s;
        if (once_s.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(be_solve_shard_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess ? 1 : 0; })) return -1;
        hipLaunchKernelGGL(be_solve_shard_kernel, dim3(1), dim3(SOL_THREADS), solve_smem_mf16(a.dims.nstate), s, a, spec);
        return 0;
    }
    if (a.ldl_mf16) hipLaunchKernelGGL((be_solve_kernel<1, true>), dim3(1), dim3(SOL_THREADS), solve_smem_mf16(a.dims.nstate), s, a, spec);
    else if (nbr * (nbr + 1) / 2 <= SOL_THREADS) hipLaunchKernelGGL((be_solve_kernel<1, false>), dim3(1), dim3(SOL_THREADS), solve_smem(a.dims.nstate), s, a, spec);
    else hipLaunchKernelGGL((be_solve_kernel<2, false>), dim3(1), dim3(SOL_THREADS), solve_smem(a.dims.nstate), s, a, spec);
    return 0;
}

// batched: every window of the table on the MF16 form (checked by the caller); smem for the largest n
int be_launch_solve_batch(const BeSolveArgs* tab_dev, int n_win, int max_n, int spec, hipStream_t s) {
    static DevOnce once;
    if (once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(be_solve_batch_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess ? 1 : 0; })) return -1;
    hipLaunchKernelGGL((be_solve_batch_kernel<1, true>), dim3(n_win), dim3(SOL_THREADS), solve_smem_mf16(max_n), s, tab_dev, spec);
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void be_accept_kernel(BeSolveArgs a) { be_accept_body(a); }
__global__ __launch_bounds__(256) void be_accept_batch_kernel(const BeSolveArgs* __restrict__ tab) { be_accept_body(tab[blockIdx.x]); }
void be_launch_accept(const BeSolveArgs& a, hipStream_t s) { hipLaunchKernelGGL(be_accept_kernel, dim3(1), dim3(256), 0, s, a); }
void be_launch_accept_batch(const BeSolveArgs* tab_dev, int n_win, hipStream_t s) { hipLaunchKernelGGL(be_accept_batch_kernel, dim3(n_win), dim3(256), 0, s, tab_dev); }

int be_solve_prepare() {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(be_reduce_kernel)) != hipSuccess) return -1;
    static DevOnce once;
    return once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(be_solve_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess ? 1 : 0; }) ? -1 : 0;
}
