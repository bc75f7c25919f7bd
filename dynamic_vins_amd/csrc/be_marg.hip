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
#include "be_mf16.h"
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
#define LM_THREADS 512

// ---------------------------------------------------------------------------------------------------------------------------
// The landmark part of the system, A_lm = sum_l (J_l^T J_l - w_l w_l^T / h_l),  b_lm = sum_l (J_l^T r_l - w_l g_l / h_l), is NOT summed from dense per-landmark
// slabs (rounds 1-2: 17.9 MB written, 18.5 MB read back per frame).  Every residual block of a landmark anchored in frame 0 touches the same few column groups:
//   common   pose of the anchor frame (6) | ex0 (6) | ex1 (6) | td (1)       19 "local" columns, touched by every block
//   frame j  pose of the observing frame (6)                                  touched by the (at most two) blocks of the landmark with fj = j
// so sum_l J_l^T J_l is block-structured — (common x common), (pose_j x common), (pose_j x pose_j), nothing between two different observing frames — and only
// the rank-one terms w_l w_l^T / h_l are dense.  be_marg_lm (one workgroup per MG_CH landmarks) therefore writes
//   part[chunk][MG_PART]   the chunk's share of the structured blocks (and of J^T r), summed over its residual blocks in block order,
//   W[l][mg_wstride(D)]    w_l | g_l | 1 / h_l (0 if h_l <= 1e-8, the reference's pseudo-inverse),
// be_marg_sum forms  - W^T diag(hinv) W  with one 16 x 16 tile per workgroup on the f64 matrix cores (v_mfma_f64_16x16x4_f64; row D of the extended product is the
// right-hand side) and adds the chunks' parts; be_marg_finish puts the structured sums at their dense positions.  Every sum has one fixed order.  ~1 MB per frame
// instead of 36 MB.
#define MG_CH 4                                  // landmarks per workgroup of be_marg_lm
#define MG_NC 19                                 // common local columns
#define MG_CC (MG_NC * (MG_NC + 1) / 2)          // 190: lower triangle common x common, then 19: common part of J^T r
#define MG_GRP (6 * MG_NC + 21 + 6)              // 141 per observing frame: pose_j x common (114) | pose_j x pose_j lower (21) | pose_j part of J^T r (6)
#define MG_G0 (MG_CC + MG_NC)                    // 209
#define MG_PART (MG_G0 + BE_NF * MG_GRP)         // 1760
#define MG_MAXF (MG_CH * BE_MAX_OBS_FACTORS)
static_assert(MG_PART <= 2 * (MG_THREADS - 64), "be_marg_finish: two structured entries per worker thread");
static_assert(MG_MAXF + 2 <= LM_THREADS - 64, "one thread of be_marg_lm per residual block of the chunk");

typedef double mg_d4 __attribute__((ext_vector_type(4)));
// local common column of (slot, comp); -1: the pose of an observing frame (or a dim no residual block touches)
__device__ __forceinline__ int mg_lc(int slot, int comp, int anchor) { return slot == anchor ? comp : slot == BE_NF ? 6 + comp : slot == BE_NF + 1 ? 12 + comp : slot == BE_NF + 2 ? 18 : -1; }
// offset of row 0 of a common local column inside a factor record (54): r2 | Ji 12 | Jj 12 | Jex0 12 | Jex1 12 | Jl 2 | Jtd 2; row 1 sits mg_cstep further
__device__ __forceinline__ int mg_coff(int lc) { return lc < 6 ? 2 + lc : lc < 12 ? 20 + lc : lc < 18 ? 26 + lc : 52; }
__device__ __forceinline__ int mg_cstep(int lc) { return lc == 18 ? 1 : 6; }
__device__ __forceinline__ void mg_tri(int t, int& hi, int& lo) {      // t = hi (hi + 1) / 2 + lo, lo <= hi
    hi = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (hi * (hi + 1) / 2 > t) --hi;
    while ((hi + 1) * (hi + 2) / 2 <= t) ++hi;
    lo = t - hi * (hi + 1) / 2;
}
// position inside a chunk's part[] of dense entry (i, j), i == D: the right-hand side; -1: structurally zero
__device__ __forceinline__ int mg_pos(int si, int ci, int sj, int cj, bool rhs, int anchor) {
    if (sj < 0 || (!rhs && si < 0)) return -1;
    const int lj = mg_lc(sj, cj, anchor);
    if (rhs) return lj >= 0 ? MG_CC + lj : MG_G0 + sj * MG_GRP + 135 + cj;
    const int li = mg_lc(si, ci, anchor);
    if (li >= 0 && lj >= 0) { const int hi = li > lj ? li : lj, lo = li > lj ? lj : li; return hi * (hi + 1) / 2 + lo; }
    if (li >= 0) return MG_G0 + sj * MG_GRP + cj * MG_NC + li;
    if (lj >= 0) return MG_G0 + si * MG_GRP + ci * MG_NC + lj;
    if (si != sj) return -1;
    const int hi = ci > cj ? ci : cj, lo = ci > cj ? cj : ci;
    return MG_G0 + si * MG_GRP + 114 + hi * (hi + 1) / 2 + lo;
}

// (slot, component) of dim i from the block offsets of the plan (BeMargArgs::pose_dim / ex_dim / td_dim); slot -1: speed-bias or out of range
__device__ __forceinline__ void mg_dim(const BeMargArgs& a, int i, int& slot, int& comp) {
    slot = -1; comp = 0;
#pragma unroll
    for (int k = 0; k < BE_NF; ++k) { const int d0 = a.pose_dim[k]; if (d0 >= 0 && i >= d0 && i < d0 + 6) { slot = k; comp = i - d0; } }
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int d0 = a.ex_dim[k]; if (d0 >= 0 && i >= d0 && i < d0 + 6) { slot = BE_NF + k; comp = i - d0; } }
    if (a.td_dim >= 0 && i == a.td_dim) { slot = BE_NF + 2; comp = 0; }
}
// W rows: w_l (D) | g_l | 1 / h_l, padded to whole 128-byte lines (the operand loads of be_marg_sum then touch four lines per instruction)
__host__ __device__ __forceinline__ int mg_wstride(int D) { return (D + 2 + 15) & ~15; }
// record offset (row 0) and row step of MFMA column c of a residual block: 0..18 common | 19 residual | 20 d/d(inverse depth) | zero columns above
__device__ __forceinline__ int mg_xoff(int c) { return c < MG_NC ? mg_coff(c) : c == MG_NC ? 0 : 50; }
__device__ __forceinline__ int mg_xstep(int c) { return c < MG_NC ? mg_cstep(c) : 1; }

__device__ __forceinline__ void be_marg_lm_body(const BeMargArgs& a, const int bx) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = a.D, D1 = mg_wstride(D), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = (a.nlm + MG_CH - 1) / MG_CH, chunk = bx;
    double* Jb = sm;                                   // (MG_MAXF + 2) x 54: the chunk's residual blocks, landmark after landmark, then zero rows
    __shared__ FrameGeom fg[BE_NF];
    __shared__ m33 ric[2];
    __shared__ d3 tic[2];
    const BeState* st = a.x;
    if (chunk == nchunk) {
        // extra block (present when the window has an IMU factor to marginalize): factor (0,1) evaluated at the solved state and whitened, beside the landmark
        // blocks.  be_marg_finish used to do this itself: a serial raw evaluation on one lane in the middle of a single-workgroup kernel.
        double* Jraw = sm; double* rr = sm + 900;
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
    __shared__ short s_slot[BE_MAX_PRIOR], s_comp[BE_MAX_PRIOR];      // dim -> (slot, component)
    __shared__ short s_cdim[MG_NC];                                    // common local column -> dim (-1: not in the system)
    __shared__ int s_first[MG_CH], s_cnt[MG_CH];
    __shared__ short s_fidx[MG_CH][BE_NF][2];                          // [landmark][observing frame][left, right] -> residual block of the chunk, -1 = none
    const int l0 = chunk * MG_CH, nk = min(MG_CH, a.nlm - l0), anchor = a.anchor;
    for (int i = tid; i < D; i += LM_THREADS) { s_slot[i] = (short)a.dim_slot[i]; s_comp[i] = (short)a.dim_comp[i]; }
    if (tid < MG_NC) s_cdim[tid] = -1;
    for (int i = tid; i < MG_CH * BE_NF * 2; i += LM_THREADS) (&s_fidx[0][0][0])[i] = -1;
    if (tid < 64) be_frame_geom_dev(st, a.nframes, fg, ric, tic, tid);
    if (tid >= 64 && tid < 64 + nk) { const BeLm L = a.lm[a.lm_sel ? a.lm_sel[l0 + tid - 64] : l0 + tid - 64]; s_first[tid - 64] = L.first; s_cnt[tid - 64] = L.count; }
    __syncthreads();
    MTS(1);
    int off[MG_CH + 1];
    off[0] = 0;
#pragma unroll
    for (int k = 0; k < MG_CH; ++k) off[k + 1] = off[k] + (k < nk ? s_cnt[k] : 0);
    const int nt = off[MG_CH];
    if (tid < nt) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < MG_CH; ++q) k += tid >= off[q] ? 1 : 0;
        const BeFactor f = a.fac[s_first[k] + tid - off[k]];
        double* o = Jb + tid * 54;
        proj_factor<true, true>(f, fg[f.fi], fg[f.fj], ric[0], tic[0], ric[1], tic[1], st->inv_depth[f.lm], st->td, o, o + 2, o + 14, o + 50, o + 26, o + 38, o + 52);
        if (f.kind == 0) for (int q = 0; q < 12; ++q) o[38 + q] = 0.0;
        double rho0, sc;
        huber1(o[0] * o[0] + o[1] * o[1], rho0, sc);
        for (int q = 0; q < 54; ++q) o[q] *= sc;
        if (f.kind != 2) s_fidx[k][f.fj][f.kind] = (short)tid;
    } else if (tid < nt + 2) {                         // two zero blocks: the matrix-core loop below runs over whole groups of four rows
        double* o = Jb + tid * 54;
        for (int q = 0; q < 54; ++q) o[q] = 0.0;
    } else if (tid >= LM_THREADS - 64) {
        const int i = tid - (LM_THREADS - 64);
        for (int d = i; d < D; d += 64) { const int sl = s_slot[d]; const int lc = sl >= 0 ? mg_lc(sl, s_comp[d], anchor) : -1; if (lc >= 0) s_cdim[lc] = (short)d; }
    }
    __syncthreads();
    MTS(2);
    double* part = a.part + (size_t)chunk * MG_PART;
    if (wave < 5) {
        // (common | r | Jl)^T (common | r | Jl) over every residual-block row of the chunk (waves 0-2: the lower tiles of the 32 x 32 product -> common x common and
        // the common part of J^T r), and (common | r | Jl)^T (Jl masked per landmark) (waves 3-4 -> w_l on the common columns, g_l, h_l): 16 x 16 x 4 fp64 MFMAs
        // over groups of four rows, row order = block order.  Lane l feeds row 4 s + (l >> 4), column l & 15 of its operand block.
        const int Ib = wave == 0 ? 0 : wave == 3 ? 0 : 1, Jb_ = wave == 2 ? 1 : 0;
        const bool wl = wave >= 3;
        const int cl = lane & 15, kq = lane >> 4;
        const int ca = Ib * 16 + cl, cbk = Jb_ * 16 + cl;
        // row rho = 4 s + kq -> block t = rho >> 1 = 2 s + (kq >> 1), row r = kq & 1
        const bool za = ca > MG_NC + 1, zb = wl ? cl >= nk : cbk > MG_NC + 1;      // zero columns: read column 0 instead and select (a predicated LDS read is a branch)
        const int base_a = (kq >> 1) * 54 + (za ? 0 : mg_xoff(ca) + (kq & 1) * mg_xstep(ca));
        const int base_b = (kq >> 1) * 54 + (wl ? 50 + (kq & 1) : zb ? 0 : mg_xoff(cbk) + (kq & 1) * mg_xstep(cbk));
        int lo = 0, hi = 0;                            // waves 3-4: the blocks of landmark (l & 15)
        if (wl && !zb) {
#pragma unroll
            for (int k = 0; k < MG_CH; ++k) if (cl == k) { lo = off[k]; hi = off[k + 1]; }
        }
        if (!wl && !zb) hi = 1 << 20;
        lo -= kq >> 1; hi -= kq >> 1;                  // compared against 2 s below
        mg_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
        const int steps = __builtin_amdgcn_readfirstlane((2 * nt + 3) >> 2);      // uniform (nt comes out of LDS: the compiler cannot know)
        const double* pa = Jb + base_a; const double* pb = Jb + base_b;
        MTS(20);
        for (int s0 = 0; s0 < steps; s0 += 8) {            // eight groups of four rows per trip: sixteen LDS reads in flight, then the eight products
            double xr[8], yr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int s = min(s0 + u, steps - 1); xr[u] = pa[s * 108]; yr[u] = pb[s * 108]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s2 = 2 * (s0 + u);
                const bool keep = (s0 + u < steps) & (s2 >= lo) & (s2 < hi);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((za | (s0 + u >= steps)) ? 0.0 : xr[u], keep ? yr[u] : 0.0, acc, 0, 0, 0);
            }
        }
        MTS(21);
        // result: register r of lane l = (row (l >> 4) + 4 r of block Ib, column l & 15)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = Ib * 16 + kq + 4 * r;
            if (!wl) {
                const int cj = cbk;
                if (ci < MG_NC && cj <= ci) part[ci * (ci + 1) / 2 + cj] = acc[r];
                else if (ci == MG_NC && cj < MG_NC) part[MG_CC + cj] = acc[r];
            } else if (cl < nk) {
                const int l = l0 + cl;
                if (ci < MG_NC) { const int d = s_cdim[ci]; if (d >= 0) a.W[(size_t)l * D1 + d] = acc[r]; }
                else if (ci == MG_NC) a.W[(size_t)l * D1 + D] = acc[r];
                else if (ci == MG_NC + 1) { const double h = acc[r]; a.lm_h[l] = h; a.W[(size_t)l * D1 + D + 1] = h > 1e-8 ? 1.0 / h : 0.0; }      // 1x1 pivot of the inverse depth, clamped like the reference's pseudo-inverse
            }
        }
    }
    MTS(3);
    // w_l on the columns that are not common: the pose of an observing frame (its left, then its right block), zero elsewhere
    for (int e = tid; e < nk * D; e += LM_THREADS) {
        const int k = e / D, i = e - k * D;
        const int slot = s_slot[i], comp = s_comp[i];
        if (slot >= 0 && mg_lc(slot, comp, anchor) >= 0) continue;
        double w = 0.0;
        if (slot >= 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { const int t = s_fidx[k][slot][q]; if (t >= 0) { const double* o = Jb + t * 54; w += o[14 + comp] * o[50]; w += o[20 + comp] * o[51]; } }
        }
        a.W[(size_t)(l0 + k) * D1 + i] = w;
    }
    MTS(4);
    // the chunk's share of the per-frame blocks: pose_j x (common | pose_j | r), summed over landmarks in order, left block before right block
    for (int e = tid; e < BE_NF * MG_GRP; e += LM_THREADS) {
        const int grp = e / MG_GRP, u = e - grp * MG_GRP;
        int oa, ob, sb;
        if (u < 114) { const int cp = u / MG_NC, lc = u - cp * MG_NC; oa = 14 + cp; ob = mg_coff(lc); sb = mg_cstep(lc); }
        else if (u < 135) { int hi, lo; mg_tri(u - 114, hi, lo); oa = 14 + hi; ob = 14 + lo; sb = 6; }
        else { oa = 14 + u - 135; ob = 0; sb = 1; }
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 2 * MG_CH; ++q) { const int t = s_fidx[q >> 1][grp][q & 1]; if (t >= 0) { const double* o = Jb + t * 54; s += o[oa] * o[ob]; s += o[oa + 6] * o[ob + sb]; } }
        part[MG_G0 + e] = s;
    }
    MTS(5);
}
__global__ __launch_bounds__(LM_THREADS) void be_marg_lm_kernel(BeMargArgs a) { be_marg_lm_body(a, blockIdx.x); }
// the marginalizations of a dv_batch group in shared launches (blockIdx.y = member, argument table in HBM; a member without one this frame has D = 0)
__global__ __launch_bounds__(LM_THREADS) void be_marg_lm_batch_kernel(const BeMargArgs* __restrict__ tab) {
    const BeMargArgs a = tab[blockIdx.y];
    if (a.D <= 0 || !(a.nlm > 0 || a.nimu > 0) || (int)blockIdx.x >= (a.nlm + MG_CH - 1) / MG_CH + (a.nimu > 0 ? 1 : 0)) return;
    be_marg_lm_body(a, blockIdx.x);
}

// be_marg_sum, two kinds of workgroups in one launch:
//   tiles    (blockIdx.x < lower tiles of the extended (D + 1) x (D + 1) system)   a.sum = - W^T diag(hinv) W  (dense D x D, both triangles, then the row of the
//            right-hand side).  Wave q takes the q-th quarter of the landmarks — all its operands in flight at once, every load instruction four whole lines — and
//            the quarters are added in order through LDS.
//   reduce   (the workgroups behind them)   a.psum[e] = sum over chunks of part[chunk][e], chunk order; coalesced.
// be_marg_finish adds psum into the dense system at the positions the structure dictates (the scattered 8-byte gathers this kernel did itself at first — 32 per
// lane, each load instruction touching up to 64 lines — kept the texture addresser of every CU busy for 7 us).
#define MG_KSTEPS 16
__device__ __forceinline__ void be_marg_sum_body(const BeMargArgs& a, const int tiles, const int bx) {
    const int D = a.D, Ws = mg_wstride(D), lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int nlm = a.nlm, nchunk = (nlm + MG_CH - 1) / MG_CH;
    if ((int)bx >= tiles) {
        const int e = ((int)bx - tiles) * 256 + threadIdx.x;
        if (e >= MG_PART) return;
        const double* pp = a.part + e;
        double sg = 0.0;
        for (int c0 = 0; c0 < nchunk; c0 += 32) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = pp[(size_t)min(c0 + u, nchunk - 1) * MG_PART];
#pragma unroll
            for (int u = 0; u < 32; ++u) if (c0 + u < nchunk) sg += v[u];
        }
        a.psum[e] = sg;
        return;
    }
    int I, J; mg_tri(bx, I, J);
    const int cl = lane & 15, kq = lane >> 4, ca = I * 16 + cl, cb = J * 16 + cl;
    __shared__ double s_acc[4][4][64];
    MTS(6);
    // acc[i][j] = sum_l W[l][16 I + i] hinv_l W[l][16 J + j], four landmarks per instruction, landmark order inside the quarter.  Columns >= D + 1 of the last
    // block hold hinv and padding: whatever they produce is not stored.
    const int per = (((nlm + 3) >> 2) + 3) & ~3, k_lo = q * per, k_hi = min(nlm, k_lo + per);
    const double* wa = a.W + ca; const double* wb = a.W + cb; const double* wh = a.W + D + 1;
    mg_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
    MTS(22);
    for (int k0 = k_lo; k0 < k_hi; k0 += 4 * MG_KSTEPS) {      // one trip while a quarter holds <= 64 landmarks
        double x[MG_KSTEPS], y[MG_KSTEPS], hv[MG_KSTEPS];
#pragma unroll
        for (int u = 0; u < MG_KSTEPS; ++u) {                  // unconditional loads (clamped row, value selected afterwards): a predicated load is a branch
            const size_t k = (size_t)min(k0 + 4 * u + kq, nlm - 1) * Ws;
            x[u] = wa[k]; y[u] = wb[k]; hv[u] = wh[k];
        }
#pragma unroll
        for (int u = 0; u < MG_KSTEPS; ++u) {
            const bool vk = k0 + 4 * u + kq < k_hi;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vk ? x[u] : 0.0, vk ? y[u] * hv[u] : 0.0, acc, 0, 0, 0);
        }
    }
    MTS(23);
    MTS(7);
#pragma unroll
    for (int r = 0; r < 4; ++r) s_acc[q][r][lane] = acc[r];
    __syncthreads();
    // after the exchange wave q owns result register q of the tile: entry (row (l >> 4) + 4 q of block I, column l & 15 of block J)
    const int i = I * 16 + kq + 4 * q, j = cb;
    if (i <= D && j < D && j <= i) {
        double rk = s_acc[0][q][lane];
        rk += s_acc[1][q][lane]; rk += s_acc[2][q][lane]; rk += s_acc[3][q][lane];
        if (i == D) a.sum[(size_t)D * D + j] = -rk;
        else { a.sum[(size_t)i * D + j] = -rk; a.sum[(size_t)j * D + i] = -rk; }
    }
    MTS(18);
}
__global__ __launch_bounds__(256) void be_marg_sum_kernel(BeMargArgs a, int tiles) { be_marg_sum_body(a, tiles, blockIdx.x); }
__global__ __launch_bounds__(256) void be_marg_sum_batch_kernel(const BeMargArgs* __restrict__ tab) {
    const BeMargArgs a = tab[blockIdx.y];
    if (a.D <= 0 || a.nlm <= 0) return;
    const int NB = (a.D + 1 + 15) / 16, tiles = NB * (NB + 1) / 2;
    if ((int)blockIdx.x >= tiles + (MG_PART + 255) / 256) return;
    be_marg_sum_body(a, tiles, blockIdx.x);
}

// dense home of entry e of the structured sums (the inverse of the layout be_marg_lm writes): (i, j) with i == D for the right-hand side; false if the block is
// not part of the system
__device__ __forceinline__ bool mg_home(const BeMargArgs& a, int e, int& i, int& j) {
    const int D = a.D, anchor = a.anchor;
    auto cdim = [&](int lc) { const int d0 = lc < 6 ? a.pose_dim[anchor] : lc < 12 ? a.ex_dim[0] : lc < 18 ? a.ex_dim[1] : a.td_dim; return d0 < 0 ? -1 : d0 + (lc < 6 ? lc : lc < 12 ? lc - 6 : lc < 18 ? lc - 12 : 0); };
    if (e < MG_CC) { int hi, lo; mg_tri(e, hi, lo); i = cdim(hi); j = cdim(lo); }
    else if (e < MG_G0) { i = D; j = cdim(e - MG_CC); }
    else {
        const int qq = e - MG_G0, grp = qq / MG_GRP, u = qq - grp * MG_GRP;
        int d0 = -1;
#pragma unroll
        for (int k = 0; k < BE_NF; ++k) if (k == grp) d0 = a.pose_dim[k];
        if (d0 < 0 || grp == anchor) return false;
        if (u < 114) { const int cp = u / MG_NC; i = d0 + cp; j = cdim(u - cp * MG_NC); }
        else if (u < 135) { int hi, lo; mg_tri(u - 114, hi, lo); i = d0 + hi; j = d0 + lo; }
        else { i = D; j = d0 + u - 135; }
    }
    return i >= 0 && j >= 0;
}

#define MG_SUM_CHUNKS 1
__device__ __forceinline__ void be_marg_finish_body(const BeMargArgs& a) {
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
    if (a.c0_mode == 2) {                               // side-stream launch: A', b' come back from global memory, then straight to the c0 factorisation
        for (int e = tid; e < n * n; e += MG_THREADS) W2[e] = a.outA[e];
        for (int i = tid; i < n; i += MG_THREADS) yv[i] = a.outb[i];
        __syncthreads();
    } else {
    // The last wave takes no share of the assembly below: it fetches the whitened IMU factor (0,1) that the extra block of be_marg_lm has prepared (the raw
    // evaluation on ONE lane and the whitening used to be a 9 us phase of this kernel, behind the prior).
    const int MGW = MG_THREADS - 64;                      // worker threads of the assembly
    const bool imu_wave = tid >= MGW;
    double* Jw = imu_ws + 450; double* rr = imu_ws + 900;
    if (imu_wave && a.nimu > 0) {
        const int lane = tid - MGW;
        for (int i = lane; i < 465; i += 64) { const double v = a.imu_w[i]; if (i < 450) Jw[i] = v; else rr[15 + i - 450] = v; }
    }
    // the structured block sums (be_marg_sum's reduce workgroups): requested first, added at their dense homes once the rank term is in LDS
    double pv[2] = { 0.0, 0.0 }; int pi_[2] = { -1, -1 }, pj_[2] = { 0, 0 };
    if (a.nlm > 0 && !imu_wave) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int e = tid + u * MGW; if (e < MG_PART && mg_home(a, e, pi_[u], pj_[u])) pv[u] = a.psum[e]; else pi_[u] = -1; }
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
    if (a.nlm > 0) {                                    // no two entries share a home: plain read-modify-write
#pragma unroll
        for (int u = 0; u < 2; ++u) if (pi_[u] >= 0) {
            const int i = pi_[u], j = pj_[u]; const double v = pv[u];
            if (i == D) bv[j] += v;
            else { A[i * D + j] += v; if (i != j) A[j * D + i] += v; }
        }
        __syncthreads();
    }
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
    if (a.mf16 && a.c0_mode == 0) {
        // ---------------- elimination of the dropped block AND c0 as ONE rank-revealing LDL^T, 16 wide, on the f64 matrix cores (be_mf16.h; round 5) ----------------
        // Extended system, e = 0 .. mf_n: [dropped dims 0..m-1, padded with identity rows to 16 mt | kept dims | right-hand side (row mf_n, as in the window solve: row n of L
        // comes out as y = D^-1 L^-1 b) | identity padding].  After the first mt block columns the trailing tiles ARE A' (and its right-hand-side row b'): every wave writes
        // its share to global memory from its registers (after_update, k = mt - 1) and the factorisation simply goes on — its remaining pivots are the LDL^T of A', so
        // c0 = b'^T A'^+ b' = sum y_k^2 d_k over the kept pivots.  Pivots <= 1e-8 are skipped (1 / d := 0) exactly as the 4-wide panel form below does.
        // One factorisation of 7 x 7 tiles (D = 97, m = 15) instead of 4 + 21 panel steps of three barriers each.
        const int mt = (m + 15) >> 4, mf_n = a.mf_n, NBm = (mf_n + 16) >> 4, e_k0 = 16 * mt;
        const int lane = tid & 63, wv = mf_wave(), cc = lane & 15, rr = lane >> 4;
        double* mfv = imu_ws + 960;
        MfLds mf; mf.Tl = sm; mf.gat = mfv; mf.dv = mf.gat + 1024; mf.iv = mf.dv + 16 * NBm; mf.yv = mf.iv + 16 * NBm;
        __shared__ int s_mf_fail;
        auto edim = [&](int e) { return e < e_k0 ? (e < m ? e : -2) : (e < e_k0 + n ? m + (e - e_k0) : (e == mf_n ? -1 : -2)); };      // dim of the D x D system | -1 right-hand side | -2 padding
        auto ext = [&](int ei, int ej) -> double {
            const int di = edim(ei), dj = edim(ej);
            const bool real_i = di >= 0, real_j = dj >= 0;
            const double va = A[(real_i ? di : 0) * D + (real_j ? dj : 0)], vb = bv[real_i ? di : (real_j ? dj : 0)];
            return (real_i && real_j) ? va : ((di == -1 && real_j) || (dj == -1 && real_i)) ? vb : (di == -1 && dj == -1) ? MF_RHO : (ei == ej ? 1.0 : 0.0);
        };
        int sI[MF_SLOTS], sJ[MF_SLOTS];
        mf_slots(a.mf_plan, wv, NBm, sI, sJ);
        mf_d4 U[MF_SLOTS];
        // (hoisting the row index out of the register loop was tried: 0.7 us less here, 48 spilled VGPRs and 3.4 us more in the factorisation loop)
#pragma unroll
        for (int s = 0; s < MF_SLOTS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) U[s][r] = sI[s] >= 0 ? ext(16 * sI[s] + cc, 16 * sJ[s] + rr + 4 * r) : 0.0;
        for (int i = tid; i < 16 * NBm; i += MG_THREADS) mf.yv[i] = 0.0;      // (the MF16 vectors lie behind the IMU workspace: no alias with A)
        if (tid == 0) s_mf_fail = 0;
        __syncthreads();                                  // every tile is in registers: the LDS image of A, b is dead, the factor's fragments take its place
        MTS(13);
        // A', b' leave the registers at the one moment they exist (after_update, k = mt - 1) — into a staging image in LDS behind the factor's fragments (the place of A's
        // tail and of W2), two 32-byte stores per tile: address arithmetic and scattered global stores at that point cost 144 spilled VGPRs and stalled the chain wave on
        // its own store queue (67 us against 62 for the panel form).  The image goes to global memory behind the factorisation, coalesced.
        double* stage = sm + (size_t)NBm * (NBm + 1) / 2 * 256;
        auto dump = [&](int k) {
            if (k != mt - 1) return;
#pragma unroll
            for (int s = 0; s < MF_SLOTS; ++s)
                if (sI[s] >= mt && sJ[s] >= mt) *reinterpret_cast<mf_d4*>(stage + (size_t)mf_tix(sI[s], sJ[s], NBm) * 256 + lane * 4) = U[s];
        };
        mf16_factor_core<true>(U, a.mf_plan, mf_n, mf, &s_mf_fail, dump);
        MTS(15);
        // A' (lower triangle of the staged tiles is the truth, mirrored: a diagonal tile holds both halves and they agree to rounding only) and b' (the right-hand-side row)
        auto staged = [&](int ei, int ej) {               // entry (ei, ej), ei >= ej, of the extended system as staged: result layout, lane (ej & 3, ei & 15), register (ej & 15) >> 2
            return stage[(size_t)mf_tix(ei >> 4, ej >> 4, NBm) * 256 + (((((ej & 3) << 4) + (ei & 15)) << 2) + ((ej & 15) >> 2))];
        };
        for (int e = tid; e < n * n + n; e += MG_THREADS) {
            if (e >= n * n) { const int kj = e - n * n; a.outb[kj] = staged(mf_n, e_k0 + kj); continue; }
            const int ki = e / n, kj = e - ki * n, hi = ki > kj ? ki : kj, lo = ki > kj ? kj : ki;
            a.outA[e] = staged(e_k0 + hi, e_k0 + lo);
        }
        if (wv == 0) {                                    // D of the dropped block -> smallest pivot / clamp flag; c0 and the rank from the kept pivots (fixed order: lane-strided, wave tree)
            double dmin = DBL_MAX, c0p = 0.0; int bad = 0, rk = 0;
            for (int e = lane; e < m; e += 64) { const double d = mf.dv[e]; dmin = fmin(dmin, d); bad |= !(d > 1e-8); }
            for (int k = lane; k < n; k += 64) { const double y = mf.yv[e_k0 + k], d = mf.dv[e_k0 + k]; c0p += y * y * d; rk += mf.iv[e_k0 + k] != 0.0; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { dmin = fmin(dmin, __shfl_xor(dmin, o)); bad |= __shfl_xor(bad, o); rk += __shfl_xor(rk, o); }
            c0p = wave_sum_f64(c0p);
            if (lane == 0) {
                const double mp = fmin(misc[0], dmin);
                a.out_scalars[0] = c0p; a.out_scalars[1] = mp; a.out_scalars[2] = (bad || misc[1] != 0.0) ? 1.0 : 0.0; a.out_scalars[3] = (double)rk;
                if (a.c0_out) a.c0_out[0] = c0p;
            }
        }
        MTS(16);
        return;
    }
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
    if (a.c0_mode == 1) {                               // c0 follows on a side stream (be_launch_marg_c0): the new prior's A', b' are what the BA stream waits for
        if (tid == 0) { a.out_scalars[1] = misc[0]; a.out_scalars[2] = misc[1]; }
        return;
    }
    __syncthreads();
    }
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
    if (tid == 0) {
        a.out_scalars[0] = c0; a.out_scalars[3] = (double)rank; if (a.c0_out) a.c0_out[0] = c0;
        if (a.c0_mode != 2) { a.out_scalars[1] = misc[0]; a.out_scalars[2] = misc[1]; }
    }
}
__global__ __launch_bounds__(MG_THREADS) void be_marg_finish_kernel(BeMargArgs a) { be_marg_finish_body(a); }
__global__ __launch_bounds__(MG_THREADS) void be_marg_finish_batch_kernel(const BeMargArgs* __restrict__ tab) {
    const BeMargArgs& a = tab[blockIdx.x];
    if (a.D <= 0) return;
    be_marg_finish_body(a);
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
#define REJ_LM 8           // landmarks per workgroup, 32 threads (one per residual block) each
__device__ __forceinline__ void be_reject_body(const BeRejectArgs& a, const int bx) {
    __shared__ m33 RsT[BE_NF]; __shared__ d3 Ps[BE_NF]; __shared__ m33 Rs[BE_NF];
    __shared__ m33 ric0, ricT[2]; __shared__ d3 tic[2];
    __shared__ double s_rp[REJ_LM][32];
    const int tid = threadIdx.x, g = tid >> 5, t = tid & 31, l = bx * REJ_LM + g;
    if (tid < a.nframes) {      // Rs = qR(normalized q), Ps: arrays_to_states on the downloaded (gauge-fixed) state
        const double* p = a.st->pose[tid];
        const m33 R = qR(qnormalized(mkq(p[6], p[3], p[4], p[5])));
        Rs[tid] = R; RsT[tid] = tr(R); Ps[tid] = mk3(p[0], p[1], p[2]);
    }
    if (a.ex_from_state) {      // free extrinsic blocks: arrays_to_states on the solved para_ex_pose
        if (tid == 32) { const double* p = a.st->ex[0]; ric0 = qR(qnormalized(mkq(p[6], p[3], p[4], p[5]))); }
        if (tid == 33 || tid == 34) { const int c = tid - 33; const double* p = a.st->ex[c]; ricT[c] = tr(qR(qnormalized(mkq(p[6], p[3], p[4], p[5])))); tic[c] = mk3(p[0], p[1], p[2]); }
    } else {
    if (tid == 32) { for (int k = 0; k < 9; ++k) ric0.m[k] = a.ric[0][k]; }
    if (tid == 33 || tid == 34) { const int c = tid - 33; m33 r; for (int k = 0; k < 9; ++k) r.m[k] = a.ric[c][k]; ricT[c] = tr(r); tic[c] = mk3(a.tic[c][0], a.tic[c][1], a.tic[c][2]); }
    }
    __syncthreads();
    if (l >= a.nlm) return;
    const BeLm L = a.lm[l];
    if (t < L.count) {
        const BeFactor f = a.fac[L.first + t];
        const double depth = 1.0 / a.st->inv_depth[l];
        const d3 pw = mul(Rs[L.anchor], mul(ric0, mk3(f.pix, f.piy, 1.0) * depth) + tic[0]) + Ps[L.anchor];
        const int cam = f.kind == 0 ? 0 : 1;
        const d3 pc = mul(ricT[cam], mul(RsT[f.fj], pw - Ps[f.fj]) - tic[cam]);
        const double rx = pc.x / pc.z - f.pjx, ry = pc.y / pc.z - f.pjy;
        s_rp[g][t] = sqrt(rx * rx + ry * ry);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();      // a landmark's 32 threads are half a wave
    if (t == 0) {
        double err = 0;
        for (int k = 0; k < L.count; ++k) err += s_rp[g][k];      // block order, as the host loop
        a.flags[l] = (err / L.count * a.focal > 3) ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void be_reject_kernel(BeRejectArgs a) { be_reject_body(a, blockIdx.x); }
__global__ __launch_bounds__(256) void be_reject_batch_kernel(const BeRejectArgs* __restrict__ tab) {
    const BeRejectArgs& a = tab[blockIdx.y];      // by reference: a by-value copy of the table entry is indexed dynamically (ric / tic) and therefore lived in SCRATCH — 37 MB of writes per launch of 16 windows (PMC, round 4)
    if (a.nlm <= 0 || (int)blockIdx.x * REJ_LM >= a.nlm) return;
    be_reject_body(a, blockIdx.x);
}
// last slot of a dv_batch round: the accept / reject decisions of all windows + their gauge fixes + state downloads in ONE launch (blockIdx.x = member)
__global__ __launch_bounds__(256) void be_accept_gauge_batch_kernel(const BeSolveArgs* __restrict__ stab, const BeGaugeArgs* __restrict__ gtab) {
    be_accept_body(stab[blockIdx.x]);
    __syncthreads();
    const BeGaugeArgs ga = gtab[blockIdx.x];
    be_gauge_body(ga);
}
void be_launch_reject(const BeRejectArgs& a, hipStream_t s) { if (a.nlm > 0) hipLaunchKernelGGL(be_reject_kernel, dim3((a.nlm + REJ_LM - 1) / REJ_LM), dim3(256), 0, s, a); }

void be_launch_gauge(const BeGaugeArgs& a, hipStream_t s) { hipLaunchKernelGGL(be_gauge_kernel, dim3(1), dim3(256), 0, s, a); }
void be_launch_accept_gauge(const BeSolveArgs& sa, const BeGaugeArgs& ga, hipStream_t s) { hipLaunchKernelGGL(be_accept_gauge_kernel, dim3(1), dim3(256), 0, s, sa, ga); }

static size_t finish_smem(int D, int n) { return ((size_t)D * D + D + std::max((size_t)n * n, (size_t)1024) + n + 16 + 960 + 1024 + 3 * 16 * MF_MAXNB) * sizeof(double); }      // A | b | W2 | y | misc | IMU factor | MF16: gather buffers, D, 1 / D, y
static size_t lm_smem() { return std::max((size_t)(MG_MAXF + 2) * 54, (size_t)1024) * sizeof(double); }
int be_marg_chunks(int nlm) { return (nlm + MG_CH - 1) / MG_CH; }
int be_marg_part() { return MG_PART; }
int be_marg_wstride(int D) { return mg_wstride(D); }

int be_launch_marg_c0(const BeMargArgs& a0, hipStream_t s) {
    BeMargArgs a = a0; a.c0_mode = 2;
    hipLaunchKernelGGL(be_marg_finish_kernel, dim3(1), dim3(MG_THREADS), finish_smem(a.D, a.D - a.m), s, a);
    return 0;
}
// the frame tails of a dv_batch group: accept + gauge + download, outlier test, (caller records its event), then the three marginalization stages
void be_launch_accept_gauge_batch(const BeSolveArgs* stab, const BeGaugeArgs* gtab, int n, hipStream_t s) { hipLaunchKernelGGL(be_accept_gauge_batch_kernel, dim3(n), dim3(256), 0, s, stab, gtab); }
void be_launch_reject_batch(const BeRejectArgs* tab, int n, int max_nlm, hipStream_t s) { if (max_nlm > 0) hipLaunchKernelGGL(be_reject_batch_kernel, dim3((max_nlm + REJ_LM - 1) / REJ_LM, n), dim3(256), 0, s, tab); }
size_t be_marg_finish_smem(int D, int n) { return finish_smem(D, n); }
int be_launch_marg_batch(const BeMargArgs* tab, int n, int max_nlm, int any_imu, int max_D, size_t max_finish_bytes, hipStream_t s) {
    static DevOnce once;
    if (once.run([] {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_marg_finish_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return 1;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_marg_lm_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return 1;
            return 0; })) return -1;
    if (max_finish_bytes > 156 * 1024 || lm_smem() > 96 * 1024) return -2;
    if (max_nlm > 0 || any_imu) hipLaunchKernelGGL(be_marg_lm_batch_kernel, dim3(be_marg_chunks(max_nlm) + (any_imu ? 1 : 0), n), dim3(LM_THREADS), lm_smem(), s, tab);
    if (max_nlm > 0) {
        const int NB = (max_D + 1 + 15) / 16, tiles = NB * (NB + 1) / 2;
        hipLaunchKernelGGL(be_marg_sum_batch_kernel, dim3(tiles + (MG_PART + 255) / 256, n), dim3(256), 0, s, tab);
    }
    hipLaunchKernelGGL(be_marg_finish_batch_kernel, dim3(n), dim3(MG_THREADS), max_finish_bytes, s, tab);
    return 0;
}
int be_launch_marg(const BeMargArgs& a, hipStream_t s) {
    static DevOnce once;
    if (once.run([] {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_marg_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return 1;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_marg_lm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return 1;
            return 0; })) return -1;
    const size_t bytes = finish_smem(a.D, a.D - a.m);
    if (bytes > 156 * 1024 || lm_smem() > 96 * 1024) return -2;
    if (a.nlm > 0 || a.nimu > 0) hipLaunchKernelGGL(be_marg_lm_kernel, dim3(be_marg_chunks(a.nlm) + (a.nimu > 0 ? 1 : 0)), dim3(LM_THREADS), lm_smem(), s, a);
    if (a.nlm > 0) {
        const int NB = (a.D + 1 + 15) / 16, tiles = NB * (NB + 1) / 2;
        hipLaunchKernelGGL(be_marg_sum_kernel, dim3(tiles + (MG_PART + 255) / 256), dim3(256), 0, s, a, tiles);
    }
    hipLaunchKernelGGL(be_marg_finish_kernel, dim3(1), dim3(MG_THREADS), bytes, s, a);
    return 0;
}

int be_marg_prepare() {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(be_gauge_kernel)) != hipSuccess) return -1;
    static DevOnce once;      // (the same attributes be_launch_marg sets on its first call)
    return once.run([] {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(be_marg_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return 1;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(be_marg_lm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ? 1 : 0; }) ? -1 : 0;
}
